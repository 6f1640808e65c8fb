#!/usr/bin/env python3
"""oracle/make_golden.py — TEST INFRASTRUCTURE.  Regenerates tests/golden/*.npz.

Every expected value in the fixtures is produced by oracle/_ref/ref_run, i.e. by the
reference's own translation units (src/{field,diffusion,advection,boundary,halo,decomp}.cpp
and include/stability.hpp) compiled where they lie under /root/reference and driven by
oracle/ref_harness.cpp under the image's MPICH (`/opt/conda/bin/mpirun -np P`).  The
fixtures are data only (inputs + expected outputs); no reference source text is stored.

Run in the build container (needs /root/reference):  python oracle/make_golden.py
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF_RUN = os.path.join(HERE, "_ref", "ref_run")
MPIRUN = "/opt/conda/bin/mpirun"


def run(mode, nranks=1, **kw):
    cmd = [REF_RUN, mode] + [f"--{k}={v!r}" if isinstance(v, float) else f"--{k}={v}"
                             for k, v in kw.items()]
    if nranks > 1:
        cmd = [MPIRUN, "-np", str(nranks)] + cmd
    r = subprocess.run(cmd, check=True, capture_output=True, text=True)
    return r.stdout


def decomp_table(nranks, nxg, nyg):
    out = run("decomp", nranks, nx=nxg, ny=nyg)
    rows = [[int(x) for x in ln.split()] for ln in out.strip().splitlines()]
    t = np.array(rows, dtype=np.int64)
    assert t.shape == (nranks, 13) and (t[:, 0] == np.arange(nranks)).all()
    return t[:, 1:]  # dims0 dims1 cx cy left right down up nx ny xoff yoff


RUN_CASES = [
    dict(name="run_mixed_bc_random", nx=48, ny=40, dx=1.0, dy=1.0, D=0.05, vx=0.5, vy=0.25,
         dt=0.1, steps=12, bc="dnpd", ic="random", seed=1, ranks=[1, 2, 4, 8]),
    dict(name="run_neumann_negv_gauss", nx=64, ny=64, dx=1.0, dy=1.0, D=0.1, vx=-0.4, vy=-0.3,
         dt=0.2, steps=20, bc="nnnn", ic="gaussian", sigma_frac=0.05, ranks=[1, 4]),
    dict(name="run_odd_pow2_spacing", nx=33, ny=17, dx=0.5, dy=0.25, D=0.01, vx=0.3, vy=-0.2,
         dt=0.05, steps=9, bc="ndpn", ic="random", seed=3, ranks=[1, 2, 4, 6]),
    dict(name="run_nonpow2_spacing", nx=20, ny=12, dx=0.7, dy=1.3, D=0.08, vx=0.6, vy=-0.9,
         dt=0.1, steps=7, bc="dddd", ic="random", seed=4, ranks=[1, 4]),
    dict(name="run_diffusion_only_periodic", nx=32, ny=32, dx=1.0, dy=1.0, D=1.0, vx=0.0, vy=0.0,
         dt=0.1, steps=10, bc="pppp", ic="gaussian", sigma_frac=0.1, ranks=[1, 4]),
    dict(name="run_advection_only", nx=40, ny=24, dx=1.0, dy=1.0, D=0.0, vx=-1.0, vy=0.5,
         dt=0.3, steps=8, bc="dnnd", ic="random", seed=6, ranks=[1, 2]),
    dict(name="run_dt_clamped", nx=16, ny=16, dx=1.0, dy=1.0, D=1.0, vx=0.0, vy=0.0,
         dt=1.0, steps=5, bc="dddd", ic="gaussian", sigma_frac=0.2, ranks=[1]),
    dict(name="run_dev_yaml_small", nx=64, ny=64, dx=1.0, dy=1.0, D=0.05, vx=0.5, vy=0.0,
         dt=0.1, steps=50, bc="dnpd", ic="gaussian", sigma_frac=0.05, ranks=[1, 4]),
    dict(name="run_tiny_1x1", nx=1, ny=1, dx=1.0, dy=1.0, D=0.1, vx=0.2, vy=-0.2,
         dt=0.1, steps=3, bc="ndnd", ic="random", seed=9, ranks=[1]),
    dict(name="run_tiny_2x5", nx=2, ny=5, dx=1.0, dy=2.0, D=0.1, vx=-0.2, vy=0.2,
         dt=0.1, steps=4, bc="nnpd", ic="random", seed=10, ranks=[1, 2]),
    dict(name="run_tiny_5x1", nx=5, ny=1, dx=2.0, dy=1.0, D=0.1, vx=0.2, vy=0.2,
         dt=0.1, steps=4, bc="dpnn", ic="random", seed=11, ranks=[1]),
    dict(name="run_wide_130x3", nx=130, ny=3, dx=1.0, dy=1.0, D=0.2, vx=0.1, vy=0.7,
         dt=0.1, steps=6, bc="nddn", ic="random", seed=12, ranks=[1, 2]),
    dict(name="run_tall_3x140", nx=3, ny=140, dx=1.0, dy=1.0, D=0.2, vx=-0.7, vy=-0.1,
         dt=0.1, steps=6, bc="dnnd", ic="random", seed=13, ranks=[1, 3]),
    # local widths that are multiples of 128: the two-steps-per-pass kernel runs on every rank
    dict(name="run_fused_256x48", nx=256, ny=48, dx=1.0, dy=1.0, D=0.05, vx=0.5, vy=-0.25,
         dt=0.1, steps=11, bc="dnpd", ic="random", seed=14, ranks=[1, 2, 4]),
    dict(name="run_fused_384x36", nx=384, ny=36, dx=1.0, dy=1.0, D=0.1, vx=-0.3, vy=0.4,
         dt=0.1, steps=8, bc="npnd", ic="random", seed=15, ranks=[1, 3]),
    # 4 x 2 process grids (the 8-GPU topology: mid-x ranks with three side and two diagonal peers), every
    # tile at least 7 cells deep so that fused passes of every depth run across the seams
    dict(name="run_np8_neumann_72x40", nx=72, ny=40, dx=1.0, dy=1.0, D=0.1, vx=-0.4, vy=-0.3,
         dt=0.2, steps=15, bc="nnnn", ic="gaussian", sigma_frac=0.08, ranks=[1, 8]),
    dict(name="run_np8_remainder_59x37", nx=59, ny=37, dx=0.5, dy=2.0, D=0.02, vx=0.3, vy=-0.5,
         dt=0.05, steps=10, bc="dpnd", ic="random", seed=21, ranks=[1, 2, 8]),
    dict(name="run_np8_nonpow2_61x29", nx=61, ny=29, dx=0.7, dy=1.3, D=0.08, vx=-0.6, vy=0.9,
         dt=0.1, steps=9, bc="npdn", ic="random", seed=22, ranks=[1, 4, 8]),
    dict(name="run_np8_fused_520x30", nx=520, ny=30, dx=1.0, dy=1.0, D=0.05, vx=0.5, vy=0.25,
         dt=0.1, steps=16, bc="dddd", ic="random", seed=23, ranks=[1, 8]),
]


def gen_run_case(c, tmp):
    nx, ny = c["nx"], c["ny"]
    kw = dict(nx=nx, ny=ny, dx=c["dx"], dy=c["dy"], D=c["D"], vx=c["vx"], vy=c["vy"],
              dt=c["dt"], steps=c["steps"], bc=c["bc"], dump_initial=1)
    save = {}
    if c["ic"] == "random":
        rng = np.random.default_rng(c["seed"])
        g0 = rng.random((ny, nx))
        icpath = os.path.join(tmp, c["name"] + ".ic.bin")
        g0.tofile(icpath)
        kw["ic"] = icpath
    else:
        kw["ic"] = "gaussian"
        kw["sigma_frac"] = c["sigma_frac"]
    glob_ref = None
    for p in c["ranks"]:
        out = os.path.join(tmp, f"{c['name']}.np{p}")
        txt = run("run", p, out=out, **kw)
        dt_eff = float([ln for ln in txt.splitlines() if ln.startswith("ranks=")][0]
                       .split("dt=")[1])
        tab = decomp_table(p, nx, ny)
        glob = np.zeros((ny, nx))
        glob0 = np.zeros((ny, nx))
        locs = []
        for r in range(p):
            lnx, lny, xo, yo = tab[r, 8], tab[r, 9], tab[r, 10], tab[r, 11]
            loc = np.fromfile(f"{out}.rank{r}.bin").reshape(lny + 2, lnx + 2)
            loc0 = np.fromfile(f"{out}.init.rank{r}.bin").reshape(lny + 2, lnx + 2)
            glob[yo:yo + lny, xo:xo + lnx] = loc[1:-1, 1:-1]
            glob0[yo:yo + lny, xo:xo + lnx] = loc0[1:-1, 1:-1]
            locs.append(loc)
        if glob_ref is None:
            glob_ref = glob
            save["u0"] = glob0          # global interior at t=0 (ny, nx)
            save["u_final"] = glob      # global interior after `steps` steps
            save["dt_effective"] = np.float64(dt_eff)
        else:
            # the reference is decomposition-invariant (same per-cell arithmetic)
            assert np.array_equal(glob, glob_ref), (c["name"], p)
            assert np.array_equal(glob0, save["u0"]), (c["name"], p)
        save[f"decomp_np{p}"] = tab
        for r, loc in enumerate(locs):
            save[f"local_np{p}_rank{r}"] = loc  # full local array, ghosts included
    meta = {k: v for k, v in c.items()}
    save["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(GOLD, c["name"] + ".npz"), **save)
    print("wrote", c["name"], "ranks", c["ranks"])


def gen_unit(tmp):
    rng = np.random.default_rng(100)
    save = {}
    cases = []
    k = 0
    for (nx, ny, dx, dy) in [(9, 7, 1.0, 1.0), (5, 4, 0.5, 2.0), (6, 3, 0.3, 0.9), (1, 1, 1.0, 1.0)]:
        u = rng.standard_normal((ny + 2, nx + 2))
        o = rng.standard_normal((ny + 2, nx + 2))
        up, op = os.path.join(tmp, "u.bin"), os.path.join(tmp, "o.bin")
        u.tofile(up)
        o.tofile(op)
        specs = [("diffusion", dict(D=0.1, dt=0.1)), ("diffusion", dict(D=0.0, dt=0.1))]
        for vx, vy in [(0.0, 0.0), (1.0, 0.0), (-1.0, 0.0), (0.0, 1.0), (0.0, -1.0), (0.3, -0.7),
                       (-0.2, 0.6)]:
            specs.append(("advection", dict(vx=vx, vy=vy, dt=0.1)))
        for opname, pr in specs:
            outp = os.path.join(tmp, "out.bin")
            run("unit", 1, op=opname, nx=nx, ny=ny, dx=dx, dy=dy, u=up, o=op, out=outp, **pr)
            res = np.fromfile(outp).reshape(ny + 2, nx + 2)
            save[f"c{k}_u"] = u
            save[f"c{k}_o"] = o
            save[f"c{k}_out"] = res
            cases.append(dict(idx=k, op=opname, nx=nx, ny=ny, dx=dx, dy=dy, **pr))
            k += 1
    # the reference's own known-answer test (tests/simulation/unit/test_diffusion.cpp:17-34):
    # 3x3, impulse 1.0 at (2,2), zero ghosts, D=dt=0.1 -> centre 1-4a, neighbours a (a=0.01)
    u = np.zeros((5, 5))
    u[2, 2] = 1.0
    o = np.zeros((5, 5))
    up, op, outp = (os.path.join(tmp, n) for n in ("u.bin", "o.bin", "out.bin"))
    u.tofile(up)
    o.tofile(op)
    run("unit", 1, op="diffusion", nx=3, ny=3, dx=1.0, dy=1.0, u=up, o=op, out=outp, D=0.1, dt=0.1)
    save[f"c{k}_u"], save[f"c{k}_o"] = u, o
    save[f"c{k}_out"] = np.fromfile(outp).reshape(5, 5)
    cases.append(dict(idx=k, op="diffusion", nx=3, ny=3, dx=1.0, dy=1.0, D=0.1, dt=0.1,
                      note="reference test_diffusion.cpp impulse"))
    save["meta"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(GOLD, "unit_steps.npz"), **save)
    print("wrote unit_steps", len(cases))


def gen_boundary(tmp):
    rng = np.random.default_rng(200)
    save = {}
    cases = []
    k = 0
    for (nx, ny) in [(4, 3), (7, 5), (1, 1), (2, 6)]:
        f = rng.standard_normal((ny + 2, nx + 2))
        fp, outp = os.path.join(tmp, "f.bin"), os.path.join(tmp, "out.bin")
        f.tofile(fp)
        for bc, val in [("dddd", 5.0), ("nnnn", 0.0), ("pppp", 3.0), ("dnpd", -2.5),
                        ("pnnp", 1.0), ("npdn", 7.25), ("ndnd", 0.5)]:
            run("boundary", 1, nx=nx, ny=ny, bc=bc, value=val, u=fp, out=outp)
            save[f"c{k}_in"] = f
            save[f"c{k}_out"] = np.fromfile(outp).reshape(ny + 2, nx + 2)
            cases.append(dict(idx=k, nx=nx, ny=ny, bc=bc, value=val))
            k += 1
    save["meta"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(GOLD, "boundary.npz"), **save)
    print("wrote boundary", len(cases))


def gen_decomp():
    save = {}
    grids = [(16, 12), (33, 17), (1000, 999)]
    sizes = list(range(1, 17)) + [18, 20, 24]
    for (nx, ny) in grids:
        for p in sizes:
            save[f"g{nx}x{ny}_np{p}"] = decomp_table(p, nx, ny)
    save["meta"] = np.array(json.dumps(dict(grids=grids, sizes=sizes, columns=[
        "dims0", "dims1", "cx", "cy", "left", "right", "down", "up", "nx_local", "ny_local",
        "x_offset", "y_offset"])))
    np.savez_compressed(os.path.join(GOLD, "decomp_table.npz"), **save)
    print("wrote decomp_table")


def gen_safedt():
    rows = []
    for dx, dy, vx, vy, D in [(1, 1, 0.5, 0, 0.05), (1, 1, 0, 0, 0), (1, 1, 0, 0, 1.0),
                              (0.5, 0.25, 0.3, -0.2, 0.01), (0.7, 1.3, 0.6, -0.9, 0.08),
                              (1, 1, -2.0, 0, 0), (1, 1, 0, 3.0, 0.2), (2.0, 0.1, 1e-3, 1e3, 1e-6),
                              (1, 1, 1.0, 0, 1.0), (1, 2, -0.4, -0.3, 0.1)]:
        v = float(run("safedt", 1, dx=float(dx), dy=float(dy), vx=float(vx), vy=float(vy),
                      D=float(D)).strip())
        rows.append([dx, dy, vx, vy, D, v])
    np.savez_compressed(os.path.join(GOLD, "safe_dt.npz"), table=np.array(rows, dtype=np.float64))
    print("wrote safe_dt")


def main():
    if not os.path.exists(REF_RUN):
        subprocess.run(["make", "-C", HERE], check=True)
    os.makedirs(GOLD, exist_ok=True)
    only = [a[len("--only="):] for a in sys.argv[1:] if a.startswith("--only=")]
    if only:  # add or refresh single run cases without touching the other fixtures
        with tempfile.TemporaryDirectory() as tmp:
            for c in RUN_CASES:
                if c["name"] in only:
                    gen_run_case(c, tmp)
        return 0
    with tempfile.TemporaryDirectory() as tmp:
        for c in RUN_CASES:
            gen_run_case(c, tmp)
        gen_unit(tmp)
        gen_boundary(tmp)
    gen_decomp()
    gen_safedt()
    sz = sum(os.path.getsize(os.path.join(GOLD, f)) for f in os.listdir(GOLD))
    print("total fixture bytes:", sz)


if __name__ == "__main__":
    sys.exit(main())

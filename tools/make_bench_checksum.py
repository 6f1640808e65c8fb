#!/usr/bin/env python3
"""tools/make_bench_checksum.py — writes tests/golden/bench_checksum.json (run on a GPU box).

bench.py's parity preflight advances the bench field — the hotspot as the DEVICE writes it (k_gaussian; exp() may
differ from glibc's in the last ulps, so the field cannot be made on the host) — by 1 + 7 + 32 steps and compares the
position-weighted 64-bit checksum of the result (csim_stepper_checksum, summed over the ranks) with the value stored
here.  That value is computed by the ORACLE: the device-made initial field is downloaded, oracle/cpu_stepper.c
(16 tiles / threads, pinned to the compiled reference by tests/test_oracle_golden.py) advances it by the same 40
steps, and the checksum is taken with numpy (csim.checksum_host).  The HIP result is cross-checked on the spot.
tests/test_gpu_bench.py::test_bench_checksum_fixture_is_what_the_oracle_computes repeats the derivation."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from oracle import cpu_oracle as ora  # noqa: E402

PHYS = dict(D=0.05, vx=0.5, vy=0.25, dt=0.1)   # bench.py PHYS
CHECK_STEPS = (1, 7, 32)                       # bench.py CHECK_STEPS


def entry(csim, nx, ny, bc, PHYS=PHYS):
    st = csim.Stepper.single(nx, ny, 1.0, 1.0, csim.bc_codes(bc))
    st.init_gaussian(1.0, 0.05, 0.5, 0.5)
    ic = st.download()
    cs_ic = st.checksum()
    assert cs_ic == csim.checksum_host(ic[1:-1, 1:-1])
    dt = min(PHYS["dt"], csim.safe_dt(1.0, 1.0, PHYS["vx"], PHYS["vy"], PHYS["D"]))
    for n in CHECK_STEPS:
        st.run(PHYS["D"], dt, PHYS["vx"], PHYS["vy"], n)
    cs_hip = st.checksum()
    st.close()
    w = ora.World(16 if min(nx, ny) >= 64 else 1, nx, ny)
    w.scatter(np.ascontiguousarray(ic[1:-1, 1:-1]))
    del ic
    w.run(PHYS["D"], PHYS["vx"], PHYS["vy"], dt, ora.bc_codes(bc), sum(CHECK_STEPS), threads=16)
    cs = csim.checksum_host(w.gather())
    assert cs == cs_hip, f"{nx}x{ny} {bc}: oracle {cs:#018x} != HIP {cs_hip:#018x}"
    return dict(nx=nx, ny=ny, bc=bc, steps=sum(CHECK_STEPS), D=PHYS["D"], vx=PHYS["vx"], vy=PHYS["vy"], dt=dt,
                checksum=cs, checksum_hex="0x%016x" % cs, checksum_ic=cs_ic,
                source="oracle/cpu_stepper.c (16 tiles) on the device-made hotspot, tools/make_bench_checksum.py")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "bench_checksum.json"))
    args = ap.parse_args()
    csim = load_package()
    csim.lib()
    csim.set_device(0)
    grids = [(16384, 16384, "dddd"), (16384, 16384, "nnnn"), (16384, 16384, "dnpd"), (1536, 1024, "dddd")]
    entries = [entry(csim, *g) for g in grids]
    # BASELINE configs[1] (diffusion only, all Periodic) and the same physics on the bench grid: bench.py --physics 1.0,0.1,0,0
    still = dict(D=1.0, vx=0.0, vy=0.0, dt=0.1)
    entries += [entry(csim, 4096, 4096, "pppp", still), entry(csim, 16384, 16384, "pppp", still)]
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(dict(note="position-weighted 64-bit checksum (csim_stepper_checksum) of the bench field after 1 + 7 + 32 steps, "
                        "computed by the oracle; see tools/make_bench_checksum.py", entries=entries),
              open(args.out, "w"), indent=1)
    print(json.dumps(entries))


if __name__ == "__main__":
    main()

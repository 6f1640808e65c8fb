#!/usr/bin/env python3
"""bench.py — the reference's headline metric on MI355X: Mcell-updates/s (and achieved HBM GB/s)
of the per-step advection–diffusion sweep on a 16384 x 16384 fp64 grid at 1/2/4/8 GPUs.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (reference src/main.cpp:101-109: halo exchange, boundary
fill, fused copy+diffusion+advection sweep, swap) over the whole grid; the grid is resident in
HBM when the timed region starts (gaussian hotspot written on the device).  N > 1 is STRONG
scaling of the same global grid: one process per GPU, 2D block decomposition by the
MPI_Dims_create rule, halos over RCCL send/recv on a second HIP stream.  torch.distributed
(gloo) is only the control plane: unique-id broadcast, barrier, max-over-ranks.

Rank 0 prints ONE JSON line.  Extra objects: `roofline` (dominant kernel = the fused sweep;
algorithmic bytes = 16 B per cell update; duration from HIP events on the compute stream
around every sweep launch of the timed region) and, at N = 1, `cpu_baseline` (the compiled
reference objects, oracle/_ref/ref_run under mpirun, on a bounded sample — or the oracle port
when that binary cannot run).
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_CELL = 16.0       # one 8-byte read of u + one 8-byte write of u'

# BASELINE.json `metric`: 16384^2 fp64; physics of configs[2]/[3]
NX = NY = 16384
PHYS = dict(D=0.05, vx=0.5, vy=0.25, dt=0.1)
BC = "dddd"


def cpu_baseline(cores: int):
    """Reference CPU path on the host cores, bounded sample of the same workload."""
    from oracle import cpu_oracle as ora
    nx = ny = int(os.environ.get("CSIM_BENCH_CPU_N", NX))
    steps = int(os.environ.get("CSIM_BENCH_CPU_STEPS", 4))
    if ora.have_reference():
        try:
            t0 = time.time()
            out = ora.ref_run("run", np_ranks=cores, timeout=600, nx=nx, ny=ny, steps=steps,
                              bc=BC, ic="gaussian", **PHYS)
            wall = time.time() - t0
            m = re.search(r"timing: total_max=([0-9.eE+-]+) s", out)
            loop_s = float(m.group(1))
            return dict(value=nx * ny * steps / loop_s / 1e6, unit="Mcell-updates/s", cores=cores,
                        kind="reference",
                        sample=f"{nx}x{ny} fp64, {steps} steps, oracle/_ref/ref_run (reference "
                               f"objects, g++ -O2) under mpirun -np {cores}; loop {loop_s:.2f} s, "
                               f"whole run {wall:.1f} s")
        except Exception as e:  # mpirun unusable on this box: fall back to the port
            sys.stderr.write(f"[bench] reference baseline unavailable ({e}); using the port\n")
    w = ora.World(cores, nx, ny, 1.0, 1.0)
    w.gaussian()
    secs = w.run(PHYS["D"], PHYS["vx"], PHYS["vy"], PHYS["dt"], ora.bc_codes(BC), steps,
                 threads=cores)
    return dict(value=nx * ny * steps / secs / 1e6, unit="Mcell-updates/s", cores=cores, kind="port",
                sample=f"{nx}x{ny} fp64, {steps} steps, oracle/cpu_stepper.c with {cores} threads "
                       f"(one tile per thread); loop {secs:.2f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--ramp-seconds", type=float, default=0.3,
                    help="untimed stepping before the warm-up steps so that the GPU has left its idle "
                         "clocks (a cold MI355X runs its first ~30 ms about 15 %% below the sustained rate)")
    ap.add_argument("--nx", type=int, default=NX)
    ap.add_argument("--ny", type=int, default=NY)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--rows-per-chunk", type=int, default=0)
    ap.add_argument("--prefetch", type=int, default=0)
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--overlap-mode", type=int, default=-1,
                    help="N > 1: -1 pick the fastest exchange schedule on this node, 0/1/2 force one")
    ap.add_argument("--lds-bytes", type=int, default=0, help="occupancy limiter experiment (see csim.h)")
    ap.add_argument("--fuse", type=int, default=-1,
                    help="time steps per HBM pass: -1 auto (deepest available), 0 off, 2..6")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test mode: ONE rank whose four neighbours are the rank itself, so that the whole N > 1 code path
    # (RCCL communicator, deep faces in 8 directions, schedule selection) runs on a single GPU
    self_torus = world == 1 and os.environ.get("CSIM_BENCH_SELF_TORUS") == "1"
    multi = world > 1 or self_torus
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    # CPU baseline first: it forks mpirun, which must happen before this process touches the GPU
    cpu = None
    if world == 1 and not self_torus and not args.no_cpu_baseline:
        cores = min(16, os.cpu_count() or 1)
        cpu = cpu_baseline(cores)

    import torch  # noqa: F401  (plumbing: torch.distributed control plane; also pins ONE HIP runtime)
    import torch.distributed as dist
    from __graft_entry__ import load_package
    csim = load_package()
    csim.lib()
    ndev = csim.device_count()
    if local_rank >= ndev and os.environ.get("CSIM_BENCH_HALO") != "gloo":
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible (one rank per GPU)")
    csim.set_device(local_rank % max(ndev, 1))

    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    dec = csim.decomp_init(world, rank, args.nx, args.ny)
    if self_torus:
        for k in range(4):
            dec.nbr[k] = 0
    st = csim.Stepper(dec, 1.0, 1.0, csim.bc_codes(BC), 0.0)
    halo = "rccl" if multi else "none"
    if multi:
        # RCCL communicator (unique id over the gloo control plane).  If it cannot be built on this
        # box the run falls back — on every rank — to host-staged faces over gloo, so that a scaling
        # number exists at all; the JSON line says which transport carried the halos.
        ok, why = 1, ""
        box = [None]
        if rank == 0 and os.environ.get("CSIM_BENCH_HALO", "rccl") != "gloo":
            try:
                box = [csim.comm_unique_id()]
            except Exception as e:  # noqa: BLE001
                why = str(e)
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            ok = 0
        else:
            try:
                st.comm_init(box[0])
            except Exception as e:  # noqa: BLE001
                ok, why = 0, str(e)
        t = torch.tensor([ok], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 0:
            halo = "gloo (host-staged; RCCL unavailable: %s)" % (why or "see other ranks")
            sys.stderr.write(f"[bench] rank {rank}: falling back to host-staged halos over gloo ({why})\n")
            st.set_option("external_halo", 1)
    for key, val in (("variant", args.variant), ("rows_per_chunk", args.rows_per_chunk),
                     ("prefetch", args.prefetch), ("overlap", 0 if args.no_overlap else 1),
                     ("fuse", args.fuse)):
        st.set_option(key, val)
    if args.lds_bytes:
        st.set_option("lds_bytes", args.lds_bytes)
    st.init_gaussian(1.0, 0.05, 0.5, 0.5)
    dt = min(PHYS["dt"], csim.safe_dt(1.0, 1.0, PHYS["vx"], PHYS["vy"], PHYS["D"]))

    nbr = list(dec.nbr)

    def advance(n):
        if halo.startswith("gloo"):
            from climate_sim_mpi_cpp_amd.host_transport import advance as advance_external
            advance_external(st, nbr, PHYS["D"], dt, PHYS["vx"], PHYS["vy"], n)
        else:
            st.run(PHYS["D"], dt, PHYS["vx"], PHYS["vy"], n)

    def global_sum():
        v = st.sum()
        if multi:
            t = torch.tensor([v], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            v = float(t.item())
        return v

    # FTCS diffusion + upwind advection conserve the total of u up to the flux through the
    # physical edges (nil here: the hotspot stays far from them), so a halo exchange that lost
    # or misplaced a face would show up as mass leaking at the tile seams through the centre
    mass0 = global_sum()

    def barrier():
        st.sync()
        if multi:
            dist.barrier()

    # untimed: clock ramp (also triggers the stepper's one-off rows-per-chunk trial), then W warm-up steps
    ramp_steps = 0
    t_ramp = time.perf_counter()
    while args.ramp_seconds > 0:
        advance(60)
        st.sync()
        ramp_steps += 60
        done = time.perf_counter() - t_ramp >= args.ramp_seconds
        if multi:  # every rank must take the same number of steps: decide together
            t = torch.tensor([1 if done else 0], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            done = bool(t.item())
        if done:
            break
    # N > 1 over RCCL: how the exchange is best hidden depends on what the RCCL kernel costs next
    # to the sweep on this node, which cannot be known beforehand: time the schedules the stepper
    # offers on a few passes each (untimed as far as `value` is concerned), keep the fastest on
    # every rank, and report all of them (SURVEY §8d config 4 asks for overlapped and
    # non-overlapped timings anyway)
    exchange_modes = None
    if multi and halo == "rccl" and not args.no_overlap and args.overlap_mode < 0:
        cands = [("overlap-1 frame first, exchange under the bulk sweep", 1, None),
                 ("overlap-0 exchange not overlapped", 0, None)]
        if os.environ.get("CSIM_BENCH_TRY_OVERLAP2") == "1":
            # the three-stream schedule has only ever run on the self-linked torus of one GPU: opt-in
            cands += [("overlap-2 frame stream beside the bulk, bulk capped at 3 workgroups/CU", 2, 41984),
                      ("overlap-2 frame stream beside the bulk", 2, 0)]
        exchange_modes = {}
        k2 = 240
        for name, ov, lds in cands:
            st.set_option("overlap", ov)
            if lds is not None:
                st.set_option("bulk_lds", lds)
            advance(24)
            barrier()
            t0 = time.perf_counter()
            advance(k2)
            st.sync()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            exchange_modes[name] = float(t.item()) / k2 * 1e3  # ms per step, max over ranks: same on all
        best = min(exchange_modes, key=exchange_modes.get)
        if exchange_modes[best] > 0.98 * exchange_modes[cands[0][0]]:
            best = cands[0][0]  # within noise of the default schedule: keep the default
        for name, ov, lds in cands:
            if name == best:
                st.set_option("overlap", ov)
                if lds is not None:
                    st.set_option("bulk_lds", lds)
        exchange_modes["chosen"] = best
    elif multi and args.overlap_mode >= 0:
        st.set_option("overlap", args.overlap_mode)
    advance(args.warmup)
    barrier()
    # HIP events around every sweep launch at N = 1; around every 8th pass at N > 1, where the two
    # event records per pass would cost ~10 % of a 170 us pass
    st.set_option("profile", 8 if multi else 1)
    st.reset_timers()
    t0 = time.perf_counter()
    advance(args.steps)
    st.sync()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if multi:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # dominant kernel = the one that advanced most of the timed steps
    kinds = {t: st.kernel_time(t) for t in (1, 2, 3, 4, 5, 6)}
    steps_per_launch = max(kinds, key=lambda t: t * kinds[t][1])
    kern_ms, launches = kinds[steps_per_launch]
    mn, mx = st.minmax()
    tuned_rows = st.get_option("tuned_rows") or (args.rows_per_chunk or "heuristic")
    mass1 = global_sum()
    mass_drift = abs(mass1 - mass0) / abs(mass0)
    if mass_drift > 1e-9 and rank == 0:
        sys.stderr.write(f"[bench] WARNING: total mass drifted by {mass_drift:.3e} (halo exchange broken?)\n")
    st.close()
    if multi:
        km = torch.tensor([kern_ms], dtype=torch.float64)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        kern_ms = float(km.item())
        dist.destroy_process_group()
    kern_avg_ms = kern_ms / max(launches, 1)

    if rank == 0:
        cells = float(args.nx) * float(args.ny)
        value = cells * args.steps / elapsed / 1e6
        local_cells = float(dec.nx_local) * float(dec.ny_local)
        # algorithmic bytes of ONE launch = 16 B x local cells x time steps that launch advances
        ach = local_cells * BYTES_PER_CELL * steps_per_launch / (kern_avg_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if world == 1 and os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                # PMC bytes were collected in separate rocprofv3 --pmc passes of this same command
                # (tools/gpu_pmc.sh); only quoted when they belong to the kernel timed here
                if (tj.get("nx") == args.nx and tj.get("ny") == args.ny
                        and tj.get("steps_per_launch") == steps_per_launch):
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "Mcell-updates/sec (16384^2 fp64 advection-diffusion sweep)",
            "value": value,
            "unit": "Mcell-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.nx}x{args.ny} fp64 gaussian hotspot, D={PHYS['D']} "
                            f"v=({PHYS['vx']},{PHYS['vy']}) dt={dt} dx=dy=1, all-Dirichlet(0), "
                            f"decomp {dec.dims[0]}x{dec.dims[1]} (local {dec.nx_local}x{dec.ny_local}), "
                            f"halo overlap {'off' if args.no_overlap else 'on'}"
                            + (" — TEST MODE: one rank linked to itself in all 8 directions" if self_torus else ""),
                "halo_transport": halo,
                "exchange_schedules_ms_per_step": exchange_modes,
                "hbm_gbs_whole_job": cells * args.steps * BYTES_PER_CELL / elapsed / 1e9,
                "field_min_max_after_run": [mn, mx],
                "relative_mass_drift": mass_drift,
                "untimed_clock_ramp_steps": ramp_steps,
                "rows_per_chunk": tuned_rows,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": ach,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic,
                # what really moved through HBM per second (PMC bytes / live kernel time): the kernel
                # advances 6 time steps per pass, so `achieved` (algorithmic) exceeds the peak while
                # the real traffic stays below it
                "traffic_gbs": (traffic / (kern_avg_ms * 1e-3) / 1e9) if traffic else None,
                "traffic_frac_of_peak": (traffic / (kern_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "kernel": ("k_sweep_dpp" if steps_per_launch == 1 else f"k_sweepO_dpp<T={steps_per_launch}>") +
                          f" (fused copy+diffusion+advection, {steps_per_launch} time step(s) per HBM pass)",
                "kernel_avg_ms": kern_avg_ms,
                "launches_timed": launches,
                "time_steps_per_launch": steps_per_launch,
                "algorithmic_bytes_per_launch": local_cells * BYTES_PER_CELL * steps_per_launch,
            },
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()

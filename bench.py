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

Rank 0 prints ONE JSON line.  Extra objects:

`roofline` (dominant kernel = the fused sweep, HBM side).  The sweep advances T time steps per HBM
  pass (temporal blocking in registers), so three byte counts exist per launch and the line names
  each one:
    traffic                        what really moved: rocprofv3 PMC bytes of this kernel (FETCH_SIZE x 2 +
                                   WRITE_SIZE, separate passes, profiles/pmc_traffic.json), looked up by the
                                   kernel instantiation (T) and grid that were timed here
    algorithmic_bytes_per_launch   what one launch must move at least: 16 B x cells (each cell read
                                   once and written once per pass)
    step_equivalent_bytes          16 B x cells x T: what a one-step-per-pass sweep would move for the
                                   same T updates (the figure SURVEY §8d's 16 B per cell-UPDATE gives)
  `achieved` = traffic / (HIP-event kernel time measured live in the timed region) and `frac` =
  achieved / 8 TB/s are therefore a true bandwidth and a true fraction (<= 1); the step-equivalent
  rate, which exceeds the HBM peak by construction, is reported beside them as
  `step_equivalent_gbs` / `step_equivalent_x_peak` and never as `frac`.
`roofline_valu`: the resource that actually binds the T >= 4 kernels — fp64 VALU issue (the
  reference's own 15 fp64 operations per cell update; contraction would change bits, except for the one
  fusion that is exact, E - 2c as fma(-2, c, E), which the kernel uses under an overflow screen: 14 issued).
`cpu_baseline` (N = 1): the compiled reference objects (oracle/_ref/ref_run under mpirun) and the
  oracle port (checked / unchecked accessor flavours) on bounded samples, host core counts stated.
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

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0       # same guide: float4 copy, what a streaming kernel can reach
BYTES_PER_CELL = 16.0       # one 8-byte read of u + one 8-byte write of u'
FP64_OPS_PER_UPDATE = 15    # non-FMA fp64 add/mul per cell update with dx = dy = 1 (csrc/kernels.hip cell<>)
# vector fp64 peak: 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz = 39.3 T non-FMA op/s (78.6 TFLOP/s as FMA)
FP64_VALU_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12

# BASELINE.json `metric`: 16384^2 fp64; physics of configs[2]/[3]
NX = NY = 16384
PHYS = dict(D=0.05, vx=0.5, vy=0.25, dt=0.1)
BC = "dddd"


# ---------------------------------------------------------------------------------------------
# CPU baseline: a reported baseline, never the target
# ---------------------------------------------------------------------------------------------
def host_cpu_info():
    """logical / physical core counts of the box and what this process may use of them"""
    logical = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except Exception:
        affinity = logical
    cores, model = set(), ""
    try:
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and not model:
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                phys = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":", 1)[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except Exception:
        pass
    quota = None
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(f).read().split()
            if f.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    return dict(nproc_logical=logical, physical_cores=len(cores) or None, affinity=affinity,
                cgroup_cpu_quota=quota, model=model)


def cpu_baseline():
    """Reference CPU path on the host cores, bounded samples of the same workload (about 20 s)."""
    from oracle import cpu_oracle as ora
    host = host_cpu_info()
    usable = host["affinity"]
    if host["cgroup_cpu_quota"]:
        usable = max(1, min(usable, int(host["cgroup_cpu_quota"])))
    # the GPU pool gives a one-GPU lease a 16-core share of the host (more threads would run on cores
    # that belong to other leases): use every usable core up to that share unless told otherwise
    share = int(os.environ.get("CSIM_BENCH_CPU_CORES", 16))
    cores = max(1, min(usable, share))
    host.update(usable=usable, used=cores,
                cap="min(usable, 16): the pool's CPU share of a one-GPU lease (override: CSIM_BENCH_CPU_CORES)")
    nx = int(os.environ.get("CSIM_BENCH_CPU_N", NX))
    steps = int(os.environ.get("CSIM_BENCH_CPU_STEPS", 4))
    small_n, small_steps = int(os.environ.get("CSIM_BENCH_CPU_SMALL_N", 4096)), 40
    variants = []

    def rate(n, k, secs):
        return n * n * k / secs / 1e6

    def reference(n, k):
        t0 = time.time()
        out = ora.ref_run("run", np_ranks=cores, timeout=600, nx=n, ny=n, steps=k, bc=BC, ic="gaussian", **PHYS)
        wall = time.time() - t0
        loop_s = float(re.search(r"timing: total_max=([0-9.eE+-]+) s", out).group(1))
        variants.append(dict(kind="reference", grid=f"{n}x{n}", steps=k, value=rate(n, k, loop_s), cores=cores,
                             how=f"oracle/_ref/ref_run (the reference's own translation units, g++ -O2, "
                                 f"bounds-checked Field::at) under mpirun -np {cores}; loop {loop_s:.2f} s, "
                                 f"whole run {wall:.1f} s"))
        return variants[-1]

    def port(n, k, checked):
        w = ora.World(cores, n, n, 1.0, 1.0)
        w.gaussian()
        secs = w.run(PHYS["D"], PHYS["vx"], PHYS["vy"], PHYS["dt"], ora.bc_codes(BC), k, threads=cores,
                     checked=checked)
        how = "Field::at-style double bounds check on every access" if checked else "row pointers, no bounds checks"
        variants.append(dict(kind="port-checked" if checked else "port-unchecked", grid=f"{n}x{n}", steps=k,
                             value=rate(n, k, secs), cores=cores,
                             how=f"oracle/cpu_stepper.c, {cores} tiles on {cores} threads, {how}; loop {secs:.2f} s"))
        return variants[-1]

    head = None
    if ora.have_reference():
        try:
            head = reference(nx, steps)
            reference(small_n, small_steps)
        except Exception as e:  # mpirun unusable on this box: the port carries the baseline
            sys.stderr.write(f"[bench] reference baseline unavailable ({e}); using the port\n")
    p_un = port(nx, steps, False)
    port(nx, steps, True)
    port(small_n, small_steps, False)
    port(small_n, small_steps, True)
    if head is None:
        head = p_un
    return dict(value=head["value"], unit="Mcell-updates/s", cores=cores,
                kind="reference" if head["kind"] == "reference" else "port",
                sample=f"{head['grid']} fp64, {head['steps']} steps of the bench workload; {head['how']}",
                host=host, variants=variants)


# ---------------------------------------------------------------------------------------------
# profile look-ups (files written by tools/gpu_pmc.sh / tools/gpu_pmc_sq.sh on a GPU box)
# ---------------------------------------------------------------------------------------------
def kernel_label(T):
    return "k_sweep_dpp" if T == 1 else f"k_sweepO_dpp<T={T}>"


def lookup_traffic(nx, ny, T, bc, rows=0):
    """PMC HBM bytes per launch of the sweep instantiation that was timed: same T, same local grid.
    Falls back to the per-cell figure of another grid of the same T (flagged) — the traffic per cell
    of a streaming sweep does not depend on the grid once it is far beyond the 256 MiB Infinity Cache."""
    f = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        entries = json.load(open(f)).get("entries", [])
    except Exception:
        return None, "profiles/pmc_traffic.json missing"
    same_t = [e for e in entries if e.get("steps_per_launch") == T]
    exact = [e for e in same_t if e.get("nx") == nx and e.get("ny") == ny]
    pick = [e for e in exact if e.get("bc") == bc] or exact
    if pick:
        # several chunk heights were profiled (more rows per chunk = fewer overhead rows re-read): nearest one
        e = min(pick, key=lambda e: abs(e.get("rows_per_chunk", 0) - rows))
        return e["hbm_bytes_per_launch"], (
            f"profiles/pmc_traffic.json '{e['kernel']}' {e['nx']}x{e['ny']} bc={e.get('bc')} rows_per_chunk="
            f"{e.get('rows_per_chunk')}: rocprofv3 --pmc "
            f"FETCH_SIZE / WRITE_SIZE in separate passes, (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH correction)")
    big = [e for e in same_t if e.get("nx", 0) * e.get("ny", 0) * 8 >= (1 << 29)]
    if big and nx * ny * 8 >= (1 << 29):
        e = min(big, key=lambda e: (e.get("bc") != bc, abs(e.get("rows_per_chunk", 0) - rows)))
        per_cell = e["hbm_bytes_per_launch"] / (e["nx"] * e["ny"])
        return per_cell * nx * ny, (f"SCALED per cell from profiles/pmc_traffic.json '{e['kernel']}' "
                                    f"{e['nx']}x{e['ny']} ({per_cell:.2f} B per cell per launch)")
    return None, f"no PMC entry for T={T} on {nx}x{ny}"


def lookup_valu(T):
    f = os.path.join(ROOT, "profiles", "sq_valu.json")
    try:
        for e in json.load(open(f)).get("entries", []):
            if e.get("steps_per_launch") == T:
                return e
    except Exception:
        pass
    return None


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
    ap.add_argument("--bc", default=BC, help="boundary mix left/right/bottom/top, e.g. dddd (default), nnnn, dnpd")
    ap.add_argument("--contract", type=int, default=0,
                    help="0 (default): the reference's own operation order, bit-identical results; 1: opt-in "
                         "contracted arithmetic (see csim.h option \"contract\"), within 1e-10 of the reference")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--rows-per-chunk", type=int, default=0)
    ap.add_argument("--prefetch", type=int, default=0)
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--overlap-mode", type=int, default=-1,
                    help="N > 1: -1 pick the fastest exchange schedule on this node; 0, 1, 3, 4, 5 force one")
    ap.add_argument("--lds-bytes", type=int, default=0, help="occupancy limiter experiment (see csim.h)")
    ap.add_argument("--fused-2c", type=int, default=-1, help="0/1: E - 2c as one fma under the overflow guard (bit-identical either way; default: on)")
    ap.add_argument("--tail-split", type=int, default=-1, help="0/1: half-height chunks at the end of a whole-field launch (default: on)")
    ap.add_argument("--fuse", type=int, default=-1,
                    help="time steps per HBM pass: -1 auto (cheapest split of the run into passes of 2..7 steps), 0 off, 2..7")
    args = ap.parse_args()
    assert len(args.bc) == 4 and set(args.bc) <= set("dnp"), "--bc takes four of d/n/p"

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test mode: ONE rank whose four neighbours are the rank itself, so that the whole N > 1 code path
    # (RCCL communicator, deep faces in 8 directions, schedule selection) runs on a single GPU
    self_torus = world == 1 and os.environ.get("CSIM_BENCH_SELF_TORUS") == "1"
    multi = world > 1 or self_torus
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    # stdout carries ONE JSON line (rank 0) and nothing else: the native libraries print banners straight to file
    # descriptor 1 (RCCL's version block, gloo's connection note), so descriptor 1 points at stderr until the line
    # is written
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    # CPU baseline first: it forks mpirun, which must happen before this process touches the GPU
    cpu = None
    if world == 1 and not self_torus and not args.no_cpu_baseline:
        cpu = cpu_baseline()

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
    st = csim.Stepper(dec, 1.0, 1.0, csim.bc_codes(args.bc), 0.0)
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
                     ("prefetch", args.prefetch), ("fuse", args.fuse)):
        st.set_option(key, val)
    if args.no_overlap:
        st.set_option("overlap", 0)   # otherwise the stepper's default (3 where the device offers it, else 1)
    if args.contract:
        st.set_option("contract", args.contract)
    if args.lds_bytes:
        st.set_option("lds_bytes", args.lds_bytes)
    if args.tail_split >= 0:
        st.set_option("tail_split", args.tail_split)
    if args.fused_2c >= 0:
        st.set_option("fused_2c", args.fused_2c)
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
            # a gloo barrier releases the ranks up to a few hundred microseconds apart — a sixth of a 20-step
            # timed region at 8 GPUs, which the first halo exchange would then spend waiting for the last rank.
            # All ranks sit on one node and CLOCK_MONOTONIC is system-wide: agree on a start time 3 ms ahead
            # and spin up to it, so that every rank enters the timed region within a microsecond of the others.
            box = [time.perf_counter() + 0.003 if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            # ... under load: a GPU that idles through these 3 ms leaves its sustained power state and runs the
            # first launches afterwards 5-15 % slower (tools/gpu_trace_steps20.sh) — a third of a 20-step region at
            # 8 GPUs.  Local launches that neither advance the field nor communicate, ending ~0.3 ms before the start.
            if halo == "rccl":
                st.keep_warm(PHYS["D"], dt, PHYS["vx"], PHYS["vy"], max(0.0, box[0] - time.perf_counter() - 0.0003))
            while time.perf_counter() < box[0]:
                pass

    # untimed: the W warm-up steps first (cold: first launches of the kernels, code loading), then the clock ramp
    # in bursts shaped like the timed run (also triggers the stepper's one-off rows-per-chunk trial).  The ramp comes
    # LAST so that nothing but the synchronisation stands between steady-state load and the timed region: a first
    # launch of a new kernel kind stalls the host for ~2 ms (code loading), the idle GPU drops out of its sustained
    # power state, and the next few launches then run 5-15 % slower than in steady state — on a 20-step timed
    # region (three launches) that was the difference between 1.45 and 1.52 M (kernel timelines: tools/gpu_trace_steps20.sh)
    st.tune(PHYS["D"], dt, PHYS["vx"], PHYS["vy"])  # the one-off chunk-height trial a first long run() would do (local, no exchange)
    advance(args.warmup)
    burst = max(1, min(args.steps, 60))
    ramp_steps = 0
    t_ramp = time.perf_counter()
    while args.ramp_seconds > 0:
        advance(burst)
        st.sync()
        ramp_steps += burst
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
        # the stepper's default (5) picks by run length: bulk-first on short runs, merged launches on long ones
        cands = [("overlap-5 default: bulk-first on runs of < 16 passes, else frame and bulk in one launch", 5),
                 ("overlap-3 frame and bulk in one launch, exchange released by an in-kernel flag", 3),
                 ("overlap-4 bulk launch first hiding this pass's exchange, then the frame launch", 4),
                 ("overlap-1 frame launch first, next pass's exchange under the bulk launch", 1),
                 ("overlap-0 exchange not overlapped", 0)]
        try:
            st.set_option("overlap", 3)
        except Exception:  # noqa: BLE001  (no hipStreamWaitValue64 / signal memory on this device)
            cands = [c for c in cands if c[1] != 3]
        exchange_modes = {}
        # trial runs shaped like the timed one (the schedules differ in what a run() call costs at its start):
        # repetitions of advance(--steps) adding up to >= 240 steps
        reps = max(1, -(-240 // max(1, args.steps))) if args.steps < 240 else 1
        k2 = min(args.steps, 240) * reps if args.steps < 240 else 240
        for name, ov in cands:
            st.set_option("overlap", ov)
            advance(min(args.steps, 24))
            barrier()
            t0 = time.perf_counter()
            if args.steps < 240:
                for _ in range(reps):
                    advance(args.steps)
            else:
                advance(240)
            st.sync()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            exchange_modes[name] = float(t.item()) / k2 * 1e3  # ms per step, max over ranks: same on all
        best = min(exchange_modes, key=exchange_modes.get)
        if exchange_modes[best] > 0.98 * exchange_modes[cands[0][0]]:
            best = cands[0][0]  # within noise of the default schedule: keep the default
        for name, ov in cands:
            if name == best:
                st.set_option("overlap", ov)
        exchange_modes["chosen"] = best
    elif multi and args.overlap_mode >= 0:
        st.set_option("overlap", args.overlap_mode)
    if multi and args.ramp_seconds > 0:  # one more burst with the schedule just chosen
        advance(burst)
        ramp_steps += burst
    # HIP events around every sweep launch at N = 1; around every 8th pass at N > 1, where the two
    # event records per pass would cost ~10 % of a 170 us pass.  (Set before the barrier: nothing but the clock
    # read stands between the synchronisation and the first timed launch, so the GPU idles as briefly as it can.)
    st.set_option("profile", 8 if multi else 1)
    barrier()
    st.reset_timers()
    t0 = time.perf_counter()
    advance(args.steps)
    st.sync()
    t1 = time.perf_counter()
    elapsed_local = elapsed = t1 - t0
    if multi:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # dominant kernel = the one that advanced most of the timed steps
    kinds = {t: st.kernel_time(t) for t in (1, 2, 3, 4, 5, 6, 7)}
    steps_per_launch = max(kinds, key=lambda t: t * kinds[t][1])
    kern_ms, launches = kinds[steps_per_launch]
    comm_ms, comm_n = st.comm_time()
    mn, mx = st.minmax()
    tuned_rows = st.get_option("tuned_rows") or (args.rows_per_chunk or "heuristic")
    last_rows = st.get_option("last_rows")
    overlap_now = st.get_option("overlap")
    mass1 = global_sum()
    mass_drift = abs(mass1 - mass0) / abs(mass0)
    if mass_drift > 1e-9 and rank == 0:
        sys.stderr.write(f"[bench] WARNING: total mass drifted by {mass_drift:.3e} (halo exchange broken?)\n")
    st.close()
    kern_avg_local = kern_ms / max(launches, 1)
    per_rank = None
    if multi:
        # every rank's own figures, so that a slow or skewed rank is visible in the one line
        mine = dict(rank=rank, coords=[dec.coords[0], dec.coords[1]], local=[dec.nx_local, dec.ny_local],
                    neighbours=list(dec.nbr), wall_ms_per_step=elapsed_local / args.steps * 1e3,
                    kernel=kernel_label(steps_per_launch), kernel_avg_ms=kern_avg_local, launches_timed=launches,
                    exchange_chain_avg_ms=(comm_ms / comm_n) if comm_n else None, exchange_chains_timed=comm_n,
                    rows_per_chunk=last_rows, overlap=overlap_now)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        km = torch.tensor([kern_ms], dtype=torch.float64)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        kern_ms = float(km.item())
        dist.destroy_process_group()
    kern_avg_ms = kern_ms / max(launches, 1)

    if rank == 0:
        cells = float(args.nx) * float(args.ny)
        value = cells * args.steps / elapsed / 1e6
        local_cells = float(dec.nx_local) * float(dec.ny_local)
        T = steps_per_launch
        secs = kern_avg_ms * 1e-3
        alg_bytes = local_cells * BYTES_PER_CELL                 # one read + one write of the field per launch
        step_eq_bytes = alg_bytes * T                            # what T one-step passes would move
        if args.contract:
            traffic, traffic_src = None, "contracted arithmetic: no PMC profile"
        else:
            traffic, traffic_src = lookup_traffic(dec.nx_local, dec.ny_local, T, args.bc, last_rows)
        if multi:
            # the PMC profiles are of the whole-field launch; a multi-rank pass is frame + bulk launches
            traffic_src += " (whole-field launch; this run splits a pass into frame + bulk launches)"
        if traffic is not None:
            ach_bytes, ach_src = traffic, "traffic (PMC)"
        else:
            ach_bytes, ach_src = alg_bytes, "algorithmic_bytes_per_launch (no PMC entry: a LOWER bound of the real traffic)"
        ach = ach_bytes / secs / 1e9
        roofline = {
            "bound": "hbm",
            # the HBM side is what this object prices (the contract's schema); what BINDS the kernel at T >= 4 is
            # fp64 VALU issue, priced in roofline_valu
            "binding_resource": ("fp64-valu (see roofline_valu)" if (T >= 4 and not args.contract) else "hbm"),
            "achieved": ach,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic,
            "achieved_from": ach_src,
            "traffic_source": traffic_src,
            "frac_of_measured_copy_peak": ach / HBM_COPY_GBS,
            "kernel": kernel_label(T) + f" (fused copy+diffusion+advection, {T} time step(s) per HBM pass)",
            "kernel_avg_ms": kern_avg_ms,
            "launches_timed": launches,
            "time_steps_per_launch": T,
            "algorithmic_bytes_per_launch": alg_bytes,
            "algorithmic_gbs": alg_bytes / secs / 1e9,
            "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
            "step_equivalent_bytes": step_eq_bytes,
            "step_equivalent_gbs": step_eq_bytes / secs / 1e9,
            "step_equivalent_x_peak": step_eq_bytes / secs / 1e9 / HBM_PEAK_GBS,
            "note": "frac = PMC traffic / live kernel time / 8 TB/s.  step_equivalent_* counts 16 B per cell-UPDATE "
                    "(SURVEY §8d) and exceeds the peak because T time levels stay in registers per pass; it is "
                    "the figure to compare with a one-step-per-pass sweep, not a bandwidth.  A deeper pass LOWERS frac while "
                    "raising the throughput (16384^2: T = 6 ~0.55, T = 7 ~0.48 at +0.9 % Mcell-updates/s): the kernel is bound "
                    "by fp64 VALU issue (roofline_valu: VALUs ~95 % busy at the clock the chip holds), not by HBM",
        }
        ops_per_update = FP64_OPS_PER_UPDATE if not args.contract else 5
        useful_tops = local_cells * T * ops_per_update / secs / 1e12
        valu = lookup_valu(T) if not args.contract else None
        roofline_valu = {
            "bound": "fp64-valu",
            "achieved": useful_tops,
            "peak": FP64_VALU_PEAK_TOPS,
            "unit": "T fp64 op/s (non-FMA add/mul)",
            "frac": useful_tops / FP64_VALU_PEAK_TOPS,
            "useful_ops_per_cell_update": ops_per_update,
            "note": "useful = the reference's own operations per cell update x updates per launch / live kernel "
                    "time; peak = 256 CUs x 4 SIMDs x 16 fp64 lanes/clk x 2.4 GHz (FMA would double the FLOP "
                    "count but change the bits)",
        }
        if valu:
            # the counters are of the whole-field launch on valu["nx"] x valu["ny"]: per cell they do not
            # depend on the grid (same strips, same chunking overheads to within a per cent)
            insts = valu["SQ_INSTS_VALU"] * local_cells / (float(valu["nx"]) * float(valu["ny"]))
            fp64_share = valu.get("fp64_share", 180.0 / 204.0)
            executed = insts * 64 * fp64_share / secs / 1e12
            roofline_valu.update(
                insts_per_launch=insts,
                insts_scaled_from_grid=None if (valu["nx"], valu["ny"]) == (dec.nx_local, dec.ny_local)
                else f"{valu['nx']}x{valu['ny']}",
                executed_fp64_tops=executed,
                executed_frac=executed / FP64_VALU_PEAK_TOPS,
                redundancy_executed_over_useful=executed / useful_tops,
                sustained_clock_ghz_under_counters=valu.get("clock_ghz"),
                # SIMD-quad-cycles of the launch = GRBM_GUI_ACTIVE / 8 XCDs / 4 x 1024 SIMDs; ACTIVE_INST_VALU counts quad-cycles
                valu_busy_frac_under_counters=(valu["SQ_ACTIVE_INST_VALU"] / (valu["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0))
                if valu.get("SQ_ACTIVE_INST_VALU") and valu.get("GRBM_GUI_ACTIVE") else None,
                source=f"profiles/sq_valu.json ({valu.get('kernel')}, {valu.get('nx')}x{valu.get('ny')}): SQ_INSTS_VALU "
                       f"per launch, {fp64_share:.3f} of them fp64 add/mul/fma (14 per cell: E - 2c, N - 2c are one exact fma "
                       f"each; rest: DPP lane shifts, 2 screening compares per row), clock = "
                       f"GRBM_GUI_ACTIVE / 8 / kernel time in that PMC run")
        n_launch_total = args.steps / T
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
                            f"v=({PHYS['vx']},{PHYS['vy']}) dt={dt} dx=dy=1, bc={args.bc} "
                            f"(left/right/bottom/top: d=Dirichlet(0) n=Neumann p=Periodic), "
                            f"decomp {dec.dims[0]}x{dec.dims[1]} (local {dec.nx_local}x{dec.ny_local}), "
                            f"halo overlap {'off' if args.no_overlap else 'on'}"
                            + (", contracted arithmetic (NOT bit-identical; opt-in)" if args.contract else "")
                            + (" — TEST MODE: one rank linked to itself in all 8 directions" if self_torus else ""),
                "halo_transport": halo,
                "exchange_schedules_ms_per_step": exchange_modes,
                "hbm_gbs_whole_job": (traffic * n_launch_total * world / elapsed / 1e9) if traffic else None,
                "hbm_gbs_whole_job_is": "PMC bytes per launch x launches of the timed region (x ranks) / wall time: "
                                        "real HBM traffic per second of the whole job",
                "step_equivalent_gbs_whole_job": cells * args.steps * BYTES_PER_CELL / elapsed / 1e9,
                "field_min_max_after_run": [mn, mx],
                "relative_mass_drift": mass_drift,
                "untimed_clock_ramp_steps": ramp_steps,
                "rows_per_chunk": tuned_rows,
                "rows_per_chunk_last_launch": last_rows,
                "per_rank": per_rank,
                "scaling_note": None if world == 1 else "N > 1 over real xGMI was never timed by the builder "
                                                        "(one-GPU lease): this line is the first measurement",
            },
            "roofline": roofline,
            "roofline_valu": roofline_valu,
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()

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
`config.repeats_ms_per_step`: the timed region is repeated `--repeats` times (default 3) back to back after ONE
  ramp; `value` / `ms_per_step` are the MEDIAN region (SURVEY §8d / BASELINE.md §4: median of >= 3 repeats).
`config.parity_preflight`: before anything is timed the run checks ITSELF: (a) N > 1: every golden case of the
  reference's own `mpirun -np N` runs (tests/golden/run_*.npz, data only) goes through the same RCCL communicator
  under every exchange schedule about to be timed and each rank compares its tile bit for bit; (b) every N: 40
  steps of the bench workload from the device-made hotspot, whose position-weighted 64-bit checksum (summed over
  the ranks) must equal the value the ORACLE computed for the same field (tests/golden/bench_checksum.json) — the
  same number at 1, 2, 4 and 8 GPUs.  A schedule that fails either check is not timed.
Stalls (N > 1): the conservative schedule (exchange not overlapped) is checked and timed FIRST; every later phase
  runs under a watchdog that, if nothing finishes in time, prints the line built from what was already measured
  (`config.stalled_schedule`) and leaves with status 3.
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


def diffusion_only():
    """vx = vy = 0: the multi-step sweep runs its seven-operation flavour (csim.h, option "fused_2c")"""
    return PHYS["vx"] == 0.0 and PHYS["vy"] == 0.0


def lookup_traffic(nx, ny, T, bc, rows=0):
    """PMC HBM bytes per launch of the sweep instantiation that was timed: same T, same local grid.
    Falls back to the per-cell figure of another grid of the same T (flagged) — the traffic per cell
    of a streaming sweep does not depend on the grid once it is far beyond the 256 MiB Infinity Cache."""
    f = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        entries = json.load(open(f)).get("entries", [])
    except Exception:
        return None, "profiles/pmc_traffic.json missing"
    # the diffusion-only flavour is its own instantiation, k_sweepO_dpp<DIV, T, 2, 2>
    still = diffusion_only() and T >= 2
    same_t = [e for e in entries if e.get("steps_per_launch") == T and (", 2, 2>" in e.get("kernel", "")) == still]
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


GOLDEN = os.path.join(ROOT, "tests", "golden")
CHECK_STEPS = (1, 7, 32)   # the checksum run: a single step, one fused pass, several passes = 40 steps


class Watchdog:
    """Deadline per phase, enforced by a thread: the main thread may be blocked inside a native call (a stream
    that never drains, a collective a dead peer never joins) where no Python-level timeout can reach it.  On expiry —
    or on SIGTERM, which the launcher sends to the surviving ranks when one rank has left — rank 0 prints the line
    built from what has been measured so far and every rank leaves through os._exit (no re-exec, no teardown of a
    process whose GPU queues are stuck)."""

    def __init__(self, rank, emit):
        import signal
        import threading
        self.rank, self.emit = rank, emit
        self.deadline, self.phase = None, "start"
        self.lock = threading.Lock()
        self.signal = signal
        try:  # SIGTERM is taken by the watchdog thread (the main thread may sit in C code for ever)
            signal.pthread_sigmask(signal.SIG_BLOCK, {signal.SIGTERM})
            self.sigs = {signal.SIGTERM}
        except Exception:  # noqa: BLE001
            self.sigs = set()
        self.thread = threading.Thread(target=self._loop, name="bench-watchdog", daemon=True)
        self.thread.start()

    def arm(self, seconds, phase):
        with self.lock:
            # the other ranks give rank 0 a head start: its line must be out before the launcher reacts to an exit
            self.deadline = time.monotonic() + seconds + (0.0 if self.rank == 0 else 8.0)
            self.phase = phase

    def disarm(self):
        with self.lock:
            self.deadline = None

    def _loop(self):
        while True:
            got = None
            if self.sigs:
                try:
                    got = self.signal.sigtimedwait(self.sigs, 0.25)
                except Exception:  # noqa: BLE001
                    time.sleep(0.25)
            else:
                time.sleep(0.25)
            with self.lock:
                expired = self.deadline is not None and time.monotonic() > self.deadline
                phase = self.phase
            if got is not None or expired:
                why = f"no progress within the deadline of phase '{phase}'" if expired else f"SIGTERM during phase '{phase}'"
                sys.stderr.write(f"[bench] rank {self.rank}: WATCHDOG: {why}\n")
                try:
                    self.emit(phase, why)
                finally:
                    os._exit(3)


def golden_cases(world):
    import glob
    import numpy as np
    out = []
    for path in sorted(glob.glob(os.path.join(GOLDEN, "run_*.npz"))):
        z = np.load(path, allow_pickle=False)
        m = json.loads(str(z["meta"]))
        if world in m["ranks"]:
            out.append((os.path.basename(path)[:-4], z, m))
    return out


def expected_checksum(nx, ny, bc, dt):
    """the oracle's checksum of the bench field after the CHECK_STEPS run (tools/make_bench_checksum.py)"""
    try:
        for e in json.load(open(os.path.join(GOLDEN, "bench_checksum.json")))["entries"]:
            if (e["nx"], e["ny"], e["bc"], e["steps"]) == (nx, ny, bc, sum(CHECK_STEPS)) and \
                    (e["D"], e["vx"], e["vy"], e["dt"]) == (PHYS["D"], PHYS["vx"], PHYS["vy"], dt):
                return e
    except Exception:  # noqa: BLE001
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--repeats", type=int, default=3,
                    help="timed regions of --steps steps each, back to back after one ramp; value = the median one")
    ap.add_argument("--ramp-seconds", type=float, default=0.3,
                    help="untimed stepping before the warm-up steps so that the GPU has left its idle "
                         "clocks (a cold MI355X runs its first ~30 ms about 15 %% below the sustained rate)")
    ap.add_argument("--nx", type=int, default=NX)
    ap.add_argument("--ny", type=int, default=NY)
    ap.add_argument("--bc", default=BC, help="boundary mix left/right/bottom/top, e.g. dddd (default), nnnn, dnpd")
    ap.add_argument("--physics", default=None,
                    help="D,dt,vx,vy (the order of csim_stepper_run, as in tools/torus_bench.py) instead of the headline workload's "
                         "0.05,0.1,0.5,0.25 — e.g. 1.0,0.1,0,0 with --nx 4096 --ny 4096 "
                         "--bc pppp = BASELINE configs[1] (diffusion only: the sweep's seven-operation flavour, HBM-bound)")
    ap.add_argument("--contract", type=int, default=0,
                    help="0 (default): the reference's own operation order, bit-identical results; 1: opt-in "
                         "contracted arithmetic (see csim.h option \"contract\"), within 1e-10 of the reference")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-preflight", action="store_true", help="skip the parity preflight (experiments only)")
    ap.add_argument("--no-safety-net", action="store_true",
                    help="N > 1: skip the host-staged timed region that precedes the RCCL communicator (see the code)")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--rows-per-chunk", type=int, default=0)
    ap.add_argument("--prefetch", type=int, default=0)
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--overlap-mode", type=int, default=-1,
                    help="N > 1: -1 pick the fastest exchange schedule on this node; 0, 1, 3, 4, 5 force one")
    ap.add_argument("--lds-bytes", type=int, default=0, help="occupancy limiter experiment (see csim.h)")
    ap.add_argument("--fused-2c", type=int, default=-1, help="0/1: E - 2c as one fma under the overflow guard (bit-identical for every non-NaN cell either way; default: on)")
    ap.add_argument("--tail-split", type=int, default=-1, help="0/1: half-height chunks at the end of a whole-field launch (default: on)")
    ap.add_argument("--fuse", type=int, default=-1,
                    help="time steps per HBM pass: -1 auto (cheapest split of the run into passes of 2..7 steps), 0 off, 2..7")
    ap.add_argument("--phase-timeout", type=float, default=float(os.environ.get("CSIM_BENCH_PHASE_TIMEOUT", 90.0)),
                    help="watchdog: seconds a phase (one schedule's preflight or timed region) may take beyond its expected time")
    args = ap.parse_args()
    assert len(args.bc) == 4 and set(args.bc) <= set("dnp"), "--bc takes four of d/n/p"
    if args.physics:
        D_, dt_, vx_, vy_ = (float(v) for v in args.physics.split(","))
        PHYS.update(D=D_, vx=vx_, vy=vy_, dt=dt_)
    assert args.repeats >= 1

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test mode: ONE rank whose four neighbours are the rank itself, so that the whole N > 1 code path
    # (RCCL communicator, deep faces in 8 directions, schedule selection) runs on a single GPU
    self_torus = world == 1 and os.environ.get("CSIM_BENCH_SELF_TORUS") == "1"
    multi = world > 1 or self_torus
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    inject_stall = os.environ.get("CSIM_BENCH_INJECT_STALL")   # test knob: the schedule of that number never returns

    # stdout carries ONE JSON line (rank 0) and nothing else: the native libraries print banners straight to file
    # descriptor 1 (RCCL's version block, gloo's connection note), so descriptor 1 points at stderr until the line
    # is written
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    # everything the line is built from; the watchdog reads it from its own thread
    S = dict(args=args, rank=rank, world=world, self_torus=self_torus, multi=multi, cpu=None, dec=None, dt=None,
             halo="none", measurements={}, chosen=None, final=[], preflight=None, stalled=None, exchange_modes=None,
             mass_drift=None, minmax=None, ramp_steps=0, tuned_rows=None)
    printed = [False]

    def emit(phase=None, why=None):
        """print THE line (once).  Called at the end of a complete run, or by the watchdog with what exists so far."""
        if printed[0] or rank != 0:
            return
        if phase is not None:
            S["stalled"] = dict(phase=phase, why=why)
        line = build_line(S)
        if line is None:
            sys.stderr.write("[bench] nothing measured yet: no line\n")
            return
        printed[0] = True
        os.dup2(stdout_fd, 1)
        os.write(1, (json.dumps(line) + "\n").encode())

    # CPU baseline first: it forks mpirun, which must happen before this process touches the GPU (and before the
    # watchdog blocks SIGTERM for every thread created from here on)
    if world == 1 and not self_torus and not args.no_cpu_baseline:
        S["cpu"] = cpu_baseline()

    wd = Watchdog(rank, emit)
    # fixed allowances of the phases (process start, communicator, first launches: generous on purpose); tests shrink them
    slack = float(os.environ.get("CSIM_BENCH_DEADLINE_SCALE", "1.0"))

    import numpy as np
    import torch  # noqa: F401  (plumbing: torch.distributed control plane; also pins ONE HIP runtime)
    import torch.distributed as dist
    from __graft_entry__ import load_package
    csim = load_package()
    csim.lib()
    ndev = csim.device_count()
    if local_rank >= ndev and os.environ.get("CSIM_BENCH_HALO") != "gloo":
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible (one rank per GPU)")
    csim.set_device(local_rank % max(ndev, 1))

    wd.arm(180 * slack + args.phase_timeout, "rendezvous and communicator")
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    dec = csim.decomp_init(world, rank, args.nx, args.ny)
    if self_torus:
        for k in range(4):
            dec.nbr[k] = 0
    S["dec"] = dec
    st = csim.Stepper(dec, 1.0, 1.0, csim.bc_codes(args.bc), 0.0)
    halo = "rccl" if multi else "none"
    dt = min(PHYS["dt"], csim.safe_dt(1.0, 1.0, PHYS["vx"], PHYS["vy"], PHYS["D"]))
    S["dt"] = dt
    if world > 1 and os.environ.get("CSIM_BENCH_HALO", "rccl") != "gloo" and not args.no_safety_net:
        # SAFETY NET, before the RCCL communicator even exists: one timed region of the same K steps with the faces
        # staged through the host over the gloo control plane (what the fall-back transport does).  It is several times
        # slower than the RCCL path and is never the reported value of a run that completes — but if building the
        # communicator, or the very first exchange over it, never returns, the watchdog has THIS to print
        # (`config.value_is`, `config.halo_transport` say so) instead of nothing.
        from climate_sim_mpi_cpp_amd.host_transport import advance as advance_external
        wd.arm(60 * slack + args.phase_timeout, "safety net: host-staged faces over gloo")
        st.set_option("external_halo", 1)
        st.init_gaussian(1.0, 0.05, 0.5, 0.5)
        nbr0 = list(dec.nbr)
        advance_external(st, nbr0, PHYS["D"], dt, PHYS["vx"], PHYS["vy"], min(args.steps, 14))
        st.sync()
        dist.barrier()
        t0 = time.perf_counter()
        advance_external(st, nbr0, PHYS["D"], dt, PHYS["vx"], PHYS["vy"], args.steps)
        st.sync()
        el = time.perf_counter() - t0
        dist.barrier()
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        S["measurements"]["gloo"] = [dict(schedule="gloo", elapsed=float(t.item()), elapsed_local=el, T=min(7, max(1, args.steps - 1)),
                                          kern_ms=0.0, launches=0, comm_ms=0.0, comm_n=0, last_rows=st.get_option("last_rows"),
                                          overlap=None)]
        st.set_option("external_halo", 0)
    if multi:
        wd.arm(180 * slack + args.phase_timeout, "RCCL communicator")
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
    S["halo"] = halo
    rccl = multi and halo == "rccl"
    for key, val in (("variant", args.variant), ("rows_per_chunk", args.rows_per_chunk),
                     ("prefetch", args.prefetch), ("fuse", args.fuse)):
        st.set_option(key, val)
    if args.contract:
        st.set_option("contract", args.contract)
    if args.lds_bytes:
        st.set_option("lds_bytes", args.lds_bytes)
    if args.tail_split >= 0:
        st.set_option("tail_split", args.tail_split)
    if args.fused_2c >= 0:
        st.set_option("fused_2c", args.fused_2c)
    nbr = list(dec.nbr)

    def advance(n, stepper=None):
        if halo.startswith("gloo"):
            from climate_sim_mpi_cpp_amd.host_transport import advance as advance_external
            advance_external(stepper or st, nbr, PHYS["D"], dt, PHYS["vx"], PHYS["vy"], n)
        else:
            (stepper or st).run(PHYS["D"], dt, PHYS["vx"], PHYS["vy"], n)

    def all_ok(flag):
        if not multi:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def global_sum():
        v = st.sum()
        if multi:
            t = torch.tensor([v], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            v = float(t.item())
        return v

    def global_checksum():
        v = st.checksum()
        if world > 1:
            parts = [None] * world
            dist.all_gather_object(parts, v)
            v = sum(parts) % (1 << 64)
        return v

    # exchange schedules of this run, the conservative one (nothing overlapped, no flag, no second launch) FIRST
    SCHED_NAMES = {0: "overlap-0 exchange not overlapped",
                   4: "overlap-4 bulk launch first hiding this pass's exchange, then the frame launch (what 5 selects)",
                   1: "overlap-1 frame launch first, next pass's exchange under the bulk launch",
                   3: "overlap-3 frame and bulk in one launch, exchange released by an in-kernel flag",
                   5: "overlap-5 default: bulk launch first hiding this pass's exchange, then the frame launch (stream relay)"}
    if not rccl:
        schedules = [None]
    elif args.no_overlap:
        schedules = [0]
    elif args.overlap_mode >= 0:
        schedules = [0, args.overlap_mode] if args.overlap_mode != 0 else [0]
    else:
        schedules = [0, 5, 1, 3]   # (4 is what 5 selects since round 3)

    def set_schedule(ov, stepper=None):
        if ov is not None:
            (stepper or st).set_option("overlap", ov)

    def maybe_stall(ov):
        if inject_stall is not None and ov is not None and str(ov) == inject_stall:
            sys.stderr.write(f"[bench] rank {rank}: CSIM_BENCH_INJECT_STALL: schedule {ov} now hangs (test knob)\n")
            while True:
                time.sleep(3600)

    # ------------------------------------------------------------------------------------------------------
    # parity preflight
    # ------------------------------------------------------------------------------------------------------
    # (over the host-staged fall-back transport the same cases run in external-halo mode: the comparison logic is the
    # same, and a two-rank run on ONE GPU — tests/test_gpu_bench.py — exercises it)
    cases = golden_cases(world) if (multi and not self_torus) else (golden_cases(1) if self_torus else [])
    cases = [c for c in cases if min(c[2]["nx"], c[2]["ny"]) >= 2]
    case_steppers = {}
    torus_ref = {}
    pre = dict(golden_cases=[c[0] for c in cases], schedules={}, ok=True,
               golden_reference=None if not cases else
                                ("the reference's own mpirun -np %d runs (tests/golden, per-rank local arrays incl. ghost "
                                 "lines)" % world) if not self_torus else
                                "schedule 0 on the same inputs (the self-linked torus is not a reference topology)",
               checksum_steps=list(CHECK_STEPS))
    exp = expected_checksum(args.nx, args.ny, args.bc, dt) if not (args.contract or self_torus) else None
    pre["checksum_expected"] = ("0x%016x" % exp["checksum"]) if exp else None
    pre["checksum_expected_from"] = (exp.get("source") if exp else
                                     "no oracle fixture for this grid / physics / topology: schedules are compared with each other")
    S["preflight"] = pre

    def preflight(ov):
        """golden cases + checksum run under schedule `ov`; returns ok (identical on every rank)"""
        name = SCHED_NAMES.get(ov, "host-staged faces over gloo" if multi else "single GPU")
        rec = dict(golden_ok=None, checksum=None, checksum_ok=None)
        pre["schedules"][name] = rec
        good = True
        st.sync()
        for cname, z, m in cases:
            key = cname
            if key not in case_steppers:
                d = csim.decomp_init(world, rank, m["nx"], m["ny"])
                if self_torus:
                    for k in range(4):
                        d.nbr[k] = 0
                ps = csim.Stepper(d, m["dx"], m["dy"], csim.bc_codes(m["bc"]))
                if rccl:
                    ps.comm_share(st)
                else:
                    ps.set_option("external_halo", 1)
                case_steppers[key] = (ps, d)
            ps, d = case_steppers[key]
            set_schedule(ov, ps)
            u = np.zeros((d.ny_local + 2, d.nx_local + 2))
            u[1:-1, 1:-1] = z["u0"][d.y_offset:d.y_offset + d.ny_local, d.x_offset:d.x_offset + d.nx_local]
            ps.upload(u)
            cdt = float(z["dt_effective"])
            if rccl:
                ps.run(m["D"], cdt, m["vx"], m["vy"], 1)                   # uneven calls: a single step, then the rest
                if m["steps"] > 1:
                    ps.run(m["D"], cdt, m["vx"], m["vy"], m["steps"] - 1)
            else:
                from climate_sim_mpi_cpp_amd.host_transport import advance as advance_external
                advance_external(ps, list(d.nbr), m["D"], cdt, m["vx"], m["vy"], m["steps"])
            ps.sync()
            got = ps.download()
            if self_torus:
                want = torus_ref.setdefault(key, got) if ov == 0 else torus_ref[key]
                same = bool(np.array_equal(got, want))
            else:
                want = z[f"local_np{world}_rank{rank}"]
                mask = np.ones(got.shape, bool)
                mask[[0, 0, -1, -1], [0, -1, 0, -1]] = False                  # corner ghosts: undefined across ranks
                same = bool(np.array_equal(got[mask], want[mask]))
            if not same:
                sys.stderr.write(f"[bench] rank {rank}: PARITY FAILURE golden case {cname} under {name}\n")
            good = good and same
        if cases:
            rec["golden_ok"] = all_ok(good)
            good = rec["golden_ok"]
        # the bench field itself: 40 steps from the device-made hotspot, checksum over all ranks
        st.init_gaussian(1.0, 0.05, 0.5, 0.5)
        rec["checksum_ic"] = "0x%016x" % global_checksum()
        for n in CHECK_STEPS:
            advance(n)
        st.sync()
        cs = global_checksum()
        rec["checksum"] = "0x%016x" % cs
        ref = exp["checksum"] if exp and not pre.get("fixture_mismatch") else pre.get("_first_checksum")
        if ref is None:                                   # first schedule and no fixture: it defines the reference
            pre["_first_checksum"] = cs
            rec["checksum_ok"] = None
        else:
            rec["checksum_ok"] = cs == ref
            if exp and ov in (0, None) and not rec["checksum_ok"]:
                # the conservative run disagrees with the fixture (another exp()? another grid?): say so and compare
                # the schedules with each other instead, so that the run still ends with a number
                pre["fixture_mismatch"] = True
                pre["_first_checksum"] = cs
                sys.stderr.write(f"[bench] WARNING: checksum {cs:#018x} differs from the oracle fixture {ref:#018x}\n")
            elif not rec["checksum_ok"]:
                sys.stderr.write(f"[bench] rank {rank}: PARITY FAILURE checksum under {name}: {cs:#018x} != {ref:#018x}\n")
                good = False
        if "_first_checksum" not in pre:
            pre["_first_checksum"] = cs
        rec["ok"] = good
        return good

    # ------------------------------------------------------------------------------------------------------
    # timing
    # ------------------------------------------------------------------------------------------------------
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

    def measure(ov, repeats=1):
        """`repeats` timed regions of exactly --steps steps under schedule ov, each bracketed by barrier + sync on both
        sides, MAX over ranks; returns the list of region records"""
        set_schedule(ov)
        advance(min(args.steps, 24))          # the schedule's own first launches (code loading) stay outside
        out = []
        for _ in range(repeats):
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
            kinds = {T: st.kernel_time(T) for T in (1, 2, 3, 4, 5, 6, 7)}
            T = max(kinds, key=lambda q: q * kinds[q][1])
            kern_ms, launches = kinds[T]
            comm_ms, comm_n = st.comm_time()
            rec = dict(schedule=ov, elapsed=elapsed, elapsed_local=elapsed_local, T=T, kern_ms=kern_ms, launches=launches,
                       comm_ms=comm_ms, comm_n=comm_n, last_rows=st.get_option("last_rows"),
                       overlap=st.get_option("overlap"))
            if multi:
                km = torch.tensor([kern_ms], dtype=torch.float64)
                dist.all_reduce(km, op=dist.ReduceOp.MAX)
                rec["kern_ms_max"] = float(km.item())
                mine = dict(rank=rank, coords=[dec.coords[0], dec.coords[1]], local=[dec.nx_local, dec.ny_local],
                            neighbours=list(dec.nbr), wall_ms_per_step=elapsed_local / args.steps * 1e3,
                            kernel=kernel_label(T), kernel_avg_ms=kern_ms / max(launches, 1), launches_timed=launches,
                            exchange_chain_avg_ms=(comm_ms / comm_n) if comm_n else None, exchange_chains_timed=comm_n,
                            rows_per_chunk=rec["last_rows"], overlap=rec["overlap"])
                per_rank = [None] * world
                dist.all_gather_object(per_rank, mine)
                rec["per_rank"] = per_rank
            out.append(rec)
        return out

    def expected_seconds(repeats=1):
        # a generous model of one phase: steps at 0.5 ms each (the slowest schedule at the smallest N) + fixed costs
        return 10.0 * slack + repeats * (args.steps + 64) * 5e-4 * max(1.0, args.nx * args.ny / (NX * NY))

    # untimed: preflight of the conservative schedule, the W warm-up steps (cold: first launches of the kernels, code
    # loading), then the clock ramp in bursts shaped like the timed run.  The ramp comes LAST so that nothing but the
    # synchronisation stands between steady-state load and the first timed region: a first launch of a new kernel kind
    # stalls the host for ~2 ms (code loading), the idle GPU drops out of its sustained power state, and the next few
    # launches then run 5-15 % slower than in steady state (kernel timelines: tools/gpu_trace_steps20.sh)
    first = schedules[0]
    wd.arm(120 * slack + args.phase_timeout, f"parity preflight of {SCHED_NAMES.get(first, 'the single-GPU path')}")
    set_schedule(first)
    if not args.no_preflight:
        ok0 = preflight(first)
        if not ok0:
            pre["ok"] = False
            sys.stderr.write("[bench] WARNING: the conservative path failed its parity preflight; timing it anyway, "
                             "the line says so (config.parity_preflight.ok = false)\n")
    st.init_gaussian(1.0, 0.05, 0.5, 0.5)
    # FTCS diffusion + upwind advection conserve the total of u up to the flux through the physical edges (nil here:
    # the hotspot stays far from them), so a face lost or misplaced later would show up as mass leaking at the seams
    mass0 = global_sum()
    wd.arm(120 * slack + args.phase_timeout + expected_seconds(3), "warm-up and clock ramp")
    st.tune(PHYS["D"], dt, PHYS["vx"], PHYS["vy"])  # the one-off chunk-height trial a first long run() would do (local, no exchange)
    advance(args.warmup)
    burst = max(1, min(args.steps, 60))
    t_ramp = time.perf_counter()
    while args.ramp_seconds > 0:
        advance(burst)
        st.sync()
        S["ramp_steps"] += burst
        done = time.perf_counter() - t_ramp >= args.ramp_seconds
        if multi:  # every rank must take the same number of steps: decide together
            t = torch.tensor([1 if done else 0], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            done = bool(t.item())
        if done:
            break
    S["tuned_rows"] = st.get_option("tuned_rows") or (args.rows_per_chunk or "heuristic")
    S["tuned_rows_by_depth"] = {str(T): st.get_option(f"tuned_rows_{T}") for T in (4, 5, 6, 7)}

    if len(schedules) == 1:
        wd.arm(args.phase_timeout + expected_seconds(args.repeats), "timed regions")
        S["chosen"] = first
        S["final"] = measure(first, args.repeats)
        S["measurements"][first] = S["final"]
    else:
        # N > 1 over RCCL: how the exchange is best hidden depends on what the RCCL kernel costs next to the sweep on
        # this node, which cannot be known beforehand.  Every schedule is first checked (preflight), then timed by the
        # full protocol once; the conservative one came first, so from here on a number exists whatever happens.
        S["exchange_modes"] = {}
        for ov in schedules:
            name = SCHED_NAMES[ov]
            if ov != first:
                if ov == 3:
                    try:
                        st.set_option("overlap", 3)
                    except Exception:  # noqa: BLE001  (no hipStreamWaitValue64 / signal memory on this device)
                        S["exchange_modes"][name] = "not offered by this device (no signal memory)"
                        continue
                wd.arm(args.phase_timeout + expected_seconds(), f"parity preflight of {name}")
                maybe_stall(ov)
                if not args.no_preflight:
                    set_schedule(ov)
                    if not preflight(ov):
                        S["exchange_modes"][name] = "EXCLUDED: failed the parity preflight"
                        pre["ok"] = False
                        continue
                    st.init_gaussian(1.0, 0.05, 0.5, 0.5)
                    advance(burst)
            wd.arm(args.phase_timeout + expected_seconds(), f"timed trial of {name}")
            S["measurements"][ov] = measure(ov)
            S["exchange_modes"][name] = S["measurements"][ov][0]["elapsed"] / args.steps * 1e3
            if S["chosen"] is None:
                S["chosen"] = ov
        timed = {ov: m[0]["elapsed"] for ov, m in S["measurements"].items() if ov != "gloo"}
        if "gloo" in S["measurements"]:
            S["exchange_modes"]["safety net: host-staged faces over gloo, timed before the communicator existed"] = \
                S["measurements"]["gloo"][0]["elapsed"] / args.steps * 1e3
        default = 5 if 5 in timed else (args.overlap_mode if args.overlap_mode in timed else first)
        best = min(timed, key=timed.get)
        if timed[best] > 0.98 * timed[default]:
            best = default  # within noise of the default schedule: keep the default
        S["chosen"] = best
        S["exchange_modes"]["chosen"] = SCHED_NAMES[best]
        wd.arm(args.phase_timeout + expected_seconds(args.repeats), f"timed regions of {SCHED_NAMES[best]}")
        set_schedule(best)
        advance(burst)
        S["ramp_steps"] += burst
        S["final"] = measure(best, args.repeats)

    wd.arm(args.phase_timeout + 30 * slack, "closing checks")
    S["minmax"] = st.minmax()
    mass1 = global_sum()
    S["mass_drift"] = abs(mass1 - mass0) / abs(mass0)
    if S["mass_drift"] > 1e-9 and rank == 0:
        sys.stderr.write(f"[bench] WARNING: total mass drifted by {S['mass_drift']:.3e} (halo exchange broken?)\n")
    for ps, _ in case_steppers.values():
        ps.close()
    st.close()
    if multi:
        dist.destroy_process_group()
    wd.disarm()
    emit()


def build_line(S):
    """the ONE JSON line from whatever has been measured (the complete run, or what exists when the watchdog fires)"""
    args, dec, world, dt = S["args"], S["dec"], S["world"], S["dt"]
    import statistics
    stalled = S["stalled"]
    regions = S["final"]
    source = "final"
    if not regions:
        # stalled before the final regions: the best COMPLETED trial (the conservative schedule was timed first)
        done = {ov: m for ov, m in S["measurements"].items() if m}
        if not done:
            return None
        ov = min(done, key=lambda k: done[k][0]["elapsed"])
        regions, source = done[ov], "trial"
    ms_list = [r["elapsed"] / args.steps * 1e3 for r in regions]
    med_ms = statistics.median(ms_list)
    rep = min(regions, key=lambda r: abs(r["elapsed"] / args.steps * 1e3 - med_ms))   # the median region's own records
    elapsed = med_ms * args.steps * 1e-3
    multi = S["multi"]
    T = rep["T"]
    # kernel time: all regions' brackets of the dominant kind together
    same_kind = [r for r in regions if r["T"] == T]
    kern_ms = sum(r.get("kern_ms_max", r["kern_ms"]) for r in same_kind)
    launches = sum(r["launches"] for r in same_kind)
    kern_avg_ms = kern_ms / max(launches, 1)
    last_rows = rep["last_rows"]

    cells = float(args.nx) * float(args.ny)
    value = cells * args.steps / elapsed / 1e6
    local_cells = float(dec.nx_local) * float(dec.ny_local)
    if launches == 0:  # the safety-net region has no kernel brackets: wall time per pass stands in (an upper bound)
        kern_avg_ms = med_ms * T
    secs = max(kern_avg_ms, 1e-9) * 1e-3
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
        ach_bytes, ach_src = traffic, "traffic (stored PMC profile)"
    else:
        ach_bytes, ach_src = alg_bytes, "algorithmic_bytes_per_launch (no PMC entry: a LOWER bound of the real traffic)"
    ach = ach_bytes / secs / 1e9
    still = diffusion_only() and T >= 2 and not args.contract
    valu_binds = T >= 4 and not args.contract and not still   # half the arithmetic: HBM binds
    roofline = {
        # this object prices the HBM side of the dominant kernel (the contract's schema); `is_binding` says whether
        # HBM is what limits it — at T >= 4 it is not: fp64 VALU issue is (roofline_valu)
        "bound": "hbm",
        "is_binding": not valu_binds,
        "binding_object": "roofline_valu" if valu_binds else "roofline",
        "achieved": ach,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": ach / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_is": "stored PMC profile (profiles/pmc_traffic.json: rocprofv3 --pmc on this kernel instantiation, grid and "
                      "chunk height), NOT collected in this run; only the kernel time is live",
        "achieved_from": ach_src,
        "traffic_source": traffic_src,
        "frac_of_measured_copy_peak": ach / HBM_COPY_GBS,
        "kernel": kernel_label(T) + (f" (fused copy+diffusion, advection term of zero velocity left out, {T} time step(s) per HBM pass)"
                                     if still else f" (fused copy+diffusion+advection, {T} time step(s) per HBM pass)"),
        "kernel_avg_ms": kern_avg_ms,
        "kernel_avg_ms_is": "HIP events on the compute stream around each RUN of equal launches / launches in it, all timed "
                            "regions together (N = 1: includes the ~5 us ghost fills between launches where a side is "
                            "Neumann; rocprofv3 --kernel-trace of the same command: profiles/)",
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
    ops_per_update = 5 if args.contract else 8 if still else FP64_OPS_PER_UPDATE   # diffusion only: the 8 operations of src/diffusion.cpp:9-16
    useful_tops = local_cells * T * ops_per_update / secs / 1e12
    valu = lookup_valu(T) if not (args.contract or still) else None
    roofline_valu = {
        "bound": "fp64-valu",
        "is_binding": valu_binds,
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
            counters_are="stored SQ profile (profiles/sq_valu.json), not collected in this run",
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
    pre = S["preflight"]
    if pre is not None:
        pre = {k: v for k, v in pre.items() if not k.startswith("_")}
        if args.no_preflight:
            pre = dict(skipped="--no-preflight")
    mn_mx = list(S["minmax"]) if S["minmax"] else None
    line = {
        "metric": "Mcell-updates/sec (16384^2 fp64 advection-diffusion sweep)",
        "value": value,
        "unit": "Mcell-updates/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": med_ms,
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
                        + (" — TEST MODE: one rank linked to itself in all 8 directions" if S["self_torus"] else ""),
            "repeats": len(regions),
            "repeats_ms_per_step": ms_list,
            "value_is": ("median of the timed regions (each: barrier + sync, exactly --steps steps, sync + barrier, max over "
                         "ranks)" if source == "final" else
                         "the best COMPLETED schedule trial (one full-protocol timed region): the run stalled before its final regions"),
            "halo_transport": S["halo"] if rep.get("schedule") != "gloo" else
                              "gloo (host-staged): the SAFETY-NET region timed before the RCCL communicator existed — the RCCL path never completed a region",
            "exchange_schedule": rep.get("schedule"),
            "exchange_schedules_ms_per_step": S["exchange_modes"],
            "parity_preflight": pre,
            "stalled_schedule": stalled,
            "hbm_gbs_whole_job": (traffic * n_launch_total * world / elapsed / 1e9) if traffic else None,
            "hbm_gbs_whole_job_is": "stored PMC bytes per launch x launches of the timed region (x ranks) / wall time: "
                                    "real HBM traffic per second of the whole job",
            "step_equivalent_gbs_whole_job": cells * args.steps * BYTES_PER_CELL / elapsed / 1e9,
            "field_min_max_after_run": mn_mx,
            "relative_mass_drift": S["mass_drift"],
            "untimed_clock_ramp_steps": S["ramp_steps"],
            "rows_per_chunk": S["tuned_rows"],
            "rows_per_chunk_by_pass_depth": S.get("tuned_rows_by_depth"),
            "rows_per_chunk_last_launch": last_rows,
            "per_rank": rep.get("per_rank"),
            "scaling_note": None if world == 1 else "N > 1 over real xGMI was never timed by the builder "
                                                    "(one-GPU lease): this line is the first measurement",
        },
        "roofline": roofline,
        "roofline_valu": roofline_valu,
    }
    if S["cpu"] is not None:
        line["cpu_baseline"] = S["cpu"]
    return line


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""tests/bench_mock_runner.py — TEST INFRASTRUCTURE: runs bench.py's real main() with a MOCK engine in place of
the HIP package, so that the multi-rank CONTROL FLOW of the benchmark — the order of the collectives every rank must
join, the parity preflight per schedule, the schedule trials, the repeats, the watchdog across processes, the one JSON
line — can run with world sizes 2 / 4 / 8 on CPUs (torch.distributed gloo), where no GPU exists.  Nothing is computed:
the mock stepper returns the golden fixtures' own arrays and a checksum that adds up to the oracle's value; timings
are sleeps.  (The measured path is tests/test_gpu_bench.py.)  Environment: MOCK_STALL_RANK / MOCK_STALL_SCHEDULE make
one rank hang inside run() under one schedule; MOCK_BAD_SCHEDULE makes one schedule return wrong tiles."""
import glob
import importlib.util
import json
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import PKG_NAME, PKG_DIR  # noqa: E402

# the real package for the pure host functions (decomposition, bc codes, safe_dt): libcsim.so loads without a GPU
spec = importlib.util.spec_from_file_location("_real_csim", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
real = importlib.util.module_from_spec(spec)
spec.loader.exec_module(real)

RANK = int(os.environ.get("RANK", "0"))
WORLD = int(os.environ.get("WORLD_SIZE", "1"))
STALL_RANK = int(os.environ.get("MOCK_STALL_RANK", "-1"))
STALL_SCHEDULE = int(os.environ.get("MOCK_STALL_SCHEDULE", "-1"))
BAD_SCHEDULE = int(os.environ.get("MOCK_BAD_SCHEDULE", "-1"))
GOLDEN = {}
for path in glob.glob(os.path.join(ROOT, "tests", "golden", "run_*.npz")):
    z = np.load(path, allow_pickle=False)
    m = json.loads(str(z["meta"]))
    GOLDEN[(m["nx"], m["ny"], m["bc"])] = z
FIXTURE = {(e["nx"], e["ny"], e["bc"]): e for e in json.load(open(os.path.join(ROOT, "tests", "golden", "bench_checksum.json")))["entries"]}
BC_LETTER = {0: "d", 1: "n", 2: "p"}


class Stepper:
    def __init__(self, dec, dx=1.0, dy=1.0, bc=(0, 0, 0, 0), bc_value=0.0):
        self.dec, self.nx, self.ny = dec, dec.nx_local, dec.ny_local
        self.bc = "".join(BC_LETTER[int(b)] for b in bc)
        self.opts = dict(overlap=5, last_rows=74, tuned_rows=74, profile=0, external_halo=0)
        self.steps = 0
        self.key = (dec.nx_global, dec.ny_global, self.bc)

    def comm_init(self, uid):
        assert uid == b"mock-unique-id"

    def comm_share(self, owner):
        assert isinstance(owner, Stepper)

    def set_option(self, key, value):
        self.opts[key] = int(value)

    def get_option(self, key):
        return self.opts.get(key, 0)

    def init_gaussian(self, *a):
        self.steps = 0

    def upload(self, host):
        self.steps = 0

    def run(self, D, dt, vx, vy, n):
        if RANK == STALL_RANK and self.opts["overlap"] == STALL_SCHEDULE:
            time.sleep(3600)            # a stream that never drains
        self.steps += n
        time.sleep(2e-5 * n)

    def download(self):
        z = GOLDEN[self.key]
        out = z[f"local_np{WORLD}_rank{RANK}"].copy()
        if self.opts["overlap"] == BAD_SCHEDULE:
            out[1, 1] += 1.0            # a schedule that computes something else
        return out

    def checksum(self):
        e = FIXTURE.get(self.key)
        if e is None or RANK != 0:
            return 0
        return e["checksum"] if self.steps else e["checksum_ic"]

    def sum(self):
        return 1.0 / WORLD

    def minmax(self):
        return 0.0, 1.0

    def kernel_time(self, T):
        return (0.16 * 3, 3) if T == 7 else (0.0, 0)

    def comm_time(self):
        return 0.1, 1

    def sync(self):
        pass

    def tune(self, *a):
        pass

    def keep_warm(self, *a):
        pass

    def reset_timers(self):
        pass

    def close(self):
        pass


mock = types.ModuleType(PKG_NAME)
mock.Stepper = Stepper
mock.lib = lambda: None
mock.device_count = lambda: max(WORLD, 1)
mock.set_device = lambda d: None
mock.decomp_init = real.decomp_init
mock.bc_codes = real.bc_codes
mock.safe_dt = real.safe_dt
mock.comm_unique_id = lambda: b"mock-unique-id"
mock.CsimError = real.CsimError
ht = types.ModuleType(PKG_NAME + ".host_transport")
ht.advance = lambda st, nbr, D, dt, vx, vy, n: st.run(D, dt, vx, vy, n)
mock.host_transport = ht
sys.modules[PKG_NAME] = mock
sys.modules[PKG_NAME + ".host_transport"] = ht

spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
bench.main()

"""oracle/cpu_oracle.py — TEST INFRASTRUCTURE (the parity checker), NOT PRODUCT.

ctypes front end of ``liboracle_cpu.so`` (oracle/cpu_stepper.c), the clean-room CPU
restatement of the reference hot path (reference src/main.cpp:101-109 and the functions it
calls).  Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may
import this module; the product path (climate-sim-mpi-cpp_amd/, include/) never does.

Arrays are numpy float64 in the reference layout: shape (ny+2, nx+2), C-contiguous, element
(i, j) at ``a[j, i]`` (reference src/field.cpp:20-25).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# CSIM_ORACLE_SO: another build of cpu_stepper.c (tools/cpu_sanitize.sh points it at an ASan/UBSan one)
LIB_PATH = os.environ.get("CSIM_ORACLE_SO") or os.path.join(HERE, "liboracle_cpu.so")
REF_RUN = os.path.join(HERE, "_ref", "ref_run")
MPIRUN = "/opt/conda/bin/mpirun"

DIRICHLET, NEUMANN, PERIODIC = 0, 1, 2
BC_CODES = {"d": DIRICHLET, "n": NEUMANN, "p": PERIODIC}

_lib = None


def build() -> None:
    """Compile the restatement (and, when /root/reference is present, oracle/_ref)."""
    subprocess.run(["make", "-s", "-C", HERE], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        L.ora_safe_dt.restype = C.c_double
        L.ora_safe_dt.argtypes = [C.c_double] * 5
        L.ora_apply_boundary.argtypes = [dp, C.c_int, C.c_int, ip, ip, C.c_double]
        L.ora_diffusion_step.argtypes = [dp, dp, C.c_int, C.c_int] + [C.c_double] * 4
        L.ora_advection_step.argtypes = [dp, dp, C.c_int, C.c_int] + [C.c_double] * 5
        L.ora_step_tile.argtypes = [dp, dp, C.c_int, C.c_int] + [C.c_double] * 6 + [ip, ip]
        L.ora_run_single.argtypes = [dp, C.c_int, C.c_int] + [C.c_double] * 6 + [ip, C.c_int]
        L.ora_dims_create.argtypes = [C.c_int, ip]
        L.ora_decomp.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, ip]
        L.ora_gaussian.argtypes = [dp] + [C.c_int] * 6 + [C.c_double] * 6
        L.ora_world_create.restype = C.c_void_p
        L.ora_world_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
        L.ora_world_destroy.argtypes = [C.c_void_p]
        L.ora_world_scatter.argtypes = [C.c_void_p, dp]
        L.ora_world_gather.argtypes = [C.c_void_p, dp]
        L.ora_world_gather_full.argtypes = [C.c_void_p, dp]
        L.ora_world_gaussian.argtypes = [C.c_void_p] + [C.c_double] * 4
        L.ora_world_tile_shape.argtypes = [C.c_void_p, C.c_int, ip]
        L.ora_world_tile_get.argtypes = [C.c_void_p, C.c_int, dp]
        L.ora_world_run.restype = C.c_double
        L.ora_world_run.argtypes = [C.c_void_p] + [C.c_double] * 4 + [ip, C.c_int, C.c_int]
        L.ora_world_run_mode.restype = C.c_double
        L.ora_world_run_mode.argtypes = [C.c_void_p] + [C.c_double] * 4 + [ip, C.c_int, C.c_int, C.c_int]
        L.ora_minmax.argtypes = [dp, C.c_size_t, dp]
        _lib = L
    return _lib


def _dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i4(v):
    return (C.c_int * 4)(*[int(x) for x in v])


def bc_codes(code: str):
    """'dnpd' -> [left, right, bottom, top] integer codes."""
    return [BC_CODES[c] for c in code.lower()]


def safe_dt(dx, dy, vx, vy, D) -> float:
    return lib().ora_safe_dt(dx, dy, vx, vy, D)


def apply_boundary(f: np.ndarray, bc, phys=(1, 1, 1, 1), value=0.0) -> None:
    ny, nx = f.shape[0] - 2, f.shape[1] - 2
    lib().ora_apply_boundary(_dp(f), nx, ny, _i4(bc), _i4(phys), value)


def diffusion_step(u, out, dx, dy, D, dt) -> None:
    ny, nx = u.shape[0] - 2, u.shape[1] - 2
    lib().ora_diffusion_step(_dp(u), _dp(out), nx, ny, dx, dy, D, dt)


def advection_step(u, out, dx, dy, vx, vy, dt) -> None:
    ny, nx = u.shape[0] - 2, u.shape[1] - 2
    lib().ora_advection_step(_dp(u), _dp(out), nx, ny, dx, dy, vx, vy, dt)


def step_tile(u, tmp, dx, dy, D, vx, vy, dt, bc, phys) -> None:
    """boundary (physical sides only) -> copy -> diffusion -> advection; result in tmp."""
    ny, nx = u.shape[0] - 2, u.shape[1] - 2
    lib().ora_step_tile(_dp(u), _dp(tmp), nx, ny, dx, dy, D, vx, vy, dt, _i4(bc), _i4(phys))


def run_single(u, dx, dy, D, vx, vy, dt, bc, steps) -> None:
    """`steps` full reference steps in place on one tile with four physical sides."""
    ny, nx = u.shape[0] - 2, u.shape[1] - 2
    lib().ora_run_single(_dp(u), nx, ny, dx, dy, D, vx, vy, dt, _i4(bc), steps)


def dims_create(size):
    d = (C.c_int * 2)()
    lib().ora_dims_create(size, d)
    return [d[0], d[1]]


def decomp(size, rank, nxg, nyg):
    o = (C.c_int * 12)()
    lib().ora_decomp(size, rank, nxg, nyg, o)
    keys = ["dims0", "dims1", "cx", "cy", "left", "right", "down", "up", "nx_local",
            "ny_local", "x_offset", "y_offset"]
    return dict(zip(keys, list(o)))


def gaussian_global(nxg, nyg, dx=1.0, dy=1.0, A=1.0, sigma_frac=0.05, xc_frac=0.5,
                    yc_frac=0.5) -> np.ndarray:
    """Global field WITH ghost ring (ghosts 0), gaussian hotspot on the interior."""
    f = np.zeros((nyg + 2, nxg + 2))
    lib().ora_gaussian(_dp(f), nxg, nyg, 0, 0, nxg, nyg, dx, dy, A, sigma_frac, xc_frac, yc_frac)
    return f


class World:
    """`size` tiles in one process = the reference under `mpirun -np size`."""

    def __init__(self, size, nxg, nyg, dx=1.0, dy=1.0):
        self.size, self.nxg, self.nyg, self.dx, self.dy = size, nxg, nyg, dx, dy
        self._w = lib().ora_world_create(size, nxg, nyg, dx, dy)

    def __del__(self):
        if getattr(self, "_w", None):
            lib().ora_world_destroy(self._w)
            self._w = None

    def scatter(self, g: np.ndarray):
        assert g.shape == (self.nyg, self.nxg)
        lib().ora_world_scatter(self._w, _dp(np.ascontiguousarray(g, dtype=np.float64)))

    def gather(self) -> np.ndarray:
        g = np.zeros((self.nyg, self.nxg))
        lib().ora_world_gather(self._w, _dp(g))
        return g

    def gather_full(self) -> np.ndarray:
        """the world as one (nyg+2, nxg+2) array incl. the ghost ring of its physical sides: the full
        local array of a 1-rank run (decomposition-invariant)."""
        g = np.zeros((self.nyg + 2, self.nxg + 2))
        lib().ora_world_gather_full(self._w, _dp(g))
        return g

    def gaussian(self, A=1.0, sigma_frac=0.05, xc_frac=0.5, yc_frac=0.5):
        lib().ora_world_gaussian(self._w, A, sigma_frac, xc_frac, yc_frac)

    def tile(self, r) -> np.ndarray:
        s = (C.c_int * 4)()
        lib().ora_world_tile_shape(self._w, r, s)
        a = np.zeros((s[1] + 2, s[0] + 2))
        lib().ora_world_tile_get(self._w, r, _dp(a))
        return a

    def run(self, D, vx, vy, dt, bc, steps, threads=1, checked=False) -> float:
        """`steps` reference steps on every tile (threads > 1: tiles spread over threads); returns the
        wall seconds.  checked=True: the bounds-checked accessor flavour (same results)."""
        return lib().ora_world_run_mode(self._w, D, vx, vy, dt, _i4(bc), steps, threads, 1 if checked else 0)


# ---- the real reference, when oracle/_ref/ref_run has been built ------------------------


def have_reference() -> bool:
    return os.path.exists(REF_RUN) and os.path.exists(MPIRUN)


def ref_run(mode: str, np_ranks: int = 1, timeout: float = 600.0, **kw) -> str:
    """Run oracle/_ref/ref_run (the compiled reference objects) under mpirun; returns stdout."""
    args = [f"--{k}={v}" for k, v in kw.items()]
    cmd = [REF_RUN, mode] + args
    if np_ranks > 1:
        cmd = [MPIRUN, "-np", str(np_ranks)] + cmd
    r = subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=timeout)
    return r.stdout

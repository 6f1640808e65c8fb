"""climate-sim-mpi-cpp_amd — thin ctypes front end of the C ABI in include/csim.h.

The product is the C-ABI shared library (csrc/ -> lib/libcsim.so: hand-written gfx950 HIP
kernels + RCCL halo exchange) and the C++17 headers in include/climate/ that mirror the
reference's own interface.  This module only exists so that tests/, bench.py and
__graft_entry__.py can reach that ABI from Python; names follow the reference
(`Field`, `Decomp2D`, `BCConfig`, `apply_boundary`, `diffusion_step`, `advection_step`,
`exchange_halos`, `safe_dt` — reference include/*.hpp).

There is no CPU fallback: if lib/libcsim.so is missing or no gfx950 device is usable the
calls raise.  (The directory name contains '-', so import it through
``__graft_entry__.load_package()``, which registers it as ``climate_sim_mpi_cpp_amd``.)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# CSIM_LIB: another build of the engine (A/B measurements of two kernel versions in tools/)
LIB_PATH = os.environ.get("CSIM_LIB") or os.path.join(HERE, "lib", "libcsim.so")
HEADER = os.path.join(ROOT, "include", "csim.h")

DIRICHLET, NEUMANN, PERIODIC = 0, 1, 2
LEFT, RIGHT, BOTTOM, TOP = 0, 1, 2, 3
NO_NEIGHBOR = -1
UNIQUE_ID_BYTES = 128
VARIANTS = {"auto": 0, "dpp": 1, "lds": 2, "naive": 3}

_BC_NAMES = {  # reference src/io.cpp:35-44 bc_from_string aliases
    "dirichlet": DIRICHLET, "fixed": DIRICHLET,
    "neumann": NEUMANN, "noflux": NEUMANN, "zero-flux": NEUMANN,
    "periodic": PERIODIC, "period": PERIODIC,
}


class CsimError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"csim error {code}: {msg}")
        self.code = code


class Decomp(C.Structure):
    """struct csim_decomp == reference Decomp2D without the communicator."""
    _fields_ = [("size", C.c_int), ("rank", C.c_int), ("dims", C.c_int * 2),
                ("coords", C.c_int * 2), ("nbr", C.c_int * 4),
                ("nx_global", C.c_int), ("ny_global", C.c_int),
                ("nx_local", C.c_int), ("ny_local", C.c_int),
                ("x_offset", C.c_int), ("y_offset", C.c_int)]

    def as_dict(self):
        return dict(dims0=self.dims[0], dims1=self.dims[1], cx=self.coords[0], cy=self.coords[1],
                    left=self.nbr[0], right=self.nbr[1], down=self.nbr[2], up=self.nbr[3],
                    nx_local=self.nx_local, ny_local=self.ny_local, x_offset=self.x_offset,
                    y_offset=self.y_offset)


class Msg(C.Structure):
    """struct csim_msg: one message of a halo exchange (peer rank, direction 0..7, doubles)."""
    _fields_ = [("peer", C.c_int), ("dir", C.c_int), ("count", C.c_long)]


def build(force: bool = False) -> str:
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        args = ["make", "-s", "-C", os.path.join(HERE, "csrc")]
        if force:
            args.append("-B")
        subprocess.run(args, check=True)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """Load lib/libcsim.so and declare every prototype of include/csim.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: run __graft_entry__.build() (there is no CPU fallback)")
    if os.environ.get("CSIM_PRELOAD_TORCH", "1") != "0":
        # torch bundles its own ROCm runtime (libamdhip64.so.7 / librccl.so.1); importing it first
        # makes libcsim.so resolve to that same copy, so a process never holds two HIP runtimes.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    d, i = C.c_double, C.c_int
    sig = {
        "csim_last_error": (C.c_char_p, []),
        "csim_abi_version": (i, []),
        "csim_device_count": (i, [ip]),
        "csim_set_device": (i, [i]),
        "csim_device_name": (i, [C.c_char_p, C.c_size_t]),
        "csim_safe_dt": (d, [d] * 5),
        "csim_decomp_init": (i, [i, i, i, i, C.POINTER(Decomp)]),
        "csim_exchange_plan": (i, [C.POINTER(Decomp), i, C.POINTER(Msg), ip, C.POINTER(Msg), ip]),
        "csim_field_create": (i, [i, i, i, d, d, C.POINTER(vp)]),
        "csim_field_destroy": (i, [vp]),
        "csim_field_upload": (i, [vp, dp]),
        "csim_field_download": (i, [vp, dp]),
        "csim_field_download_interior": (i, [vp, dp]),
        "csim_field_fill": (i, [vp, d]),
        "csim_field_copy": (i, [vp, vp]),
        "csim_field_swap": (i, [vp, vp]),
        "csim_field_minmax": (i, [vp, dp]),
        "csim_field_sum": (i, [vp, dp]),
        "csim_field_linf_diff": (i, [vp, vp, dp]),
        "csim_apply_boundary": (i, [vp, ip, ip, d]),
        "csim_diffusion_step": (i, [vp, vp, d, d]),
        "csim_advection_step": (i, [vp, vp, d, d, d]),
        "csim_fused_step": (i, [vp, vp, d, d, d, d]),
        "csim_stepper_create": (i, [C.POINTER(Decomp), d, d, ip, d, C.POINTER(vp)]),
        "csim_stepper_destroy": (i, [vp]),
        "csim_comm_unique_id": (i, [vp, C.c_size_t]),
        "csim_stepper_comm_init": (i, [vp, vp, C.c_size_t]),
        "csim_stepper_upload": (i, [vp, dp]),
        "csim_stepper_download": (i, [vp, dp]),
        "csim_stepper_download_interior": (i, [vp, dp]),
        "csim_stepper_snapshot_begin": (i, [vp]),
        "csim_stepper_snapshot_wait": (i, [vp, C.POINTER(dp)]),
        "csim_stepper_init_gaussian": (i, [vp, d, d, d, d]),
        "csim_stepper_exchange_halos": (i, [vp]),
        "csim_stepper_halo_pack": (i, [vp, C.POINTER(dp)]),
        "csim_stepper_halo_unpack": (i, [vp, C.POINTER(dp)]),
        "csim_stepper_fuse_limit": (i, [vp, ip]),
        "csim_stepper_faces_neighbors": (i, [vp, i, ip, ip]),
        "csim_stepper_faces_pack": (i, [vp, i, C.POINTER(dp)]),
        "csim_stepper_faces_unpack": (i, [vp, i, C.POINTER(dp)]),
        "csim_stepper_run": (i, [vp, d, d, d, d, i]),
        "csim_stepper_tune": (i, [vp, d, d, d, d]),
        "csim_stepper_keep_warm": (i, [vp, d, d, d, d, d]),
        "csim_pass_schedule": (i, [i, i, C.c_long, i, ip, i, C.POINTER(C.c_long)]),
        "csim_pass_schedule_for": (i, [i, i, C.c_long, i, i, ip, i, C.POINTER(C.c_long)]),
        "csim_stepper_sync": (i, [vp]),
        "csim_stepper_checksum": (i, [vp, C.POINTER(C.c_ulonglong)]),
        "csim_stepper_comm_share": (i, [vp, vp]),
        "csim_stepper_minmax": (i, [vp, dp]),
        "csim_stepper_sum": (i, [vp, dp]),
        "csim_stepper_set_option": (i, [vp, C.c_char_p, C.c_long]),
        "csim_stepper_get_option": (i, [vp, C.c_char_p, C.POINTER(C.c_long)]),
        "csim_stepper_kernel_time": (i, [vp, i, dp, C.POINTER(C.c_long)]),
        "csim_stepper_comm_time": (i, [vp, dp, C.POINTER(C.c_long)]),
        "csim_stepper_reset_timers": (i, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def declared_symbols():
    """Every function name declared in include/csim.h (parsed from the header text)."""
    import re
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"\b(csim_[a-z0-9_]+)\s*\(", txt)))


def _ck(rc):
    if rc != 0:
        raise CsimError(rc, lib().csim_last_error().decode())


def _dp(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i4(v):
    return (C.c_int * 4)(*[int(x) for x in v])


def bc_from_string(s: str) -> int:
    try:
        return _BC_NAMES[s.lower()]
    except KeyError:
        raise RuntimeError("Unknown BC type: " + s)


def bc_codes(code: str):
    """'dnpd' -> [left, right, bottom, top]."""
    m = {"d": DIRICHLET, "n": NEUMANN, "p": PERIODIC}
    return [m[c] for c in code.lower()]


def device_count() -> int:
    n = C.c_int(0)
    _ck(lib().csim_device_count(C.byref(n)))
    return n.value


def set_device(dev: int) -> None:
    _ck(lib().csim_set_device(dev))


def device_name() -> str:
    buf = C.create_string_buffer(256)
    _ck(lib().csim_device_name(buf, 256))
    return buf.value.decode()


def safe_dt(dx, dy, vx, vy, D) -> float:
    return lib().csim_safe_dt(dx, dy, vx, vy, D)


def decomp_init(size, rank, nx_global, ny_global) -> Decomp:
    d = Decomp()
    _ck(lib().csim_decomp_init(size, rank, nx_global, ny_global, C.byref(d)))
    return d


def exchange_plan(dec: Decomp, depth: int):
    """(sends, recvs) of one halo exchange of this rank, each an ordered list of (peer, dir, count)."""
    sends, recvs = (Msg * 8)(), (Msg * 8)()
    ns, nr = C.c_int(0), C.c_int(0)
    _ck(lib().csim_exchange_plan(C.byref(dec), depth, sends, C.byref(ns), recvs, C.byref(nr)))
    return ([(m.peer, m.dir, m.count) for m in sends[:ns.value]],
            [(m.peer, m.dir, m.count) for m in recvs[:nr.value]])


def pass_schedule(nsteps: int, smallest_tile: int = 1 << 30, fuse: int = -1, tile_cells: int = 0, diffusion_only: bool = False):
    """time steps per HBM pass of a run of nsteps (csim_pass_schedule_for), as a list"""
    n = C.c_long(0)
    args = (nsteps, min(smallest_tile, 1 << 30), tile_cells, fuse, int(diffusion_only))
    _ck(lib().csim_pass_schedule_for(*args, None, 0, C.byref(n)))
    buf = (C.c_int * max(1, n.value))()
    _ck(lib().csim_pass_schedule_for(*args, buf, n.value, C.byref(n)))
    return list(buf[:n.value])


class Field:
    """Device mirror of the reference `struct Field` (include/field.hpp:5-21)."""

    def __init__(self, nx, ny, halo=1, dx=1.0, dy=1.0):
        self.nx_local, self.ny_local, self.halo, self.dx, self.dy = nx, ny, halo, dx, dy
        h = C.c_void_p()
        _ck(lib().csim_field_create(nx, ny, halo, dx, dy, C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib().csim_field_destroy(self._h)
            self._h = None

    def nx_total(self):
        return self.nx_local + 2 * self.halo

    def ny_total(self):
        return self.ny_local + 2 * self.halo

    def upload(self, host: np.ndarray):
        assert host.shape == (self.ny_total(), self.nx_total())
        _ck(lib().csim_field_upload(self._h, _dp(np.ascontiguousarray(host, dtype=np.float64))))
        return self

    def download(self) -> np.ndarray:
        out = np.empty((self.ny_total(), self.nx_total()))
        _ck(lib().csim_field_download(self._h, _dp(out)))
        return out

    def download_interior(self) -> np.ndarray:
        out = np.empty((self.ny_local, self.nx_local))
        _ck(lib().csim_field_download_interior(self._h, _dp(out)))
        return out

    def fill(self, v):
        _ck(lib().csim_field_fill(self._h, v))

    def copy_from(self, other: "Field"):
        _ck(lib().csim_field_copy(self._h, other._h))

    def swap(self, other: "Field"):
        _ck(lib().csim_field_swap(self._h, other._h))

    def minmax(self):
        o = (C.c_double * 2)()
        _ck(lib().csim_field_minmax(self._h, o))
        return o[0], o[1]

    def sum(self):
        o = C.c_double()
        _ck(lib().csim_field_sum(self._h, C.byref(o)))
        return o.value

    def linf_diff(self, other: "Field"):
        o = C.c_double()
        _ck(lib().csim_field_linf_diff(self._h, other._h, C.byref(o)))
        return o.value


def checksum_host(interior: np.ndarray, x_offset=0, y_offset=0, nx_global=None) -> int:
    """numpy restatement of csim_stepper_checksum for an (ny, nx) interior block (the checker's side)"""
    a = np.ascontiguousarray(interior, dtype=np.float64)
    ny, nx = a.shape
    nxg = nx if nx_global is None else nx_global
    g = (np.arange(ny, dtype=np.uint64)[:, None] + np.uint64(y_offset)) * np.uint64(nxg) + \
        (np.arange(nx, dtype=np.uint64)[None, :] + np.uint64(x_offset))
    with np.errstate(over="ignore"):
        w = np.uint64(0x9E3779B97F4A7C15) + np.uint64(2) * g
        return int((a.view(np.uint64) * w).sum(dtype=np.uint64))


def apply_boundary(f: Field, bc, is_physical=(1, 1, 1, 1), value=0.0):
    _ck(lib().csim_apply_boundary(f._h, _i4(bc), _i4(is_physical), value))


def diffusion_step(u: Field, out: Field, D, dt):
    _ck(lib().csim_diffusion_step(u._h, out._h, D, dt))


def advection_step(u: Field, out: Field, vx, vy, dt):
    _ck(lib().csim_advection_step(u._h, out._h, vx, vy, dt))


def fused_step(u: Field, out: Field, D, dt, vx, vy):
    _ck(lib().csim_fused_step(u._h, out._h, D, dt, vx, vy))


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    _ck(lib().csim_comm_unique_id(buf, UNIQUE_ID_BYTES))
    return buf.raw


class Stepper:
    """The time loop of reference src/main.cpp:93-118 (minus I/O) on one GPU / one rank."""

    def __init__(self, dec: Decomp, dx=1.0, dy=1.0, bc=(0, 0, 0, 0), bc_value=0.0):
        self.dec = dec
        self.nx, self.ny = dec.nx_local, dec.ny_local
        h = C.c_void_p()
        _ck(lib().csim_stepper_create(C.byref(dec), dx, dy, _i4(bc), bc_value, C.byref(h)))
        self._h = h

    @classmethod
    def single(cls, nx, ny, dx=1.0, dy=1.0, bc=(0, 0, 0, 0), bc_value=0.0):
        return cls(decomp_init(1, 0, nx, ny), dx, dy, bc, bc_value)

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "_h", None):
            lib().csim_stepper_destroy(self._h)
            self._h = None

    def comm_init(self, unique_id: bytes):
        buf = C.create_string_buffer(unique_id, UNIQUE_ID_BYTES)
        _ck(lib().csim_stepper_comm_init(self._h, buf, UNIQUE_ID_BYTES))

    def comm_share(self, owner: "Stepper"):
        """borrow another stepper's RCCL communicator (same rank; the owner must outlive this one)"""
        _ck(lib().csim_stepper_comm_share(self._h, owner._h))

    def checksum(self) -> int:
        """position-weighted 64-bit checksum of the local interior (csim_stepper_checksum)"""
        v = C.c_ulonglong(0)
        _ck(lib().csim_stepper_checksum(self._h, C.byref(v)))
        return v.value

    def upload(self, host: np.ndarray):
        assert host.shape == (self.ny + 2, self.nx + 2)
        _ck(lib().csim_stepper_upload(self._h, _dp(np.ascontiguousarray(host, dtype=np.float64))))

    def download(self) -> np.ndarray:
        out = np.empty((self.ny + 2, self.nx + 2))
        _ck(lib().csim_stepper_download(self._h, _dp(out)))
        return out

    def download_interior(self) -> np.ndarray:
        out = np.empty((self.ny, self.nx))
        _ck(lib().csim_stepper_download_interior(self._h, _dp(out)))
        return out

    def fuse_limit(self) -> int:
        d = C.c_int(0)
        _ck(lib().csim_stepper_fuse_limit(self._h, C.byref(d)))
        return d.value

    def faces_neighbors(self, depth):
        peers, lens = (C.c_int * 8)(), (C.c_int * 8)()
        _ck(lib().csim_stepper_faces_neighbors(self._h, depth, peers, lens))
        return list(peers), list(lens)

    def faces_pack(self, depth):
        """faces of the current field of the given depth per direction (None where no peer)."""
        peers, lens = self.faces_neighbors(depth)
        bufs = [np.empty(lens[d]) if peers[d] >= 0 else None for d in range(8)]
        arr = (C.POINTER(C.c_double) * 8)(*[_dp(b) if b is not None else None for b in bufs])
        _ck(lib().csim_stepper_faces_pack(self._h, depth, arr))
        return bufs

    def faces_unpack(self, depth, faces):
        keep = [np.ascontiguousarray(b, dtype=np.float64) if b is not None else None for b in faces]
        arr = (C.POINTER(C.c_double) * 8)(*[_dp(b) if b is not None else None for b in keep])
        _ck(lib().csim_stepper_faces_unpack(self._h, depth, arr))

    def snapshot_begin(self):
        _ck(lib().csim_stepper_snapshot_begin(self._h))

    def snapshot_wait(self) -> np.ndarray:
        """copy of the interior captured by the last snapshot_begin()"""
        ptr = C.POINTER(C.c_double)()
        _ck(lib().csim_stepper_snapshot_wait(self._h, C.byref(ptr)))
        return np.ctypeslib.as_array(ptr, shape=(self.ny, self.nx)).copy()

    def init_gaussian(self, A=1.0, sigma_frac=0.05, xc_frac=0.5, yc_frac=0.5):
        _ck(lib().csim_stepper_init_gaussian(self._h, A, sigma_frac, xc_frac, yc_frac))

    def exchange_halos(self):
        _ck(lib().csim_stepper_exchange_halos(self._h))

    def _side_len(self, k):
        return self.ny if k < 2 else self.nx

    def halo_pack(self):
        """edge lines of the current field per side (None on physical sides)."""
        bufs = [np.empty(self._side_len(k)) if self.dec.nbr[k] >= 0 else None for k in range(4)]
        arr = (C.POINTER(C.c_double) * 4)(*[_dp(b) if b is not None else None for b in bufs])
        _ck(lib().csim_stepper_halo_pack(self._h, arr))
        return bufs

    def halo_unpack(self, lines):
        """stage the neighbours' edge lines (list of 4, None on physical sides)."""
        keep = [np.ascontiguousarray(b, dtype=np.float64) if b is not None else None for b in lines]
        for k, b in enumerate(keep):
            assert b is None or b.shape == (self._side_len(k),)
        arr = (C.POINTER(C.c_double) * 4)(*[_dp(b) if b is not None else None for b in keep])
        _ck(lib().csim_stepper_halo_unpack(self._h, arr))

    def run(self, D, dt, vx, vy, nsteps):
        _ck(lib().csim_stepper_run(self._h, D, dt, vx, vy, nsteps))

    def sync(self):
        _ck(lib().csim_stepper_sync(self._h))

    def minmax(self):
        o = (C.c_double * 2)()
        _ck(lib().csim_stepper_minmax(self._h, o))
        return o[0], o[1]

    def sum(self):
        o = C.c_double()
        _ck(lib().csim_stepper_sum(self._h, C.byref(o)))
        return o.value

    def set_option(self, key: str, value: int):
        _ck(lib().csim_stepper_set_option(self._h, key.encode(), int(value)))

    def tune(self, D, dt, vx, vy):
        _ck(lib().csim_stepper_tune(self._h, D, dt, vx, vy))

    def keep_warm(self, D, dt, vx, vy, seconds):
        """load without effect on the field for about `seconds` (see csim_stepper_keep_warm)"""
        _ck(lib().csim_stepper_keep_warm(self._h, D, dt, vx, vy, float(seconds)))

    def get_option(self, key: str) -> int:
        v = C.c_long()
        _ck(lib().csim_stepper_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    def kernel_time(self, steps_per_launch=None):
        """(total ms, launches) of the timed sweep launches of one kind (1 or 2 steps per launch);
        with None: (total ms, launches, time steps covered) over both kinds."""
        def one(t):
            ms, n = C.c_double(), C.c_long()
            _ck(lib().csim_stepper_kernel_time(self._h, t, C.byref(ms), C.byref(n)))
            return ms.value, n.value
        if steps_per_launch is not None:
            return one(steps_per_launch)
        parts = [(t,) + one(t) for t in (1, 2, 3, 4, 5, 6, 7)]
        return (sum(p[1] for p in parts), sum(p[2] for p in parts), sum(p[0] * p[2] for p in parts))

    def comm_time(self):
        """(total ms, passes) of the sampled comm-stream chains (pack + RCCL exchange + unpack + ghost fill)"""
        ms, n = C.c_double(), C.c_long()
        _ck(lib().csim_stepper_comm_time(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def reset_timers(self):
        _ck(lib().csim_stepper_reset_timers(self._h))

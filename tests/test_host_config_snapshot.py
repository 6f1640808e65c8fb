"""Host side of the drop-in (no GPU): the YAML/CLI configuration surface and the NetCDF snapshot
container, through climate-sim-mpi-cpp_amd/driver/csim_hosttool.  Expectations restate the
reference's tests/simulation/unit/test_io.cpp (nested/flat YAML, `--k=v` / `--k v`, BC aliases,
validation errors, metadata strings) and SURVEY §0 quirks Q2/Q4."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRV = os.path.join(ROOT, "climate-sim-mpi-cpp_amd", "driver")
# CSIM_HOSTTOOL: another build of the host tool (tools/cpu_sanitize.sh points it at an ASan/UBSan one)
TOOL = os.environ.get("CSIM_HOSTTOOL") or os.path.join(DRV, "csim_hosttool")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "climate-sim-mpi-cpp_amd", "csrc")], check=True)
    subprocess.run(["make", "-s", "-C", DRV, "csim_hosttool"], check=True)


def tool(*args, env=None, check=True):
    e = dict(os.environ, **(env or {}))
    r = subprocess.run([TOOL, *args], capture_output=True, text=True, env=e)
    if check:
        assert r.returncode == 0, r.stderr
    return r


def cfg(*args):
    return json.loads(tool("print-config", *args).stdout)


def test_defaults_match_reference_structs():
    c = cfg()  # reference include/io.hpp:10-39
    assert (c["nx"], c["ny"], c["dx"], c["dy"]) == (256, 256, 1.0, 1.0)
    assert (c["D"], c["vx"], c["vy"], c["dt"], c["steps"], c["out_every"]) == (0.0, 0.0, 0.0, 0.1, 100, 50)
    assert c["bc"] == ["dirichlet"] * 4 and c["output_prefix"] == "snap"
    assert c["ic"] == dict(mode="preset", preset="gaussian_hotspot", A=1.0, sigma_frac=0.05,
                           xc_frac=0.5, yc_frac=0.5, path="", var="")


def test_dev_yaml_nested_blocks_and_flow_maps():
    c = cfg("--config", os.path.join(ROOT, "configs", "dev.yaml"))
    assert (c["nx"], c["ny"], c["D"], c["vx"], c["vy"], c["dt"], c["steps"], c["out_every"]) == \
        (512, 512, 0.05, 0.5, 0.0, 0.1, 1000, 100)
    assert c["bc"] == ["dirichlet", "neumann", "periodic", "dirichlet"]
    assert c["output_prefix"] == "dev"


def test_flat_yaml_and_scalar_bc(tmp_path):
    p = tmp_path / "flat.yaml"
    p.write_text("nx: 32\nny: 16\ndx: 0.5\ndy: 0.25\nD: 0.2\nvx: -1\nvy: 2\ndt: 0.01\nsteps: 5\n"
                 "out_every: 2\nbc: period   # alias\noutput_prefix: flat\n"
                 "ic:\n  preset: constant_zero\n  sigma_frac: 0.2\n  params:\n    A: 9.0\n")
    c = cfg(f"--config={p}")
    assert (c["nx"], c["ny"], c["dx"], c["dy"], c["D"], c["vx"], c["vy"]) == (32, 16, 0.5, 0.25, 0.2, -1.0, 2.0)
    assert (c["dt"], c["steps"], c["out_every"]) == (0.01, 5, 2)
    assert c["bc"] == ["periodic"] * 4 and c["output_prefix"] == "flat"
    assert c["ic"]["preset"] == "constant_zero" and c["ic"]["sigma_frac"] == 0.2
    assert c["ic"]["A"] == 1.0  # ic.params is not read (Q4)


def test_cli_forms_precedence_and_quirks(tmp_path):
    p = tmp_path / "c.yaml"
    p.write_text("grid: { nx: 64, ny: 48 }\ntime: { dt: 0.05, steps: 9, out_every: 3 }\nbc: neumann\n")
    c = cfg("--config", str(p), "--nx=128", "--steps", "11", "--bc.left=fixed", "--bc.top", "zero-flux",
            "--bc=periodic", "--D", "0.3", "--ic.sigma_frac=0.1", "--output.prefix=run7", "--unknown=1")
    assert (c["nx"], c["ny"], c["steps"], c["dt"], c["out_every"], c["D"]) == (128, 48, 11, 0.05, 3, 0.3)
    # --bc=periodic is not a flag of the reference (Q2): YAML neumann stays, sides overridden
    assert c["bc"] == ["dirichlet", "neumann", "neumann", "neumann"]
    assert c["ic"]["sigma_frac"] == 0.1 and c["output_prefix"] == "run7"


@pytest.mark.parametrize("args,msg", [
    (["--nx=0"], "nx/ny must be > 0"), (["--dy=-1"], "dx/dy must be > 0"), (["--dt=0"], "dt must be > 0"),
    (["--steps=0"], "steps must be > 0"), (["--out_every=0"], "out_every must be >= 1"),
    (["--bc.left=robin"], "Unknown BC type: robin")])
def test_validation_errors(args, msg):
    r = tool("print-config", *args, check=False)
    assert r.returncode != 0 and msg in r.stderr


# ---- NetCDF snapshot container ------------------------------------------------------------------
def parse_cdf(path):
    """independent parser of the classic netCDF header (CDF-1/2/5), written from the format spec"""
    b = open(path, "rb").read()
    assert b[:3] == b"CDF"
    ver = b[3]
    pos = 4

    def i32():
        nonlocal pos
        v = struct.unpack(">i", b[pos:pos + 4])[0]
        pos += 4
        return v

    def i64():
        nonlocal pos
        v = struct.unpack(">q", b[pos:pos + 8])[0]
        pos += 8
        return v

    nn = i64 if ver == 5 else i32
    off = i32 if ver == 1 else i64

    def name():
        nonlocal pos
        n = nn()
        s = b[pos:pos + n].decode()
        pos += (n + 3) // 4 * 4
        return s

    def atts():
        nonlocal pos
        tag, n = i32(), nn()
        out = {}
        for _ in range(n if tag == 12 else 0):
            k = name()
            t, cnt = i32(), nn()
            assert t == 2
            out[k] = b[pos:pos + cnt].decode()
            pos += (cnt + 3) // 4 * 4
        return out

    numrecs = nn()
    assert i32() == 10
    dims = [(name(), nn()) for _ in range(nn())]
    gatts = atts()
    assert i32() == 11
    vars_ = {}
    for _ in range(nn()):
        vn = name()
        dimids = [nn() for _ in range(nn())]
        atts()
        t, vsize, begin = i32(), nn(), off()
        vars_[vn] = dict(dimids=dimids, type=t, vsize=vsize, begin=begin)
    return dict(version=ver, numrecs=numrecs, dims=dims, gatts=gatts, vars=vars_, raw=b)


def test_cdf5_snapshot_layout_and_metadata(tmp_path):
    nx, ny, nrec = 7, 5, 3
    rng = np.random.default_rng(1)
    data = rng.standard_normal((nrec, ny, nx))
    raw = tmp_path / "raw.bin"
    data.tofile(raw)
    out = tmp_path / "snap.nc"
    tool("nc-write", str(out), str(raw), str(nrec), f"--nx={nx}", f"--ny={ny}", "--dt=0.1", "--steps=12",
         "--D=0.05", "--vx=0.5", "--vy=-0.25", "--bc.right=neumann")
    h = parse_cdf(out)
    assert h["version"] == 5 and h["numrecs"] == nrec       # NC_64BIT_DATA, like reference io.cpp:386
    assert h["dims"] == [("time", 0), ("y", ny), ("x", nx)]  # reference io.cpp:389-391
    v = h["vars"]["u"]
    assert v["dimids"] == [0, 1, 2] and v["type"] == 6 and v["vsize"] == nx * ny * 8
    got = np.frombuffer(h["raw"], dtype=">f8", count=nrec * ny * nx, offset=v["begin"]).reshape(nrec, ny, nx)
    assert np.array_equal(got, data)
    a = h["gatts"]  # reference io.cpp:439-447 (std::to_string formatting)
    assert list(a) == ["description", "grid", "dt", "steps", "D", "velocity", "boundary_conditions"]
    assert a["description"] == "climate-sim-mpi-cpp" and a["grid"] == "7 x 5"
    assert a["dt"] == "0.100000" and a["steps"] == "12" and a["D"] == "0.050000"
    assert a["velocity"] == "(0.500000,-0.250000)"
    assert a["boundary_conditions"] == "left=dirichlet right=neumann bottom=dirichlet top=dirichlet"
    # our reader (used for ic.mode=file) on the same file
    back = tmp_path / "back.bin"
    r = tool("nc-read", str(out), "u", "2", str(back))
    assert r.stdout.split() == [str(ny), str(nx)]
    assert np.array_equal(np.fromfile(back).reshape(ny, nx), data[2])
    assert "grid=7 x 5" in tool("nc-attrs", str(out)).stdout


def test_cdf2_flavour_is_readable_by_scipy(tmp_path):
    """the same writer in 64-bit-offset (CDF-2) mode, read by an independent implementation"""
    from scipy.io import netcdf_file
    nx, ny, nrec = 6, 4, 2
    data = np.arange(nrec * ny * nx, dtype=float).reshape(nrec, ny, nx) / 7.0
    raw = tmp_path / "raw.bin"
    data.tofile(raw)
    out = tmp_path / "snap2.nc"
    tool("nc-write", str(out), str(raw), str(nrec), f"--nx={nx}", f"--ny={ny}", env={"CSIM_NC_VERSION": "2"})
    with netcdf_file(str(out), "r", mmap=False) as f:
        assert f.dimensions == {"time": None, "y": ny, "x": nx}
        assert f.variables["u"].shape == (nrec, ny, nx)
        assert np.array_equal(f.variables["u"][:], data)
        assert f.description == b"climate-sim-mpi-cpp" and f.grid == b"6 x 4"


def test_windowed_ic_read_never_holds_the_global_array(tmp_path):
    """ic.mode=file on a decomposed run: every rank reads only rows y_offset.. / columns x_offset.. of
    the record (per-rank start/count like reference src/io.cpp:402-418).  The read must not
    raise the process's peak RSS by anything near the size of the global array."""
    nx, ny = 2048, 1536                     # 25 MB record
    rng = np.random.default_rng(5)
    data = rng.standard_normal((2, ny, nx))
    raw = tmp_path / "raw.bin"
    data.tofile(raw)
    out = tmp_path / "ic.nc"
    tool("nc-write", str(out), str(raw), "2", f"--nx={nx}", f"--ny={ny}")
    for (y0, x0, wy, wx) in [(0, 0, 3, 5), (700, 1000, 96, 130), (ny - 64, nx - 100, 64, 100), (0, 0, 1, nx)]:
        back = tmp_path / "win.bin"
        r = tool("nc-read-window", str(out), "u", "1", str(y0), str(x0), str(wy), str(wx), str(back))
        got_ny, got_nx, rss_kib = (int(v) for v in r.stdout.split())
        assert (got_ny, got_nx) == (ny, nx)
        w = np.fromfile(back).reshape(wy + 2, wx + 2)
        assert np.array_equal(w[1:-1, 1:-1], data[1, y0:y0 + wy, x0:x0 + wx])
        ring = np.ones(w.shape, bool)
        ring[1:-1, 1:-1] = False
        assert (w[ring] == -7.0).all()           # nothing outside the interior was touched
        assert rss_kib * 1024 < nx * ny * 8 // 8, rss_kib   # growth of the peak RSS: the 25 MB record was never resident
    r = subprocess.run([TOOL, "nc-read-window", str(out), "u", "0", "10", "10", str(ny), "4", str(tmp_path / "x.bin")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "window outside" in r.stderr

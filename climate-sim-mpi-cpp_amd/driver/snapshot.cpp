// snapshot.cpp — CDF-5 ("64-bit data") classic NetCDF writer/reader, see include/climate/snapshot.hpp.
// Format reference: the netCDF classic format specification (CDF-1/2/5): all header integers
// big-endian; in CDF-5 every NON_NEG / OFFSET field is 64-bit, list tags and nc_type stay 32-bit;
// names and attribute values are padded to 4 bytes.  Metadata strings match reference
// src/io.cpp:428-448 (std::to_string formatting).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <stdexcept>

#include "climate/snapshot.hpp"

namespace {

constexpr int32_t NC_DIMENSION = 10, NC_VARIABLE = 11, NC_ATTRIBUTE = 12;
constexpr int32_t NC_CHAR = 2, NC_DOUBLE = 6;

struct Writer {
    std::string buf;
    int version;  // 5 (default, like the reference) or 2 (64-bit offsets, readable by scipy)
    void i32(int32_t v) {
        for (int s = 24; s >= 0; s -= 8) buf.push_back(static_cast<char>((static_cast<uint32_t>(v) >> s) & 0xff));
    }
    void i64(int64_t v) {
        for (int s = 56; s >= 0; s -= 8) buf.push_back(static_cast<char>((static_cast<uint64_t>(v) >> s) & 0xff));
    }
    void nonneg(int64_t v) { version == 5 ? i64(v) : i32(static_cast<int32_t>(v)); }
    void offset(int64_t v) { version == 1 ? i32(static_cast<int32_t>(v)) : i64(v); }
    void padded(const std::string& s) {
        buf += s;
        while (buf.size() % 4) buf.push_back('\0');
    }
    void name(const std::string& s) {
        nonneg(static_cast<int64_t>(s.size()));
        padded(s);
    }
    void text_attr(const std::string& k, const std::string& v) {
        name(k);
        i32(NC_CHAR);
        nonneg(static_cast<int64_t>(v.size()));
        padded(v);
    }
};

struct OpenFile {
    int fd = -1;
    int version = 5;
    int64_t begin = 0, recsize = 0, numrecs_pos = 4;
    int nx_global = 0, ny_global = 0;
    int rank = 0;
    int64_t numrecs = 0;
    MPI_Comm comm = MPI_COMM_NULL;
};
std::map<int, OpenFile> g_files;
int g_next = 1;

std::string header(const Decomp2D& dec, const SimConfig& cfg, int version, int64_t& begin, int64_t& recsize) {
    // two passes: the variable's `begin` is the (4-byte aligned) header length itself
    recsize = static_cast<int64_t>(dec.ny_global) * dec.nx_global * 8;
    begin = 0;
    std::string out;
    for (int pass = 0; pass < 2; ++pass) {
        Writer w{std::string(), version};
        w.buf = std::string("CDF") + static_cast<char>(version);
        w.nonneg(0);  // numrecs, patched on close
        w.i32(NC_DIMENSION);
        w.nonneg(3);
        w.name("time");
        w.nonneg(0);  // record dimension
        w.name("y");
        w.nonneg(dec.ny_global);
        w.name("x");
        w.nonneg(dec.nx_global);
        w.i32(NC_ATTRIBUTE);
        w.nonneg(7);
        w.text_attr("description", "climate-sim-mpi-cpp");
        w.text_attr("grid", std::to_string(cfg.nx) + " x " + std::to_string(cfg.ny));
        w.text_attr("dt", std::to_string(cfg.dt));
        w.text_attr("steps", std::to_string(cfg.steps));
        w.text_attr("D", std::to_string(cfg.D));
        w.text_attr("velocity", "(" + std::to_string(cfg.vx) + "," + std::to_string(cfg.vy) + ")");
        w.text_attr("boundary_conditions", "left=" + bc_to_string(cfg.bc.left) + " right=" +
                                               bc_to_string(cfg.bc.right) + " bottom=" +
                                               bc_to_string(cfg.bc.bottom) + " top=" +
                                               bc_to_string(cfg.bc.top));
        w.i32(NC_VARIABLE);
        w.nonneg(1);
        w.name("u");
        w.nonneg(3);
        w.nonneg(0);
        w.nonneg(1);
        w.nonneg(2);
        w.i32(0);  // no variable attributes: ABSENT = ZERO tag + ZERO count
        w.nonneg(0);
        w.i32(NC_DOUBLE);
        w.nonneg(recsize);
        w.offset(begin);
        out = w.buf;
        begin = static_cast<int64_t>(out.size());
    }
    return out;
}

void barrier(MPI_Comm comm) {
#ifdef CSIM_WITH_MPI
    MPI_Barrier(comm);
#else
    (void)comm;
#endif
}

uint64_t bswap(uint64_t v) { return __builtin_bswap64(v); }

bool pwrite_all(int fd, const void* p, size_t n, int64_t off) {
    const char* c = static_cast<const char*>(p);
    while (n) {
        const ssize_t k = ::pwrite(fd, c, n, off);
        if (k <= 0) return false;
        c += k;
        off += k;
        n -= static_cast<size_t>(k);
    }
    return true;
}

}  // namespace

int open_netcdf_parallel(const std::string& filename, const Decomp2D& dec, const SimConfig& cfg,
                         MPI_Comm comm, int& ncid, int& varid) {
    OpenFile f;
    const char* ver = std::getenv("CSIM_NC_VERSION");
    f.version = (ver && std::string(ver) == "2") ? 2 : 5;
    f.rank = dec.world_rank;
    f.comm = comm;
    f.nx_global = dec.nx_global;
    f.ny_global = dec.ny_global;
#ifndef CSIM_WITH_MPI
    if (dec.world_size > 1)
        throw std::runtime_error("open_netcdf_parallel: multi-rank output needs a -DCSIM_WITH_MPI build");
#endif
    const std::string hdr = header(dec, cfg, f.version, f.begin, f.recsize);
    if (f.rank == 0) {  // NC_CLOBBER
        f.fd = ::open(filename.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0644);
        if (f.fd < 0 || !pwrite_all(f.fd, hdr.data(), hdr.size(), 0))
            throw std::runtime_error("ncmpi_create: cannot write " + filename);
    }
    barrier(comm);
    if (f.rank != 0) {
        f.fd = ::open(filename.c_str(), O_RDWR);
        if (f.fd < 0) throw std::runtime_error("ncmpi_create: cannot open " + filename);
    }
    ncid = g_next++;
    varid = 0;
    g_files[ncid] = f;
    return NC_NOERR;
}

bool write_interior_netcdf(int ncid, int varid, const double* interior, const Decomp2D& dec, int step) {
    (void)varid;
    auto it = g_files.find(ncid);
    if (it == g_files.end() || step < 0) {
        std::cerr << "Rank write failed: bad ncid/step\n";
        return false;
    }
    OpenFile& f = it->second;
    std::vector<uint64_t> row(static_cast<size_t>(dec.nx_local));
    for (int j = 0; j < dec.ny_local; ++j) {
        const double* src = interior + static_cast<size_t>(j) * dec.nx_local;
        for (int i = 0; i < dec.nx_local; ++i) {
            uint64_t bits;
            std::memcpy(&bits, &src[i], 8);
            row[i] = bswap(bits);
        }
        const int64_t off = f.begin + static_cast<int64_t>(step) * f.recsize +
                            (static_cast<int64_t>(dec.y_offset + j) * f.nx_global + dec.x_offset) * 8;
        if (!pwrite_all(f.fd, row.data(), row.size() * 8, off)) {
            std::cerr << "Rank write failed: pwrite\n";
            return false;
        }
    }
    if (step + 1 > f.numrecs) f.numrecs = step + 1;
    return true;
}

bool write_field_netcdf(int ncid, int varid, const Field& f, const Decomp2D& dec, int step) {
    std::vector<double> buf(static_cast<size_t>(dec.nx_local) * dec.ny_local);
    for (int j = 0; j < dec.ny_local; ++j)
        for (int i = 0; i < dec.nx_local; ++i)
            buf[static_cast<size_t>(j) * dec.nx_local + i] = f.at(i + f.halo, j + f.halo);
    return write_interior_netcdf(ncid, varid, buf.data(), dec, step);
}

void close_netcdf_parallel(int ncid) {
    auto it = g_files.find(ncid);
    if (it == g_files.end()) return;
    OpenFile& f = it->second;
    barrier(f.comm);
    if (f.rank == 0) {
        Writer w{std::string(), f.version};
        w.nonneg(f.numrecs);
        pwrite_all(f.fd, w.buf.data(), w.buf.size(), 4);
    }
    ::close(f.fd);
    g_files.erase(it);
}

// ---- reader --------------------------------------------------------------------------------------
namespace {
struct Reader {
    std::string b;
    size_t p = 0;
    int version = 1;
    int32_t i32() {
        if (p + 4 > b.size()) throw std::runtime_error("netcdf: truncated header");
        uint32_t v = 0;
        for (int k = 0; k < 4; ++k) v = (v << 8) | static_cast<unsigned char>(b[p++]);
        return static_cast<int32_t>(v);
    }
    int64_t i64() {
        if (p + 8 > b.size()) throw std::runtime_error("netcdf: truncated header");
        uint64_t v = 0;
        for (int k = 0; k < 8; ++k) v = (v << 8) | static_cast<unsigned char>(b[p++]);
        return static_cast<int64_t>(v);
    }
    int64_t nonneg() { return version == 5 ? i64() : i32(); }
    int64_t offset() { return version == 1 ? i32() : i64(); }
    std::string str(int64_t n) {
        if (p + static_cast<size_t>(n) > b.size()) throw std::runtime_error("netcdf: truncated header");
        std::string s = b.substr(p, static_cast<size_t>(n));
        p += static_cast<size_t>(n);
        p = (p + 3) / 4 * 4;
        return s;
    }
    std::string name() { return str(nonneg()); }
};

int64_t type_size(int32_t t) {
    switch (t) {
        case 1: case 2: case 7: return 1;
        case 3: case 8: return 2;
        case 4: case 5: case 9: return 4;
        default: return 8;
    }
}

struct VarInfo {
    std::vector<int64_t> dimids;
    int32_t type = 0;
    int64_t vsize = 0, begin = 0;
};
struct Parsed {
    int version = 1;
    int64_t numrecs = 0;
    std::vector<std::pair<std::string, int64_t>> dims;
    std::vector<std::pair<std::string, std::string>> gatts;
    std::map<std::string, VarInfo> vars;
    int64_t recsize = 0;  // sum of record variables' vsize
};

std::vector<std::pair<std::string, std::string>> read_atts(Reader& r) {
    std::vector<std::pair<std::string, std::string>> out;
    const int32_t tag = r.i32();
    const int64_t n = r.nonneg();
    if (tag == 0) return out;
    if (tag != NC_ATTRIBUTE) throw std::runtime_error("netcdf: bad attribute list");
    for (int64_t k = 0; k < n; ++k) {
        const std::string nm = r.name();
        const int32_t t = r.i32();
        const int64_t cnt = r.nonneg();
        const std::string raw = r.str(cnt * type_size(t));
        out.emplace_back(nm, t == NC_CHAR ? raw : std::string());
    }
    return out;
}

Parsed parse(const std::string& filename) {
    std::ifstream f(filename, std::ios::binary);
    if (!f) throw std::runtime_error("netcdf: cannot open " + filename);
    Reader r;
    r.b.resize(1 << 20);
    f.read(&r.b[0], static_cast<std::streamsize>(r.b.size()));
    r.b.resize(static_cast<size_t>(f.gcount()));
    if (r.b.size() < 8 || r.b.compare(0, 3, "CDF") != 0) throw std::runtime_error("netcdf: not a classic file");
    Parsed P;
    P.version = r.version = static_cast<unsigned char>(r.b[3]);
    if (P.version != 1 && P.version != 2 && P.version != 5) throw std::runtime_error("netcdf: unsupported version");
    r.p = 4;
    P.numrecs = r.nonneg();
    int32_t tag = r.i32();
    int64_t n = r.nonneg();
    if (tag == NC_DIMENSION)
        for (int64_t k = 0; k < n; ++k) {
            const std::string nm = r.name();
            P.dims.emplace_back(nm, r.nonneg());
        }
    P.gatts = read_atts(r);
    tag = r.i32();
    n = r.nonneg();
    if (tag == NC_VARIABLE)
        for (int64_t k = 0; k < n; ++k) {
            const std::string nm = r.name();
            VarInfo v;
            const int64_t rank = r.nonneg();
            for (int64_t d = 0; d < rank; ++d) v.dimids.push_back(r.nonneg());
            read_atts(r);
            v.type = r.i32();
            v.vsize = r.nonneg();
            v.begin = r.offset();
            const bool rec = !v.dimids.empty() && P.dims[static_cast<size_t>(v.dimids[0])].second == 0;
            if (rec) P.recsize += v.vsize;
            P.vars[nm] = v;
        }
    return P;
}
}  // namespace

std::vector<std::pair<std::string, std::string>> read_netcdf_attrs(const std::string& filename) {
    return parse(filename).gatts;
}

namespace {
// locate record `step` of a double variable with dims ([time,] y, x): file offset of its first element
int64_t locate_2d(const Parsed& P, const std::string& var, int step, int& ny, int& nx) {
    auto it = P.vars.find(var);
    if (it == P.vars.end()) throw std::runtime_error("netcdf: no variable " + var);
    const VarInfo& v = it->second;
    if (v.type != NC_DOUBLE) throw std::runtime_error("netcdf: variable is not double");
    const bool rec = !v.dimids.empty() && P.dims[static_cast<size_t>(v.dimids[0])].second == 0;
    const size_t nd = v.dimids.size();
    if ((rec && nd != 3) || (!rec && nd != 2)) throw std::runtime_error("netcdf: expected ([time,] y, x)");
    ny = static_cast<int>(P.dims[static_cast<size_t>(v.dimids[nd - 2])].second);
    nx = static_cast<int>(P.dims[static_cast<size_t>(v.dimids[nd - 1])].second);
    if (rec && (step < 0 || step >= P.numrecs)) throw std::runtime_error("netcdf: record out of range");
    return v.begin + (rec ? static_cast<int64_t>(step) * P.recsize : 0);
}

void pread_doubles(int fd, double* dst, size_t count, int64_t off) {
    size_t got = 0;
    const size_t want = count * 8;
    while (got < want) {
        const ssize_t k = ::pread(fd, reinterpret_cast<char*>(dst) + got, want - got, off + static_cast<int64_t>(got));
        if (k <= 0) break;
        got += static_cast<size_t>(k);
    }
    if (got != want) throw std::runtime_error("netcdf: short read");
    for (size_t k = 0; k < count; ++k) {
        uint64_t bits;
        std::memcpy(&bits, &dst[k], 8);
        bits = bswap(bits);
        std::memcpy(&dst[k], &bits, 8);
    }
}
}  // namespace

void read_netcdf_2d(const std::string& filename, const std::string& var, int step, int& ny, int& nx,
                    std::vector<double>& out) {
    const Parsed P = parse(filename);
    const int64_t off = locate_2d(P, var, step, ny, nx);
    out.resize(static_cast<size_t>(ny) * nx);
    const int fd = ::open(filename.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("netcdf: cannot open " + filename);
    try {
        pread_doubles(fd, out.data(), out.size(), off);
    } catch (...) {
        ::close(fd);
        throw;
    }
    ::close(fd);
}

// Per-rank window of the same record: rows y0 .. y0+wy-1, columns x0 .. x0+wx-1, one pread per row
// straight into `dst` (row stride `dst_stride` doubles).  Nothing of global size is allocated — the
// counterpart of the per-rank start/count of the reference's writer (src/io.cpp:402-418).
void read_netcdf_window(const std::string& filename, const std::string& var, int step, int y0, int x0,
                        int wy, int wx, double* dst, size_t dst_stride, int& ny, int& nx) {
    const Parsed P = parse(filename);
    const int64_t off = locate_2d(P, var, step, ny, nx);
    if (y0 < 0 || x0 < 0 || wy < 0 || wx < 0 || y0 + wy > ny || x0 + wx > nx)
        throw std::runtime_error("netcdf: window outside the variable");
    const int fd = ::open(filename.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("netcdf: cannot open " + filename);
    try {
        for (int j = 0; j < wy; ++j)
            pread_doubles(fd, dst + static_cast<size_t>(j) * dst_stride, static_cast<size_t>(wx),
                          off + (static_cast<int64_t>(y0 + j) * nx + x0) * 8);
    } catch (...) {
        ::close(fd);
        throw;
    }
    ::close(fd);
}

void netcdf_dims_2d(const std::string& filename, const std::string& var, int& ny, int& nx) {
    const Parsed P = parse(filename);
    (void)locate_2d(P, var, 0, ny, nx);
}

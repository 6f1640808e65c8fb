// oracle/ref_harness.cpp — TEST INFRASTRUCTURE, NOT PRODUCT.
//
// Drives the REAL reference objects (compiled from /root/reference/src/*.cpp where
// they lie, see oracle/Makefile) so that golden vectors and the "reference" CPU
// baseline come from the reference's own arithmetic.  Nothing under oracle/ is
// imported, linked or executed by the product path (include/, climate-sim-mpi-cpp_amd/).
//
// The loop in mode "run" replays the call order of the reference driver
// (reference src/main.cpp:62-109): Decomp2D::init -> Field u,tmp(h=1) -> IC ->
// per step { exchange_halos; apply_boundary(.,.,bc,0.0); copy u->tmp;
// diffusion_step; advection_step; swap }.  NetCDF/YAML are not involved (PnetCDF and
// yaml-cpp are absent from the image, so src/io.cpp, src/init.cpp and src/main.cpp
// are not buildable here; the gaussian IC below restates src/init.cpp:12-33).
//
// Modes (first argument):
//   run        full multi-rank time loop; dumps per-rank local fields (ghosts included)
//   unit       one diffusion_step and/or advection_step on rank 0 (ring-copy/accumulate)
//   boundary   apply_boundary alone on rank 0 with a caller-chosen fill value
//   decomp     prints Decomp2D of every rank as one text line each
//   safedt     evaluates the reference's safe_dt (include/stability.hpp:5-16)
#include <mpi.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "advection.hpp"
#include "boundary.hpp"
#include "decomp.hpp"
#include "diffusion.hpp"
#include "field.hpp"
#include "halo.hpp"
#include "stability.hpp"

namespace {

struct Args {
    std::map<std::string, std::string> kv;
    bool has(const std::string& k) const { return kv.count(k) != 0; }
    std::string str(const std::string& k, const std::string& d = "") const {
        auto it = kv.find(k);
        return it == kv.end() ? d : it->second;
    }
    double num(const std::string& k, double d) const {
        auto it = kv.find(k);
        return it == kv.end() ? d : std::strtod(it->second.c_str(), nullptr);
    }
    int integer(const std::string& k, int d) const {
        auto it = kv.find(k);
        return it == kv.end() ? d : std::atoi(it->second.c_str());
    }
};

Args parse(int argc, char** argv, int first) {
    Args a;
    for (int i = first; i < argc; ++i) {
        std::string s = argv[i];
        if (s.rfind("--", 0) != 0) continue;
        auto eq = s.find('=');
        if (eq == std::string::npos)
            a.kv[s.substr(2)] = "1";
        else
            a.kv[s.substr(2, eq - 2)] = s.substr(eq + 1);
    }
    return a;
}

BCType bc_of(char c) {
    switch (c) {
        case 'd': case 'D': return BCType::Dirichlet;
        case 'n': case 'N': return BCType::Neumann;
        default: return BCType::Periodic;
    }
}

// "dnpd" -> left,right,bottom,top
BCConfig bc_from_code(const std::string& code) {
    BCConfig bc;
    std::string c = code.size() == 4 ? code : "dddd";
    bc.left = bc_of(c[0]);
    bc.right = bc_of(c[1]);
    bc.bottom = bc_of(c[2]);
    bc.top = bc_of(c[3]);
    return bc;
}

std::vector<double> read_doubles(const std::string& path, size_t n) {
    std::vector<double> v(n);
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) {
        std::fprintf(stderr, "cannot open %s\n", path.c_str());
        MPI_Abort(MPI_COMM_WORLD, 2);
    }
    size_t got = std::fread(v.data(), sizeof(double), n, f);
    std::fclose(f);
    if (got != n) {
        std::fprintf(stderr, "short read on %s (%zu of %zu)\n", path.c_str(), got, n);
        MPI_Abort(MPI_COMM_WORLD, 2);
    }
    return v;
}

void write_doubles(const std::string& path, const double* p, size_t n) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) {
        std::fprintf(stderr, "cannot write %s\n", path.c_str());
        MPI_Abort(MPI_COMM_WORLD, 2);
    }
    std::fwrite(p, sizeof(double), n, f);
    std::fclose(f);
}

// gaussian hotspot exactly as the reference preset (src/init.cpp:12-33, defaults include/io.hpp:13-16)
void ic_gaussian(const Decomp2D& dec, Field& u, int nxg, int nyg, double dx, double dy, double A,
                 double sigma_frac, double xc_frac, double yc_frac) {
    const double Lx = nxg * dx, Ly = nyg * dy;
    const double xc = xc_frac * Lx, yc = yc_frac * Ly;
    const double sig = sigma_frac * std::min(Lx, Ly);
    for (int j = 0; j < u.ny_local; ++j) {
        const double y = (dec.y_offset + j + 0.5) * dy;
        for (int i = 0; i < u.nx_local; ++i) {
            const double x = (dec.x_offset + i + 0.5) * dx;
            const double r2 = (x - xc) * (x - xc) + (y - yc) * (y - yc);
            u.at(i + u.halo, j + u.halo) = A * std::exp(-r2 / (2.0 * sig * sig));
        }
    }
}

// global (ny x nx, row-major) interior block -> local interior
void ic_from_global(const Decomp2D& dec, Field& u, const std::vector<double>& g, int nxg) {
    for (int j = 0; j < u.ny_local; ++j)
        for (int i = 0; i < u.nx_local; ++i)
            u.at(i + u.halo, j + u.halo) =
                g[static_cast<size_t>(dec.y_offset + j) * nxg + (dec.x_offset + i)];
}

int mode_run(const Args& a, int rank, int size) {
    const int nxg = a.integer("nx", 64), nyg = a.integer("ny", 64);
    const double dx = a.num("dx", 1.0), dy = a.num("dy", 1.0);
    const double D = a.num("D", 0.0), vx = a.num("vx", 0.0), vy = a.num("vy", 0.0);
    double dt = a.num("dt", 0.1);
    const int steps = a.integer("steps", 10);
    const BCConfig bc = bc_from_code(a.str("bc", "dddd"));
    const std::string out = a.str("out", "");
    const bool clamp = a.integer("clamp", 1) != 0;

    if (clamp) {  // reference src/main.cpp:42-49
        const double lim = safe_dt(dx, dy, vx, vy, D);
        if (dt > lim) dt = lim;
    }

    Decomp2D dec;
    dec.init(MPI_COMM_WORLD, nxg, nyg);
    const int halo = 1;
    Field u(dec.nx_local, dec.ny_local, halo, dx, dy);
    Field tmp(dec.nx_local, dec.ny_local, halo, dx, dy);
    u.fill(0.0);
    tmp.fill(0.0);

    const std::string ic = a.str("ic", "gaussian");
    if (ic == "gaussian") {
        ic_gaussian(dec, u, nxg, nyg, dx, dy, a.num("A", 1.0), a.num("sigma_frac", 0.05),
                    a.num("xc_frac", 0.5), a.num("yc_frac", 0.5));
    } else if (ic == "zero") {
    } else {  // path to a raw fp64 global interior (ny x nx)
        auto g = read_doubles(ic, static_cast<size_t>(nxg) * nyg);
        ic_from_global(dec, u, g, nxg);
    }

    if (!out.empty() && a.integer("dump_initial", 0))
        write_doubles(out + ".init.rank" + std::to_string(rank) + ".bin", u.data.data(),
                      u.data.size());

    MPI_Barrier(MPI_COMM_WORLD);
    const double t0 = MPI_Wtime();
    double sum_step = 0.0;
    for (int n = 0; n < steps; ++n) {
        const double ts = MPI_Wtime();
        exchange_halos(u, dec, MPI_COMM_WORLD);
        apply_boundary(u, dec, bc, 0.0);
        std::copy(u.data.begin(), u.data.end(), tmp.data.begin());
        diffusion_step(u, tmp, D, dt);
        advection_step(u, tmp, vx, vy, dt);
        std::swap(u.data, tmp.data);
        sum_step += MPI_Wtime() - ts;
    }
    const double total = MPI_Wtime() - t0;

    double total_max = 0.0, step_worst = 0.0;
    double avg_step = sum_step / std::max(1, steps);
    MPI_Reduce(&total, &total_max, 1, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD);
    MPI_Reduce(&avg_step, &step_worst, 1, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD);

    // per-rank interior sum / global max as cheap known answers
    double lsum = 0.0, lmax = -1e300;
    for (int j = 0; j < u.ny_local; ++j)
        for (int i = 0; i < u.nx_local; ++i) {
            const double v = u.at(i + halo, j + halo);
            lsum += v;
            lmax = std::max(lmax, v);
        }
    double gmax = 0.0;
    MPI_Reduce(&lmax, &gmax, 1, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD);

    if (!out.empty())
        write_doubles(out + ".rank" + std::to_string(rank) + ".bin", u.data.data(), u.data.size());

    if (rank == 0) {
        std::printf("ranks=%d dims=%dx%d dt=%.17g\n", size, dec.dims[0], dec.dims[1], dt);
        std::printf("known: max=%.17g sum_rank0=%.17g\n", gmax, lsum);
        // same line format as reference src/main.cpp:131-132
        std::printf("timing: total_max=%g s, worst_avg_step=%g s\n", total_max, step_worst);
        std::printf("mcells_per_s=%.6g\n",
                    static_cast<double>(nxg) * nyg * steps / std::max(total_max, 1e-12) / 1e6);
    }
    dec.finalize();
    return 0;
}

int mode_unit(const Args& a, int rank) {
    if (rank != 0) return 0;
    const int nx = a.integer("nx", 8), ny = a.integer("ny", 8);
    const double dx = a.num("dx", 1.0), dy = a.num("dy", 1.0);
    Field u(nx, ny, 1, dx, dy), o(nx, ny, 1, dx, dy);
    auto uin = read_doubles(a.str("u"), u.data.size());
    auto oin = read_doubles(a.str("o"), o.data.size());
    u.data = uin;
    o.data = oin;
    const std::string op = a.str("op", "diffusion");
    if (op == "diffusion")
        diffusion_step(u, o, a.num("D", 0.1), a.num("dt", 0.1));
    else if (op == "advection")
        advection_step(u, o, a.num("vx", 0.0), a.num("vy", 0.0), a.num("dt", 0.1));
    else {
        std::fprintf(stderr, "unknown op\n");
        return 2;
    }
    write_doubles(a.str("out"), o.data.data(), o.data.size());
    return 0;
}

int mode_boundary(const Args& a, int rank) {
    const int nx = a.integer("nx", 4), ny = a.integer("ny", 3);
    Decomp2D dec;
    dec.init(MPI_COMM_WORLD, nx, ny);
    if (rank == 0) {
        Field f(dec.nx_local, dec.ny_local, 1, 1.0, 1.0);
        f.data = read_doubles(a.str("u"), f.data.size());
        apply_boundary(f, dec, bc_from_code(a.str("bc", "dddd")), a.num("value", 0.0));
        write_doubles(a.str("out"), f.data.data(), f.data.size());
    }
    dec.finalize();
    return 0;
}

int mode_decomp(const Args& a, int rank, int size) {
    const int nxg = a.integer("nx", 16), nyg = a.integer("ny", 12);
    Decomp2D dec;
    dec.init(MPI_COMM_WORLD, nxg, nyg);
    int row[13] = {rank,           dec.dims[0],   dec.dims[1],   dec.coords[0], dec.coords[1],
                   dec.nbr_lr[0],  dec.nbr_lr[1], dec.nbr_du[0], dec.nbr_du[1], dec.nx_local,
                   dec.ny_local,   dec.x_offset,  dec.y_offset};
    for (int k = 5; k <= 8; ++k)
        if (row[k] == MPI_PROC_NULL) row[k] = -1;
    std::vector<int> all(static_cast<size_t>(13) * size);
    MPI_Gather(row, 13, MPI_INT, all.data(), 13, MPI_INT, 0, MPI_COMM_WORLD);
    if (rank == 0)
        for (int r = 0; r < size; ++r) {
            for (int k = 0; k < 13; ++k) std::printf("%d%c", all[r * 13 + k], k == 12 ? '\n' : ' ');
        }
    dec.finalize();
    return 0;
}

int mode_safedt(const Args& a, int rank) {
    if (rank == 0)
        std::printf("%.17g\n", safe_dt(a.num("dx", 1.0), a.num("dy", 1.0), a.num("vx", 0.0),
                                      a.num("vy", 0.0), a.num("D", 0.0)));
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    MPI_Init(&argc, &argv);
    int rank = 0, size = 1;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &size);
    int rc = 2;
    const std::string mode = argc > 1 ? argv[1] : "";
    const Args a = parse(argc, argv, 2);
    if (mode == "run")
        rc = mode_run(a, rank, size);
    else if (mode == "unit")
        rc = mode_unit(a, rank);
    else if (mode == "boundary")
        rc = mode_boundary(a, rank);
    else if (mode == "decomp")
        rc = mode_decomp(a, rank, size);
    else if (mode == "safedt")
        rc = mode_safedt(a, rank);
    else if (rank == 0)
        std::fprintf(stderr, "usage: ref_run run|unit|boundary|decomp|safedt --key=value ...\n");
    MPI_Finalize();
    return rc;
}

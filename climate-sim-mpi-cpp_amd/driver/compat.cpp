// compat.cpp — the reference's free functions (include/climate/*.hpp) implemented on the C ABI
// of include/csim.h.  This is the translation unit a maintainer links in place of the reference's
// src/{field,decomp,halo,boundary,diffusion,advection}.cpp; it contains no GPU code itself.
//
// The functions are stateless like the reference's: host Field in, host Field out.  Each call
// uploads its operands, runs the HIP kernel and downloads the result — correct for unit tests
// and unmodified callers, but PCIe-bound; a time loop should hold the field in HBM with
// climate::Stepper (include/climate/stepper.hpp).
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "climate/advection.hpp"
#include "climate/boundary.hpp"
#include "climate/decomp.hpp"
#include "climate/diffusion.hpp"
#include "climate/halo.hpp"
#include "climate/stepper.hpp"

namespace {

void ck(int rc) {
    if (rc == CSIM_OK) return;
    const std::string msg = csim_last_error();
    if (rc == CSIM_ERR_ARG) throw std::invalid_argument("csim: " + msg);
    throw std::runtime_error("csim: " + msg);
}

// RAII device mirror of one host Field
struct Mirror {
    csim_field* h = nullptr;
    explicit Mirror(const Field& f) { ck(csim_field_create(f.nx_local, f.ny_local, f.halo, f.dx, f.dy, &h)); }
    ~Mirror() { csim_field_destroy(h); }
    Mirror(const Mirror&) = delete;
    Mirror& operator=(const Mirror&) = delete;
    void up(const Field& f) { ck(csim_field_upload(h, f.data.data())); }
    void down(Field& f) { ck(csim_field_download(h, f.data.data())); }
};

int g_size = 0, g_rank = 0;  // 0 = not set

#ifndef CSIM_WITH_MPI
int env_int(const char* name, int fallback) {
    const char* v = std::getenv(name);
    return v && *v ? std::atoi(v) : fallback;
}
#endif

}  // namespace

namespace csim {

void set_world(int size, int rank) {
    g_size = size;
    g_rank = rank;
}

void get_world(MPI_Comm comm, int& size, int& rank) {
#ifdef CSIM_WITH_MPI
    MPI_Comm_size(comm, &size);
    MPI_Comm_rank(comm, &rank);
#else
    (void)comm;
    if (g_size > 0) {
        size = g_size;
        rank = g_rank;
        return;
    }
    size = env_int("WORLD_SIZE", env_int("OMPI_COMM_WORLD_SIZE", env_int("PMI_SIZE", 1)));
    rank = env_int("RANK", env_int("OMPI_COMM_WORLD_RANK", env_int("PMI_RANK", 0)));
#endif
}

}  // namespace csim

// reference src/decomp.cpp:5-34
void Decomp2D::init(MPI_Comm comm_world, int nx_global_, int ny_global_) {
    csim::get_world(comm_world, world_size, world_rank);
    csim_decomp d;
    ck(csim_decomp_init(world_size, world_rank, nx_global_, ny_global_, &d));
    cart_comm = comm_world;
    dims[0] = d.dims[0];
    dims[1] = d.dims[1];
    coords[0] = d.coords[0];
    coords[1] = d.coords[1];
    auto nb = [](int r) { return r == CSIM_NO_NEIGHBOR ? MPI_PROC_NULL : r; };
    nbr_lr[0] = nb(d.nbr[CSIM_LEFT]);
    nbr_lr[1] = nb(d.nbr[CSIM_RIGHT]);
    nbr_du[0] = nb(d.nbr[CSIM_BOTTOM]);
    nbr_du[1] = nb(d.nbr[CSIM_TOP]);
    nx_global = d.nx_global;
    ny_global = d.ny_global;
    nx_local = d.nx_local;
    ny_local = d.ny_local;
    x_offset = d.x_offset;
    y_offset = d.y_offset;
}

void Decomp2D::finalize() { cart_comm = MPI_COMM_NULL; }

csim_decomp Decomp2D::c_abi() const {
    csim_decomp d{};
    d.size = world_size;
    d.rank = world_rank;
    d.dims[0] = dims[0];
    d.dims[1] = dims[1];
    d.coords[0] = coords[0];
    d.coords[1] = coords[1];
    auto nb = [](int r) { return r == MPI_PROC_NULL ? CSIM_NO_NEIGHBOR : r; };
    d.nbr[CSIM_LEFT] = nb(nbr_lr[0]);
    d.nbr[CSIM_RIGHT] = nb(nbr_lr[1]);
    d.nbr[CSIM_BOTTOM] = nb(nbr_du[0]);
    d.nbr[CSIM_TOP] = nb(nbr_du[1]);
    d.nx_global = nx_global;
    d.ny_global = ny_global;
    d.nx_local = nx_local;
    d.ny_local = ny_local;
    d.x_offset = x_offset;
    d.y_offset = y_offset;
    return d;
}

// reference src/boundary.cpp:12-54
void apply_boundary(Field& f, const Decomp2D& dec, const BCConfig& bc, double value) {
    if (f.halo != 1) throw std::invalid_argument("csim: only halo == 1 is supported");
    const int codes[4] = {climate::bc_code(bc.left), climate::bc_code(bc.right),
                          climate::bc_code(bc.bottom), climate::bc_code(bc.top)};
    const int phys[4] = {dec.nbr_lr[0] == MPI_PROC_NULL, dec.nbr_lr[1] == MPI_PROC_NULL,
                         dec.nbr_du[0] == MPI_PROC_NULL, dec.nbr_du[1] == MPI_PROC_NULL};
    Mirror m(f);
    m.up(f);
    ck(csim_apply_boundary(m.h, codes, phys, value));
    m.down(f);
}

// reference src/diffusion.cpp:3-26 (every cell of `out` is written: interior + ring)
void diffusion_step(const Field& u, Field& out, double D, double dt) {
    Mirror mu(u), mo(out);
    mu.up(u);
    ck(csim_diffusion_step(mu.h, mo.h, D, dt));
    mo.down(out);
}

// reference src/advection.cpp:5-34 (accumulates onto `out`, so `out` is uploaded too)
void advection_step(const Field& u, Field& out, double vx, double vy, double dt) {
    Mirror mu(u), mo(out);
    mu.up(u);
    mo.up(out);
    ck(csim_advection_step(mu.h, mo.h, vx, vy, dt));
    mo.down(out);
}

// reference src/halo.cpp:6-50 on host fields
void exchange_halos(Field& f, const Decomp2D& dec, MPI_Comm comm) {
    const bool alone = dec.nbr_lr[0] == MPI_PROC_NULL && dec.nbr_lr[1] == MPI_PROC_NULL &&
                       dec.nbr_du[0] == MPI_PROC_NULL && dec.nbr_du[1] == MPI_PROC_NULL;
    if (alone) return;
#ifdef CSIM_WITH_MPI
    const int h = f.halo, nx = f.nx_local, ny = f.ny_local;
    std::vector<double> sl(ny), sr(ny), rl(ny), rr(ny);
    for (int j = 0; j < ny; ++j) {
        sl[j] = f.at(h, h + j);
        sr[j] = f.at(h + nx - 1, h + j);
    }
    MPI_Request rq[8];
    int n = 0;
    const int left = dec.nbr_lr[0], right = dec.nbr_lr[1], down = dec.nbr_du[0], up = dec.nbr_du[1];
    if (left != MPI_PROC_NULL) {
        MPI_Irecv(rl.data(), ny, MPI_DOUBLE, left, 100, comm, &rq[n++]);
        MPI_Isend(sl.data(), ny, MPI_DOUBLE, left, 101, comm, &rq[n++]);
    }
    if (right != MPI_PROC_NULL) {
        MPI_Irecv(rr.data(), ny, MPI_DOUBLE, right, 101, comm, &rq[n++]);
        MPI_Isend(sr.data(), ny, MPI_DOUBLE, right, 100, comm, &rq[n++]);
    }
    if (down != MPI_PROC_NULL) {  // rows are contiguous in the host layout (interior span)
        MPI_Irecv(&f.at(h, 0), nx, MPI_DOUBLE, down, 200, comm, &rq[n++]);
        MPI_Isend(&f.at(h, h), nx, MPI_DOUBLE, down, 201, comm, &rq[n++]);
    }
    if (up != MPI_PROC_NULL) {
        MPI_Irecv(&f.at(h, h + ny), nx, MPI_DOUBLE, up, 201, comm, &rq[n++]);
        MPI_Isend(&f.at(h, h + ny - 1), nx, MPI_DOUBLE, up, 200, comm, &rq[n++]);
    }
    MPI_Waitall(n, rq, MPI_STATUSES_IGNORE);
    for (int j = 0; j < ny; ++j) {
        if (left != MPI_PROC_NULL) f.at(0, h + j) = rl[j];
        if (right != MPI_PROC_NULL) f.at(h + nx, h + j) = rr[j];
    }
#else
    (void)f;
    (void)comm;
    throw std::runtime_error(
        "exchange_halos on host Fields across ranks needs a -DCSIM_WITH_MPI build; "
        "use climate::Stepper for the GPU-resident RCCL exchange");
#endif
}

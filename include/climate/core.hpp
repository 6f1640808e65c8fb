// include/climate/core.hpp — the reference's public types and free functions for the hot path,
// source-compatible, in ONE header: Field (reference include/field.hpp), Decomp2D
// (include/decomp.hpp), BCType / BCConfig / apply_boundary (include/boundary.hpp),
// diffusion_step (include/diffusion.hpp), advection_step (include/advection.hpp),
// exchange_halos (include/halo.hpp).  The reference's individual header names
// (field.hpp, decomp.hpp, ...) exist next to this file and simply forward here, so
// `#include "field.hpp"` etc. keep working with -Iinclude/climate.
//
// Every free function runs on the GPU through the C ABI (include/csim.h); it is stateless like
// the reference's (host Field in, host Field out: upload -> HIP kernel -> download).  A time loop
// should keep the field resident in HBM with climate::Stepper (stepper.hpp).
#pragma once
#include <cstddef>
#include <stdexcept>
#include <vector>

#include "csim.h"
#include "mpi_shim.hpp"

// ---- Field ----------------------------------------------------------------------------------
// Dense row-major fp64 array with a ghost ring of width `halo`; element (i, j) lives at
// data[j * nx_total() + i].  Members are public because callers copy / swap `data` directly
// (reference src/main.cpp:104,109); a bad index throws std::out_of_range (src/field.cpp:14-29).
struct Field {
    int nx_local, ny_local, halo;
    double dx, dy;
    std::vector<double> data;

    Field(int nx, int ny, int h, double dx_, double dy_)
        : nx_local(nx), ny_local(ny), halo(h), dx(dx_), dy(dy_),
          data(static_cast<std::size_t>(nx + 2 * h) * static_cast<std::size_t>(ny + 2 * h), 0.0) {}

    int nx_total() const { return nx_local + 2 * halo; }
    int ny_total() const { return ny_local + 2 * halo; }

    std::size_t idx(int i, int j) const {
        const bool inside = i >= 0 && j >= 0 && i < nx_total() && j < ny_total();
        if (!inside) throw std::out_of_range("Field index out of range");
        return static_cast<std::size_t>(j) * static_cast<std::size_t>(nx_total()) + static_cast<std::size_t>(i);
    }
    double& at(int i, int j) { return data.at(idx(i, j)); }
    const double& at(int i, int j) const { return data.at(idx(i, j)); }
    void fill(double value) { data.assign(data.size(), value); }
};

// ---- Decomp2D ---------------------------------------------------------------------------------
// 2D Cartesian block decomposition with the reference's field names (src/decomp.cpp:5-34).  The
// topology is re-derived without MPI (csim_decomp_init reproduces MPI_Dims_create +
// MPI_Cart_create(periods 0,0, reorder 0); pinned against the real MPI library in
// tests/golden/decomp_table.npz), so `cart_comm` is only kept for source compatibility.
struct Decomp2D {
    MPI_Comm cart_comm = MPI_COMM_NULL;
    int dims[2]{0, 0}, coords[2]{0, 0};             // dims[0] splits x, dims[1] splits y
    int nbr_lr[2]{MPI_PROC_NULL, MPI_PROC_NULL};    // x-, x+ neighbour ranks
    int nbr_du[2]{MPI_PROC_NULL, MPI_PROC_NULL};    // y-, y+ neighbour ranks
    int nx_global = 0, ny_global = 0;
    int nx_local = 0, ny_local = 0;                 // remainder goes to the last block
    int x_offset = 0, y_offset = 0;
    int world_size = 1, world_rank = 0;             // extension: what init() saw

    void init(MPI_Comm comm_world, int nx_global_, int ny_global_);
    void finalize();
    csim_decomp c_abi() const;                      // the same topology as the C ABI's struct
};

// ---- boundary conditions ------------------------------------------------------------------------
enum class BCType { Dirichlet, Neumann, Periodic };

struct BCConfig {
    BCType left = BCType::Dirichlet, right = BCType::Dirichlet;
    BCType bottom = BCType::Dirichlet, top = BCType::Dirichlet;
};

// ghost fill on the sides whose neighbour is MPI_PROC_NULL, order left,right,bottom,top
// (reference src/boundary.cpp:12-54; Periodic is a no-op there)
void apply_boundary(Field& f, const Decomp2D& dec, const BCConfig& bc, double value = 0.0);

// ---- the two numerical kernels --------------------------------------------------------------------
// FTCS 5-point diffusion of u into out + copy of the outer ring (reference src/diffusion.cpp:3-26)
void diffusion_step(const Field& u, Field& out, double D, double dt);
// first-order upwind advection ACCUMULATED onto out (reference src/advection.cpp:5-34)
void advection_step(const Field& u, Field& out, double vx, double vy, double dt);

// ---- halo exchange on a HOST field (reference src/halo.cpp:6-50) ------------------------------------
// One rank: nothing to do.  Several ranks: needs a -DCSIM_WITH_MPI build (the faces travel over
// MPI exactly like the reference); the GPU-resident exchange over RCCL/xGMI is climate::Stepper's.
void exchange_halos(Field& f, const Decomp2D& dec, MPI_Comm comm);

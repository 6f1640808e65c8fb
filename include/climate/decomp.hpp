// include/climate/decomp.hpp — mirror of reference include/decomp.hpp:4-17.  The topology is
// re-derived without MPI (csim_decomp_init reproduces MPI_Dims_create + MPI_Cart_create(periods
// 0,0, reorder 0), pinned against the real MPI library in tests/golden/decomp_table.npz), so
// `cart_comm` is kept only for source compatibility.
#pragma once
#include "csim.h"
#include "mpi_shim.hpp"

struct Decomp2D {
    MPI_Comm cart_comm = MPI_COMM_NULL;
    int dims[2]{0, 0};
    int coords[2]{0, 0};
    int nbr_lr[2]{MPI_PROC_NULL, MPI_PROC_NULL};
    int nbr_du[2]{MPI_PROC_NULL, MPI_PROC_NULL};

    int nx_global = 0, ny_global = 0;
    int nx_local = 0, ny_local = 0;
    int x_offset = 0, y_offset = 0;

    int world_size = 1, world_rank = 0;  // extension: what init() saw

    void init(MPI_Comm comm_world, int nx_global_, int ny_global_);
    void finalize();

    csim_decomp c_abi() const;  // the same topology as the C ABI's struct
};

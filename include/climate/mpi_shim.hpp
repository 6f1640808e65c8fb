// include/climate/mpi_shim.hpp — lets the reference-compatible headers compile with or
// without an MPI installation.  With -DCSIM_WITH_MPI the real <mpi.h> is used; without it the
// few MPI names the reference's public headers mention (include/decomp.hpp:2,5,8-9,
// include/halo.hpp:7) become plain ints, and the world size/rank come from the launcher's
// environment (WORLD_SIZE/RANK as set by torchrun, or PMI/OMPI variables) or csim::set_world().
#pragma once

#ifdef CSIM_WITH_MPI
#include <mpi.h>
#else
using MPI_Comm = int;
constexpr MPI_Comm MPI_COMM_NULL = -1;
constexpr MPI_Comm MPI_COMM_WORLD = 0;
constexpr int MPI_PROC_NULL = -1;
#endif

namespace csim {
// process-wide (size, rank) used by Decomp2D::init in builds without MPI
void set_world(int size, int rank);
void get_world(MPI_Comm comm, int& size, int& rank);
}  // namespace csim

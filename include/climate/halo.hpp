// include/climate/halo.hpp — mirror of reference include/halo.hpp:7.
#pragma once
#include "decomp.hpp"
#include "field.hpp"
#include "mpi_shim.hpp"

// 1-cell face exchange with the four Cartesian neighbours (reference src/halo.cpp:6-50) on a HOST
// field.  One rank: nothing to do.  Several ranks: needs a -DCSIM_WITH_MPI build (the faces
// travel over MPI exactly like the reference); the GPU-resident exchange over RCCL/xGMI lives in
// climate::Stepper, which is what a time loop should use.
void exchange_halos(Field& f, const Decomp2D& dec, MPI_Comm comm);

// include/climate/boundary.hpp — mirror of reference include/boundary.hpp:5-14.
#pragma once
#include "decomp.hpp"
#include "field.hpp"

enum class BCType { Dirichlet, Neumann, Periodic };

struct BCConfig {
    BCType left = BCType::Dirichlet;
    BCType right = BCType::Dirichlet;
    BCType bottom = BCType::Dirichlet;
    BCType top = BCType::Dirichlet;
};

// ghost fill on the sides whose neighbour is MPI_PROC_NULL, order left,right,bottom,top
// (reference src/boundary.cpp:12-54); runs on the GPU (csim_apply_boundary).
void apply_boundary(Field& f, const Decomp2D& dec, const BCConfig& bc, double value = 0.0);

// include/climate/advection.hpp — mirror of reference include/advection.hpp:4.
#pragma once
#include "field.hpp"

// first-order upwind advection ACCUMULATED onto out (reference src/advection.cpp:5-34), on the GPU.
void advection_step(const Field& u, Field& out, double vx, double vy, double dt);

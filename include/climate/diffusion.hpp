// include/climate/diffusion.hpp — mirror of reference include/diffusion.hpp:4.
#pragma once
#include "field.hpp"

// FTCS 5-point diffusion of u into out + ring copy (reference src/diffusion.cpp:3-26), on the GPU.
void diffusion_step(const Field& u, Field& out, double D, double dt);

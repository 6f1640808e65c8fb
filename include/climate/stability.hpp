// include/climate/stability.hpp — mirror of reference include/stability.hpp:5-16.
#pragma once
#include "csim.h"

inline double safe_dt(double dx, double dy, double vx, double vy, double D) {
    return csim_safe_dt(dx, dy, vx, vy, D);
}

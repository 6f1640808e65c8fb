// include/climate/field.hpp — source-compatible mirror of reference include/field.hpp:5-21.
// Same public members and methods (callers copy / swap `data` directly, reference
// src/main.cpp:104,109), same layout data[j*nx_total()+i], same std::out_of_range on a bad index
// (reference src/field.cpp:14-29).  The host vector stays the caller's storage; GPU work happens
// inside the free functions of diffusion.hpp / advection.hpp / boundary.hpp (stateless
// upload -> kernel -> download through include/csim.h) or, for the time loop, in climate::Stepper.
#pragma once
#include <cstddef>
#include <stdexcept>
#include <vector>

struct Field {
    int nx_local, ny_local;
    int halo;
    double dx, dy;
    std::vector<double> data;

    Field(int nx, int ny, int h, double dx_, double dy_)
        : nx_local(nx), ny_local(ny), halo(h), dx(dx_), dy(dy_),
          data(static_cast<std::size_t>(nx + 2 * h) * static_cast<std::size_t>(ny + 2 * h), 0.0) {}

    int nx_total() const { return nx_local + 2 * halo; }
    int ny_total() const { return ny_local + 2 * halo; }

    std::size_t idx(int i, int j) const {
        if (i < 0 || j < 0 || i >= nx_total() || j >= ny_total())
            throw std::out_of_range("Field index out of range");
        return static_cast<std::size_t>(j) * static_cast<std::size_t>(nx_total()) + static_cast<std::size_t>(i);
    }
    double& at(int i, int j) { return data.at(idx(i, j)); }
    const double& at(int i, int j) const { return data.at(idx(i, j)); }

    void fill(double value) { data.assign(data.size(), value); }
};

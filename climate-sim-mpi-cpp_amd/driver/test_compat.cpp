// test_compat.cpp — the reference's own unit-test bodies (tests/simulation/unit/
// test_{field,diffusion,advection,boundary,stability}.cpp) restated against the
// source-compatible headers in include/climate/, plus a Stepper-vs-free-function check.
// gtest is not available offline, so this is a plain executable: exit code 0 = all passed.
// Needs a GPU (the free functions run HIP kernels through the C ABI).
#include <cmath>
#include <cstdio>
#include <stdexcept>

#include "climate/advection.hpp"
#include "climate/boundary.hpp"
#include "climate/decomp.hpp"
#include "climate/diffusion.hpp"
#include "climate/field.hpp"
#include "climate/halo.hpp"
#include "climate/stability.hpp"
#include "climate/stepper.hpp"

static int g_fail = 0;
#define EXPECT(cond)                                                     \
    do {                                                                 \
        if (!(cond)) {                                                   \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);  \
            ++g_fail;                                                    \
        }                                                                \
    } while (0)

static void test_field() {  // reference test_field.cpp:5-26
    Field f(4, 3, 1, 1.0, 1.0);
    EXPECT(f.data.size() == static_cast<size_t>(f.nx_total() * f.ny_total()));
    for (int j = 0; j < f.ny_total(); ++j)
        for (int i = 0; i < f.nx_total(); ++i) f.at(i, j) = 10.0 * j + i;
    EXPECT(f.data[1 * f.nx_total() + 2] == 12.0);
    bool threw = false;
    try {
        f.at(-1, 0) = 1.0;
    } catch (const std::out_of_range&) {
        threw = true;
    }
    EXPECT(threw);
    threw = false;
    try {
        f.at(f.nx_total(), 0) = 1.0;
    } catch (const std::out_of_range&) {
        threw = true;
    }
    EXPECT(threw);
}

static void test_diffusion() {  // reference test_diffusion.cpp:17-34
    Field u(3, 3, 1, 1.0, 1.0), v(3, 3, 1, 1.0, 1.0);
    u.at(2, 2) = 1.0;
    const double D = 0.1, dt = 0.1, alpha = D * dt / (u.dx * u.dx);
    diffusion_step(u, v, D, dt);
    EXPECT(std::fabs(v.at(2, 2) - (1.0 - 4 * alpha)) < 1e-12);
    EXPECT(std::fabs(v.at(1, 2) - alpha) < 1e-12);
    EXPECT(std::fabs(v.at(3, 2) - alpha) < 1e-12);
    EXPECT(std::fabs(v.at(2, 1) - alpha) < 1e-12);
    EXPECT(std::fabs(v.at(2, 3) - alpha) < 1e-12);
}

static void test_advection() {  // reference test_advection.cpp:13-71
    const int nx = 8, ny = 8;
    Field u(nx, ny, 1, 1.0, 1.0);
    u.fill(0.0);
    u.at(nx / 2 + 1, ny / 2 + 1) = 1.0;
    const double vs[5][2] = {{0, 0}, {1, 0}, {-1, 0}, {0, 1}, {0, -1}};
    for (int k = 0; k < 5; ++k) {
        Field out(nx, ny, 1, 1.0, 1.0);
        out.fill(0.0);
        advection_step(u, out, vs[k][0], vs[k][1], 0.1);
        if (k == 0) {
            for (int j = 1; j <= ny; ++j)
                for (int i = 1; i <= nx; ++i) EXPECT(out.at(i, j) == 0.0);
        } else {
            EXPECT(out.at(nx / 2 + 1, ny / 2 + 1) != 0.0);
        }
    }
}

static void test_boundary() {  // reference test_boundary.cpp:8-69
    const int NX = 4, NY = 3, h = 1;
    Decomp2D dec;
    csim::set_world(1, 0);
    dec.init(MPI_COMM_WORLD, NX, NY);
    EXPECT(dec.dims[0] * dec.dims[1] == 1);
    Field f(NX, NY, h, 1.0, 1.0);
    f.fill(-1.0);
    for (int j = h; j < h + NY; ++j)
        for (int i = h; i < h + NX; ++i) f.at(i, j) = 10.0;
    BCConfig dir;
    apply_boundary(f, dec, dir, 5.0);
    for (int j = 0; j < f.ny_total(); ++j) {
        EXPECT(f.at(0, j) == 5.0);
        EXPECT(f.at(h + NX, j) == 5.0);
    }
    for (int i = 0; i < f.nx_total(); ++i) {
        EXPECT(f.at(i, 0) == 5.0);
        EXPECT(f.at(i, h + NY) == 5.0);
    }
    f.fill(-1.0);
    for (int j = h; j < h + NY; ++j)
        for (int i = h; i < h + NX; ++i) f.at(i, j) = static_cast<double>(j);
    BCConfig neu;
    neu.left = neu.right = neu.bottom = neu.top = BCType::Neumann;
    apply_boundary(f, dec, neu, 0.0);
    for (int j = 0; j < f.ny_total(); ++j) {
        EXPECT(f.at(0, j) == f.at(h, j));
        EXPECT(f.at(h + NX, j) == f.at(h + NX - 1, j));
    }
    for (int i = 0; i < f.nx_total(); ++i) {
        EXPECT(f.at(i, 0) == f.at(i, h));
        EXPECT(f.at(i, h + NY) == f.at(i, h + NY - 1));
    }
    exchange_halos(f, dec, MPI_COMM_WORLD);  // single rank: no-op
    dec.finalize();
}

static void test_stability() {  // reference test_stability.cpp:5-27
    EXPECT(safe_dt(1, 1, 0.5, 0.5, 0.1) > 0);
    EXPECT(safe_dt(1, 1, 2.0, 0, 0) < safe_dt(1, 1, 1.0, 0, 0));
    EXPECT(safe_dt(1, 1, 0, 0, 2.0) < safe_dt(1, 1, 0, 0, 1.0));
}

// the reference's loop body vs climate::Stepper on the same input: bit-identical
static void test_stepper_matches_reference_loop() {
    const int nx = 256, ny = 96, steps = 7;
    const double D = 0.05, vx = 0.5, vy = -0.25, dt = 0.1;
    Decomp2D dec;
    csim::set_world(1, 0);
    dec.init(MPI_COMM_WORLD, nx, ny);
    BCConfig bc;
    bc.right = BCType::Neumann;
    bc.bottom = BCType::Periodic;
    Field u(nx, ny, 1, 1.0, 1.0), tmp(nx, ny, 1, 1.0, 1.0), w(nx, ny, 1, 1.0, 1.0);
    unsigned s = 12345u;
    for (int j = 1; j <= ny; ++j)
        for (int i = 1; i <= nx; ++i) {
            s = s * 1664525u + 1013904223u;
            u.at(i, j) = (s >> 8) / 16777216.0;
        }
    climate::Stepper st(dec, bc, 1.0, 1.0);
    st.upload(u);
    st.run(D, dt, vx, vy, steps);
    st.download(w);
    for (int n = 0; n < steps; ++n) {  // reference src/main.cpp:101-109 with the free functions
        exchange_halos(u, dec, MPI_COMM_WORLD);
        apply_boundary(u, dec, bc, 0.0);
        std::copy(u.data.begin(), u.data.end(), tmp.data.begin());
        diffusion_step(u, tmp, D, dt);
        advection_step(u, tmp, vx, vy, dt);
        std::swap(u.data, tmp.data);
    }
    size_t bad = 0;
    for (size_t k = 0; k < u.data.size(); ++k) bad += (u.data[k] != w.data[k]);
    EXPECT(bad == 0);
}

int main() {
    int ndev = 0;
    if (csim_device_count(&ndev) != CSIM_OK || ndev < 1) {
        std::printf("no GPU: %s\n", csim_last_error());
        return 77;
    }
    test_field();
    test_diffusion();
    test_advection();
    test_boundary();
    test_stability();
    test_stepper_matches_reference_loop();
    std::printf("test_compat: %s (%d failures)\n", g_fail ? "FAILED" : "all passed", g_fail);
    return g_fail ? 1 : 0;
}

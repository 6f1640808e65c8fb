// test_compat_mpi.cpp — the reference's two MPI unit tests restated against the drop-in headers
// (include/climate/) in the -DCSIM_WITH_MPI build, run under `mpirun -np P`:
//   Unit_Decomp.GridDimsAndNeighbors  (reference tests/simulation/unit/test_decomp_mpi.cpp:6-35)
//   Unit_Halo.AdaptiveFaces           (reference tests/simulation/unit/test_halo.cpp:8-66)
// plus two checks the reference's own tests leave open: the decomposition against MPI's OWN Cartesian
// topology (MPI_Dims_create + MPI_Cart_create / MPI_Cart_shift, the calls of reference src/decomp.cpp:9-22, made
// here directly), and the physical ghost lines / corner ghosts being left untouched by exchange_halos.
// Host code only: exchange_halos on host Fields is the MPI branch of driver/compat.cpp; no GPU is touched.
// gtest is not available offline, so this is a plain executable: exit code 0 = all passed on this rank.
#include <mpi.h>

#include <cstdio>

#include "climate/decomp.hpp"
#include "climate/field.hpp"
#include "climate/halo.hpp"

static int g_fail = 0, g_rank = 0;
#define EXPECT(cond)                                                                     \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            std::printf("FAIL rank %d %s:%d  %s\n", g_rank, __FILE__, __LINE__, #cond); \
            ++g_fail;                                                                    \
        }                                                                                \
    } while (0)

static void test_decomp(int world_size) {  // reference test_decomp_mpi.cpp:6-35
    Decomp2D d;
    d.init(MPI_COMM_WORLD, 16, 12);
    EXPECT(d.dims[0] * d.dims[1] == world_size);
    EXPECT(d.coords[0] >= 0 && d.coords[1] >= 0);
    EXPECT(d.coords[0] < d.dims[0] && d.coords[1] < d.dims[1]);
    const bool on_left = d.coords[0] == 0, on_right = d.coords[0] == d.dims[0] - 1;
    const bool on_down = d.coords[1] == 0, on_up = d.coords[1] == d.dims[1] - 1;
    if (!on_left) EXPECT(d.nbr_lr[0] != MPI_PROC_NULL);
    if (!on_right) EXPECT(d.nbr_lr[1] != MPI_PROC_NULL);
    if (!on_down) EXPECT(d.nbr_du[0] != MPI_PROC_NULL);
    if (!on_up) EXPECT(d.nbr_du[1] != MPI_PROC_NULL);
    // and the edges the other way round, which the reference's test leaves open
    if (on_left) EXPECT(d.nbr_lr[0] == MPI_PROC_NULL);
    if (on_right) EXPECT(d.nbr_lr[1] == MPI_PROC_NULL);
    if (on_down) EXPECT(d.nbr_du[0] == MPI_PROC_NULL);
    if (on_up) EXPECT(d.nbr_du[1] == MPI_PROC_NULL);

    // the same numbers from MPI itself (reference src/decomp.cpp:9-22)
    int dims[2] = {0, 0}, periods[2] = {0, 0}, coords[2] = {0, 0}, lr[2], du[2], cart_rank = -1;
    MPI_Dims_create(world_size, 2, dims);
    MPI_Comm cart;
    MPI_Cart_create(MPI_COMM_WORLD, 2, dims, periods, 0, &cart);
    MPI_Comm_rank(cart, &cart_rank);
    MPI_Cart_coords(cart, cart_rank, 2, coords);
    MPI_Cart_shift(cart, 0, 1, &lr[0], &lr[1]);
    MPI_Cart_shift(cart, 1, 1, &du[0], &du[1]);
    EXPECT(cart_rank == g_rank);
    EXPECT(d.dims[0] == dims[0] && d.dims[1] == dims[1]);
    EXPECT(d.coords[0] == coords[0] && d.coords[1] == coords[1]);
    EXPECT(d.nbr_lr[0] == lr[0] && d.nbr_lr[1] == lr[1] && d.nbr_du[0] == du[0] && d.nbr_du[1] == du[1]);
    // block sizes and offsets (reference src/decomp.cpp:24-33): remainder on the last rank of an axis
    const int bx = 16 / dims[0], by = 12 / dims[1];
    EXPECT(d.nx_local == bx + (coords[0] == dims[0] - 1 ? 16 % dims[0] : 0));
    EXPECT(d.ny_local == by + (coords[1] == dims[1] - 1 ? 12 % dims[1] : 0));
    EXPECT(d.x_offset == coords[0] * bx && d.y_offset == coords[1] * by);
    MPI_Comm_free(&cart);
    d.finalize();
}

static void test_halo(int rank) {  // reference test_halo.cpp:24-62
    const int NXG = 8, NYG = 8;
    Decomp2D dec;
    dec.init(MPI_COMM_WORLD, NXG, NYG);
    const int h = 1;
    Field f(dec.nx_local, dec.ny_local, h, 1.0, 1.0);
    f.fill(-1.0);
    for (int j = h; j < h + dec.ny_local; ++j)
        for (int i = h; i < h + dec.nx_local; ++i) f.at(i, j) = static_cast<double>(rank);
    exchange_halos(f, dec, MPI_COMM_WORLD);
    if (dec.nbr_lr[0] != MPI_PROC_NULL)
        for (int j = h; j < h + dec.ny_local; ++j) EXPECT(f.at(0, j) == static_cast<double>(dec.nbr_lr[0]));
    if (dec.nbr_lr[1] != MPI_PROC_NULL)
        for (int j = h; j < h + dec.ny_local; ++j) EXPECT(f.at(h + dec.nx_local, j) == static_cast<double>(dec.nbr_lr[1]));
    if (dec.nbr_du[0] != MPI_PROC_NULL)
        for (int i = h; i < h + dec.nx_local; ++i) EXPECT(f.at(i, 0) == static_cast<double>(dec.nbr_du[0]));
    if (dec.nbr_du[1] != MPI_PROC_NULL)
        for (int i = h; i < h + dec.nx_local; ++i) EXPECT(f.at(i, h + dec.ny_local) == static_cast<double>(dec.nbr_du[1]));
    // what the reference's test does not look at: physical sides keep their fill value (src/halo.cpp posts nothing
    // towards MPI_PROC_NULL), the interior is untouched, and so are the four corner ghosts
    if (dec.nbr_lr[0] == MPI_PROC_NULL)
        for (int j = h; j < h + dec.ny_local; ++j) EXPECT(f.at(0, j) == -1.0);
    if (dec.nbr_lr[1] == MPI_PROC_NULL)
        for (int j = h; j < h + dec.ny_local; ++j) EXPECT(f.at(h + dec.nx_local, j) == -1.0);
    if (dec.nbr_du[0] == MPI_PROC_NULL)
        for (int i = h; i < h + dec.nx_local; ++i) EXPECT(f.at(i, 0) == -1.0);
    if (dec.nbr_du[1] == MPI_PROC_NULL)
        for (int i = h; i < h + dec.nx_local; ++i) EXPECT(f.at(i, h + dec.ny_local) == -1.0);
    for (int j = h; j < h + dec.ny_local; ++j)
        for (int i = h; i < h + dec.nx_local; ++i) EXPECT(f.at(i, j) == static_cast<double>(rank));
    EXPECT(f.at(0, 0) == -1.0 && f.at(h + dec.nx_local, 0) == -1.0);
    EXPECT(f.at(0, h + dec.ny_local) == -1.0 && f.at(h + dec.nx_local, h + dec.ny_local) == -1.0);
    dec.finalize();
}

int main(int argc, char** argv) {
    MPI_Init(&argc, &argv);
    int size = 0;
    MPI_Comm_rank(MPI_COMM_WORLD, &g_rank);
    MPI_Comm_size(MPI_COMM_WORLD, &size);
    test_decomp(size);
    if (size >= 2) test_halo(g_rank);  // (the reference skips AdaptiveFaces on one rank, test_halo.cpp:20-23)
    int total = 0;
    MPI_Allreduce(&g_fail, &total, 1, MPI_INT, MPI_SUM, MPI_COMM_WORLD);
    if (g_rank == 0) std::printf("test_compat_mpi: %d ranks, %s (%d failures)\n", size, total ? "FAILED" : "all passed", total);
    MPI_Finalize();
    return total ? 1 : 0;
}

/* oracle/cpu_stepper.c — TEST INFRASTRUCTURE (the parity checker), NOT PRODUCT.
 *
 * A clean-room CPU restatement, in plain C, of the one hot path of
 * antoniorizzoeng/climate-sim-mpi-cpp:  per time step
 *     exchange_halos -> apply_boundary -> copy -> diffusion_step -> advection_step -> swap
 * (reference src/main.cpp:101-109).  Each function cites the reference lines it follows.
 * It keeps the reference's structure (separate copy / diffusion / advection passes over a
 * dense row-major array with a 1-cell ghost ring) and the reference's exact floating-point
 * association order, and is built with -ffp-contract=off so that it is bit-identical to
 * the reference objects (pinned by tests/golden/, generated from oracle/_ref/ref_run).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (include/csim.h, climate-sim-mpi-cpp_amd/) never does.
 *
 * Layout (reference src/field.cpp:20-25): element (i,j), 0<=i<nx+2, 0<=j<ny+2, lives at
 * f[j*(nx+2)+i]; i is the contiguous axis; halo width is 1 (reference src/main.cpp:65).
 */
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define AT(f, i, j, nxt) ((f)[(size_t)(j) * (size_t)(nxt) + (size_t)(i)])

enum { ORA_DIRICHLET = 0, ORA_NEUMANN = 1, ORA_PERIODIC = 2 };
enum { ORA_LEFT = 0, ORA_RIGHT = 1, ORA_BOTTOM = 2, ORA_TOP = 3 };

/* ---- stability (reference include/stability.hpp:5-16) ------------------------------- */
double ora_safe_dt(double dx, double dy, double vx, double vy, double D) {
    const double ax = fabs(vx), ay = fabs(vy);
    const double adv = (ax > 0 ? ax / dx : 0.0) + (ay > 0 ? ay / dy : 0.0);
    const double dt_adv = adv > 0 ? 1.0 / adv : INFINITY;
    const double lapw = 1.0 / (dx * dx) + 1.0 / (dy * dy);
    const double dt_dif = D > 0 ? 1.0 / (2.0 * D * lapw) : INFINITY;
    return dt_adv < dt_dif ? dt_adv : dt_dif;
}

/* ---- boundary fill (reference src/boundary.cpp:12-54) -------------------------------
 * Sides in the order left, right, bottom, top; only sides flagged physical (neighbour ==
 * MPI_PROC_NULL) are touched; columns span every row j=0..ny+1 and rows every i=0..nx+1,
 * so corners end with the row rule.  Periodic does nothing. */
void ora_apply_boundary(double* f, int nx, int ny, const int bc[4], const int phys[4],
                        double value) {
    const int nxt = nx + 2, nyt = ny + 2;
    if (phys[ORA_LEFT]) {
        if (bc[ORA_LEFT] == ORA_DIRICHLET)
            for (int j = 0; j < nyt; ++j) AT(f, 0, j, nxt) = value;
        else if (bc[ORA_LEFT] == ORA_NEUMANN)
            for (int j = 0; j < nyt; ++j) AT(f, 0, j, nxt) = AT(f, 1, j, nxt);
    }
    if (phys[ORA_RIGHT]) {
        if (bc[ORA_RIGHT] == ORA_DIRICHLET)
            for (int j = 0; j < nyt; ++j) AT(f, nx + 1, j, nxt) = value;
        else if (bc[ORA_RIGHT] == ORA_NEUMANN)
            for (int j = 0; j < nyt; ++j) AT(f, nx + 1, j, nxt) = AT(f, nx, j, nxt);
    }
    if (phys[ORA_BOTTOM]) {
        if (bc[ORA_BOTTOM] == ORA_DIRICHLET)
            for (int i = 0; i < nxt; ++i) AT(f, i, 0, nxt) = value;
        else if (bc[ORA_BOTTOM] == ORA_NEUMANN)
            for (int i = 0; i < nxt; ++i) AT(f, i, 0, nxt) = AT(f, i, 1, nxt);
    }
    if (phys[ORA_TOP]) {
        if (bc[ORA_TOP] == ORA_DIRICHLET)
            for (int i = 0; i < nxt; ++i) AT(f, i, ny + 1, nxt) = value;
        else if (bc[ORA_TOP] == ORA_NEUMANN)
            for (int i = 0; i < nxt; ++i) AT(f, i, ny + 1, nxt) = AT(f, i, ny, nxt);
    }
}

/* ---- diffusion (reference src/diffusion.cpp:3-26) -----------------------------------
 * out = u + (dt*D) * ( (E - 2u + W)/(dx*dx) + (N - 2u + S)/(dy*dy) ) on the interior, then
 * the outer ring of u is copied to out. "E - 2.0*c + W" associates as (E - 2.0*c) + W and
 * "dt * D * lap" as (dt*D)*lap. */
void ora_diffusion_step(const double* u, double* out, int nx, int ny, double dx, double dy,
                        double D, double dt) {
    const int nxt = nx + 2, nyt = ny + 2;
    const double dx2 = dx * dx, dy2 = dy * dy;
    const double k = dt * D;
    for (int j = 1; j <= ny; ++j) {
        const double* rs = u + (size_t)(j - 1) * nxt;
        const double* rc = u + (size_t)j * nxt;
        const double* rn = u + (size_t)(j + 1) * nxt;
        double* ro = out + (size_t)j * nxt;
        for (int i = 1; i <= nx; ++i) {
            const double c = rc[i];
            const double lap = (rc[i + 1] - 2.0 * c + rc[i - 1]) / dx2 +
                               (rn[i] - 2.0 * c + rs[i]) / dy2;
            ro[i] = c + k * lap;
        }
    }
    for (int i = 0; i < nxt; ++i) {
        AT(out, i, 0, nxt) = AT(u, i, 0, nxt);
        AT(out, i, nyt - 1, nxt) = AT(u, i, nyt - 1, nxt);
    }
    for (int j = 0; j < nyt; ++j) {
        AT(out, 0, j, nxt) = AT(u, 0, j, nxt);
        AT(out, nxt - 1, j, nxt) = AT(u, nxt - 1, j, nxt);
    }
}

/* ---- advection (reference src/advection.cpp:5-34) ------------------------------------
 * First-order upwind, ACCUMULATED onto out: out += (-dt) * (vx*dudx + vy*dudy);
 * backward difference when v >= 0, forward otherwise. */
void ora_advection_step(const double* u, double* out, int nx, int ny, double dx, double dy,
                        double vx, double vy, double dt) {
    const int nxt = nx + 2;
    const double mdt = -dt;
    for (int j = 1; j <= ny; ++j) {
        const double* rs = u + (size_t)(j - 1) * nxt;
        const double* rc = u + (size_t)j * nxt;
        const double* rn = u + (size_t)(j + 1) * nxt;
        double* ro = out + (size_t)j * nxt;
        for (int i = 1; i <= nx; ++i) {
            const double c = rc[i];
            const double dudx = (vx >= 0.0) ? (c - rc[i - 1]) / dx : (rc[i + 1] - c) / dx;
            const double dudy = (vy >= 0.0) ? (c - rs[i]) / dy : (rn[i] - c) / dy;
            const double adv = vx * dudx + vy * dudy;
            ro[i] += mdt * adv;
        }
    }
}

/* ---- bounds-checked accessor flavour (CPU-baseline variant only) ------------------------
 * The reference reaches every element through Field::at = check_bounds (i, j against the local
 * extents, src/field.cpp:14-18) + std::vector::at (index against size, :27-29).  These two
 * functions restate diffusion_step / advection_step with that double check on all 5 + 1 and 3 + 2
 * accesses per cell, so that bench.py can report the port both ways ("checked" is how the reference
 * objects behave, "unchecked" what the same loops cost without the accessor). Same arithmetic. */
static inline size_t idx_checked(int i, int j, int nxt, int nyt) {
    if (i < 0 || i >= nxt || j < 0 || j >= nyt) abort();       /* check_bounds */
    const size_t k = (size_t)j * (size_t)nxt + (size_t)i;
    if (k >= (size_t)nxt * (size_t)nyt) abort();                /* vector::at */
    return k;
}
#define ATC(f, i, j) ((f)[idx_checked((i), (j), nxt, nyt)])

void ora_diffusion_step_checked(const double* u, double* out, int nx, int ny, double dx, double dy,
                                double D, double dt) {
    const int nxt = nx + 2, nyt = ny + 2;
    for (int j = 1; j <= ny; ++j)
        for (int i = 1; i <= nx; ++i) {
            const double uij = ATC(u, i, j);
            const double lap = (ATC(u, i + 1, j) - 2.0 * uij + ATC(u, i - 1, j)) / (dx * dx) +
                               (ATC(u, i, j + 1) - 2.0 * uij + ATC(u, i, j - 1)) / (dy * dy);
            ATC(out, i, j) = uij + dt * D * lap;
        }
    for (int i = 0; i < nxt; ++i) {
        ATC(out, i, 0) = ATC(u, i, 0);
        ATC(out, i, nyt - 1) = ATC(u, i, nyt - 1);
    }
    for (int j = 0; j < nyt; ++j) {
        ATC(out, 0, j) = ATC(u, 0, j);
        ATC(out, nxt - 1, j) = ATC(u, nxt - 1, j);
    }
}

void ora_advection_step_checked(const double* u, double* out, int nx, int ny, double dx, double dy,
                                double vx, double vy, double dt) {
    const int nxt = nx + 2, nyt = ny + 2;
    for (int j = 1; j <= ny; ++j)
        for (int i = 1; i <= nx; ++i) {
            double dudx, dudy;
            if (vx >= 0.0)
                dudx = (ATC(u, i, j) - ATC(u, i - 1, j)) / dx;
            else
                dudx = (ATC(u, i + 1, j) - ATC(u, i, j)) / dx;
            if (vy >= 0.0)
                dudy = (ATC(u, i, j) - ATC(u, i, j - 1)) / dy;
            else
                dudy = (ATC(u, i, j + 1) - ATC(u, i, j)) / dy;
            const double adv = vx * dudx + vy * dudy;
            ATC(out, i, j) += (-dt) * adv;
        }
}

/* One reference time step on ONE tile whose ghosts already hold neighbour data on the
 * non-physical sides: boundary -> copy -> diffusion -> advection (src/main.cpp:102-107).
 * The caller swaps u and tmp afterwards (src/main.cpp:109). */
void ora_step_tile(double* u, double* tmp, int nx, int ny, double dx, double dy, double D,
                   double vx, double vy, double dt, const int bc[4], const int phys[4]) {
    ora_apply_boundary(u, nx, ny, bc, phys, 0.0);
    memcpy(tmp, u, sizeof(double) * (size_t)(nx + 2) * (size_t)(ny + 2));
    ora_diffusion_step(u, tmp, nx, ny, dx, dy, D, dt);
    ora_advection_step(u, tmp, nx, ny, dx, dy, vx, vy, dt);
}

/* Single-tile convenience: `steps` full steps in place (result in u). */
void ora_run_single(double* u, int nx, int ny, double dx, double dy, double D, double vx,
                    double vy, double dt, const int bc[4], int steps) {
    const size_t n = (size_t)(nx + 2) * (size_t)(ny + 2);
    double* a = u;
    double* b = (double*)malloc(n * sizeof(double));
    const int phys[4] = {1, 1, 1, 1};
    memcpy(b, u, n * sizeof(double));
    for (int s = 0; s < steps; ++s) {
        ora_step_tile(a, b, nx, ny, dx, dy, D, vx, vy, dt, bc, phys);
        double* t = a;
        a = b;
        b = t;
    }
    if (a != u) {
        memcpy(u, a, n * sizeof(double));
        free(a);
    } else {
        free(b);
    }
}

/* ---- decomposition (reference src/decomp.cpp:5-34) -----------------------------------
 * MPI_Dims_create(size, 2) restated: the most balanced factor pair, non-increasing
 * (1->1x1, 2->2x1, 4->2x2, 8->4x2); pinned against the real MPI library by
 * tests/golden/decomp_table.npz.  Cartesian ranks are row-major over (coords[0],coords[1])
 * with dims[0] splitting x; no periodic wrap; remainder goes to the last block. */
void ora_dims_create(int size, int dims[2]) {
    int b = 1;
    for (int f = 1; (long)f * f <= size; ++f)
        if (size % f == 0) b = f;
    dims[0] = size / b;
    dims[1] = b;
}

/* out[12] = dims0 dims1 cx cy left right down up nx_local ny_local x_off y_off (-1 = none) */
void ora_decomp(int size, int rank, int nxg, int nyg, int out[12]) {
    int dims[2];
    ora_dims_create(size, dims);
    const int cx = rank / dims[1], cy = rank % dims[1];
    const int bx = nxg / dims[0], by = nyg / dims[1];
    out[0] = dims[0];
    out[1] = dims[1];
    out[2] = cx;
    out[3] = cy;
    out[4] = cx > 0 ? (cx - 1) * dims[1] + cy : -1;
    out[5] = cx < dims[0] - 1 ? (cx + 1) * dims[1] + cy : -1;
    out[6] = cy > 0 ? cx * dims[1] + (cy - 1) : -1;
    out[7] = cy < dims[1] - 1 ? cx * dims[1] + (cy + 1) : -1;
    out[8] = bx + (cx == dims[0] - 1 ? nxg % dims[0] : 0);
    out[9] = by + (cy == dims[1] - 1 ? nyg % dims[1] : 0);
    out[10] = cx * bx;
    out[11] = cy * by;
}

/* ---- gaussian hotspot (reference src/init.cpp:12-33; NEXT-1 row, used for inputs) ---- */
void ora_gaussian(double* f, int nx, int ny, int x_off, int y_off, int nxg, int nyg, double dx,
                  double dy, double A, double sigma_frac, double xc_frac, double yc_frac) {
    const int nxt = nx + 2;
    const double Lx = nxg * dx, Ly = nyg * dy;
    const double xc = xc_frac * Lx, yc = yc_frac * Ly;
    const double sig = sigma_frac * (Lx < Ly ? Lx : Ly);
    for (int j = 0; j < ny; ++j) {
        const double y = (y_off + j + 0.5) * dy;
        for (int i = 0; i < nx; ++i) {
            const double x = (x_off + i + 0.5) * dx;
            const double r2 = (x - xc) * (x - xc) + (y - yc) * (y - yc);
            AT(f, i + 1, j + 1, nxt) = A * exp(-r2 / (2.0 * sig * sig));
        }
    }
}

/* ---- multi-tile world: emulates `mpirun -np size` inside one process ------------------
 * Tiles exchange 1-cell faces exactly like reference src/halo.cpp:28-43 (interior span of
 * each face; the reference's row messages also carry the two corner ghosts, whose values
 * are formally undefined there and never read by the stencils — SURVEY Q7 — so corners are
 * left alone here).  One thread per tile when threads>1. */
typedef struct {
    int nx, ny, xo, yo;
    int nbr[4]; /* left right down up, -1 = physical edge */
    double *u, *tmp;
} ora_tile;

typedef struct {
    int size, nxg, nyg;
    double dx, dy;
    ora_tile* t;
} ora_world;

ora_world* ora_world_create(int size, int nxg, int nyg, double dx, double dy) {
    ora_world* w = (ora_world*)calloc(1, sizeof(ora_world));
    w->size = size;
    w->nxg = nxg;
    w->nyg = nyg;
    w->dx = dx;
    w->dy = dy;
    w->t = (ora_tile*)calloc((size_t)size, sizeof(ora_tile));
    for (int r = 0; r < size; ++r) {
        int d[12];
        ora_decomp(size, r, nxg, nyg, d);
        ora_tile* t = &w->t[r];
        t->nx = d[8];
        t->ny = d[9];
        t->xo = d[10];
        t->yo = d[11];
        for (int k = 0; k < 4; ++k) t->nbr[k] = d[4 + k];
        const size_t n = (size_t)(t->nx + 2) * (size_t)(t->ny + 2);
        t->u = (double*)calloc(n, sizeof(double));
        t->tmp = (double*)calloc(n, sizeof(double));
    }
    return w;
}

void ora_world_destroy(ora_world* w) {
    if (!w) return;
    for (int r = 0; r < w->size; ++r) {
        free(w->t[r].u);
        free(w->t[r].tmp);
    }
    free(w->t);
    free(w);
}

/* global interior (nyg x nxg, row-major) -> tiles */
void ora_world_scatter(ora_world* w, const double* g) {
    for (int r = 0; r < w->size; ++r) {
        ora_tile* t = &w->t[r];
        for (int j = 0; j < t->ny; ++j)
            memcpy(&AT(t->u, 1, j + 1, t->nx + 2), g + (size_t)(t->yo + j) * w->nxg + t->xo,
                   sizeof(double) * (size_t)t->nx);
    }
}

void ora_world_gather(const ora_world* w, double* g) {
    for (int r = 0; r < w->size; ++r) {
        const ora_tile* t = &w->t[r];
        for (int j = 0; j < t->ny; ++j)
            memcpy(g + (size_t)(t->yo + j) * w->nxg + t->xo, &AT(t->u, 1, j + 1, t->nx + 2),
                   sizeof(double) * (size_t)t->nx);
    }
}

/* the world as ONE array with ghost ring, (nyg+2) x (nxg+2): interiors of all tiles plus the ghost
 * lines of their PHYSICAL sides (span 1..n of the tile; the four global corners come from the
 * corner tiles) — i.e. the full local array a 1-rank run of the reference would hold, which is
 * decomposition-invariant (each ghost cell depends only on the boundary rule and its adjacent
 * interior cell; the reference's side order L,R,B,T only matters at the global corners). */
void ora_world_gather_full(const ora_world* w, double* g) {
    const int gx = w->nxg + 2;
    for (int r = 0; r < w->size; ++r) {
        const ora_tile* t = &w->t[r];
        const int nxt = t->nx + 2;
        const int pl = t->nbr[ORA_LEFT] < 0, pr = t->nbr[ORA_RIGHT] < 0;
        const int pb = t->nbr[ORA_BOTTOM] < 0, pt = t->nbr[ORA_TOP] < 0;
        for (int j = 0; j <= t->ny + 1; ++j) {
            const int row_ok = (j >= 1 && j <= t->ny) || (j == 0 && pb) || (j == t->ny + 1 && pt);
            if (!row_ok) continue;
            for (int i = 0; i <= t->nx + 1; ++i) {
                const int col_ok = (i >= 1 && i <= t->nx) || (i == 0 && pl) || (i == t->nx + 1 && pr);
                if (col_ok) AT(g, t->xo + i, t->yo + j, gx) = AT(t->u, i, j, nxt);
            }
        }
    }
}

void ora_world_gaussian(ora_world* w, double A, double sigma_frac, double xc_frac,
                        double yc_frac) {
    for (int r = 0; r < w->size; ++r) {
        ora_tile* t = &w->t[r];
        ora_gaussian(t->u, t->nx, t->ny, t->xo, t->yo, w->nxg, w->nyg, w->dx, w->dy, A,
                     sigma_frac, xc_frac, yc_frac);
    }
}

/* copy of tile r's full local array (ghosts included) */
void ora_world_tile_shape(const ora_world* w, int r, int out[4]) {
    out[0] = w->t[r].nx;
    out[1] = w->t[r].ny;
    out[2] = w->t[r].xo;
    out[3] = w->t[r].yo;
}
void ora_world_tile_get(const ora_world* w, int r, double* dst) {
    const ora_tile* t = &w->t[r];
    memcpy(dst, t->u, sizeof(double) * (size_t)(t->nx + 2) * (size_t)(t->ny + 2));
}

/* pull model: tile r fills ITS ghosts from neighbours' edge interior cells */
static void tile_pull_halos(ora_world* w, int r) {
    ora_tile* t = &w->t[r];
    const int nxt = t->nx + 2;
    if (t->nbr[ORA_LEFT] >= 0) {
        const ora_tile* n = &w->t[t->nbr[ORA_LEFT]];
        for (int j = 1; j <= t->ny; ++j) AT(t->u, 0, j, nxt) = AT(n->u, n->nx, j, n->nx + 2);
    }
    if (t->nbr[ORA_RIGHT] >= 0) {
        const ora_tile* n = &w->t[t->nbr[ORA_RIGHT]];
        for (int j = 1; j <= t->ny; ++j) AT(t->u, t->nx + 1, j, nxt) = AT(n->u, 1, j, n->nx + 2);
    }
    if (t->nbr[ORA_BOTTOM] >= 0) {
        const ora_tile* n = &w->t[t->nbr[ORA_BOTTOM]];
        for (int i = 1; i <= t->nx; ++i) AT(t->u, i, 0, nxt) = AT(n->u, i, n->ny, n->nx + 2);
    }
    if (t->nbr[ORA_TOP] >= 0) {
        const ora_tile* n = &w->t[t->nbr[ORA_TOP]];
        for (int i = 1; i <= t->nx; ++i) AT(t->u, i, t->ny + 1, nxt) = AT(n->u, i, 1, n->nx + 2);
    }
}

typedef struct {
    ora_world* w;
    int r0, r1, steps;
    double D, vx, vy, dt;
    int bc[4];
    pthread_barrier_t* bar;
    int checked; /* 1: the bounds-checked accessor flavour of diffusion / advection */
} ora_job;

static void* world_worker(void* arg) {
    ora_job* jb = (ora_job*)arg;
    ora_world* w = jb->w;
    for (int s = 0; s < jb->steps; ++s) {
        for (int r = jb->r0; r < jb->r1; ++r) tile_pull_halos(w, r);
        if (jb->bar) pthread_barrier_wait(jb->bar);
        for (int r = jb->r0; r < jb->r1; ++r) {
            ora_tile* t = &w->t[r];
            int phys[4];
            for (int k = 0; k < 4; ++k) phys[k] = t->nbr[k] < 0;
            if (jb->checked) {
                ora_apply_boundary(t->u, t->nx, t->ny, jb->bc, phys, 0.0);
                memcpy(t->tmp, t->u, sizeof(double) * (size_t)(t->nx + 2) * (size_t)(t->ny + 2));
                ora_diffusion_step_checked(t->u, t->tmp, t->nx, t->ny, w->dx, w->dy, jb->D, jb->dt);
                ora_advection_step_checked(t->u, t->tmp, t->nx, t->ny, w->dx, w->dy, jb->vx, jb->vy, jb->dt);
            } else {
                ora_step_tile(t->u, t->tmp, t->nx, t->ny, w->dx, w->dy, jb->D, jb->vx, jb->vy, jb->dt,
                              jb->bc, phys);
            }
            double* x = t->u;
            t->u = t->tmp;
            t->tmp = x;
        }
        if (jb->bar) pthread_barrier_wait(jb->bar);
    }
    return NULL;
}

/* returns wall seconds; checked = 1 selects the bounds-checked accessor flavour */
double ora_world_run_mode(ora_world* w, double D, double vx, double vy, double dt, const int bc[4],
                          int steps, int threads, int checked) {
    if (threads < 1) threads = 1;
    if (threads > w->size) threads = w->size;
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    if (threads == 1) {
        ora_job jb = {w, 0, w->size, steps, D, vx, vy, dt, {bc[0], bc[1], bc[2], bc[3]}, NULL, checked};
        world_worker(&jb);
    } else {
        pthread_barrier_t bar;
        pthread_barrier_init(&bar, NULL, (unsigned)threads);
        pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
        ora_job* jobs = (ora_job*)calloc((size_t)threads, sizeof(ora_job));
        for (int k = 0; k < threads; ++k) {
            ora_job jb = {w,  (int)((long)w->size * k / threads), (int)((long)w->size * (k + 1) / threads),
                          steps, D, vx, vy, dt, {bc[0], bc[1], bc[2], bc[3]}, &bar, checked};
            jobs[k] = jb;
            pthread_create(&th[k], NULL, world_worker, &jobs[k]);
        }
        for (int k = 0; k < threads; ++k) pthread_join(th[k], NULL);
        pthread_barrier_destroy(&bar);
        free(th);
        free(jobs);
    }
    clock_gettime(CLOCK_MONOTONIC, &b);
    return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}

double ora_world_run(ora_world* w, double D, double vx, double vy, double dt, const int bc[4],
                     int steps, int threads) {
    return ora_world_run_mode(w, D, vx, vy, dt, bc, steps, threads, 0);
}

/* reductions used around the loop (reference src/main.cpp:73-77 takes min/max over the whole
 * local array, ghosts included) */
void ora_minmax(const double* f, size_t n, double out[2]) {
    double mn = f[0], mx = f[0];
    for (size_t k = 1; k < n; ++k) {
        if (f[k] < mn) mn = f[k];
        if (f[k] > mx) mx = f[k];
    }
    out[0] = mn;
    out[1] = mx;
}

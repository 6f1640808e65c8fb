// api.cpp — the C ABI of include/csim.h, part 1: library / device, host-side scalars, MPI-free decomposition,
// the exchange plan, the device Field mirror and the reference-granularity operators.  (The time-loop stepper:
// stepper.cpp, passes.cpp, planner.cpp, profile.cpp; kernels: kernels.hip.)  Compiled with hipcc; host code only.
#include "stepper.hpp"

namespace csim {

static thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
static bool pow2(double x) {
    if (!(x > 0.0) || !std::isnormal(x)) return false;
    int e = 0;
    return std::frexp(x, &e) == 0.5 && std::isnormal(1.0 / x);
}

Phys make_phys(double dx, double dy, double D, double dt, double vx, double vy, bool contract) {
    Phys p;
    p.kdiff = dt * D;
    p.mdt = -dt;
    p.vx = vx;
    p.vy = vy;
    p.dx = dx;
    p.dy = dy;
    p.dx2 = dx * dx;
    p.dy2 = dy * dy;
    p.rdx = 1.0 / dx;
    p.rdy = 1.0 / dy;
    p.rdx2 = 1.0 / p.dx2;
    p.rdy2 = 1.0 / p.dy2;
    if (dx == 1.0 && dy == 1.0)
        p.div_mode = 0;
    else if (pow2(dx) && pow2(dy) && pow2(p.dx2) && pow2(p.dy2))
        p.div_mode = 1;
    else
        p.div_mode = 2;
    // coefficient form of the same update (option "contract"): collect what multiplies each of the five
    // points in  c + dt D ((E - 2c + W)/dx^2 + (N - 2c + S)/dy^2) - dt (vx dudx + vy dudy)  with the
    // upwind differences dudx = (c - W)/dx for vx >= 0, (E - c)/dx otherwise (dudy likewise)
    {
        const long double kx = static_cast<long double>(dt) * D / (static_cast<long double>(dx) * dx);
        const long double ky = static_cast<long double>(dt) * D / (static_cast<long double>(dy) * dy);
        const long double cx = static_cast<long double>(dt) * vx / dx, cy = static_cast<long double>(dt) * vy / dy;
        const long double acx = cx < 0 ? -cx : cx, acy = cy < 0 ? -cy : cy;
        p.a0 = static_cast<double>(1.0L - 2.0L * kx - 2.0L * ky - acx - acy);
        p.aW = static_cast<double>(kx + (vx >= 0.0 ? acx : 0.0L));
        p.aE = static_cast<double>(kx + (vx >= 0.0 ? 0.0L : acx));
        p.aS = static_cast<double>(ky + (vy >= 0.0 ? acy : 0.0L));
        p.aN = static_cast<double>(ky + (vy >= 0.0 ? 0.0L : acy));
    }
    // fused E - 2c (diffuse_term<., true>): max|u| grows by at most g per step — the sum of the magnitudes of
    // what multiplies the five points, with room for the 15 roundings — so a tile whose inputs stay below
    // 2^1022 / g^MAX_FUSE cannot reach 2^1023 at any level of a pass.  Parameters so far from stable that
    // this bound drops below 2^900 switch the fused form off.
    {
        const double g = (1.0 + 4.0 * std::fabs(p.kdiff) * (1.0 / p.dx2 + 1.0 / p.dy2) +
                          2.0 * std::fabs(p.mdt) * (std::fabs(vx) / std::fabs(dx) + std::fabs(vy) / std::fabs(dy))) * (1.0 + 1e-9);
        const double thr = std::ldexp(1.0, 1022) / std::pow(g, MAX_FUSE);
        p.fast_thr = (std::isfinite(g) && thr >= std::ldexp(1.0, 900)) ? thr : 0.0;
    }
    if (contract) p.div_mode = 3;
    return p;
}

int finish_partials(const double* scratch_dev, int nblocks, int kind, double out[2],
                           hipStream_t st) {
    std::vector<double> h(2 * REDUCE_BLOCKS);
    CSIM_HIP(hipMemcpyAsync(h.data(), scratch_dev, sizeof(double) * 2 * REDUCE_BLOCKS,
                            hipMemcpyDeviceToHost, st));
    CSIM_HIP(hipStreamSynchronize(st));
    double r0 = h[0], r1 = h[REDUCE_BLOCKS];
    for (int k = 1; k < nblocks; ++k) {
        if (kind == 0) {
            r0 = std::fmin(r0, h[k]);
            r1 = std::fmax(r1, h[REDUCE_BLOCKS + k]);
        } else if (kind == 1) {
            r0 += h[k];
        } else {
            r0 = std::fmax(r0, h[k]);
        }
    }
    out[0] = r0;
    out[1] = r1;
    return CSIM_OK;
}

int reduce_blocks(int nrows) { return nrows < REDUCE_BLOCKS ? nrows : REDUCE_BLOCKS; }

int upload_2d(double* d, int nx, int ny, int pitch, const double* host) {
    CSIM_HIP(hipMemcpy2D(d + (LPAD - 1), sizeof(double) * pitch, host, sizeof(double) * (nx + 2),
                         sizeof(double) * (nx + 2), ny + 2, hipMemcpyHostToDevice));
    return CSIM_OK;
}
int download_2d(const double* d, int nx, int ny, int pitch, double* host) {
    CSIM_HIP(hipMemcpy2D(host, sizeof(double) * (nx + 2), d + (LPAD - 1), sizeof(double) * pitch,
                         sizeof(double) * (nx + 2), ny + 2, hipMemcpyDeviceToHost));
    return CSIM_OK;
}
int download_interior_2d(const double* d, int nx, int ny, int pitch, double* host) {
    CSIM_HIP(hipMemcpy2D(host, sizeof(double) * nx, d + pitch + LPAD, sizeof(double) * pitch,
                         sizeof(double) * nx, ny, hipMemcpyDeviceToHost));
    return CSIM_OK;
}

// The 8 peers of a tile (L R B T BL BR TL TR): the four sides from the decomposition, a diagonal
// only where both adjacent sides have neighbours.  A size-1 decomposition whose sides were pointed
// at rank 0 is the self-linked test torus.
void neighbours8(const csim_decomp& dec, int nbr8[8]) {
    for (int k = 0; k < 4; ++k) nbr8[k] = dec.nbr[k];
    const int cx = dec.coords[0], cy = dec.coords[1], py = dec.dims[1];
    auto diag = [&](int sx, int sy, int dx_, int dy_) {
        if (dec.nbr[sx] < 0 || dec.nbr[sy] < 0) return -1;
        return dec.size == 1 ? 0 : (cx + dx_) * py + (cy + dy_);
    };
    nbr8[4] = diag(CSIM_LEFT, CSIM_BOTTOM, -1, -1);
    nbr8[5] = diag(CSIM_RIGHT, CSIM_BOTTOM, +1, -1);
    nbr8[6] = diag(CSIM_LEFT, CSIM_TOP, -1, +1);
    nbr8[7] = diag(CSIM_RIGHT, CSIM_TOP, +1, +1);
}

// doubles in the face of direction d at depth H on an nx x ny tile (depth 1: the interior span of an
// edge line, reference src/halo.cpp:12-18; deeper: ghost entries ride along, corners are H x H blocks)
size_t face_doubles(int d, int H, int nx, int ny) {
    if (H == 1) return d < 2 ? static_cast<size_t>(ny) : static_cast<size_t>(nx);
    return d < 2 ? static_cast<size_t>(H) * (ny + 2) : d < 4 ? static_cast<size_t>(H) * (nx + 2)
                                                       : static_cast<size_t>(H) * H;
}

bool valid_bc(const int bc[4]) {
    for (int s = 0; s < 4; ++s)
        if (bc[s] < CSIM_BC_DIRICHLET || bc[s] > CSIM_BC_PERIODIC) return false;
    return true;
}

}  // namespace csim

using namespace csim;

extern "C" {

const char* csim_last_error(void) { return g_err.c_str(); }

int csim_abi_version(void) { return CSIM_ABI_VERSION; }

int csim_device_count(int* count) {
    CSIM_REQUIRE(count, "count is null");
    CSIM_HIP(hipGetDeviceCount(count));
    return CSIM_OK;
}

int csim_set_device(int device) {
    CSIM_HIP(hipSetDevice(device));
    return CSIM_OK;
}

int csim_device_name(char* buf, size_t n) {
    CSIM_REQUIRE(buf && n > 0, "bad buffer");
    int dev = 0;
    CSIM_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    CSIM_HIP(hipGetDeviceProperties(&prop, dev));
    std::snprintf(buf, n, "%s", prop.gcnArchName);
    return CSIM_OK;
}

// reference include/stability.hpp:5-16
double csim_safe_dt(double dx, double dy, double vx, double vy, double D) {
    const double inf = std::numeric_limits<double>::infinity();
    const double ax = std::fabs(vx), ay = std::fabs(vy);
    const double rate = (ax > 0 ? ax / dx : 0.0) + (ay > 0 ? ay / dy : 0.0);
    const double dt_adv = rate > 0 ? 1.0 / rate : inf;
    const double w = 1.0 / (dx * dx) + 1.0 / (dy * dy);
    const double dt_diff = D > 0 ? 1.0 / (2.0 * D * w) : inf;
    return std::min(dt_adv, dt_diff);
}

// reference src/decomp.cpp:5-34 with MPI_Dims_create / MPI_Cart_* re-derived:
// most balanced factor pair, larger factor first; rank = coords[0]*dims[1] + coords[1].
int csim_decomp_init(int size, int rank, int nx_global, int ny_global, csim_decomp* out) {
    CSIM_REQUIRE(out, "out is null");
    CSIM_REQUIRE(size >= 1 && rank >= 0 && rank < size, "bad size/rank");
    CSIM_REQUIRE(nx_global > 0 && ny_global > 0, "nx/ny must be > 0");
    int small = 1;
    for (int f = 1; static_cast<long>(f) * f <= size; ++f)
        if (size % f == 0) small = f;
    csim_decomp d{};
    d.size = size;
    d.rank = rank;
    d.dims[0] = size / small;
    d.dims[1] = small;
    d.coords[0] = rank / d.dims[1];
    d.coords[1] = rank % d.dims[1];
    const int cx = d.coords[0], cy = d.coords[1];
    d.nbr[CSIM_LEFT] = cx > 0 ? (cx - 1) * d.dims[1] + cy : CSIM_NO_NEIGHBOR;
    d.nbr[CSIM_RIGHT] = cx + 1 < d.dims[0] ? (cx + 1) * d.dims[1] + cy : CSIM_NO_NEIGHBOR;
    d.nbr[CSIM_BOTTOM] = cy > 0 ? cx * d.dims[1] + (cy - 1) : CSIM_NO_NEIGHBOR;
    d.nbr[CSIM_TOP] = cy + 1 < d.dims[1] ? cx * d.dims[1] + (cy + 1) : CSIM_NO_NEIGHBOR;
    d.nx_global = nx_global;
    d.ny_global = ny_global;
    const int bx = nx_global / d.dims[0], by = ny_global / d.dims[1];
    d.nx_local = bx + (cx == d.dims[0] - 1 ? nx_global % d.dims[0] : 0);
    d.ny_local = by + (cy == d.dims[1] - 1 ? ny_global % d.dims[1] : 0);
    d.x_offset = cx * bx;
    d.y_offset = cy * by;
    CSIM_REQUIRE(d.nx_local > 0 && d.ny_local > 0, "more ranks than cells along an axis");
    *out = d;
    return CSIM_OK;
}

// THE message order of one halo exchange — the only place it is decided (post_exchange and
// post_exchange2 walk this plan).  Sends go out in direction order; receives are posted in
// opposite-direction order, so that the k-th message a rank sends to a given peer is the k-th one
// that peer expects from it even when one peer sits in several directions (2-wide process grids, the
// self-linked torus): RCCL matches the sends and receives of a pair in posting order.
int csim_exchange_plan(const csim_decomp* dec, int depth, csim_msg sends[8], int* nsend, csim_msg recvs[8],
                       int* nrecv) {
    CSIM_REQUIRE(dec && sends && nsend && recvs && nrecv, "null argument");
    CSIM_REQUIRE(depth >= 1 && depth <= MAX_FUSE, "depth must be 1..7");
    static const int recv_order1[4] = {CSIM_RIGHT, CSIM_LEFT, CSIM_TOP, CSIM_BOTTOM};
    static const int recv_order8[8] = {1, 0, 3, 2, 7, 6, 5, 4};
    int nbr8[8];
    neighbours8(*dec, nbr8);
    const int ndir = depth == 1 ? 4 : 8;
    const int* order = depth == 1 ? recv_order1 : recv_order8;
    int ns = 0, nr = 0;
    for (int d = 0; d < ndir; ++d)
        if (nbr8[d] >= 0)
            sends[ns++] = csim_msg{nbr8[d], d, static_cast<long>(face_doubles(d, depth, dec->nx_local, dec->ny_local))};
    for (int q = 0; q < ndir; ++q) {
        const int d = order[q];
        if (nbr8[d] >= 0)
            recvs[nr++] = csim_msg{nbr8[d], d, static_cast<long>(face_doubles(d, depth, dec->nx_local, dec->ny_local))};
    }
    *nsend = ns;
    *nrecv = nr;
    return CSIM_OK;
}

// ---- Field -------------------------------------------------------------------------------------
int csim_field_create(int nx, int ny, int halo, double dx, double dy, csim_field** out) {
    CSIM_REQUIRE(out, "out is null");
    *out = nullptr;
    CSIM_REQUIRE(nx > 0 && ny > 0, "nx/ny must be > 0");
    CSIM_REQUIRE(halo == 1, "only halo == 1 is supported (reference src/main.cpp:65)");
    CSIM_REQUIRE(dx > 0 && dy > 0, "dx/dy must be > 0");
    csim_field* f = new csim_field;
    f->nx = nx;
    f->ny = ny;
    f->halo = halo;
    f->dx = dx;
    f->dy = dy;
    f->pitch = pitch_for(nx);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&f->alloc), f->bytes());
    if (e == hipSuccess) e = hipMemset(f->alloc, 0, f->bytes());
    if (e == hipSuccess) f->d = f->alloc + static_cast<size_t>(GHOST_EXTRA) * f->pitch;
    if (e == hipSuccess)
        e = hipMalloc(reinterpret_cast<void**>(&f->scratch), sizeof(double) * 2 * REDUCE_BLOCKS);
    if (e != hipSuccess) {
        if (f->alloc) (void)hipFree(f->alloc);
        delete f;
        return fail(CSIM_ERR_HIP, std::string("csim_field_create: ") + hipGetErrorString(e));
    }
    *out = f;
    return CSIM_OK;
}

int csim_field_destroy(csim_field* f) {
    if (!f) return CSIM_OK;
    if (f->alloc) (void)hipFree(f->alloc);
    if (f->scratch) (void)hipFree(f->scratch);
    delete f;
    return CSIM_OK;
}

int csim_field_upload(csim_field* f, const double* host) {
    CSIM_REQUIRE(f && host, "null argument");
    return upload_2d(f->d, f->nx, f->ny, f->pitch, host);
}

int csim_field_download(const csim_field* f, double* host) {
    CSIM_REQUIRE(f && host, "null argument");
    return download_2d(f->d, f->nx, f->ny, f->pitch, host);
}

int csim_field_download_interior(const csim_field* f, double* host) {
    CSIM_REQUIRE(f && host, "null argument");
    return download_interior_2d(f->d, f->nx, f->ny, f->pitch, host);
}

int csim_field_fill(csim_field* f, double value) {
    CSIM_REQUIRE(f, "null field");
    CSIM_HIP(launch_fill(f->d, f->nx, f->ny, f->pitch, value, nullptr));
    CSIM_HIP(hipStreamSynchronize(nullptr));
    return CSIM_OK;
}

static bool same_shape(const csim_field* a, const csim_field* b) {
    return a && b && a->nx == b->nx && a->ny == b->ny;
}

int csim_field_copy(csim_field* dst, const csim_field* src) {
    CSIM_REQUIRE(same_shape(dst, src), "fields differ in shape");
    CSIM_HIP(hipMemcpy(dst->alloc, src->alloc, src->bytes(), hipMemcpyDeviceToDevice));
    return CSIM_OK;
}

int csim_field_swap(csim_field* a, csim_field* b) {
    CSIM_REQUIRE(same_shape(a, b), "fields differ in shape");
    std::swap(a->alloc, b->alloc);
    std::swap(a->d, b->d);
    return CSIM_OK;
}

int csim_field_minmax(const csim_field* f, double out[2]) {
    CSIM_REQUIRE(f && out, "null argument");
    CSIM_HIP(launch_minmax(f->d, f->nx, f->ny, f->pitch, f->scratch, nullptr));
    return finish_partials(f->scratch, reduce_blocks(f->ny + 2), 0, out, nullptr);
}

int csim_field_sum(const csim_field* f, double* out) {
    CSIM_REQUIRE(f && out, "null argument");
    double r[2];
    CSIM_HIP(launch_sum(f->d, f->nx, f->ny, f->pitch, f->scratch, nullptr));
    int rc = finish_partials(f->scratch, reduce_blocks(f->ny), 1, r, nullptr);
    *out = r[0];
    return rc;
}

int csim_field_linf_diff(const csim_field* a, const csim_field* b, double* out) {
    CSIM_REQUIRE(same_shape(a, b) && out, "fields differ in shape");
    double r[2];
    CSIM_HIP(launch_linf(a->d, b->d, a->nx, a->ny, a->pitch, a->scratch, nullptr));
    int rc = finish_partials(a->scratch, reduce_blocks(a->ny), 2, r, nullptr);
    *out = r[0];
    return rc;
}

// ---- operators ----------------------------------------------------------------------------------
int csim_apply_boundary(csim_field* f, const int bc[4], const int is_physical[4], double value) {
    CSIM_REQUIRE(f && bc && is_physical, "null argument");
    CSIM_REQUIRE(valid_bc(bc), "unknown boundary type");
    GhostArgs g{};
    for (int s = 0; s < 4; ++s) {
        g.bc[s] = bc[s];
        g.phys[s] = is_physical[s] != 0;
        g.recv[s] = nullptr;
    }
    g.value = value;
    CSIM_HIP(launch_ghost_fill(f->d, nullptr, f->nx, f->ny, f->pitch, g, nullptr));
    CSIM_HIP(hipStreamSynchronize(nullptr));
    return CSIM_OK;
}

int csim_diffusion_step(const csim_field* u, csim_field* out, double D, double dt) {
    CSIM_REQUIRE(same_shape(u, out), "fields differ in shape");
    CSIM_REQUIRE(u->d != out->d, "u and out must be distinct fields");
    const Phys p = make_phys(u->dx, u->dy, D, dt, 0.0, 0.0);
    CSIM_HIP(launch_diffusion_only(u->d, out->d, u->nx, u->ny, u->pitch, p, nullptr));
    CSIM_HIP(launch_ring_copy(u->d, out->d, u->nx, u->ny, u->pitch, nullptr));
    CSIM_HIP(hipStreamSynchronize(nullptr));
    return CSIM_OK;
}

int csim_advection_step(const csim_field* u, csim_field* out, double vx, double vy, double dt) {
    CSIM_REQUIRE(same_shape(u, out), "fields differ in shape");
    CSIM_REQUIRE(u->d != out->d, "u and out must be distinct fields");
    const Phys p = make_phys(u->dx, u->dy, 0.0, dt, vx, vy);
    CSIM_HIP(launch_advection_only(u->d, out->d, u->nx, u->ny, u->pitch, p, nullptr));
    CSIM_HIP(hipStreamSynchronize(nullptr));
    return CSIM_OK;
}

int csim_fused_step(const csim_field* u, csim_field* out, double D, double dt, double vx, double vy) {
    CSIM_REQUIRE(same_shape(u, out), "fields differ in shape");
    CSIM_REQUIRE(u->d != out->d, "u and out must be distinct fields");
    const Phys p = make_phys(u->dx, u->dy, D, dt, vx, vy);
    SweepCfg cfg;
    CSIM_HIP(launch_ring_copy(u->d, out->d, u->nx, u->ny, u->pitch, nullptr));
    CSIM_HIP(launch_sweep(u->d, out->d, u->nx, u->ny, u->pitch, p, cfg, nullptr));
    CSIM_HIP(hipStreamSynchronize(nullptr));
    return CSIM_OK;
}

}  // extern "C"

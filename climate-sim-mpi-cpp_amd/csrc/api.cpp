// api.cpp — the C ABI of include/csim.h: device Field mirror, the reference-granularity
// operators, MPI-free decomposition, and the time-loop stepper with its RCCL halo exchange.
// Compiled with hipcc; host code only (kernels live in kernels.hip).
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "internal.hpp"

namespace csim {

static thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define CSIM_NCCL(expr)                                                                        \
    do {                                                                                       \
        ncclResult_t r_ = (expr);                                                              \
        if (r_ != ncclSuccess)                                                                 \
            return ::csim::fail(CSIM_ERR_RCCL, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
    } while (0)

#define CSIM_REQUIRE(cond, msg) \
    do {                        \
        if (!(cond)) return ::csim::fail(CSIM_ERR_ARG, msg); \
    } while (0)

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

static int finish_partials(const double* scratch_dev, int nblocks, int kind, double out[2],
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

static int reduce_blocks(int nrows) { return nrows < REDUCE_BLOCKS ? nrows : REDUCE_BLOCKS; }

static int upload_2d(double* d, int nx, int ny, int pitch, const double* host) {
    CSIM_HIP(hipMemcpy2D(d + (LPAD - 1), sizeof(double) * pitch, host, sizeof(double) * (nx + 2),
                         sizeof(double) * (nx + 2), ny + 2, hipMemcpyHostToDevice));
    return CSIM_OK;
}
static int download_2d(const double* d, int nx, int ny, int pitch, double* host) {
    CSIM_HIP(hipMemcpy2D(host, sizeof(double) * (nx + 2), d + (LPAD - 1), sizeof(double) * pitch,
                         sizeof(double) * (nx + 2), ny + 2, hipMemcpyDeviceToHost));
    return CSIM_OK;
}
static int download_interior_2d(const double* d, int nx, int ny, int pitch, double* host) {
    CSIM_HIP(hipMemcpy2D(host, sizeof(double) * nx, d + pitch + LPAD, sizeof(double) * pitch,
                         sizeof(double) * nx, ny, hipMemcpyDeviceToHost));
    return CSIM_OK;
}

}  // namespace csim

using namespace csim;

// ---- stepper handle ----------------------------------------------------------------------------
struct csim_stepper {
    csim_decomp dec{};
    double dx = 1.0, dy = 1.0;
    int bc[4]{0, 0, 0, 0};
    int phys[4]{1, 1, 1, 1};
    double bc_value = 0.0;
    int nx = 0, ny = 0, pitch = 0;
    double* buf[2]{nullptr, nullptr};  // allocations incl. the device-only ghost layers, see internal.hpp
    double* cur = nullptr;             // views (row j = 0) into buf[], ping-pong
    double* nxt = nullptr;
    double* scratch = nullptr;
    double* send[4]{nullptr, nullptr, nullptr, nullptr};
    double* recv[4]{nullptr, nullptr, nullptr, nullptr};
    double* fin[4]{nullptr, nullptr, nullptr, nullptr};  // FinLines of the last fused pass of a run (all sides)
    hipStream_t s_comp = nullptr, s_comm = nullptr;
    // Relay (bulk-first passes): the two streams swap roles every pass — the stream that carried a pass's exchange and
    // frame launch also takes the NEXT pass's bulk launch — so `tail` names the stream on which the current field
    // state is ordered.  Every entry point that is not a relay pass settles it back onto s_comp first (settle()).
    // On multi-rank steppers both streams have the same (high) priority: they carry the same kinds of work in turn.
    hipStream_t tail = nullptr;
    hipEvent_t ev_tail = nullptr;
    int relay = 1;
    hipEvent_t ev_edge = nullptr, ev_recv = nullptr, ev_ready = nullptr;
    ncclComm_t comm = nullptr;
    bool comm_borrowed = false;  // csim_stepper_comm_share: another stepper owns `comm`
    long sync_timeout_ms = 0;    // > 0: csim_stepper_sync gives up after that long (CSIM_ERR_TIMEOUT)
    bool stall_armed = false;    // option "test_stall": the comm stream is parked on a value only the host will write
    bool multi = false;       // has at least one neighbour
    bool halo_fresh = false;  // recv[] holds the neighbours' edge lines of `cur`
    bool edge_async = false;  // ... and the exchange that delivers them was posted on s_comm (ev_recv marks its end)
    // depth-2 faces for two-steps-per-pass on several ranks; directions L R B T BL BR TL TR
    int nbr8[8]{-1, -1, -1, -1, -1, -1, -1, -1};
    size_t cap2[8]{0, 0, 0, 0, 0, 0, 0, 0};  // staging capacity (faces of depth MAX_FUSE)
    // doubles in the face of direction d at depth H
    size_t face_len(int d, int H) const;
    double* send2[8]{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double* recv2[8]{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_edge2 = nullptr, ev_recv2 = nullptr;
    // overlap mode 3: frame + bulk in one launch; the frame wavefronts publish the pass number in `frame_flag`
    // (signal memory) and the comm stream waits on it with hipStreamWaitValue64
    unsigned* frame_counter = nullptr;
    unsigned long long* frame_flag = nullptr;
    unsigned long long pass_no = 0;
    int frame_fence = 0, frame_prio = 1;  // experiment switches of mode 3, see FrameSync
    int fused_2c = 1;         // k_sweepO_dpp's interior body fuses E - 2c into one fma under the overflow guard (Phys::fast_thr)
    int fused_2c_active = 0;  // read-only: whether the last run's parameters allowed it
    int direct_faces = 1;                 // merged launch: the frame wavefronts fill send2[] themselves (no pack kernel)
    bool bulk_first_run = false;          // the current csim_stepper_run uses pass_fused_bulk_first
    // asynchronous snapshot of the interior (device staging copy + pinned host buffer + I/O stream)
    double* snap_d = nullptr;
    double* snap_h = nullptr;
    hipStream_t s_io = nullptr;
    hipEvent_t ev_snap_src = nullptr, ev_snap_copied = nullptr;
    bool snap_pending = false;
    int last_rows = 0;    // chunk height the last fused whole-field / bulk launch used
    long tile_cells = 0;  // cells of the decomposition's base tile (same on every rank): picks the preferred depth
    int fuse_cap = 1;     // deepest pass every rank of the decomposition can run (same on all ranks)
    int faces_depth = 0;  // recv2[] holds the neighbours' faces of `cur` of this depth (0 = none)
    SweepCfg cfg;
    int overlap = 5;        // 0: exchange serial; 1: frame launch, then bulk launch hiding the NEXT pass's exchange;
                            // 3: frame and bulk in ONE launch (needs signal memory, else as 1);
                            // 4: bulk launch first, hiding THIS pass's exchange, then the frame (pass_fused_bulk_first);
                            // 5 (default): as 4 (until round 3: 4 on runs of fewer than 16 passes, 3 otherwise)
    bool ring_ok = false;      // single rank without a Neumann side: the ghost ring (Dirichlet value / untouched
                               // Periodic ghosts) is constant and both buffers already hold it — no more ghost fills
    bool pre_unpacked = false; // the comm stream already unpacked the faces in recv2[] and filled the ghosts
                               // for the next fused pass (ev_recv2 marks the end of that)
    int fuse = -1;  // time steps per HBM pass: -1 auto (cheapest split, see plan_passes), 0/1 off, 2..7 depth
    int contract = 0;  // 1: opt-in contracted arithmetic (5-point FMA stencil), NOT bit-identical to the reference
    int external = 0;  // halos are carried by the caller (csim_stepper_halo_pack/_unpack), not RCCL
    int profile = 0;        // 0 off, k >= 1: HIP events around every k-th pass
    bool prof_active = false;
    long prof_slot = -1;
    unsigned long prof_counter = 0;
    int autotune = 1;     // pick rows_per_chunk (when 0 = auto) by timing trial launches on this GPU
    bool tuned = false;
    int tuned_T[MAX_FUSE + 1]{};  // chunk height found by the trial for passes of that depth (0 = not tried: cfg.tuned_rows re-snapped)
    void forget_tuning() {
        tuned = false;
        cfg.tuned_rows = 0;
        for (int& t : tuned_T) t = 0;
    }
    std::vector<hipEvent_t> ev_pool;  // start/stop pairs around sweep launches
    std::vector<int> ev_steps;        // time steps covered by each timed launch
    std::vector<long> ev_count;       // launches bracketed by each pair (see prof_begin: runs of equal launches)
    size_t ev_used = 0;
    int prof_open_kind = 0;           // > 0: a bracket of launches of that kind is open on the compute stream
    long prof_open_slot = -1;
    static constexpr int PROF_COMM = MAX_FUSE + 1;  // comm-stream chain of a pass: pack, RCCL group, unpack, ghost fill
    double prof_ms[MAX_FUSE + 2]{};     // indexed by time steps per launch (1..MAX_FUSE), [PROF_COMM]
    long prof_launches[MAX_FUSE + 2]{};
    size_t bytes() const { return sizeof(double) * static_cast<size_t>(ny + 2 + 2 * GHOST_EXTRA) * pitch; }
    // whole-allocation pointer of a view
    double* base(double* view) const { return view - static_cast<size_t>(GHOST_EXTRA) * pitch; }
};

static size_t face_doubles(int d, int H, int nx, int ny);
size_t csim_stepper::face_len(int d, int H) const { return face_doubles(d, H, nx, ny); }

// the field state back onto the compute stream (see csim_stepper::tail)
static int settle(csim_stepper* s) {
    if (s->tail == nullptr || s->tail == s->s_comp) {
        s->tail = s->s_comp;
        return CSIM_OK;
    }
    CSIM_HIP(hipEventRecord(s->ev_tail, s->tail));
    CSIM_HIP(hipStreamWaitEvent(s->s_comp, s->ev_tail, 0));
    s->tail = s->s_comp;
    return CSIM_OK;
}
#define CSIM_SETTLE(s_)            \
    do {                           \
        int rc_ = settle(s_);      \
        if (rc_) return rc_;       \
    } while (0)

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

// The 8 peers of a tile (L R B T BL BR TL TR): the four sides from the decomposition, a diagonal
// only where both adjacent sides have neighbours.  A size-1 decomposition whose sides were pointed
// at rank 0 is the self-linked test torus.
static void neighbours8(const csim_decomp& dec, int nbr8[8]) {
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
static size_t face_doubles(int d, int H, int nx, int ny) {
    if (H == 1) return d < 2 ? static_cast<size_t>(ny) : static_cast<size_t>(nx);
    return d < 2 ? static_cast<size_t>(H) * (ny + 2) : d < 4 ? static_cast<size_t>(H) * (nx + 2)
                                                       : static_cast<size_t>(H) * H;
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
static bool valid_bc(const int bc[4]) {
    for (int s = 0; s < 4; ++s)
        if (bc[s] < CSIM_BC_DIRICHLET || bc[s] > CSIM_BC_PERIODIC) return false;
    return true;
}

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

// ---- stepper -------------------------------------------------------------------------------------
int csim_stepper_create(const csim_decomp* dec, double dx, double dy, const int bc[4],
                        double bc_value, csim_stepper** out) {
    CSIM_REQUIRE(out, "out is null");
    *out = nullptr;
    CSIM_REQUIRE(dec && bc, "null argument");
    CSIM_REQUIRE(dec->nx_local > 0 && dec->ny_local > 0, "empty local tile");
    CSIM_REQUIRE(dx > 0 && dy > 0, "dx/dy must be > 0");
    CSIM_REQUIRE(valid_bc(bc), "unknown boundary type");
    csim_stepper* s = new csim_stepper;
    s->dec = *dec;
    s->dx = dx;
    s->dy = dy;
    s->bc_value = bc_value;
    s->nx = dec->nx_local;
    s->ny = dec->ny_local;
    s->pitch = pitch_for(s->nx);
    for (int k = 0; k < 4; ++k) {
        s->bc[k] = bc[k];
        s->phys[k] = dec->nbr[k] < 0;
        if (!s->phys[k]) s->multi = true;
    }
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) {
        if (e == hipSuccess) e = r;
        return e == hipSuccess;
    };
    ok(hipMalloc(reinterpret_cast<void**>(&s->buf[0]), s->bytes())) &&
        ok(hipMalloc(reinterpret_cast<void**>(&s->buf[1]), s->bytes())) &&
        ok(hipMemset(s->buf[0], 0, s->bytes())) && ok(hipMemset(s->buf[1], 0, s->bytes())) &&
        ok(hipMalloc(reinterpret_cast<void**>(&s->scratch), sizeof(double) * 2 * REDUCE_BLOCKS)) &&
        ok(hipEventCreateWithFlags(&s->ev_tail, hipEventDisableTiming)) &&
        ok(hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming)) &&
        ok(hipEventCreateWithFlags(&s->ev_edge, hipEventDisableTiming)) &&
        ok(hipEventCreateWithFlags(&s->ev_recv, hipEventDisableTiming));
    for (int k = 0; k < 4 && e == hipSuccess; ++k) {
        const size_t nf = sizeof(double) * static_cast<size_t>(k < 2 ? s->ny : s->nx);
        ok(hipMalloc(reinterpret_cast<void**>(&s->fin[k]), nf)) && ok(hipMemset(s->fin[k], 0, nf));
    }
    for (int k = 0; k < 4 && e == hipSuccess; ++k) {
        if (s->phys[k]) continue;
        const size_t n = sizeof(double) * static_cast<size_t>(k < 2 ? s->ny : s->nx);
        ok(hipMalloc(reinterpret_cast<void**>(&s->send[k]), n)) &&
            ok(hipMalloc(reinterpret_cast<void**>(&s->recv[k]), n)) &&
            ok(hipMemset(s->send[k], 0, n)) && ok(hipMemset(s->recv[k], 0, n));
    }
    // The fused-pass depth must be decided identically on every rank (the face exchange is
    // collective in effect): depth <= the smallest tile of the decomposition.
    {
        const int px = dec->dims[0] > 0 ? dec->dims[0] : 1, py = dec->dims[1] > 0 ? dec->dims[1] : 1;
        const int gx = dec->nx_global > 0 ? dec->nx_global : s->nx, gy = dec->ny_global > 0 ? dec->ny_global : s->ny;
        const int bx = gx / px, by = gy / py;
        // the size classes of the pass planner are single-rank measurements; across ranks every pass carries an
        // exchange whose cost does not shrink with the depth, so shallow passes lose what they gain (20-step
        // runs on the 4096 x 8192 self-torus: 4 x 5 932 k, 7 + 7 + 6 1 176 k): multi-rank steppers plan with the
        // mid-size table (preferred depth 6, depth 7 where it saves a pass)
        s->tile_cells = s->multi ? 0 : static_cast<long>(s->nx) * s->ny;
        const int min_tile = s->multi ? std::min(bx, by) : MAX_FUSE;
        s->fuse_cap = std::max(1, std::min(MAX_FUSE, min_tile));
    }
    neighbours8(*dec, s->nbr8);  // diagonal peers only where both adjacent sides have neighbours
    for (int d = 0; d < 8 && e == hipSuccess; ++d) {
        if (s->nbr8[d] < 0) continue;
        s->cap2[d] = s->face_len(d, MAX_FUSE);
        const size_t n = sizeof(double) * s->cap2[d];
        ok(hipMalloc(reinterpret_cast<void**>(&s->send2[d]), n)) &&
            ok(hipMalloc(reinterpret_cast<void**>(&s->recv2[d]), n)) &&
            ok(hipMemset(s->send2[d], 0, n)) && ok(hipMemset(s->recv2[d], 0, n));
    }
    if (e == hipSuccess) {
        // the exchange goes on a high-priority stream, so its small kernels are dispatched ahead of the
        // bulk sweep that is hiding them
        int lo = 0, hi = 0;  // numerically lower = higher priority
        ok(hipDeviceGetStreamPriorityRange(&lo, &hi)) &&
            ok(hipStreamCreateWithPriority(&s->s_comm, hipStreamNonBlocking, hi)) &&
            (s->multi ? ok(hipStreamCreateWithPriority(&s->s_comp, hipStreamNonBlocking, hi))
                      : ok(hipStreamCreateWithFlags(&s->s_comp, hipStreamNonBlocking)));
        s->tail = s->s_comp;
    }
    if (e == hipSuccess) {
        ok(hipEventCreateWithFlags(&s->ev_edge2, hipEventDisableTiming)) &&
            ok(hipEventCreateWithFlags(&s->ev_recv2, hipEventDisableTiming));
    }
    if (e == hipSuccess && s->multi) {
        ok(hipMalloc(reinterpret_cast<void**>(&s->frame_counter), sizeof(unsigned))) &&
            ok(hipMemset(s->frame_counter, 0, sizeof(unsigned)));
        // signal memory: absent or refused -> mode 3 is simply not offered (set_option reports it)
        if (e == hipSuccess) {
            int can = 0;
            int dev = 0;
            if (hipGetDevice(&dev) == hipSuccess &&
                hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev) == hipSuccess && can) {
                void* p = nullptr;
                if (hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory) == hipSuccess) {
                    s->frame_flag = static_cast<unsigned long long*>(p);
                    *s->frame_flag = 0;  // host-visible
                } else {
                    (void)hipGetLastError();
                }
            }
        }
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) {
        s->cur = s->buf[0] + static_cast<size_t>(GHOST_EXTRA) * s->pitch;
        s->nxt = s->buf[1] + static_cast<size_t>(GHOST_EXTRA) * s->pitch;
    }
    if (e != hipSuccess) {
        csim_stepper_destroy(s);
        return fail(CSIM_ERR_HIP, std::string("csim_stepper_create: ") + hipGetErrorString(e));
    }
    *out = s;
    return CSIM_OK;
}

int csim_stepper_destroy(csim_stepper* s) {
    if (!s) return CSIM_OK;
    if (s->s_comp) (void)hipStreamSynchronize(s->s_comp);
    if (s->s_comm) (void)hipStreamSynchronize(s->s_comm);
    if (s->s_io) (void)hipStreamSynchronize(s->s_io);
    if (s->snap_d) (void)hipFree(s->snap_d);
    if (s->snap_h) (void)hipHostFree(s->snap_h);
    if (s->ev_snap_src) (void)hipEventDestroy(s->ev_snap_src);
    if (s->ev_snap_copied) (void)hipEventDestroy(s->ev_snap_copied);
    if (s->s_io) (void)hipStreamDestroy(s->s_io);
    if (s->comm && !s->comm_borrowed) (void)ncclCommDestroy(s->comm);
    for (hipEvent_t ev : s->ev_pool) (void)hipEventDestroy(ev);
    for (int k = 0; k < 4; ++k) {
        if (s->send[k]) (void)hipFree(s->send[k]);
        if (s->recv[k]) (void)hipFree(s->recv[k]);
        if (s->fin[k]) (void)hipFree(s->fin[k]);
    }
    for (int d = 0; d < 8; ++d) {
        if (s->send2[d]) (void)hipFree(s->send2[d]);
        if (s->recv2[d]) (void)hipFree(s->recv2[d]);
    }
    if (s->frame_counter) (void)hipFree(s->frame_counter);
    if (s->frame_flag) (void)hipFree(s->frame_flag);
    if (s->ev_edge2) (void)hipEventDestroy(s->ev_edge2);
    if (s->ev_recv2) (void)hipEventDestroy(s->ev_recv2);
    if (s->ev_edge) (void)hipEventDestroy(s->ev_edge);
    if (s->ev_recv) (void)hipEventDestroy(s->ev_recv);
    if (s->s_comp) (void)hipStreamDestroy(s->s_comp);
    if (s->s_comm) (void)hipStreamDestroy(s->s_comm);
    if (s->ev_ready) (void)hipEventDestroy(s->ev_ready);
    if (s->ev_tail) (void)hipEventDestroy(s->ev_tail);
    if (s->buf[0]) (void)hipFree(s->buf[0]);
    if (s->buf[1]) (void)hipFree(s->buf[1]);
    if (s->scratch) (void)hipFree(s->scratch);
    delete s;
    return CSIM_OK;
}

int csim_comm_unique_id(void* id, size_t nbytes) {
    static_assert(sizeof(ncclUniqueId) == CSIM_UNIQUE_ID_BYTES, "ncclUniqueId size");
    CSIM_REQUIRE(id && nbytes >= sizeof(ncclUniqueId), "id buffer too small");
    ncclUniqueId u;
    CSIM_NCCL(ncclGetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return CSIM_OK;
}

int csim_stepper_comm_init(csim_stepper* s, const void* id, size_t nbytes) {
    CSIM_REQUIRE(s && id && nbytes >= sizeof(ncclUniqueId), "bad argument");
    if (s->comm) return fail(CSIM_ERR_STATE, "communicator already initialised");
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    CSIM_NCCL(ncclCommInitRank(&s->comm, s->dec.size, u, s->dec.rank));
    return CSIM_OK;
}

// Several steppers of one rank on one communicator (e.g. small parity cases run beside the production tile:
// building a communicator costs ~1 s at 8 ranks).  `s` borrows `owner`'s communicator and never destroys it;
// `owner` must outlive `s`.  The steppers must not have exchanges in flight at the same time (RCCL matches the
// messages of a rank pair in posting order): sync one before running the other.
int csim_stepper_comm_share(csim_stepper* s, csim_stepper* owner) {
    CSIM_REQUIRE(s && owner && s != owner, "bad argument");
    if (s->comm) return fail(CSIM_ERR_STATE, "communicator already initialised");
    if (!owner->comm) return fail(CSIM_ERR_STATE, "owner has no communicator: csim_stepper_comm_init first");
    CSIM_REQUIRE(s->dec.size == owner->dec.size && s->dec.rank == owner->dec.rank, "steppers of different ranks / world sizes");
    s->comm = owner->comm;
    s->comm_borrowed = true;
    return CSIM_OK;
}

int csim_stepper_upload(csim_stepper* s, const double* host) {
    CSIM_REQUIRE(s && host, "null argument");
    CSIM_SETTLE(s);
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    CSIM_HIP(hipStreamSynchronize(s->s_comm));
    int rc = upload_2d(s->cur, s->nx, s->ny, s->pitch, host);
    if (rc) return rc;
    // both ping-pong buffers start with the same ghost ring (reference main.cpp:104 copies u->tmp)
    // (a device-to-device hipMemcpy may return before it has run, and the stepper's streams do
    // not synchronise with the null stream: order the copy on the compute stream and wait)
    CSIM_HIP(hipDeviceSynchronize());
    CSIM_HIP(hipMemcpyAsync(s->base(s->nxt), s->base(s->cur), s->bytes(), hipMemcpyDeviceToDevice,
                            s->s_comp));
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    s->halo_fresh = false;
    s->faces_depth = 0;
    s->ring_ok = false;
    return CSIM_OK;
}

int csim_stepper_download(csim_stepper* s, double* host) {
    CSIM_REQUIRE(s && host, "null argument");
    CSIM_SETTLE(s);
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    return download_2d(s->cur, s->nx, s->ny, s->pitch, host);
}

int csim_stepper_download_interior(csim_stepper* s, double* host) {
    CSIM_REQUIRE(s && host, "null argument");
    CSIM_SETTLE(s);
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    return download_interior_2d(s->cur, s->nx, s->ny, s->pitch, host);
}

// Snapshot without stalling the time loop (the reference packs and writes the interior inside the
// step loop, src/io.cpp:402-424 called from src/main.cpp:96-99).  _begin enqueues a device-side
// copy of the current interior (the ping-pong buffers are free to move on after ~1 ms) followed
// by an asynchronous D2H into a pinned buffer on a third stream, and returns at once; the caller
// keeps enqueuing steps and calls _wait when it wants the data (pointer valid until the next
// _begin).  Layout: ny_local x nx_local, row-major — what write_field_netcdf packs.
int csim_stepper_snapshot_begin(csim_stepper* s) {
    CSIM_REQUIRE(s, "null stepper");
    CSIM_SETTLE(s);
    const size_t bytes = sizeof(double) * static_cast<size_t>(s->nx) * s->ny;
    // each piece is created once; a failed allocation is reported and retried by the next call
    if (!s->s_io) CSIM_HIP(hipStreamCreateWithFlags(&s->s_io, hipStreamNonBlocking));
    if (!s->ev_snap_src) CSIM_HIP(hipEventCreateWithFlags(&s->ev_snap_src, hipEventDisableTiming));
    if (!s->ev_snap_copied) CSIM_HIP(hipEventCreateWithFlags(&s->ev_snap_copied, hipEventDisableTiming));
    if (!s->snap_d) CSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s->snap_d), bytes));
    if (!s->snap_h) CSIM_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->snap_h), bytes, hipHostMallocDefault));
    if (s->snap_pending) CSIM_HIP(hipStreamSynchronize(s->s_io));  // previous snapshot still in flight
    CSIM_HIP(hipEventRecord(s->ev_snap_src, s->s_comp));
    CSIM_HIP(hipStreamWaitEvent(s->s_io, s->ev_snap_src, 0));
    CSIM_HIP(hipMemcpy2DAsync(s->snap_d, sizeof(double) * s->nx, s->cur + s->pitch + LPAD,
                              sizeof(double) * s->pitch, sizeof(double) * s->nx, s->ny,
                              hipMemcpyDeviceToDevice, s->s_io));
    CSIM_HIP(hipEventRecord(s->ev_snap_copied, s->s_io));
    // the sweeps may overwrite the source buffer only after the staging copy has read it
    CSIM_HIP(hipStreamWaitEvent(s->s_comp, s->ev_snap_copied, 0));
    CSIM_HIP(hipMemcpyAsync(s->snap_h, s->snap_d, bytes, hipMemcpyDeviceToHost, s->s_io));
    s->snap_pending = true;
    return CSIM_OK;
}

int csim_stepper_snapshot_wait(csim_stepper* s, const double** host_interior) {
    CSIM_REQUIRE(s && host_interior, "null argument");
    if (!s->snap_pending) return fail(CSIM_ERR_STATE, "no snapshot in flight: csim_stepper_snapshot_begin first");
    CSIM_HIP(hipStreamSynchronize(s->s_io));
    s->snap_pending = false;
    *host_interior = s->snap_h;
    return CSIM_OK;
}

int csim_stepper_init_gaussian(csim_stepper* s, double A, double sigma_frac, double xc_frac,
                               double yc_frac) {
    CSIM_REQUIRE(s, "null stepper");
    CSIM_SETTLE(s);
    CSIM_HIP(hipStreamSynchronize(s->s_comm));
    CSIM_HIP(hipMemsetAsync(s->base(s->cur), 0, s->bytes(), s->s_comp));
    CSIM_HIP(launch_gaussian(s->cur, s->nx, s->ny, s->pitch, s->dec.x_offset, s->dec.y_offset,
                             s->dec.nx_global, s->dec.ny_global, s->dx, s->dy, A, sigma_frac,
                             xc_frac, yc_frac, s->s_comp));
    CSIM_HIP(hipMemcpyAsync(s->base(s->nxt), s->base(s->cur), s->bytes(), hipMemcpyDeviceToDevice,
                            s->s_comp));
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    s->halo_fresh = false;
    s->faces_depth = 0;
    s->ring_ok = false;
    return CSIM_OK;
}

// one grouped RCCL exchange (replaces the <= 8 MPI requests + MPI_Waitall of reference src/halo.cpp:28-46):
// depth 1 = the staged edge lines of the four sides, depth 2..7 = the deep faces of all eight
// directions (diagonal ranks are direct xGMI peers too), in the order csim_exchange_plan fixes.
static int post_plan(csim_stepper* s, int depth, hipStream_t st) {
    if (!s->comm) return fail(CSIM_ERR_STATE, "halo exchange needs csim_stepper_comm_init first");
    csim_msg sends[8], recvs[8];
    int ns = 0, nr = 0;
    int rc = csim_exchange_plan(&s->dec, depth, sends, &ns, recvs, &nr);
    if (rc) return rc;
    double* const* sbuf = depth == 1 ? s->send : s->send2;
    double* const* rbuf = depth == 1 ? s->recv : s->recv2;
    CSIM_NCCL(ncclGroupStart());
    // A failed send/recv must not leave the group open (every later RCCL call of this thread would be
    // swallowed into it): stop posting, close the group, then report the first failure.
    ncclResult_t bad = ncclSuccess;
    const char* what = "";
    for (int k = 0; k < ns && bad == ncclSuccess; ++k) {
        bad = ncclSend(sbuf[sends[k].dir], static_cast<size_t>(sends[k].count), ncclDouble, sends[k].peer, s->comm, st);
        what = "ncclSend";
    }
    for (int k = 0; k < nr && bad == ncclSuccess; ++k) {
        bad = ncclRecv(rbuf[recvs[k].dir], static_cast<size_t>(recvs[k].count), ncclDouble, recvs[k].peer, s->comm, st);
        what = "ncclRecv";
    }
    const ncclResult_t end = ncclGroupEnd();
    if (bad != ncclSuccess)
        return fail(CSIM_ERR_RCCL, std::string(what) + " (halo exchange, depth " + std::to_string(depth) + "): " + ncclGetErrorString(bad));
    if (end != ncclSuccess) return fail(CSIM_ERR_RCCL, std::string("ncclGroupEnd: ") + ncclGetErrorString(end));
    return CSIM_OK;
}
static int post_exchange(csim_stepper* s, hipStream_t st) { return post_plan(s, 1, st); }
static int post_exchange2(csim_stepper* s, int H, hipStream_t st) { return post_plan(s, H, st); }

// halos of the CURRENT field: pack its edge lines, exchange, leave them staged in recv[]
static int refresh_halos(csim_stepper* s) {
    CSIM_HIP(launch_pack(s->cur, s->nx, s->ny, s->pitch, s->send, s->s_comp));
    int rc = post_exchange(s, s->s_comp);
    if (rc) return rc;
    s->halo_fresh = true;
    s->edge_async = false;
    return CSIM_OK;
}

static GhostArgs ghost_args(const csim_stepper* s) {
    GhostArgs g{};
    for (int k = 0; k < 4; ++k) {
        g.bc[k] = s->bc[k];
        g.phys[k] = s->phys[k];
        g.recv[k] = s->phys[k] ? nullptr : s->recv[k];
    }
    g.value = s->bc_value;
    return g;
}

// The sweeps never write ghost cells, so once a single-rank field without Neumann sides has had its
// ring filled (in both ping-pong buffers) the ring stays what every later apply_boundary would make it.
static bool ring_is_static(const csim_stepper* s) {
    if (s->multi) return false;
    for (int k = 0; k < 4; ++k)
        if (s->bc[k] == CSIM_BC_NEUMANN) return false;
    return true;
}

// External transport (e.g. the reference's own MPI): the caller moves the edge lines between
// ranks.  pack: edge lines of the current field -> host buffers (ny doubles for left/right, nx
// for bottom/top; entries of physical sides are ignored).  unpack: the neighbours' lines -> the
// staging buffers the next step's ghost fill reads.
int csim_stepper_halo_pack(csim_stepper* s, double* const host_send[4]) {
    CSIM_REQUIRE(s && host_send, "null argument");
    CSIM_SETTLE(s);
    if (!s->multi) return CSIM_OK;
    CSIM_HIP(launch_pack(s->cur, s->nx, s->ny, s->pitch, s->send, s->s_comp));
    for (int k = 0; k < 4; ++k) {
        if (s->phys[k]) continue;
        CSIM_REQUIRE(host_send[k], "missing host buffer for a neighbour side");
        const size_t n = sizeof(double) * static_cast<size_t>(k < 2 ? s->ny : s->nx);
        CSIM_HIP(hipMemcpyAsync(host_send[k], s->send[k], n, hipMemcpyDeviceToHost, s->s_comp));
    }
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    return CSIM_OK;
}

int csim_stepper_halo_unpack(csim_stepper* s, const double* const host_recv[4]) {
    CSIM_REQUIRE(s && host_recv, "null argument");
    if (!s->multi) return CSIM_OK;
    for (int k = 0; k < 4; ++k) {
        if (s->phys[k]) continue;
        CSIM_REQUIRE(host_recv[k], "missing host buffer for a neighbour side");
        const size_t n = sizeof(double) * static_cast<size_t>(k < 2 ? s->ny : s->nx);
        CSIM_HIP(hipMemcpyAsync(s->recv[k], host_recv[k], n, hipMemcpyHostToDevice, s->s_comp));
    }
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    s->halo_fresh = true;
    return CSIM_OK;
}

// deep-face flavour of the external transport, for csim_stepper_run(.., depth) in external mode
static bool depth_ok(const csim_stepper* s, int depth) {
    return depth >= 2 && depth <= MAX_FUSE && depth <= s->nx && depth <= s->ny;
}

static int fused_depth(const csim_stepper* s);

// deepest fused pass this stepper can run (1 = single steps only); identical on every rank of a
// decomposition, so external-transport callers can schedule their passes the way run() does
int csim_stepper_fuse_limit(const csim_stepper* s, int* depth) {
    CSIM_REQUIRE(s && depth, "null argument");
    *depth = fused_depth(s);
    return CSIM_OK;
}

int csim_stepper_faces_neighbors(const csim_stepper* s, int depth, int peers[8], int lengths[8]) {
    CSIM_REQUIRE(s && peers && lengths, "null argument");
    CSIM_REQUIRE(depth_ok(s, depth), "face depth must be 2..7 and fit the tile");
    for (int d = 0; d < 8; ++d) {
        peers[d] = s->nbr8[d];
        lengths[d] = s->nbr8[d] >= 0 ? static_cast<int>(s->face_len(d, depth)) : 0;
    }
    return CSIM_OK;
}

int csim_stepper_faces_pack(csim_stepper* s, int depth, double* const host_send[8]) {
    CSIM_REQUIRE(s && host_send, "null argument");
    CSIM_SETTLE(s);
    CSIM_REQUIRE(depth_ok(s, depth), "face depth must be 2..7 and fit the tile");
    if (!s->multi) return CSIM_OK;
    CSIM_HIP(launch_halo2_pack(s->cur, s->nx, s->ny, s->pitch, depth, s->send2, s->s_comp));
    for (int d = 0; d < 8; ++d) {
        if (s->nbr8[d] < 0) continue;
        CSIM_REQUIRE(host_send[d], "missing host buffer for a neighbour direction");
        CSIM_HIP(hipMemcpyAsync(host_send[d], s->send2[d], sizeof(double) * s->face_len(d, depth),
                                hipMemcpyDeviceToHost, s->s_comp));
    }
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    return CSIM_OK;
}

int csim_stepper_faces_unpack(csim_stepper* s, int depth, const double* const host_recv[8]) {
    CSIM_REQUIRE(s && host_recv, "null argument");
    CSIM_REQUIRE(depth_ok(s, depth), "face depth must be 2..7 and fit the tile");
    if (!s->multi) return CSIM_OK;
    for (int d = 0; d < 8; ++d) {
        if (s->nbr8[d] < 0) continue;
        CSIM_REQUIRE(host_recv[d], "missing host buffer for a neighbour direction");
        CSIM_HIP(hipMemcpyAsync(s->recv2[d], host_recv[d], sizeof(double) * s->face_len(d, depth),
                                hipMemcpyHostToDevice, s->s_comp));
    }
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    s->faces_depth = depth;
    return CSIM_OK;
}

int csim_stepper_exchange_halos(csim_stepper* s) {
    CSIM_REQUIRE(s, "null stepper");
    CSIM_SETTLE(s);
    if (!s->multi) return CSIM_OK;
    int rc = refresh_halos(s);
    if (rc) return rc;
    // unpack only (no boundary rule): physical sides are left alone, like reference halo.cpp
    GhostArgs g = ghost_args(s);
    for (int k = 0; k < 4; ++k)
        if (g.phys[k]) g.bc[k] = CSIM_BC_PERIODIC;
    CSIM_HIP(launch_ghost_fill(s->cur, s->nxt, s->nx, s->ny, s->pitch, g, s->s_comp));
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    return CSIM_OK;
}

static int prof_close(csim_stepper* s);

static int prof_fold(csim_stepper* s) {
    if (s->ev_used == 0) return CSIM_OK;
    int rc0 = prof_close(s);
    if (rc0) return rc0;
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    CSIM_HIP(hipStreamSynchronize(s->s_comm));
    for (size_t k = 0; k + 1 < s->ev_used; k += 2) {
        float ms = 0.f;
        CSIM_HIP(hipEventElapsedTime(&ms, s->ev_pool[k], s->ev_pool[k + 1]));
        const int t = s->ev_steps[k / 2];
        s->prof_ms[t] += ms;
        s->prof_launches[t] += s->ev_count[k / 2];
    }
    s->ev_used = 0;
    return CSIM_OK;
}

// one start/stop event pair of the current (sampled) pass: start recorded now on `st`; kind = time
// steps of the sweep launch (1..MAX_FUSE) or PROF_COMM for the comm-stream chain of a pass
static int prof_start(csim_stepper* s, int kind, hipStream_t st, long* slot) {
    *slot = -1;
    if (!s->prof_active) return CSIM_OK;
    while (s->ev_pool.size() < s->ev_used + 2) {
        hipEvent_t ev;
        // timing only: without the system-scope fence a default event performs when it is recorded (cache
        // write-back and invalidation between the kernels it brackets — the very thing being timed)
        CSIM_HIP(hipEventCreateWithFlags(&ev, hipEventDisableSystemFence));
        s->ev_pool.push_back(ev);
    }
    if (s->ev_steps.size() < s->ev_pool.size() / 2) s->ev_steps.resize(s->ev_pool.size() / 2);
    if (s->ev_count.size() < s->ev_pool.size() / 2) s->ev_count.resize(s->ev_pool.size() / 2);
    s->ev_steps[s->ev_used / 2] = kind;
    s->ev_count[s->ev_used / 2] = 1;
    CSIM_HIP(hipEventRecord(s->ev_pool[s->ev_used], st));
    *slot = static_cast<long>(s->ev_used);
    s->ev_used += 2;
    return CSIM_OK;
}

static int prof_stop(csim_stepper* s, long slot, hipStream_t st) {
    if (slot < 0) return CSIM_OK;
    CSIM_HIP(hipEventRecord(s->ev_pool[static_cast<size_t>(slot) + 1], st));
    return CSIM_OK;
}

// the stop event of an open bracket (see prof_begin)
static int prof_close(csim_stepper* s) {
    if (s->prof_open_kind == 0) return CSIM_OK;
    s->prof_open_kind = 0;
    return prof_stop(s, s->prof_open_slot, s->s_comp);
}

static int prof_begin(csim_stepper* s, int steps, hipStream_t st = nullptr) {
    if (!st) st = s->s_comp;
    constexpr size_t POOL = 2048;
    if (s->profile == 1 && !s->multi) {
        // Single rank, every pass timed: ONE bracket per run of equal launches instead of one per launch.  An
        // event between two launches is a barrier: the next launch cannot start its first wavefronts while the
        // previous one drains, which costs ~5 % of a 1.2 ms launch (kernel timelines of bench.py --steps 20) —
        // the measurement would slow down what it measures.  The figure reported per kind is then the
        // start-to-end time of the run divided by its launches (ghost fills between them included: ~5 us).
        s->prof_active = false;
        if (s->prof_open_kind == steps) {
            s->ev_count[static_cast<size_t>(s->prof_open_slot) / 2] += 1;
            return CSIM_OK;
        }
        int rc = prof_close(s);
        if (rc) return rc;
        if (s->ev_used + 4 > POOL) {
            rc = prof_fold(s);
            if (rc) return rc;
        }
        s->prof_active = true;  // prof_start looks at it
        rc = prof_start(s, steps, s->s_comp, &s->prof_open_slot);
        s->prof_active = false;
        if (rc == CSIM_OK) s->prof_open_kind = steps;
        return rc;
    }
    // profile = k > 1: only every k-th pass is bracketed (two event records cost a few microseconds
    // of stream time each, which shows on the ~170 us passes of a small multi-rank tile)
    s->prof_active = s->profile > 0 && (s->prof_counter++ % s->profile) == 0;
    if (!s->prof_active) return CSIM_OK;
    if (s->ev_used + 4 > POOL) {
        int rc = prof_fold(s);
        if (rc) return rc;
    }
    return prof_start(s, steps, st, &s->prof_slot);
}

static int prof_end(csim_stepper* s, hipStream_t st = nullptr) {
    if (!s->prof_active) return CSIM_OK;
    int rc = prof_stop(s, s->prof_slot, st ? st : s->s_comp);
    s->prof_active = false;
    return rc;
}


// ONE reference step: exchange_halos + apply_boundary + fused sweep + swap
static int pass_single(csim_stepper* s, const Phys& p, const GhostArgs& g) {
    CSIM_SETTLE(s);
    const bool rccl = s->multi && !s->external;
    if (rccl) {
        if (!s->halo_fresh) {
            int rc = refresh_halos(s);  // on s_comp: ordered before the ghost fill
            if (rc) return rc;
        } else if (s->edge_async) {
            // whatever "overlap" says NOW: the exchange in flight was posted on the comm stream
            CSIM_HIP(hipStreamWaitEvent(s->s_comp, s->ev_recv, 0));
        }
        s->edge_async = false;
    }
    // exchange_halos (unpack) + apply_boundary, mirrored into the partner buffer
    if (!s->ring_ok) {
        CSIM_HIP(launch_ghost_fill(s->cur, s->nxt, s->nx, s->ny, s->pitch, g, s->s_comp));
        s->ring_ok = ring_is_static(s);
    }
    if (rccl && s->overlap) {
        // edge lines of the NEXT field first, so their exchange overlaps the full sweep
        CSIM_HIP(launch_edge_pack(s->cur, s->nx, s->ny, s->pitch, p, s->send, s->s_comp));
        CSIM_HIP(hipEventRecord(s->ev_edge, s->s_comp));
        CSIM_HIP(hipStreamWaitEvent(s->s_comm, s->ev_edge, 0));
        int rc = post_exchange(s, s->s_comm);
        if (rc) return rc;
        CSIM_HIP(hipEventRecord(s->ev_recv, s->s_comm));
        s->edge_async = true;
    }
    int rc = prof_begin(s, 1);
    if (rc) return rc;
    CSIM_HIP(launch_sweep(s->cur, s->nxt, s->nx, s->ny, s->pitch, p, s->cfg, s->s_comp));
    rc = prof_end(s);
    if (rc) return rc;
    std::swap(s->cur, s->nxt);
    if (s->multi && (s->external || !s->overlap)) s->halo_fresh = false;  // exchange again next step
    s->faces_depth = 0;
    return CSIM_OK;
}

// T = 2..7 reference steps in one HBM pass.  Several ranks: faces of depth T (8 directions) are
// staged in recv2[]; when the next pass is fused too (with `next_T` steps), the frame tiles are
// computed first and the comm stream packs and exchanges their depth-next_T faces while the bulk
// of the sweep is still running.
static hipError_t launch_fused(csim_stepper* s, const Phys& p, const int kind[4], int T, int part,
                               hipStream_t st, bool final_pass = false, int lds_bytes = 0,
                               const FrameSync* sync = nullptr) {
    SweepCfg cfg = s->cfg;
    if (lds_bytes > 0) cfg.lds_bytes = lds_bytes;
    if (T >= 2 && T <= MAX_FUSE && s->tuned_T[T] > 0) cfg.tuned_rows = s->tuned_T[T];  // this depth had its own trial
    cfg.rows_used = &s->last_rows;
    return launch_sweepO(s->cur, s->nxt, s->nx, s->ny, s->pitch, p, cfg, kind, s->bc_value, T, part, st,
                         final_pass ? s->fin : nullptr, sync);
}

// final_pass (overlapped-strip kernels only): the last pass of a run.  The kernel also emits the
// FinLines (level T-1 = the state before the last step) and a closing ghost fill turns them into
// the ghost ring the reference leaves behind — halos and boundary values of the state BEFORE the
// last step (src/main.cpp:102-104 + src/diffusion.cpp:18-25) — without a trailing one-step pass.
// Bulk-first pass (overlap 4, and overlap 5 on short runs): the exchange of THIS pass's faces runs under
// THIS pass's bulk sweep, which needs nothing from the neighbours, and the frame tiles follow once the
// faces are in:
//
//   comm stream     wait(field ready) -> pack faces of `cur` -> RCCL group -> unpack -> ghost fill -> record(recv)
//   compute stream  BULK tiles -> wait(recv) -> FRAME tiles (-> FinLines on the last pass)
//
// No pass of a run — not even the first — waits for an exchange that nothing hides (the frame-first
// schedules 1 and 3 start the exchange of pass p+1 under pass p, so pass 1 of every csim_stepper_run call
// pays its exchange in full: ~100 us of a ~160 us pass on the 8-GPU tile).  The price is two launches per
// pass with the thin frame launch last (~7 us per pass against the merged launch), so it wins on runs
// of fewer than ~16 passes, e.g. the three passes of a 20-step run.
static int pass_fused_bulk_first(csim_stepper* s, const Phys& p, int T, bool final_pass) {
    int kind[4];
    for (int k = 0; k < 4; ++k) kind[k] = s->phys[k] ? s->bc[k] : 3;
    GhostArgs g = ghost_args(s);
    for (int k = 0; k < 4; ++k) g.recv[k] = nullptr;  // neighbour sides come from the deep faces
    s->pre_unpacked = false;
    // Relay (option "relay", default on): X = the stream the field state is ordered on carries the bulk; Y, the other
    // one, carries the exchange chain and the frame launch — and, in the next pass, the bulk, which then follows the
    // frame launch on the SAME stream without an event in between.  The one cross-stream wait per pass that remains
    // on the way of data (Y's chain waits for X's state) sits under the bulk.  Without the relay the frame launch waits
    // for the chain through an event (~12 us from the record to the launch it releases) and the next pass's bulk
    // follows the frame through another record / wait pair (~8 us): 20 of the ~205 us of a 7-step pass on the 8-GPU
    // tile (profiles/r03_timeline_torus20.txt).
    if (!s->relay) CSIM_SETTLE(s);
    hipStream_t X = s->tail ? s->tail : s->s_comp;
    hipStream_t Y = X == s->s_comp ? s->s_comm : s->s_comp;
    // everything enqueued so far on X produced `cur` (and the partner buffer's ring)
    CSIM_HIP(hipEventRecord(s->ev_ready, X));
    CSIM_HIP(hipStreamWaitEvent(Y, s->ev_ready, 0));
    int rc = prof_begin(s, T, X);
    if (rc) return rc;
    // the bulk goes out first: the GPU starts on it while the host is still enqueuing the exchange
    CSIM_HIP(launch_fused(s, p, kind, T, 2, X));  // nothing to launch on tiles that are all frame
    if (s->relay) CSIM_HIP(hipEventRecord(s->ev_edge2, X));  // the bulk's end, for whatever follows the frame on Y
    long comm_slot = -1;
    rc = prof_start(s, csim_stepper::PROF_COMM, Y, &comm_slot);
    if (rc) return rc;
    CSIM_HIP(launch_halo2_pack(s->cur, s->nx, s->ny, s->pitch, T, s->send2, Y));
    rc = post_exchange2(s, T, Y);
    if (rc) return rc;
    CSIM_HIP(launch_halo2_unpack(s->cur, s->nx, s->ny, s->pitch, T, s->recv2, Y));
    CSIM_HIP(launch_ghost_fill(s->cur, s->nxt, s->nx, s->ny, s->pitch, g, Y, T));
    rc = prof_stop(s, comm_slot, Y);
    if (rc) return rc;
    hipStream_t F = Y;  // the frame launch follows the chain on its own stream
    if (!s->relay) {
        CSIM_HIP(hipEventRecord(s->ev_recv2, Y));
        CSIM_HIP(hipStreamWaitEvent(X, s->ev_recv2, 0));
        F = X;
    }
    CSIM_HIP(launch_fused(s, p, kind, T, 1, F, final_pass));
    rc = prof_end(s, F);
    if (rc) return rc;
    if (s->relay) {
        CSIM_HIP(hipStreamWaitEvent(Y, s->ev_edge2, 0));  // the field is complete on Y once the bulk is done too
        s->tail = Y;
    }
    std::swap(s->cur, s->nxt);
    s->halo_fresh = false;
    s->faces_depth = 0;
    if (final_pass) {
        GhostArgs gf = ghost_args(s);
        for (int k = 0; k < 4; ++k) {
            gf.recv[k] = s->phys[k] ? nullptr : s->fin[k];
            gf.adj[k] = s->phys[k] ? s->fin[k] : nullptr;
        }
        CSIM_HIP(launch_ghost_fill(s->cur, s->nxt, s->nx, s->ny, s->pitch, gf, s->tail));
    }
    return CSIM_OK;
}

static int pass_fused(csim_stepper* s, const Phys& p, int T, int next_T, bool final_pass = false) {
    const bool rccl = s->multi && !s->external;
    if (rccl && s->bulk_first_run && s->faces_depth == 0)
        return pass_fused_bulk_first(s, p, T, final_pass);
    CSIM_SETTLE(s);
    int kind[4];
    for (int k = 0; k < 4; ++k) kind[k] = s->phys[k] ? s->bc[k] : 3;
    GhostArgs g = ghost_args(s);
    for (int k = 0; k < 4; ++k) g.recv[k] = nullptr;  // neighbour sides come from the deep faces
    const bool prepared = s->multi && s->faces_depth == T && s->pre_unpacked;
    s->pre_unpacked = false;
    if (s->multi) {
        if (s->faces_depth != T) {
            if (s->external)
                return fail(CSIM_ERR_STATE, "external halo transport: csim_stepper_faces_unpack (same depth) first");
            CSIM_HIP(launch_halo2_pack(s->cur, s->nx, s->ny, s->pitch, T, s->send2, s->s_comp));
            int rc = post_exchange2(s, T, s->s_comp);
            if (rc) return rc;
        } else if (rccl && s->overlap) {
            CSIM_HIP(hipStreamWaitEvent(s->s_comp, s->ev_recv2, 0));
        }
        if (!prepared) CSIM_HIP(launch_halo2_unpack(s->cur, s->nx, s->ny, s->pitch, T, s->recv2, s->s_comp));
    }
    if (!prepared && !s->ring_ok) {
        CSIM_HIP(launch_ghost_fill(s->cur, s->nxt, s->nx, s->ny, s->pitch, g, s->s_comp, s->multi ? T : 0));
        s->ring_ok = ring_is_static(s);
    }
    if (s->ring_ok) final_pass = false;  // nothing to rebuild after the last step: the ring is constant
    int rc = prof_begin(s, T);
    if (rc) return rc;
    if (rccl && s->overlap && next_T >= 2) {
        bool direct = false;
        if ((s->overlap == 3 || s->overlap == 5) && s->frame_flag) {
            // ONE launch: the frame tiles are the first blocks of the grid, the bulk tiles fill the rest of
            // the chip at once; the last frame wavefront publishes this pass's number and the comm stream,
            // parked on that value by the command processor, starts the exchange under the running kernel
            FrameSync fs;
            fs.counter = s->frame_counter;
            fs.flag = s->frame_flag;
            fs.pass = ++s->pass_no;
            fs.fence = s->frame_fence;
            fs.prio = s->frame_prio;
            direct = s->direct_faces && s->frame_fence == 0;
            if (direct) {  // the frame wavefronts write the next pass's faces into send2[] before they count themselves
                for (int d = 0; d < 8; ++d) fs.face[d] = s->send2[d];
                fs.face_depth = next_T;
            }
            CSIM_HIP(launch_fused(s, p, kind, T, 3, s->s_comp, false, 0, &fs));
            CSIM_HIP(hipStreamWaitValue64(s->s_comm, s->frame_flag, fs.pass, hipStreamWaitValueGte, ~0ull));
        } else {
            // FRAME tiles first (thin tiles along the four edges, ~15 us), then the BULK on the same
            // stream; as soon as the frame is done the comm stream packs the NEXT pass's faces from it
            // and runs the exchange, which the bulk hides
            CSIM_HIP(launch_fused(s, p, kind, T, 1, s->s_comp));
            CSIM_HIP(hipEventRecord(s->ev_edge2, s->s_comp));
            CSIM_HIP(hipStreamWaitEvent(s->s_comm, s->ev_edge2, 0));
        }
        long comm_slot = -1;
        rc = prof_start(s, csim_stepper::PROF_COMM, s->s_comm, &comm_slot);
        if (rc) return rc;
        if (!direct) CSIM_HIP(launch_halo2_pack(s->nxt, s->nx, s->ny, s->pitch, next_T, s->send2, s->s_comm));
        rc = post_exchange2(s, next_T, s->s_comm);
        if (rc) return rc;
        if (s->overlap == 1 || s->overlap == 3 || s->overlap == 5) {
            // the comm stream goes on to prepare the next pass — unpack of the faces into the new
            // field's halo cells, ghost fill of both buffers' rings — while the bulk is still
            // sweeping: those cells are disjoint from everything the bulk reads or writes, and
            // the frame cells the Neumann rule reads are final (the exchange waited for them)
            CSIM_HIP(launch_halo2_unpack(s->nxt, s->nx, s->ny, s->pitch, next_T, s->recv2, s->s_comm));
            CSIM_HIP(launch_ghost_fill(s->nxt, s->cur, s->nx, s->ny, s->pitch, g, s->s_comm, next_T));
            s->pre_unpacked = true;
        }
        rc = prof_stop(s, comm_slot, s->s_comm);
        if (rc) return rc;
        CSIM_HIP(hipEventRecord(s->ev_recv2, s->s_comm));
        if (!((s->overlap == 3 || s->overlap == 5) && s->frame_flag)) CSIM_HIP(launch_fused(s, p, kind, T, 2, s->s_comp));
        s->faces_depth = next_T;
    } else {
        CSIM_HIP(launch_fused(s, p, kind, T, 0, s->s_comp, final_pass));
        s->faces_depth = 0;
    }
    rc = prof_end(s);
    if (rc) return rc;
    std::swap(s->cur, s->nxt);
    s->halo_fresh = false;
    if (final_pass) {
        GhostArgs gf = ghost_args(s);
        for (int k = 0; k < 4; ++k) {
            gf.recv[k] = s->phys[k] ? nullptr : s->fin[k];  // the neighbour's edge line before the last step
            gf.adj[k] = s->phys[k] ? s->fin[k] : nullptr;   // own adjacent interior line before the last step
        }
        CSIM_HIP(launch_ghost_fill(s->cur, s->nxt, s->nx, s->ny, s->pitch, gf, s->s_comp));
    }
    return CSIM_OK;
}

// Rows per chunk of the fused sweep by trial: how a launch's wavefronts tile the 256 CUs (rounds
// of 4096 resident wavefronts, overhead rows per chunk) depends on the tile shape in a way no
// closed formula caught (tools/sweep_variants.py scans), so the stepper times the candidates on
// its own tile once: cur -> nxt launches WITHOUT a swap, i.e. the field is not advanced and the
// scratch interior written to nxt is overwritten by the next real pass.  Results never depend on
// the choice.  Ranks tune independently (no communication involved).
static int tune_rows(csim_stepper* s, const Phys& p, int T, bool preferred_depth = true) {
    if (preferred_depth) s->tuned = true;
    // small tiles: a launch takes a few tens of microseconds whatever the chunking, the trial
    // would cost more than it can win
    if (static_cast<long>(s->nx) * s->ny < (1L << 22)) return CSIM_OK;
    std::vector<int> cand;
    for (int ry = 6; ry <= 236 && ry <= s->ny; ry += (ry < 30 ? 4 : 6)) {
        const int snapped = ry + (6 - (ry + 2 * (T - 1)) % 6) % 6;
        if (snapped <= s->ny && (cand.empty() || cand.back() != snapped)) cand.push_back(snapped);
    }
    if (cand.size() < 2) return CSIM_OK;
    int kind[4];
    for (int k = 0; k < 4; ++k) kind[k] = s->phys[k] ? s->bc[k] : 3;
    struct EventPair {  // destroyed on every return path
        hipEvent_t a = nullptr, b = nullptr;
        ~EventPair() {
            if (a) (void)hipEventDestroy(a);
            if (b) (void)hipEventDestroy(b);
        }
    } ev;
    CSIM_HIP(hipEventCreateWithFlags(&ev.a, hipEventDisableSystemFence));  // timing only
    CSIM_HIP(hipEventCreateWithFlags(&ev.b, hipEventDisableSystemFence));
    const hipEvent_t e0 = ev.a, e1 = ev.b;
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    SweepCfg cfg = s->cfg;
    // what a pass of this stepper launches with the chunk height under trial: the whole tile on one rank, the BULK of
    // the tile (everything but the thin frame tiles, whose height is fixed) on a rank with neighbours
    const int part = s->multi ? 2 : 0;
    auto trial = [&](int ry, float* ms) -> int {
        cfg.tuned_rows = ry;
        CSIM_HIP(hipEventRecord(e0, s->s_comp));
        CSIM_HIP(launch_sweepO(s->cur, s->nxt, s->nx, s->ny, s->pitch, p, cfg, kind, s->bc_value, T, part, s->s_comp));
        CSIM_HIP(hipEventRecord(e1, s->s_comp));
        CSIM_HIP(hipEventSynchronize(e1));
        CSIM_HIP(hipEventElapsedTime(ms, e0, e1));
        return CSIM_OK;
    };
    // bring the clocks up first (a cold GPU runs its first ~20 ms well below the sustained rate)
    float ms = 0.f, spent = 0.f;
    for (int k = 0; k < 64 && spent < 30.f; ++k) {
        int rc = trial(cand[cand.size() / 2], &ms);
        if (rc) return rc;
        spent += ms;
    }
    std::vector<float> best(cand.size(), 1e30f);
    for (int round = 0; round < 3; ++round)
        for (size_t c = 0; c < cand.size(); ++c) {
            const size_t idx = (round & 1) ? cand.size() - 1 - c : c;  // alternate the order: drift cancels
            int rc = trial(cand[idx], &ms);
            if (rc) return rc;
            best[idx] = std::min(best[idx], ms);
        }
    // second stage: the candidates differ by a per cent or two, which is also the noise of three launches —
    // the four fastest get five more rounds each before the minimum decides
    std::vector<size_t> order(cand.size());
    for (size_t c = 0; c < order.size(); ++c) order[c] = c;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return best[a] < best[b]; });
    const size_t finalists = std::min<size_t>(4, order.size());
    for (int round = 0; round < 5; ++round)
        for (size_t q = 0; q < finalists; ++q) {
            const size_t idx = order[(round & 1) ? finalists - 1 - q : q];
            int rc = trial(cand[idx], &ms);
            if (rc) return rc;
            best[idx] = std::min(best[idx], ms);
        }
    size_t arg = order[0];
    for (size_t q = 1; q < finalists; ++q)
        if (best[order[q]] < best[arg]) arg = order[q];
    s->tuned_T[T] = cand[arg];
    if (preferred_depth) s->cfg.tuned_rows = cand[arg];
    return CSIM_OK;
}

// depth of the fused passes of this stepper with the current options (1 = single steps only): what
// the option "fuse" asks for, or pref_fuse(tile) — the depth with the lowest cost per step — in auto mode
static int fused_depth(const csim_stepper* s) {
    const int depth = std::min(s->fuse < 0 ? pref_fuse(s->tile_cells) : s->fuse, s->fuse_cap);
    const bool dpp_family = s->cfg.variant == VAR_AUTO || s->cfg.variant == VAR_DPP;
    return depth >= 2 && dpp_family ? depth : 1;
}

// Pass depths of a run of K steps.  "fuse" = N: as few passes as possible of balanced depth <= N.  Auto:
// the cheapest split by a small dynamic programme over the measured cost of one time step inside a pass
// of depth T relative to T = 6 (16384^2, DESIGN.md §7: shallow passes are HBM-bound and cost almost as much
// as a deep one: T = 5 costs 9 % more per step than T = 6; T = 7 pays 12.5 % instead of 9.4 % overlap columns
// but moves fewer bytes per step: 0.9 % cheaper than T = 6 on tiles of >= 2e8 cells, 6 % dearer on small ones,
// where it is still used when it saves a whole pass: 20 steps = 7 + 7 + 6 instead of 4 x 5) plus a small fixed
// cost per pass.  A run of
// two or more steps never contains a single-step pass unless it must (tiles only two cells deep and an odd
// K): one step alone costs 4.5 steps of a deep pass, so the programme avoids it by itself.  The result
// depends on (K, cap) only, so every rank of a decomposition derives the same schedule.
// (tools/depth_ab.py, profiles/r02_depth_ab.jsonl; relative to the preferred depth of the size class)
static const double STEP_COST_BIG[MAX_FUSE + 1] = {0.0, 4.52, 2.32, 1.55, 1.24, 1.09, 1.0, 0.991};    // >= 2e8 cells (16384^2, 32768^2)
static const double STEP_COST_MID[MAX_FUSE + 1] = {0.0, 4.52, 2.32, 1.55, 1.20, 1.04, 1.0, 1.005};    // 5e7 .. 2e8 (8192^2, 8192 x 16384)
static const double STEP_COST_MIDSMALL[MAX_FUSE + 1] = {0.0, 4.0, 2.0, 1.40, 1.12, 1.0, 1.0, 1.10};   // 1.2e7 .. 5e7 (4096^2, 4096 x 8192)
static const double STEP_COST_SMALL[MAX_FUSE + 1] = {0.0, 3.0, 1.5, 1.03, 1.0, 1.0, 1.2, 1.22};       // < 1.2e7 (relative to T = 4)
static const double* step_cost_table(long tile_cells) {
    if (tile_cells >= BIG_TILE_CELLS) return STEP_COST_BIG;
    if (tile_cells <= 0 || tile_cells >= 50000000L) return STEP_COST_MID;  // (0 = size unknown)
    return tile_cells >= SMALL_TILE_CELLS ? STEP_COST_MIDSMALL : STEP_COST_SMALL;
}
static const double PASS_COST = 0.1;   // launch and inter-kernel gap, in time steps of the preferred depth
// The plan is `lead` passes of depth `lead_depth` followed by the passes listed in `tail` (a run of 10^9
// steps must not materialise 10^8 entries).
struct PassPlan {
    long lead = 0;
    int lead_depth = 1;
    std::vector<int> tail;
    long size() const { return lead + static_cast<long>(tail.size()); }
    int at(long k) const { return k < lead ? lead_depth : tail[static_cast<size_t>(k - lead)]; }
};
static void plan_passes(int K, int cap, bool balanced, long tile_cells, PassPlan& plan) {
    const double* step_cost = step_cost_table(tile_cells);
    plan = PassPlan{};
    std::vector<int>& out = plan.tail;
    if (K <= 0) return;
    if (cap < 2) {
        plan.lead = K;
        plan.lead_depth = 1;
        return;
    }
    if (balanced) {
        // as few passes as possible, of balanced depth: all but the last few are of depth `cap`
        if (K > 4 * cap) {
            plan.lead = (K - 4 * cap) / cap;
            plan.lead_depth = cap;
        }
        int remaining = K - static_cast<int>(plan.lead) * cap;
        while (remaining > 0) {
            const int npass = (remaining + cap - 1) / cap;
            const int t = remaining < 2 ? 1 : (remaining + npass - 1) / npass;
            out.push_back(t);
            remaining -= t;
        }
        return;
    }
    const int pref = std::min(cap, pref_fuse(tile_cells));
    // long runs: passes of the preferred depth, the last <= 8 * pref steps are planned
    if (K > 8 * pref) plan.lead = (K - 8 * pref + pref - 1) / pref;
    plan.lead_depth = pref;
    const int R = K - static_cast<int>(plan.lead) * pref;
    std::vector<double> best(static_cast<size_t>(R) + 1, 1e300);
    std::vector<int> pick(static_cast<size_t>(R) + 1, 0);
    best[0] = 0.0;
    for (int k = 1; k <= R; ++k)
        for (int t = 1; t <= std::min(cap, k); ++t) {
            if (t == 1 && cap >= 3 && K >= 2) continue;  // every k >= 2 splits into 2s and 3s: no single-step pass
            const double c = best[static_cast<size_t>(k - t)] + t * step_cost[t] + PASS_COST;
            if (c < best[static_cast<size_t>(k)]) {
                best[static_cast<size_t>(k)] = c;
                pick[static_cast<size_t>(k)] = t;
            }
        }
    std::vector<int> tail;
    for (int k = R; k > 0; k -= pick[static_cast<size_t>(k)]) tail.push_back(pick[static_cast<size_t>(k)]);
    std::sort(tail.begin(), tail.end(), [](int a, int b) { return a > b; });  // deep passes first
    out.insert(out.end(), tail.begin(), tail.end());
}

// the pass schedule as pure host arithmetic (no GPU): what csim_stepper_run(nsteps) will launch on a
// decomposition whose smallest tile is `smallest_tile` cells deep, with option "fuse" = `fuse`
int csim_pass_schedule(int nsteps, int smallest_tile, long tile_cells, int fuse, int* depths, int max_depths,
                       long* npasses) {
    CSIM_REQUIRE(npasses && nsteps >= 0 && smallest_tile >= 1, "bad argument");
    CSIM_REQUIRE(fuse >= -1 && fuse <= MAX_FUSE, "fuse must be -1 (auto) or 0..7");
    CSIM_REQUIRE(max_depths == 0 || depths, "depths is null");
    const int fuse_cap = std::max(1, std::min(MAX_FUSE, smallest_tile));
    const int depth = std::min(fuse < 0 ? pref_fuse(tile_cells) : fuse, fuse_cap);
    const int cap = depth < 2 ? 1 : fuse < 0 ? std::min(MAX_FUSE, fuse_cap) : depth;
    PassPlan plan;
    plan_passes(nsteps, cap, fuse >= 0, tile_cells, plan);
    *npasses = plan.size();
    for (long k = 0; k < plan.size() && k < max_depths; ++k) depths[k] = plan.at(k);
    return CSIM_OK;
}

// the one-off trial of csim_stepper_run's first long call, on request (e.g. before a timed loop)
int csim_stepper_tune(csim_stepper* s, double D, double dt, double vx, double vy) {
    CSIM_REQUIRE(s, "null stepper");
    CSIM_SETTLE(s);
    const int depth = fused_depth(s);
    if (depth < 2 || s->cfg.rows_per_chunk != 0) return CSIM_OK;
    Phys p = make_phys(s->dx, s->dy, D, dt, vx, vy, s->contract != 0);
    if (!s->fused_2c) p.fast_thr = 0.0;
    if (!s->tuned) {
        int rc = tune_rows(s, p, depth);
        if (rc) return rc;
    }
    // the other depths an automatic pass plan mixes in (20 steps = 7 + 7 + 6, remainders of 4 and 5): each has its own
    // balance of overhead rows per chunk against rounds of wavefronts, so each gets its own trial
    if (s->fuse < 0)
        for (int T = std::min(MAX_FUSE, s->fuse_cap); T >= 4; --T)
            if (T != depth && s->tuned_T[T] == 0) {
                int rc = tune_rows(s, p, T, false);
                if (rc) return rc;
            }
    return CSIM_OK;
}

// Load without effect: whole-tile launches cur -> nxt of the multi-step sweep without a swap (what tune_rows
// does), one at a time, until the next one would end after `seconds`.  No exchange, no ghost fill: the scratch
// interior left in nxt is overwritten by the next real pass.
int csim_stepper_keep_warm(csim_stepper* s, double D, double dt, double vx, double vy, double seconds) {
    CSIM_REQUIRE(s, "null stepper");
    CSIM_REQUIRE(seconds >= 0.0 && seconds <= 10.0, "seconds must be in [0, 10]");
    CSIM_SETTLE(s);
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    const int depth = fused_depth(s);
    if (depth < 2) return CSIM_OK;
    Phys p = make_phys(s->dx, s->dy, D, dt, vx, vy, s->contract != 0);
    if (!s->fused_2c) p.fast_thr = 0.0;
    int kind[4];
    for (int k = 0; k < 4; ++k) kind[k] = s->phys[k] ? s->bc[k] : 3;
    CSIM_HIP(hipStreamSynchronize(s->s_comm));
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    double last = 0.0;
    for (int n = 0; n < 100000; ++n) {
        const double before = elapsed();
        if (before + 1.25 * last >= seconds) break;  // the next launch would run past the deadline
        CSIM_HIP(launch_sweepO(s->cur, s->nxt, s->nx, s->ny, s->pitch, p, s->cfg, kind, s->bc_value, depth, 0, s->s_comp));
        CSIM_HIP(hipStreamSynchronize(s->s_comp));
        last = elapsed() - before;
    }
    return CSIM_OK;
}

int csim_stepper_run(csim_stepper* s, double D, double dt, double vx, double vy, int nsteps) {
    CSIM_REQUIRE(s, "null stepper");
    CSIM_REQUIRE(nsteps >= 0, "nsteps must be >= 0");
    // Up to MAX_FUSE steps per HBM pass where possible (across ranks: a tile at least as large as
    // the face depth).
    const int depth = fused_depth(s);
    const bool can_fuse = depth >= 2;
    const bool auto_depth = s->fuse < 0;
    const int cap = !can_fuse ? 1 : auto_depth ? std::min(MAX_FUSE, s->fuse_cap) : depth;
    if (s->multi && s->external) {
        // the caller carries the faces: one step (depth-1 faces) or one fused pass per call
        if (nsteps == 1 && !s->halo_fresh)
            return fail(CSIM_ERR_STATE, "external halo transport: csim_stepper_halo_unpack first");
        if (nsteps >= 2 && !(can_fuse && nsteps <= cap))
            return fail(CSIM_ERR_STATE, "external halo transport: a call advances 1 step or one fused pass");
    } else if (s->multi && !s->comm) {
        return fail(CSIM_ERR_STATE, "multi-rank stepper needs csim_stepper_comm_init before run");
    }
    // Invariant every schedule relies on: a run starts with no deep faces staged and nothing pre-unpacked (each
    // run's last pass has no successor, so it ends that way; external mode stages faces explicitly per call).
    if (!s->external && (s->faces_depth != 0 || s->pre_unpacked))
        return fail(CSIM_ERR_STATE, "internal: csim_stepper_run entered with faces of a fused pass in flight "
                                    "(an earlier call failed half-way?): upload or re-initialise the field");
    Phys p = make_phys(s->dx, s->dy, D, dt, vx, vy, s->contract != 0);
    if (!s->fused_2c) p.fast_thr = 0.0;
    s->fused_2c_active = p.fast_thr > 0.0 && p.div_mode != 3;
    const GhostArgs g = ghost_args(s);
    if (s->multi && s->external && nsteps >= 2) return pass_fused(s, p, nsteps, 0);
    // The ghost ring left in the field must be exactly the reference's: the halos / boundary values
    // of the state before the LAST step (src/main.cpp:104 + src/diffusion.cpp:18-25).
    // Every pass is fused, the last one as `final_pass` (see pass_fused); a run of two or more
    // steps never contains a single-step pass.
    if (can_fuse && s->autotune && !s->tuned && s->cfg.rows_per_chunk == 0 && nsteps >= 4 * depth) {
        CSIM_SETTLE(s);
        int rc = tune_rows(s, p, depth);
        if (rc) return rc;
    }
    PassPlan plan;
    plan_passes(nsteps, cap, !auto_depth, s->tile_cells, plan);
    // exchange schedule of this run: bulk-first (4, and the default 5).  Until round 3 the default went bulk-first only on
    // runs of fewer than 16 passes and merged (3) otherwise; with the relay (pass_fused_bulk_first) bulk-first is the
    // faster one at every run length on every per-GPU tile of the 16384^2 run (self-linked torus, 1200-step runs:
    // 4096 x 8192 1.26-1.27 M against 1.11-1.19 M merged, 8192 x 16384 1.49-1.50 M against 1.45 M, 8192^2 equal), and
    // it needs nothing but stream order and events: no in-kernel flag, no hipStreamWaitValue64, no write-through stores.
    s->bulk_first_run = s->multi && !s->external && (s->overlap == 4 || s->overlap == 5);
    for (long k = 0; k < plan.size(); ++k) {
        const int t = plan.at(k);
        int rc;
        if (t >= 2) {
            const bool last = k + 1 == plan.size();
            const int nt = last ? 0 : plan.at(k + 1);
            rc = pass_fused(s, p, t, nt >= 2 ? nt : 0, last);
        } else {
            rc = pass_single(s, p, g);
        }
        if (rc) return rc;
    }
    return prof_close(s);
}

// Wait for both streams.  With an RCCL communicator the wait polls instead of blocking, so that an asynchronous
// communicator error (a peer that died, a failed transport: ncclCommGetAsyncError) ends it with CSIM_ERR_RCCL
// instead of a silent hang on a stream nobody will ever complete — the reference's MPI_Waitall
// (src/halo.cpp:46) would abort the job through the MPI error handler.  Option "sync_timeout_ms" > 0 bounds the
// wait (CSIM_ERR_TIMEOUT; the streams stay as they are).
static int wait_stream(csim_stepper* s, hipStream_t st, const std::chrono::steady_clock::time_point& t0) {
    if (!s->comm && s->sync_timeout_ms <= 0 && !s->stall_armed) {
        CSIM_HIP(hipStreamSynchronize(st));
        return CSIM_OK;
    }
    for (unsigned long spin = 0;; ++spin) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return CSIM_OK;
        if (q != hipErrorNotReady) return fail(CSIM_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(q));
        (void)hipGetLastError();  // hipErrorNotReady is sticky in the last-error slot
        if ((spin & 63) == 63) {
            if (s->comm) {
                ncclResult_t async = ncclSuccess;
                const ncclResult_t r = ncclCommGetAsyncError(s->comm, &async);
                if (r != ncclSuccess || (async != ncclSuccess && async != ncclInProgress))
                    return fail(CSIM_ERR_RCCL, std::string("halo exchange failed asynchronously: ") +
                                                   ncclGetErrorString(r != ncclSuccess ? r : async));
            }
            if (s->sync_timeout_ms > 0) {
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (ms > static_cast<double>(s->sync_timeout_ms))
                    return fail(CSIM_ERR_TIMEOUT, "csim_stepper_sync: streams still busy after " +
                                                      std::to_string(s->sync_timeout_ms) + " ms (option sync_timeout_ms)");
            }
        }
    }
}

int csim_stepper_sync(csim_stepper* s) {
    CSIM_REQUIRE(s, "null stepper");
    const auto t0 = std::chrono::steady_clock::now();
    int rc = wait_stream(s, s->s_comp, t0);
    if (rc) return rc;
    return wait_stream(s, s->s_comm, t0);
}

// Position-weighted 64-bit checksum of the local interior (k_checksum): the per-rank values of a decomposition
// add up modulo 2^64 to the checksum of the same global field on one rank.
int csim_stepper_checksum(csim_stepper* s, unsigned long long* out) {
    CSIM_REQUIRE(s && out, "null argument");
    CSIM_SETTLE(s);
    const long nxg = s->dec.nx_global > 0 ? s->dec.nx_global : s->nx;
    CSIM_HIP(launch_checksum(s->cur, s->nx, s->ny, s->pitch, s->dec.x_offset, s->dec.y_offset, nxg, s->scratch, s->s_comp));
    const int nb = reduce_blocks(s->ny);
    std::vector<unsigned long long> h(static_cast<size_t>(nb));
    CSIM_HIP(hipMemcpyAsync(h.data(), s->scratch, sizeof(unsigned long long) * nb, hipMemcpyDeviceToHost, s->s_comp));
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    unsigned long long acc = 0;
    for (int k = 0; k < nb; ++k) acc += h[static_cast<size_t>(k)];
    *out = acc;
    return CSIM_OK;
}

int csim_stepper_minmax(csim_stepper* s, double out[2]) {
    CSIM_REQUIRE(s && out, "null argument");
    CSIM_SETTLE(s);
    CSIM_HIP(launch_minmax(s->cur, s->nx, s->ny, s->pitch, s->scratch, s->s_comp));
    return finish_partials(s->scratch, reduce_blocks(s->ny + 2), 0, out, s->s_comp);
}

int csim_stepper_sum(csim_stepper* s, double* out) {
    CSIM_REQUIRE(s && out, "null argument");
    CSIM_SETTLE(s);
    double r[2];
    CSIM_HIP(launch_sum(s->cur, s->nx, s->ny, s->pitch, s->scratch, s->s_comp));
    int rc = finish_partials(s->scratch, reduce_blocks(s->ny), 1, r, s->s_comp);
    *out = r[0];
    return rc;
}

int csim_stepper_set_option(csim_stepper* s, const char* key, long value) {
    CSIM_REQUIRE(s && key, "null argument");
    const std::string k(key);
    if (k == "variant") {
        CSIM_REQUIRE(value >= VAR_AUTO && value <= VAR_NAIVE, "unknown variant");
        s->cfg.variant = static_cast<int>(value);
    } else if (k == "rows_per_chunk") {
        CSIM_REQUIRE(value >= 0, "rows_per_chunk must be >= 0");
        s->cfg.rows_per_chunk = static_cast<int>(value);
    } else if (k == "prefetch") {
        CSIM_REQUIRE(value >= 0 && value <= 8, "prefetch must be 0..8");
        s->cfg.prefetch = static_cast<int>(value);
    } else if (k == "xcd_swizzle") {
        s->cfg.xcd_swizzle = value != 0;
    } else if (k == "tail_split") {
        CSIM_REQUIRE(value >= 0 && value <= 2, "tail_split must be 0, 1 or 2");
        s->cfg.tail_split = static_cast<int>(value);
        s->forget_tuning();  // the best chunk height depends on it
    } else if (k == "overlap") {
        CSIM_REQUIRE(value >= 0 && value <= 5 && value != 2, "overlap must be 0, 1, 3, 4 or 5");
        // the schedules hand state to each other only through "nothing in flight": every csim_stepper_run ends that way
        if (!s->external && (s->faces_depth != 0 || s->pre_unpacked))
            return fail(CSIM_ERR_STATE, "internal: exchange schedule changed with faces of a fused pass in flight");
        if (value == 3 && s->multi && !s->frame_flag)
            return fail(CSIM_ERR_STATE, "overlap 3 needs hipStreamWaitValue64 / signal memory, which this device or runtime refused");
        s->overlap = static_cast<int>(value);
    } else if (k == "frame_fence") {
        CSIM_REQUIRE(value >= 0 && value <= 2, "frame_fence must be 0..2");
        s->frame_fence = static_cast<int>(value);
    } else if (k == "frame_rows") {
        CSIM_REQUIRE(value >= 0 && value <= 4096, "frame_rows must be 0..4096");
        s->cfg.frame_rows = static_cast<int>(value);
    } else if (k == "frame_prio") {
        s->frame_prio = value != 0;
    } else if (k == "relay") {
        CSIM_SETTLE(s);
        s->relay = value != 0;
    } else if (k == "direct_faces") {
        s->direct_faces = value != 0;
    } else if (k == "fused_2c") {
        s->fused_2c = value != 0;
    } else if (k == "external_halo") {
        s->external = value != 0;
        s->halo_fresh = false;
    } else if (k == "contract") {
        CSIM_REQUIRE(value == 0 || value == 1, "contract must be 0 (reference operation order, default) or 1");
        if (s->contract != static_cast<int>(value)) s->forget_tuning();  // another kernel: its best chunk height is found anew
        s->contract = static_cast<int>(value);
    } else if (k == "fuse") {
        CSIM_REQUIRE(value >= -1 && value <= MAX_FUSE, "fuse must be -1 (auto) or 0..7");
        s->fuse = static_cast<int>(value);
    } else if (k == "lds_bytes") {
        CSIM_REQUIRE(value >= 0 && value <= 65536, "lds_bytes must be 0..65536");
        s->cfg.lds_bytes = static_cast<int>(value);
    } else if (k == "autotune") {
        s->autotune = value != 0;
        s->forget_tuning();
    } else if (k == "tuned_rows" || k == "last_rows") {  // read back through csim_stepper_get_option
        return fail(CSIM_ERR_ARG, k + " is read-only");
    } else if (k == "sync_timeout_ms") {
        CSIM_REQUIRE(value >= 0, "sync_timeout_ms must be >= 0");
        s->sync_timeout_ms = value;
    } else if (k == "test_stall") {
        // Test hook for the stall handling of callers (bench.py's watchdog, csim_stepper_sync's timeout): 1 parks the
        // comm stream on a value of the signal word that no kernel ever publishes — exactly what a lost flag or a
        // dead peer looks like from the host —, 0 releases it from the host and restores the word.
        CSIM_REQUIRE(value == 0 || value == 1, "test_stall must be 0 or 1");
        if (!s->frame_flag) return fail(CSIM_ERR_STATE, "test_stall needs a multi-rank stepper with signal memory");
        constexpr unsigned long long NEVER = 1ull << 62;
        if (value == 1 && !s->stall_armed) {
            CSIM_HIP(hipStreamWaitValue64(s->s_comm, s->frame_flag, NEVER, hipStreamWaitValueGte, ~0ull));
            s->stall_armed = true;
        } else if (value == 0 && s->stall_armed) {
            __atomic_store_n(s->frame_flag, NEVER, __ATOMIC_SEQ_CST);
            CSIM_HIP(hipStreamSynchronize(s->s_comm));
            __atomic_store_n(s->frame_flag, s->pass_no, __ATOMIC_SEQ_CST);
            s->stall_armed = false;
        }
    } else if (k == "profile") {
        CSIM_REQUIRE(value >= 0 && value <= 1024, "profile must be 0..1024");
        s->profile = static_cast<int>(value);
        s->prof_counter = 0;
    } else {
        return fail(CSIM_ERR_ARG, "unknown option: " + k);
    }
    return CSIM_OK;
}

int csim_stepper_get_option(const csim_stepper* s, const char* key, long* value) {
    CSIM_REQUIRE(s && key && value, "null argument");
    const std::string k(key);
    if (k == "variant") *value = s->cfg.variant;
    else if (k == "rows_per_chunk") *value = s->cfg.rows_per_chunk;
    else if (k == "tuned_rows") *value = s->cfg.tuned_rows;
    else if (k.size() == 12 && k.compare(0, 11, "tuned_rows_") == 0 && k[11] >= '2' && k[11] <= '0' + MAX_FUSE)
        *value = s->tuned_T[k[11] - '0'];  // "tuned_rows_2" .. "tuned_rows_7": the trial's result for passes of that depth (0 = none)
    else if (k == "last_rows") *value = s->last_rows;
    else if (k == "prefetch") *value = s->cfg.prefetch;
    else if (k == "xcd_swizzle") *value = s->cfg.xcd_swizzle;
    else if (k == "tail_split") *value = s->cfg.tail_split;
    else if (k == "overlap") *value = s->overlap;
    else if (k == "direct_faces") *value = s->direct_faces;
    else if (k == "relay") *value = s->relay;
    else if (k == "fused_2c") *value = s->fused_2c;
    else if (k == "fused_2c_active") *value = s->fused_2c_active;
    else if (k == "frame_rows") *value = s->cfg.frame_rows;
    else if (k == "external_halo") *value = s->external;
    else if (k == "fuse") *value = s->fuse;
    else if (k == "contract") *value = s->contract;
    else if (k == "autotune") *value = s->autotune;
    else if (k == "profile") *value = s->profile;
    else if (k == "sync_timeout_ms") *value = s->sync_timeout_ms;
    else if (k == "test_stall") *value = s->stall_armed;
    else if (k == "faces_in_flight") *value = s->faces_depth != 0 || s->pre_unpacked;
    else return fail(CSIM_ERR_ARG, "unknown option: " + k);
    return CSIM_OK;
}

int csim_stepper_kernel_time(csim_stepper* s, int steps_per_launch, double* total_ms,
                             long* launches) {
    CSIM_REQUIRE(s && total_ms && launches, "null argument");
    CSIM_REQUIRE(steps_per_launch >= 1 && steps_per_launch <= MAX_FUSE, "steps_per_launch must be 1..7");
    int rc = prof_fold(s);
    if (rc) return rc;
    *total_ms = s->prof_ms[steps_per_launch];
    *launches = s->prof_launches[steps_per_launch];
    return CSIM_OK;
}

int csim_stepper_reset_timers(csim_stepper* s) {
    CSIM_REQUIRE(s, "null stepper");
    int rc = prof_fold(s);
    if (rc) return rc;
    for (int t = 0; t <= csim_stepper::PROF_COMM; ++t) {
        s->prof_ms[t] = 0.0;
        s->prof_launches[t] = 0;
    }
    return CSIM_OK;
}

int csim_stepper_comm_time(csim_stepper* s, double* total_ms, long* passes) {
    CSIM_REQUIRE(s && total_ms && passes, "null argument");
    int rc = prof_fold(s);
    if (rc) return rc;
    *total_ms = s->prof_ms[csim_stepper::PROF_COMM];
    *passes = s->prof_launches[csim_stepper::PROF_COMM];
    return CSIM_OK;
}

}  // extern "C"

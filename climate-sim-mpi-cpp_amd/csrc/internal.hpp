// internal.hpp — shared declarations of the HIP engine behind include/csim.h.
//
// Device layout of a Field (MI355X-first, not the host layout): rows are padded to a pitch
// so that the FIRST INTERIOR column (i = 1) of every row starts on a 128-byte boundary and a
// 64-lane wavefront reading 2 doubles per lane covers exactly 8 full 128-byte lines:
//
//     element (i, j), 0 <= i <= nx+1, 0 <= j <= ny+1   ->   view[j * pitch + (LPAD - 1) + i]
//
//     | 15 unused | ghost i=0 | interior i=1..nx (128-B aligned) | ghost i=nx+1 | pad ... |
//
// `view` points GHOST_EXTRA rows into the allocation: rows -6..-1 and ny+2..ny+7 (and the pad
// columns left of i = 0 / right of i = nx+1) exist as device-only ghost layers, which the
// multi-step-per-pass sweeps read (as halo data on neighbour sides, as don't-care otherwise).
//
// pitch = LPAD + round_up(nx + 1, 128) + 128 doubles; for nx = 16384 that is 16656 doubles =
// 133248 B, an odd multiple of 128 B, so vertically adjacent rows do not alias onto one HBM
// channel.  Pads are zero and never written.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdio>
#include <string>

#include "csim.h"

namespace csim {

constexpr int LPAD = 16;        // doubles in front of the first interior column
constexpr int WAVE_COLS = 128;  // columns one wavefront covers per row (64 lanes x 2 doubles)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline int pitch_for(int nx) { return LPAD + round_up(nx + 1, WAVE_COLS) + WAVE_COLS; }

int fail(int code, const std::string& msg);

#define CSIM_HIP(expr)                                                                       \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return ::csim::fail(CSIM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// physics constants of one sweep, precomputed on the host exactly as the reference forms them
struct Phys {
    double kdiff;  // dt * D            (reference src/diffusion.cpp:15  "dt * D * lap")
    double mdt;    // -dt               (reference src/advection.cpp:31  "(-dt) * adv")
    double vx, vy;
    double dx, dy;      // divisors of the upwind differences
    double dx2, dy2;    // dx*dx, dy*dy: divisors of the second differences
    double rdx, rdy, rdx2, rdy2;  // exact reciprocals (only used when all four are powers of two)
    int div_mode;       // 0: dx == dy == 1 (x/1 is x)   1: exact reciprocal multiply   2: IEEE divide
                        // 3: option "contract" — the coefficient form below (not bit-identical)
    double a0, aW, aE, aS, aN;  // u' = a0 c + aW W + aE E + aS S + aN N, the same update as one 5-point stencil
    double fast_thr;    // > 0: k_sweepO_dpp may fuse E - 2c into one fma on tiles whose loaded values are all below
                        // this magnitude (no 2c of any level can overflow then); 0: always the plain form
};
Phys make_phys(double dx, double dy, double D, double dt, double vx, double vy, bool contract = false);

// kernel variants of the fused sweep (option "variant")
enum { VAR_AUTO = 0, VAR_DPP = 1, VAR_LDS = 2, VAR_NAIVE = 3 };

struct SweepCfg {
    int variant = VAR_AUTO;
    int rows_per_chunk = 0;  // 0 = auto
    int tuned_rows = 0;      // auto mode: rows per chunk found by the stepper's on-device trial (0 = heuristic)
    int lds_bytes = 0;       // dynamic LDS requested per workgroup of the fused sweep: an occupancy limiter
                             // (41 KB -> 3 workgroups per CU instead of 4), the kernel never touches it
    int prefetch = 0;        // 0 = auto (rows kept in flight per wavefront)
    int xcd_swizzle = 1;
    int tail_split = 1;        // fused launches of two or more rounds of wavefronts end with a region of half-height
                               // chunks (see Tiling); 0 off, 2 experiment (half + quarter height)
    int frame_rows = 0;        // multi-rank pass: chunk height of the frame's side strips (0 = as thin as the bands)
    int* rows_used = nullptr;  // out: chunk height of the last whole-field / bulk launch (option "last_rows")
};

// ---- kernel launchers (kernels.hip) --------------------------------------------------------
// All pointers are device pointers in the padded layout above.
hipError_t launch_sweep(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                        const SweepCfg& cfg, hipStream_t st);
// T = 2..7 fused time steps per pass (overlapped strips).  kind[s] = CSIM_BC_* on physical sides, 3 on
// neighbour sides; part: 0 = every tile, 1 = frame tiles only, 2 = all but the frame tiles.
// fin_lines (last pass of a run, all four or nullptr): per side the level T-1 line the final ghost
// fill needs — see FinLines in kernels.hip
// part 3 = frame and bulk in ONE grid (frame tiles dispatched first); with `sync` the frame wavefronts
// count themselves on sync->counter and the last one stores sync->pass into sync->flag (signal memory a
// stream can wait on with hipStreamWaitValue64), see k_sweepO_dpp
struct FrameSync {
    unsigned* counter = nullptr;
    unsigned long long* flag = nullptr;
    unsigned long long pass = 0;
    unsigned nframe = 0;  // filled in by the launcher
    int fence = 0;        // 0 write-through result stores + drain (default), 1 plain stores + agent-scope fence per
                          // wavefront (slow: +40 us per pass)
    int prio = 1;         // frame wavefronts raise their issue priority
    // direct faces (face_depth > 0): the frame wavefronts also copy the cells that form the faces of the NEXT pass
    // (depth face_depth, k_halo2_pack's layout) straight into the send buffers; nullptr = no peer in that direction
    double* face[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int face_depth = 0;
};
hipError_t launch_sweepO(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                         const SweepCfg& cfg, const int kind[4], double value, int T, int part,
                         hipStream_t st, double* const fin_lines[4] = nullptr, const FrameSync* sync = nullptr);
constexpr int MAX_FUSE = 7;       // deepest temporal blocking (123 VGPRs: still 4 waves/SIMD; 8 would drop to 3)
// depth with the lowest measured cost per time step (tools/depth_ab.py, profiles/r02_depth_ab.jsonl): 7 on tiles
// of >= 2e8 cells (+0.9 % over 6 at 16384^2 and 32768^2); 6 in between (7 loses 7 % on a 4096 x 8192 tile, which is
// a single round of wavefronts: more overhead rows per chunk and nothing to amortise them); 4 on tiles below
// 1.2e7 cells, where a launch is a fraction of one round and the length of a wavefront's dependent chain
// decides (+17...22 % over 6 at 512^2, 1024^2, 3072^2)
constexpr long BIG_TILE_CELLS = 200000000L;
constexpr long SMALL_TILE_CELLS = 12000000L;
// `still` — the diffusion-only flavour (v == 0: half the arithmetic, so the sweep is HBM-bound and every extra level
// per pass pays): 7 at every size (profiles/r03_diffusion_only_depth.jsonl: +8 % over 6 at 4096^2, +13 % at 8192^2 and 16384^2)
inline int pref_fuse(long tile_cells, bool still = false) {
    if (still) return 7;
    return tile_cells >= BIG_TILE_CELLS ? 7 : (tile_cells > 0 && tile_cells < SMALL_TILE_CELLS) ? 4 : 6;
}
constexpr int GHOST_EXTRA = 6;    // device-only ghost layers beyond the reference's one (= MAX_FUSE-1)
// faces of depth H = 2..7 (8 directions: L R B T BL BR TL TR; nullptr = no neighbour there);
// sizes H*(ny+2) (L,R), H*(nx+2) (B,T), H*H (corners)
hipError_t launch_halo2_pack(const double* f, int nx, int ny, int pitch, int depth,
                             double* const send[8], hipStream_t st);
hipError_t launch_halo2_unpack(double* f, int nx, int ny, int pitch, int depth,
                               double* const recv[8], hipStream_t st);

hipError_t launch_diffusion_only(const double* in, double* out, int nx, int ny, int pitch,
                                 const Phys& p, hipStream_t st);
hipError_t launch_advection_only(const double* in, double* out, int nx, int ny, int pitch,
                                 const Phys& p, hipStream_t st);
hipError_t launch_ring_copy(const double* in, double* out, int nx, int ny, int pitch, hipStream_t st);
hipError_t launch_fill(double* f, int nx, int ny, int pitch, double v, hipStream_t st);

struct GhostArgs {
    int bc[4];
    int phys[4];          // side is a physical edge (no neighbour)
    double value;
    const double* recv[4];  // per side: staged halo from the neighbour (nullptr on physical sides)
    const double* adj[4];   // per side: adjacent interior line to read instead of the field (nullptr = the field)
};
// boundary fill (+ unpack of received halos) written to `a` and, when b != nullptr, to `b` too
// ext_depth > 0 additionally continues each physical Dirichlet/Neumann edge over that many halo
// cells of an adjacent neighbour side (needed by fused passes across ranks; reads halo cells a
// preceding launch_halo2_unpack wrote)
hipError_t launch_ghost_fill(double* a, double* b, int nx, int ny, int pitch, const GhostArgs& g,
                             hipStream_t st, int ext_depth = 0);
// updated values of the four edge lines of the NEXT field, computed from `in` and written
// straight into the send staging buffers (nullptr = side not needed)
hipError_t launch_edge_pack(const double* in, int nx, int ny, int pitch, const Phys& p,
                            double* const send[4], hipStream_t st);
// plain pack of the current edge lines (used by csim_stepper_exchange_halos)
hipError_t launch_pack(const double* in, int nx, int ny, int pitch, double* const send[4],
                       hipStream_t st);
hipError_t launch_gaussian(double* f, int nx, int ny, int pitch, int x_off, int y_off, int nxg,
                           int nyg, double dx, double dy, double A, double sigma_frac,
                           double xc_frac, double yc_frac, hipStream_t st);

// reductions: partial results per block in `scratch` (>= 2 * REDUCE_BLOCKS doubles), finished on host
constexpr int REDUCE_BLOCKS = 1024;
hipError_t launch_minmax(const double* f, int nx, int ny, int pitch, double* scratch, hipStream_t st);
hipError_t launch_sum(const double* f, int nx, int ny, int pitch, double* scratch, hipStream_t st);
hipError_t launch_linf(const double* a, const double* b, int nx, int ny, int pitch, double* scratch,
                       hipStream_t st);
// position-weighted 64-bit checksum of the interior (k_checksum): min(ny, REDUCE_BLOCKS) u64 partials in `scratch`
hipError_t launch_checksum(const double* f, int nx, int ny, int pitch, long x_off, long y_off, long nx_global,
                           double* scratch, hipStream_t st);

}  // namespace csim

// ---- opaque handle types -------------------------------------------------------------------
struct csim_field {
    int nx = 0, ny = 0, halo = 1;
    double dx = 1.0, dy = 1.0;
    int pitch = 0;
    double* alloc = nullptr;    // (ny + 2 + 2 * GHOST_EXTRA) * pitch doubles
    double* d = nullptr;        // view: alloc + GHOST_EXTRA * pitch, i.e. row j = 0 of the reference layout
    double* scratch = nullptr;  // reduction partials
    size_t bytes() const {
        return sizeof(double) * static_cast<size_t>(ny + 2 + 2 * csim::GHOST_EXTRA) * pitch;
    }
};

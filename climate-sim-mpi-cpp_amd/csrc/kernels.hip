// kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the advection–diffusion hot path.
//
// The path is HBM-bandwidth-bound by nature (16 algorithmic bytes and ~14 fp64 flops per cell
// update), so there is no MFMA here.  What matters:
//   * 16-byte-per-lane coalesced row accesses on 128-byte-aligned rows, each cell read once and
//     written once per PASS: vertical reuse in registers while a wavefront marches up its column
//     strip, horizontal reuse through cross-lane DPP moves (an LDS-staged variant is kept for
//     comparison), row loads kept in flight to cover HBM latency, XCD-aware block->tile map;
//   * temporal blocking: up to seven time levels stay in registers per pass (k_sweepO_dpp, the
//     default), which divides the HBM traffic per step by as much and leaves the kernel bound by the
//     reference's own fp64 add/mul stream.
// Kernels, in file order: k_sweep_dpp (1 step/pass), k_sweepO_dpp (2-7 steps/pass, overlapped
// strips, DEFAULT), k_sweep_lds (the LDS-staged design, measured alternative), k_sweep_naive
// (strawman), then the small kernels (ghost fill / extend, edge and face packing,
// reference-granularity operators, reductions, the bit-identity checksum).
//
// Arithmetic follows the reference's association order exactly (reference
// src/diffusion.cpp:9-16, src/advection.cpp:13-33) and this file is compiled with
// -ffp-contract=off, so every kernel is bit-identical to the reference CPU path.
#include <algorithm>
#include <type_traits>

#include "internal.hpp"

#pragma clang fp contract(off)

namespace csim {

// -------------------------------------------------------------------------------------------
// per-cell update:  o = c + (dt*D)*lap;  o = o + (-dt)*(vx*dudx + vy*dudy)
//   lap  = ((E - 2c) + W)/(dx*dx) + ((N - 2c) + S)/(dy*dy)
//   dudx = vx >= 0 ? (c - W)/dx : (E - c)/dx      (dudy likewise)
// DIV 0: dx == dy == 1, x/1 == x.  DIV 1: all divisors are powers of two, so x * (1/d) is the
// correctly rounded quotient too (bit-identical to x/d).  DIV 2: true IEEE fp64 division.
// -------------------------------------------------------------------------------------------
// FAST: E - 2c as ONE operation, fma(-2, c, E).  2c is exact in binary floating point (subnormals included), so
// the fused form rounds the same real number E - 2c once, exactly like the subtraction does — unless 2c
// overflows (|c| >= 2^1023), where the reference gets +-inf and the fma a finite number.  Only k_sweepO_dpp's
// interior body uses it, under a guard that re-runs the tile with the plain form if that could happen (see
// sweepO_march); it removes one of the 15 fp64 operations per cell.
template <int DIV, bool FAST = false>
__device__ __forceinline__ double diffuse_term(double c, double W, double E, double S, double N,
                                               const Phys& p) {
    double lx, ly;
    if (FAST) {
        lx = __builtin_fma(-2.0, c, E) + W;
        ly = __builtin_fma(-2.0, c, N) + S;
    } else {
        const double tc = 2.0 * c;
        lx = (E - tc) + W;
        ly = (N - tc) + S;
    }
    if (DIV == 1) {
        lx = lx * p.rdx2;
        ly = ly * p.rdy2;
    } else if (DIV == 2) {
        lx = lx / p.dx2;
        ly = ly / p.dy2;
    }
    const double lap = lx + ly;
    return c + p.kdiff * lap;
}

// SX / SY: upwind direction known at compile time (1: v >= 0, 0: v < 0, -1: decided at run
// time).  The compute-bound multi-step kernels are instantiated per sign so that neither both
// differences nor a per-lane select are evaluated.
template <int DIV, int SX = -1, int SY = -1>
__device__ __forceinline__ double advect_term(double c, double W, double E, double S, double N,
                                              const Phys& p) {
    double gx, gy;
    if (SX == 1)
        gx = c - W;
    else if (SX == 0)
        gx = E - c;
    else
        gx = (p.vx >= 0.0) ? (c - W) : (E - c);
    if (SY == 1)
        gy = c - S;
    else if (SY == 0)
        gy = N - c;
    else
        gy = (p.vy >= 0.0) ? (c - S) : (N - c);
    if (DIV == 1) {
        gx = gx * p.rdx;
        gy = gy * p.rdy;
    } else if (DIV == 2) {
        gx = gx / p.dx;
        gy = gy / p.dy;
    }
    const double adv = p.vx * gx + p.vy * gy;
    return p.mdt * adv;
}

// DIV 3 — option "contract" (opt-in, NOT bit-identical): the same update written as the 5-point stencil
// it is, a0 c + aW W + aE E + aS S + aN N with host-made coefficients (make_phys), evaluated as one
// multiply and four FMAs instead of 15 non-FMA operations.  Differs from the reference's rounding by a
// few ulp per step (tests: L_inf < 1e-10 after 1000 steps, the north-star tolerance).
template <int DIV, int SX = -1, int SY = -1, bool FAST = false>
__device__ __forceinline__ double cell(double c, double W, double E, double S, double N,
                                       const Phys& p) {
    if (DIV == 3) {
        double o = p.a0 * c;
        o = __builtin_fma(p.aW, W, o);
        o = __builtin_fma(p.aE, E, o);
        o = __builtin_fma(p.aS, S, o);
        return __builtin_fma(p.aN, N, o);
    }
    const double o = diffuse_term<DIV, FAST>(c, W, E, S, N, p);
    // SX = 2 / SY = 2 — vx == 0 / vy == 0 (both: BASELINE configs[1], diffusion only; one: e.g. the reference's own
    // configs/dev.yaml, vy = 0).  The reference still evaluates o + (-dt) * (vx * dudx + vy * dudy).  With finite
    // differences a product with a zero velocity is +0 or -0; adding it to the other product changes nothing unless that
    // one is a zero too, and then only the SIGN of the zero sum; (-dt) times a zero is a zero; and o + (+-0) is o bit for
    // bit unless o is -0 — and c + k * lap can only be -0 where c itself is -0, level after level down to a LOADED -0.
    // So the screened interior body (FAST: every loaded value finite and below the threshold; here also: none of them -0)
    // leaves the operations of a zero component out (3 of 14 for one, all 7 for both); every other body of such an
    // instantiation evaluates them as v >= 0.
    if (FAST && (SX == 2 || SY == 2)) {
        if (SX == 2 && SY == 2) return o;
        double g;  // the one live component, as advect_term forms it
        if (SY == 2)
            g = SX == 1 ? c - W : E - c;
        else
            g = SY == 1 ? c - S : N - c;
        if (DIV == 1) g = g * (SY == 2 ? p.rdx : p.rdy);
        const double adv = (SY == 2 ? p.vx : p.vy) * g;
        return o + p.mdt * adv;
    }
    return o + advect_term<DIV, (SX == 2 ? 1 : SX), (SY == 2 ? 1 : SY)>(c, W, E, S, N, p);
}

// ---- cross-lane neighbour moves (DPP, no LDS traffic) ---------------------------------------
// wave_shr:1  lane i <- lane i-1, lane 0 keeps `edge`;  wave_shl:1  lane i <- lane i+1, lane 63
// keeps `edge` (bound_ctrl off: lanes without a source keep the old value).
__device__ __forceinline__ double from_prev_lane(double src, double edge) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(src), 0x138, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(src), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_next_lane(double src, double edge) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(src), 0x130, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(src), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Blocks b and b+8 share an XCD (round-robin dispatch), so give every XCD one contiguous run of
// tile ids: x-adjacent strips and y-adjacent chunks then hit the same 4 MiB L2.  Bijective for
// any grid size.  Placement only affects speed, never results.
__device__ __forceinline__ int xcd_remap(int b, int nb, int enable) {
    if (!enable || nb < 16) return b;
    const int per = nb >> 3, rem = nb & 7;
    const int xcd = b & 7, q = b >> 3;
    return xcd < rem ? xcd * (per + 1) + q : rem * (per + 1) + (xcd - rem) * per + q;
}

// write-through flavour (agent-scope relaxed atomic stores, `global_store ... sc1`): the values are in
// memory, visible to every XCD, once the wavefront's s_waitcnt vmcnt(0) returns — no L2 write-back
// (buffer_wbl2) needed.  Used by the frame tiles of a merged launch, whose outputs later kernels on
// another stream read while this kernel is still running.
__device__ __forceinline__ void store_pair_wt(double* dst, double ox, double oy, int nvalid) {
    if (nvalid >= 1)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), static_cast<unsigned long long>(__double_as_longlong(ox)),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nvalid >= 2)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst + 1), static_cast<unsigned long long>(__double_as_longlong(oy)),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Direct faces (merged launch): one cell (column i in 0..nx+1, row j in 0..ny+1; 0 and n+1 = ghost lines) of the
// field a frame tile has just written goes into every face of the NEXT pass it belongs to.  Indexing is
// k_halo2_pack's for depth H: column faces [c][j] over rows 0..ny+1, row faces [r][i] over columns 0..nx+1 (the
// ghost entries travel along: Periodic ghosts are never rewritten), corner blocks [r][c] of interior cells.
__device__ __forceinline__ void face_store_cell(const FrameSync& fs, int i, int j, double v, int nx, int ny) {
    const int H = fs.face_depth;
    const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(v));
    auto put = [&](double* face, int idx) {  // write-through, like the tile's own result stores
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(face + idx), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    const bool in_i = i >= 1 && i <= nx, in_j = j >= 1 && j <= ny;
    const bool l = in_i && i <= H, r = in_i && i >= nx - H + 1, b = in_j && j <= H, t = in_j && j >= ny - H + 1;
    const int cl = i - 1, cr = i - (nx - H + 1), rb = j - 1, rt = j - (ny - H + 1);
    if (l && fs.face[0]) put(fs.face[0], cl * (ny + 2) + j);
    if (r && fs.face[1]) put(fs.face[1], cr * (ny + 2) + j);
    if (b && fs.face[2]) put(fs.face[2], rb * (nx + 2) + i);
    if (t && fs.face[3]) put(fs.face[3], rt * (nx + 2) + i);
    if (l && b && fs.face[4]) put(fs.face[4], rb * H + cl);
    if (r && b && fs.face[5]) put(fs.face[5], rb * H + cr);
    if (l && t && fs.face[6]) put(fs.face[6], rt * H + cl);
    if (r && t && fs.face[7]) put(fs.face[7], rt * H + cr);
}

__device__ __forceinline__ void store_pair(double* dst, double ox, double oy, int nvalid) {
    if (nvalid >= 2) {
        *reinterpret_cast<double2*>(dst) = make_double2(ox, oy);
    } else if (nvalid == 1) {
        dst[0] = ox;
    }
}

// -------------------------------------------------------------------------------------------
// VAR_DPP — the default fused sweep.
// One wavefront owns a strip of 128 interior columns (2 per lane, one 16-byte load per lane
// per row = 8 full 128-byte lines per wave) and marches up `ry` rows keeping rows j-1, j, j+1
// of its own columns in registers, with PF further rows already in flight.  W/E neighbours
// come from the adjacent lanes by DPP; only lane 0 / lane 63 fetch the one column outside the
// strip (an L1/L2 hit: the neighbouring wave streams that line at the same time).  No LDS, no
// barriers, ~40 VGPRs -> 8 waves/SIMD.  HBM traffic per cell: 8 B read (+2/ry halo rows) + 8 B
// written.
// -------------------------------------------------------------------------------------------
template <int DIV, int PF>
__global__ __launch_bounds__(256) void k_sweep_dpp(const double* __restrict__ in,
                                                   double* __restrict__ out, int nx, int ny,
                                                   int pitch, int ry, int nwgx, int swz, Phys p) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int lin = xcd_remap(blockIdx.x, gridDim.x, swz);
    const int wgx = lin % nwgx, chunk = lin / nwgx;
    const int c0 = (wgx * 4 + wave) * WAVE_COLS;
    if (c0 >= nx) return;  // wave-uniform
    const int jb = chunk * ry + 1;
    const int je = min(jb + ry - 1, ny);
    const int col = c0 + 2 * lane;
    const int nvalid = nx - col;
    const size_t xoff = static_cast<size_t>(LPAD + col);
    const bool edge_lane = (lane == 0) || (lane == 63);
    const size_t eoff = static_cast<size_t>(LPAD + c0 + (lane == 0 ? -1 : WAVE_COLS));

    auto ld2 = [&](int j) {
        return *reinterpret_cast<const double2*>(in + static_cast<size_t>(j) * pitch + xoff);
    };
    auto lde = [&](int j) {
        double e = 0.0;
        if (edge_lane) e = in[static_cast<size_t>(j) * pitch + eoff];
        return e;
    };

    double2 S = ld2(jb - 1);
    double2 C = ld2(jb);
    double eC = lde(jb);
    double2 q[PF];
    double eq[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        q[u] = make_double2(0.0, 0.0);
        eq[u] = 0.0;
        const int r = jb + 1 + u;
        if (r <= je + 1) {
            q[u] = ld2(r);
            eq[u] = lde(r);
        }
    }
    for (int j = jb; j <= je; j += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int jj = j + u;
            if (jj <= je) {  // wave-uniform
                const double2 N = q[u];
                const double eN = eq[u];
                const int r = jj + 1 + PF;
                if (r <= je + 1) {
                    q[u] = ld2(r);
                    eq[u] = lde(r);
                }
                const double Wx = from_prev_lane(C.y, eC);
                const double Ey = from_next_lane(C.x, eC);
                const double ox = cell<DIV>(C.x, Wx, C.y, S.x, N.x, p);
                const double oy = cell<DIV>(C.y, C.x, Ey, S.y, N.y, p);
                store_pair(out + static_cast<size_t>(jj) * pitch + xoff, ox, oy, nvalid);
                S = C;
                C = N;
                eC = eN;
            }
        }
    }
}

// kind[s] of a fused pass: CSIM_BC_* on physical sides, 3 where the side has a neighbour rank
// (plain stencil on the stored deep halo)
struct Bc2 {
    int kind[4];  // per side: CSIM_BC_* or 3 (= not a physical edge)
    double value;
};

// -------------------------------------------------------------------------------------------
// VAR_OVERLAP — T time steps per pass with OVERLAPPED strips (the default multi-step kernel).
// A wavefront loads 128 consecutive columns (2 per lane, 16-byte aligned) but only the inner
// 128 - 2*TP of them (TP = T rounded up to even) are its outputs: level l is valid on local
// columns [l, 127 - l], so no extra-column bookkeeping is needed at all — the W/E neighbours are
// plain DPP lane shifts (the invalid outermost lanes simply compute don't-care values) and the
// strips overlap by 2*TP columns (6 % redundant work at T = 4) instead of paying one extra
// wave-wide cell update per level (50 %).  A row of a level is ONE double2 per lane, so the whole
// T-level pipeline fits in ~107 VGPRs at T = 6.
//   - level l+1 of row r needs level l of rows r-1..r+1: the march starts T-1 rows below the chunk
//     and ends T-1 rows above it (the device layout keeps GHOST_EXTRA extra ghost rows/columns);
//   - where a strip/chunk touches a PHYSICAL edge, the ghost value of an intermediate level is not a
//     stencil result but the boundary rule applied to that level (reference src/boundary.cpp:23-53
//     run at the start of the next step): Dirichlet -> value, Neumann -> adjacent interior of the
//     same level, Periodic (no-op) -> the stored ghost, unchanged.  kind 3 = the side
//     has a neighbour rank: plain stencil on the stored deep halo.
// Ghost COLUMNS are ordinary lanes here, patched by the boundary rule on wavefronts that contain a
// physical edge.  Any nx works.
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ double shift_from_prev(double src) {  // lane i <- lane i-1 (lane 0: 0)
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), 0x138, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shift_from_next(double src) {  // lane i <- lane i+1 (lane 63: 0)
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), 0x130, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

#ifdef CSIM_TRACE
// tools/wavetrace.hip only: start/end time (100 MHz wall clock) and placement of every wavefront
__device__ unsigned long long* g_wave_trace = nullptr;
struct WaveTrace {
    int slot, lane;
    unsigned long long t0;
    __device__ WaveTrace(int s, int l) : slot(s), lane(l), t0(wall_clock64()) {}
    __device__ ~WaveTrace() {
        if (lane == 0 && g_wave_trace) {
            g_wave_trace[3 * slot] = t0;
            g_wave_trace[3 * slot + 1] = wall_clock64();
            g_wave_trace[3 * slot + 2] = (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg(63508)) << 32) |
                                         static_cast<unsigned>(__builtin_amdgcn_s_getreg(63492));
        }
    }
};
#endif

// An empty volatile asm cannot be speculated, so the block it sits in stays a real (wave-uniform)
// branch instead of being if-converted into per-lane selects on the hot path.
__device__ __forceinline__ void keep_branch() { asm volatile(""); }

// Last pass of a run: the kernel also leaves, per side, the line of level T-1 (the state before
// the last step) that the reference's final halo exchange + apply_boundary would have read, so
// that the ghost ring of the result can be made exactly the reference's without a trailing
// single-step pass.  Physical side: the adjacent interior line (column 1 / nx, row 1 / ny);
// neighbour side: the ghost line itself (column 0 / nx+1, row 0 / ny+1), which this rank computes
// anyway from the deep faces — bitwise what the neighbour holds there.
struct FinLines {
    double* line[4];  // left/right: ny entries; bottom/top: nx entries; all nullptr = off
};

// Tiles of one launch: up to four rectangular regions of (strip, chunk) tiles, numbered
// consecutively; wavefront w of block b owns tile 4 b + w.  One region (all strips x all rows) is
// the whole-field launch; a multi-rank pass splits the field into the FRAME (bottom band, top
// band, left strip(s), right strip(s): thin tiles, finished early so that the faces can travel
// while the rest computes) and the BULK (everything else).
struct TileRegion {
    int t_end;          // tiles [t_end of the previous region, t_end)
    int strip0, nstrip; // strips strip0 .. strip0 + nstrip - 1
    int j0, j1, ry;     // rows j0 .. j1 in chunks of ry
};
struct Tiling {
    TileRegion r[8];
    int nregions, ntiles;
    // merged launch (frame + bulk in one grid): tiles [0, frame_tiles) are the frame, owned by blocks
    // [0, frame_blocks) in plain order so that they are dispatched first and spread over all XCDs; the
    // bulk tiles follow from tile 4 * frame_blocks on, XCD-remapped among themselves.  0 = not merged.
    int frame_tiles, frame_blocks;
    // TAIL region: the last tail_blocks blocks own, in plain order, the tiles of the last region(s) — the top
    // eighth of the (bulk of the) field cut into chunks of half the height, dispatched last, so that the
    // chip drains in half-height steps instead of idling behind the last full-height wavefronts
    // (17 468 wavefronts are 4.26 rounds of 4096 slots on 16384^2: the partial last round was 7 % of the
    // launch).  The main tiles before them fill their blocks exactly and are XCD-remapped.  0 = no tail.
    int tail_blocks;
};


// The by-value argument block of k_sweepO_dpp (behind the two field pointers, which stay direct __restrict__
// parameters).  Everything a wavefront needs BEFORE or DURING its march is read from the parameter as usual; what it
// needs only rarely or only AFTER the march — the FinLines pointers, the whole FrameSync — is read from the
// kernel-argument segment at the point of use (LateArgs), so that those ~40 scalars are not kept alive (and
// spilled to VGPR lanes: 101 SGPR spills, 524 v_readlane/v_writelane per edge group of six in round 2) across the
// loop that is the kernel.
struct SweepArgs {
    int nx, ny, pitch, nstrips, swz;
    Tiling tl;
    Phys p;
    Bc2 bc;
    FinLines fin;
    FrameSync fs;
};
constexpr int SWEEP_ARGS_KERNARG_OFFSET = 16;  // two pointers precede it; alignof(SweepArgs) == 8

struct LateArgs {
    typedef const SweepArgs __attribute__((address_space(4))) * Ptr;
    Ptr a;
    __device__ __forceinline__ static LateArgs get() {
        typedef const char __attribute__((address_space(4))) * Bytes;
        LateArgs l;
        l.a = (Ptr)((Bytes)__builtin_amdgcn_kernarg_segment_ptr() + SWEEP_ARGS_KERNARG_OFFSET);
        return l;
    }
    // an opaque copy of the pointer: the loads through it stay where they are written (the scalar data cache
    // serves them; the kernel-argument segment is a few hundred bytes)
    __device__ __forceinline__ Ptr here() const {
        Ptr q = a;
        asm volatile("" : "+s"(q));
        return q;
    }
    __device__ __forceinline__ double* fin_line(int side) const { return here()->fin.line[side]; }
};

template <int T>
struct OverlapGeom {
    static constexpr int TP = 2 * ((T + 1) / 2);       // T rounded up to even
    static constexpr int STRIDE = WAVE_COLS - 2 * TP;  // output columns per wavefront
};

// FAST (interior body only): the cell update with E - 2c, N - 2c fused (diffuse_term<., true>), bit-identical to
// the plain form as long as no 2c overflows.  Guard: every value the tile loads is compared with p.fast_thr =
// 2^1022 / g^MAX_FUSE, g = the host's bound on the growth of max|u| per step; below it no level of the pass can
// reach 2^1023.  The march returns true if any lane saw a value that is not below the threshold (NaN and Inf
// included) and the caller then repeats the tile with the plain form — same loads, same stores, the reference's
// own operations.  Cost: two compares per lane and loaded row against 2 T multiplies saved.
//
// The EDGE bodies (wavefronts whose strip or chunk touches a physical edge, and the frame tiles of a final pass) are
// the interior body plus PATCHES that put the boundary rule where a level's ghost cells come out.  In the GENERIC
// flavour (M_GENERIC below; the straight-line flavours 0..6 compile their one patch in) every patch sits
// in a wave-uniform branch and works on values made opaque INSIDE that branch (pin / pin2: an empty asm the value
// passes through), so that the compiler can neither hoist the patch's moves, lane shifts and selects out of the
// branch nor turn the branch into per-lane selects executed by every level-row — which is what it did to the
// plain `if` blocks of round 2: 667 v_cndmask + 312 DPP moves + 524 v_readlane/v_writelane (101 spilled scalars)
// per group of six at T = 7, 2.2 x the instructions of the interior body.  What remains in the steady state:
//   * ghost COLUMNS (first / last strips): per level-row and side one per-lane select (Dirichlet / Periodic: the
//     ghost keeps its value; 2 v_cndmask) or a lane shift + select (Neumann; 2 DPP + 2 v_cndmask);
//   * ghost ROWS (physical bottom / top) only come out in the first ~2T and the last ~2T iterations of a chunk that
//     reaches the edge: the row tests run only in groups of six that can contain one (`rows_here`), elsewhere a
//     level-row pays one scalar branch;
//   * the FinLines emission of a run's last pass: scalar tests at level T-1, with the four line pointers read from
//     the kernel-argument segment where they are used (LateArgs) instead of living in scalars across the march.
__device__ __forceinline__ void pin(double& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin2(double2& v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }

// MODE: which body this instantiation is — ONE march loop each, so that every flavour is register-allocated like the
// interior body (several bodies chained in one function grew the kernel from 124 to 179-194 VGPRs, i.e. from 4 to 2
// wavefronts per SIMD for EVERY tile):
//   M_FAST / M_PLAIN  interior body with / without the fused E - 2c (see FAST above)
//   M_GENERIC         edge body with every patch behind run-time tests: tiles that can produce ghost ROWS of a
//                     physical edge (the launcher keeps them thin: bottom / top bands) and strips that hold BOTH ghost
//                     columns; also every edge tile of the instantiations that are not specialised (SPECIALISE_EDGES)
//   0..6              edge body of a strip with at most ONE ghost column and no ghost rows, its patch compiled in,
//                     straight-line like the interior body: 0 none (a final pass's frame tile that only emits
//                     FinLines), 1 / 2 left ghost kept / Neumann, 3 / 4 right ghost kept in .x / .y, 5 / 6 right
//                     ghost Neumann in .x / .y
constexpr int M_FAST = -3, M_PLAIN = -2, M_GENERIC = -1;

template <int DIV, int T, int MODE, int SX, int SY>
__device__ __forceinline__ bool sweepO_march(const double* __restrict__ in, double* __restrict__ out,
                                             int nx, int ny, int pitch, int jb, int je, int g0, int lane,
                                             int kl, int kr, const Phys& p, int kb, int kt,
                                             LateArgs late, bool fin_any, bool fin_l, bool fin_r, bool wt) {
    constexpr int TP = OverlapGeom<T>::TP;
    constexpr int STRIDE = OverlapGeom<T>::STRIDE;
    constexpr bool EDGE = MODE >= M_GENERIC, FAST = MODE == M_FAST, GENERIC = MODE == M_GENERIC;
    constexpr int CASE = MODE;
    // this lane's two columns, 0-based interior index (-1 = left ghost, nx = right ghost)
    const int gx = g0 + 2 * lane, gy = gx + 1;
    const ptrdiff_t xoff = LPAD + gx;
    // kb / kt: kind of the bottom / top side, 3 = neighbour rank: plain stencil
    // output lanes: local columns [TP, TP + STRIDE), clipped to the interior
    const bool out_lane = 2 * lane >= TP && 2 * lane < TP + STRIDE && gx < nx;
    const int nvalid = nx - gx;
    // lanes that hold a ghost column of a physical edge (EDGE bodies only).  g0 is even, so the right ghost column
    // (index nx) is the .x of its lane when nx is even and the .y when nx is odd: wave-uniform
    const bool ghost_ly = kl != 3 && gy == -1;
    const bool ghost_rx = kr != 3 && gx == nx, ghost_ry = kr != 3 && gy == nx;
    const bool right_in_x = (nx & 1) == 0;
    const bool ghost_cols = kl != 3 || kr != 3;  // wave-uniform

    auto load = [&](int j) {
        return *reinterpret_cast<const double2*>(in + static_cast<ptrdiff_t>(j) * pitch + xoff);
    };

    const int r_first = jb - (T - 1);
    const int niter = (je - jb + 1) + 2 * (T - 1);
    const int last_row = r_first + niter;
    // Every load below is unconditional (row index clamped to the last row the chunk needs) and
    // the march runs whole groups of six iterations without a per-iteration exit test: a memory
    // operation that may or may not have been issued makes the compiler wait for ALL of them
    // (s_waitcnt vmcnt(0)) at the top of every iteration, which would serialise the row prefetch.
    // The <= 5 surplus iterations of a chunk whose niter is not a multiple of six compute rows
    // beyond je that are never stored (the host picks ry so that only a ragged last chunk has any).
    double2 L0[6];
    double2 L[T][3];
#pragma unroll
    for (int q = 0; q < 6; ++q) L0[q] = load(min(r_first - 1 + q, last_row));
    bool big = false;  // FAST: some loaded value is not below the threshold
    auto screen = [&](const double2& v) {
        big |= !(__builtin_fabs(v.x) < p.fast_thr);
        big |= !(__builtin_fabs(v.y) < p.fast_thr);
        if (SX == 2 || SY == 2) {  // flavours without (part of) the advection term (see cell): a loaded -0 sends the tile to the plain body too
            big |= __builtin_amdgcn_class(v.x, 0x20);
            big |= __builtin_amdgcn_class(v.y, 0x20);
        }
    };
    if (FAST) {  // every later row is screened when it is the `n` of level 1
        screen(L0[0]);
        screen(L0[1]);
    }
#pragma unroll
    for (int l = 0; l < T; ++l)
#pragma unroll
        for (int q = 0; q < 3; ++q) L[l][q] = make_double2(0.0, 0.0);

    // One group = six iterations.  Level l has nothing valid to produce before iteration 2 (l - 1)
    // (its first needed row, jb - (T - l), comes out exactly then), so the first two groups are
    // separate copies of the body in which the not-yet-started levels are left out at compile time:
    // 30 of the 6 (ry + 10) level-rows of a chunk at T = 6.  (The edge body keeps the single generic
    // copy: it is rare and its code is three times the size.)
    auto group = [&](auto gtag, int k0) {
        constexpr int G = decltype(gtag)::value;
        // GENERIC: can this group of six contain a ghost row of a physical edge?  Level l = 1..T-1 produces ghost row 0
        // at iteration l + T - 2 - jb (patched from row 1 one iteration later) and ghost row ny+1 at iteration
        // ny + l + T - 1 - jb: only in groups with  k0 <= 2T - 2 - jb  (bottom)  or  k0 + 5 >= ny + T - jb  (top)
        const bool rows_here = GENERIC && ((kb != 3 && k0 <= 2 * T - 2 - jb) || (kt != 3 && k0 + 5 >= ny + T - jb));
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            {
                const int r = r_first + k0 + u;
#pragma unroll
                for (int l = 1; l <= T; ++l) {
                    if (G < 2 && 6 * G + u < 2 * (l - 1)) continue;  // compile-time: level not started yet
                    const int rho = r - l + 1;
                    const double2 s = (l == 1) ? L0[u % 6] : L[l - 1][(u + 1) % 3];
                    const double2 c = (l == 1) ? L0[(u + 1) % 6] : L[l - 1][(u + 2) % 3];
                    const double2 n = (l == 1) ? L0[(u + 2) % 6] : L[l - 1][u % 3];
                    // The stencil is evaluated on every lane and row (branch-free, the same code as
                    // the interior body); where the result is a ghost cell of a physical edge it is
                    // then replaced by the boundary rule.
                    double2 o;
                    {
                        const double Wx = shift_from_prev(c.y);
                        const double Ey = shift_from_next(c.x);
                        o.x = cell<DIV, SX, SY, FAST>(c.x, Wx, c.y, s.x, n.x, p);
                        o.y = cell<DIV, SX, SY, FAST>(c.y, c.x, Ey, s.y, n.y, p);
                        if (FAST && l == 1) screen(n);
                    }
                    if (EDGE && !GENERIC && l < T) {  // the strip's one ghost column, unconditionally
                        if (CASE == 1) o.y = ghost_ly ? c.y : o.y;
                        if (CASE == 2) {
                            const double nb = shift_from_next(o.x);
                            o.y = ghost_ly ? nb : o.y;
                        }
                        if (CASE == 3) o.x = ghost_rx ? c.x : o.x;
                        if (CASE == 4) o.y = ghost_ry ? c.y : o.y;
                        if (CASE == 5) {
                            const double pb = shift_from_prev(o.y);
                            o.x = ghost_rx ? pb : o.x;
                        }
                        if (CASE == 6) o.y = ghost_ry ? o.x : o.y;
                    }
                    if (GENERIC && l < T) {
                        bool ghost_row = false;
                        if (rows_here) {
                            const bool gb = rho == 0 && kb != 3, gt = rho == ny + 1 && kt != 3;
                            ghost_row = gb || gt;
                            if (ghost_row) {  // ghost ROW of this level
                                const int kk = gb ? kb : kt;
                                if (kk != CSIM_BC_NEUMANN) {  // Dirichlet / Periodic ghosts keep their level-0 value (bc.value / stored)
                                    double2 t = c;
                                    pin2(t);
                                    o = t;
                                } else if (gt) {  // Neumann top: row ny of this level
                                    double2 t = L[l][(u + 2) % 3];
                                    pin2(t);
                                    o = t;
                                }
                                // (Neumann bottom: patched below as soon as row 1 of this level exists)
                            }
                        }
                        if (!ghost_row && ghost_cols) {  // ghost COLUMNS of this level (first / last strip)
                            if (kl == CSIM_BC_NEUMANN) {  // left ghost (.y of its lane) := column 0 (.x of the next lane)
                                double t = o.x;
                                pin(t);
                                const double nb = shift_from_next(t);
                                o.y = ghost_ly ? nb : o.y;
                            } else if (kl != 3) {  // Dirichlet / Periodic: unchanged through the levels
                                double t = c.y;
                                pin(t);
                                o.y = ghost_ly ? t : o.y;
                            }
                            if (kr == CSIM_BC_NEUMANN) {  // right ghost := column nx-1 (.y of the previous lane, or the lane's own .x)
                                if (right_in_x) {
                                    double t = o.y;
                                    pin(t);
                                    const double pb = shift_from_prev(t);
                                    o.x = ghost_rx ? pb : o.x;
                                } else {
                                    double t = o.x;
                                    pin(t);
                                    o.y = ghost_ry ? t : o.y;
                                }
                            } else if (kr != 3) {
                                if (right_in_x) {
                                    double t = c.x;
                                    pin(t);
                                    o.x = ghost_rx ? t : o.x;
                                } else {
                                    double t = c.y;
                                    pin(t);
                                    o.y = ghost_ry ? t : o.y;
                                }
                            }
                        }
                    }
                    if (EDGE && T >= 2 && l == T - 1 && fin_any) {  // see FinLines
                        keep_branch();
                        if (rho >= jb && rho <= je) {
                            if (fin_l && (kl != 3 ? gx == 0 : gy == -1)) late.fin_line(CSIM_LEFT)[rho - 1] = kl != 3 ? o.x : o.y;
                            if (fin_r) {
                                const int col = kr != 3 ? nx - 1 : nx;
                                if (gx == col) late.fin_line(CSIM_RIGHT)[rho - 1] = o.x;
                                if (gy == col) late.fin_line(CSIM_RIGHT)[rho - 1] = o.y;
                            }
                        }
                        if (jb == 1 && rho == (kb != 3 ? 1 : 0) && out_lane) store_pair(late.fin_line(CSIM_BOTTOM) + gx, o.x, o.y, nvalid);
                        if (je == ny && rho == (kt != 3 ? ny : ny + 1) && out_lane) store_pair(late.fin_line(CSIM_TOP) + gx, o.x, o.y, nvalid);
                    }
                    if (l < T) {
                        if (GENERIC && rows_here && rho == 1 && kb == CSIM_BC_NEUMANN) {  // ghost row 0 := row 1
                            double2 t = o;
                            pin2(t);
                            L[l][(u + 2) % 3] = t;
                        }
                        L[l][u % 3] = o;
                    } else if (rho >= jb && rho <= je && out_lane) {
                        if (wt) {  // wave-uniform: frame tile of a merged launch
                            keep_branch();
                            store_pair_wt(out + static_cast<ptrdiff_t>(rho) * pitch + xoff, o.x, o.y, nvalid);
                        } else {
                            store_pair(out + static_cast<ptrdiff_t>(rho) * pitch + xoff, o.x, o.y, nvalid);
                        }
                    }
                }
                L0[u % 6] = load(min(r + 5, last_row));  // row r-1 is dead: its slot takes row r+5
            }
        }
    };
    using G2 = std::integral_constant<int, 2>;
    if (EDGE) {
        for (int k0 = 0; k0 < niter; k0 += 6) group(G2{}, k0);
    } else {
        group(std::integral_constant<int, 0>{}, 0);
        if (niter > 6) group(std::integral_constant<int, 1>{}, 6);
        for (int k0 = 12; k0 < niter; k0 += 6) group(G2{}, k0);
    }
    return FAST && __builtin_amdgcn_ballot_w64(big) != 0;
}

// Which instantiations get the straight-line edge flavours (seven more march bodies, ~13 KB of code each): the
// arithmetic modes and depths that long runs are made of.  The others (IEEE division, contracted arithmetic, the
// shallow depths of remainder passes) run every edge tile through the generic body, as round 2 did.
template <int DIV, int T>
struct SPECIALISE_EDGES {
    static constexpr bool value = (DIV == 0 || DIV == 1) && T >= 4;
};

template <int DIV, int T, int SX, int SY>
__global__ __launch_bounds__(256) void k_sweepO_dpp(const double* __restrict__ in, double* __restrict__ out, SweepArgs a) {
    constexpr int TP = OverlapGeom<T>::TP;
    constexpr int STRIDE = OverlapGeom<T>::STRIDE;
    const int nx = a.nx, ny = a.ny, pitch = a.pitch, nstrips = a.nstrips;
    const LateArgs late = LateArgs::get();
    const int lane = threadIdx.x & 63;
    // readfirstlane: tells the compiler the wave index (and the strip, edge kinds and row range
    // derived from it) is wave-uniform, so those tests become scalar branches
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef CSIM_TRACE
    WaveTrace trace_scope(blockIdx.x * 4 + wave, lane);
#endif
    // blocks [0, frame_blocks): the frame tiles in plain order; the last tail_blocks blocks: the tail tiles in
    // plain order; the blocks in between own the main tiles, XCD-remapped among themselves
    int tile;
    bool frame_tile = false;
    {
        const int b = blockIdx.x;
        if (b < a.tl.frame_blocks) {
            tile = 4 * b + wave;
            if (tile >= a.tl.frame_tiles) return;  // padding of the last frame block
            frame_tile = true;
        } else {
            const int lb = b - a.tl.frame_blocks, nb_mid = gridDim.x - a.tl.frame_blocks - a.tl.tail_blocks;
            tile = a.tl.frame_tiles + (lb < nb_mid ? xcd_remap(lb, nb_mid, a.swz) : lb) * 4 + wave;
        }
    }
    if (tile >= a.tl.ntiles) return;  // wave-uniform
    int t0 = 0, strip0 = a.tl.r[0].strip0, nstrip = a.tl.r[0].nstrip, j0 = a.tl.r[0].j0, j1 = a.tl.r[0].j1, ry = a.tl.r[0].ry;
#pragma unroll
    for (int q = 1; q < 8; ++q)
        if (q < a.tl.nregions && tile >= a.tl.r[q - 1].t_end) {
            t0 = a.tl.r[q - 1].t_end;
            strip0 = a.tl.r[q].strip0, nstrip = a.tl.r[q].nstrip, j0 = a.tl.r[q].j0, j1 = a.tl.r[q].j1, ry = a.tl.r[q].ry;
        }
    const int local = tile - t0;
    const int strip = strip0 + local % nstrip;
    const int chunk = local / nstrip;
    const bool first = strip == 0, last = strip == nstrips - 1;
    const int jb = j0 + chunk * ry;
    const int je = min(jb + ry - 1, j1);
    const int g0 = strip * STRIDE - TP;
    // a strip meets the left ghost column iff it is the first one; the right ghost column (index
    // nx) lies inside every strip whose 128 loaded columns reach it
    const int kl = first ? a.bc.kind[CSIM_LEFT] : 3;
    const int kr = g0 + WAVE_COLS > nx ? a.bc.kind[CSIM_RIGHT] : 3;
    const int kb = a.bc.kind[CSIM_BOTTOM], kt = a.bc.kind[CSIM_TOP];
    // on the last pass of a run the frame tiles also take the edge body: they emit the FinLines
    const bool fin_frame = a.fin.line[CSIM_BOTTOM] != nullptr && (first || last || jb == 1 || je == ny);
    const bool edge = kl != 3 || kr != 3 || (kb != 3 && jb - (T - 1) < 1) || (kt != 3 && je + (T - 1) > ny) || fin_frame;
    if (frame_tile && a.fs.prio) __builtin_amdgcn_s_setprio(3);  // the faces wait for these: issue ahead of the co-resident bulk
    const bool signalling = frame_tile && a.fs.flag != nullptr;  // merged launch: this wavefront counts itself below
    const bool wt = signalling && a.fs.fence == 0;
    if (edge) {
        // Edge wavefronts run a little longer than interior ones and would finish last, leaving the rest of the chip
        // idle (29 % of a 4096 x 8192 launch with round 1's edge body, tools/wavetrace.hip): give them issue priority.
        __builtin_amdgcn_s_setprio(3);
        const bool rows = (kb != 3 && jb - (T - 1) < 1) || (kt != 3 && je + (T - 1) > ny);
        int col_case = 0;
        if (kl != 3 && kr != 3)
            col_case = 7;
        else if (kl != 3)
            col_case = kl == CSIM_BC_NEUMANN ? 2 : 1;
        else if (kr != 3)
            col_case = (kr == CSIM_BC_NEUMANN ? 5 : 3) + (nx & 1);  // g0 is even: the right ghost column is a .x iff nx is even
        if (!SPECIALISE_EDGES<DIV, T>::value || rows || col_case == 7) col_case = -1;
#define CSIM_MARCH(MODE_) \
    sweepO_march<DIV, T, MODE_, SX, SY>(in, out, nx, ny, pitch, jb, je, g0, lane, kl, kr, a.p, kb, kt, late, fin_frame, first, last, wt)
        switch (col_case) {
            case 0: if (SPECIALISE_EDGES<DIV, T>::value) CSIM_MARCH(0); break;
            case 1: if (SPECIALISE_EDGES<DIV, T>::value) CSIM_MARCH(1); break;
            case 2: if (SPECIALISE_EDGES<DIV, T>::value) CSIM_MARCH(2); break;
            case 3: if (SPECIALISE_EDGES<DIV, T>::value) CSIM_MARCH(3); break;
            case 4: if (SPECIALISE_EDGES<DIV, T>::value) CSIM_MARCH(4); break;
            case 5: if (SPECIALISE_EDGES<DIV, T>::value) CSIM_MARCH(5); break;
            case 6: if (SPECIALISE_EDGES<DIV, T>::value) CSIM_MARCH(6); break;
            default: CSIM_MARCH(M_GENERIC); break;
        }
    } else {
        bool redo = true;
        if (DIV != 3 && a.p.fast_thr > 0.0) redo = CSIM_MARCH(M_FAST);
        if (redo) {
            keep_branch();
            CSIM_MARCH(M_PLAIN);
        }
    }
#undef CSIM_MARCH
    if (signalling) {
        // Merged launch: the comm stream is parked on `flag` (hipStreamWaitValue64) and goes on to pack and
        // send the next pass's faces as soon as EVERY frame tile is in memory — while this very kernel is
        // still sweeping the bulk.  The consumers are later kernels on another stream and may run on any XCD,
        // so a frame tile's outputs must be in memory, not in this XCD's write-back L2, before it counts
        // itself: its result stores are write-through (store_pair_wt) and only have to be drained here.  (An
        // agent-scope release fence, i.e. buffer_wbl2 per wavefront, also works but writes back the dirty
        // output lines of the whole bulk each time: measured +40 us per 165 us pass; kept as fence = 1.)  The
        // wavefront that completes the count re-arms the counter and publishes the pass number (system scope:
        // the waiting side reads it through the command processor).
        // Everything from here on is read from the kernel-argument segment NOW (LateArgs): nothing of FrameSync was
        // alive during the march.
        const LateArgs::Ptr ka = late.here();
        FrameSync fs;
        fs.counter = ka->fs.counter, fs.flag = ka->fs.flag, fs.pass = ka->fs.pass, fs.nframe = ka->fs.nframe;
        fs.fence = ka->fs.fence, fs.face_depth = ka->fs.face_depth;
#pragma unroll
        for (int d = 0; d < 8; ++d) fs.face[d] = ka->fs.face[d];
        if (fs.fence == 0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (fs.fence == 0 && fs.face_depth > 0) {
            // Direct faces: this wavefront copies the part of its tile that belongs to a face of the NEXT pass
            // (and the ghost entries beside it) into the send buffers, so the comm stream can post the RCCL
            // group at the flag without a pack kernel in between.  Done after the march by reading the tile
            // back (its stores are drained and written through; the loads bypass the vector L1): the same
            // stores inside the march loop cost 20 more VGPRs, i.e. one wavefront per SIMD.
            const int H = fs.face_depth;
            const int gx = g0 + 2 * lane;
            const bool out_lane = 2 * lane >= TP && 2 * lane < TP + STRIDE && gx < nx;
            const bool rows_near = jb <= H || je >= ny - H + 1;              // wave-uniform
            const bool cols_near = g0 + TP < H || g0 + TP + STRIDE > nx - H;  // wave-uniform
            if (rows_near || cols_near) {
                const int jf0 = jb == 1 ? 0 : jb, jf1 = je == ny ? ny + 1 : je;
                auto ldf = [&](const double* q) {
                    return __longlong_as_double(static_cast<long long>(__hip_atomic_load(
                        reinterpret_cast<const unsigned long long*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
                };
                for (int rho = jf0; rho <= jf1; ++rho) {
                    if (!out_lane) continue;
                    const double* src = out + static_cast<ptrdiff_t>(rho) * pitch + LPAD + gx;  // cell (gx + 1, rho)
                    face_store_cell(fs, gx + 1, rho, ldf(src), nx, ny);
                    if (gx + 1 < nx) face_store_cell(fs, gx + 2, rho, ldf(src + 1), nx, ny);
                    if (gx == 0) face_store_cell(fs, 0, rho, ldf(src - 1), nx, ny);
                    if (gx + 1 == nx) face_store_cell(fs, nx + 1, rho, ldf(src + 1), nx, ny);
                    if (gx + 2 == nx) face_store_cell(fs, nx + 1, rho, ldf(src + 2), nx, ny);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        if (fs.fence == 1) __threadfence();  // plain result stores + agent-scope fence per wavefront (measured alternative)
        // Ordering (ISA level; no C++ happens-before is claimed): every store of this wavefront that a consumer may
        // read — tile results and face copies — is an agent-scope write-through store (global_store ... sc1), and the
        // s_waitcnt vmcnt(0) above returns only once each of them has been acknowledged by memory.  The counter
        // increment below is therefore issued after the data is globally visible; it can be relaxed, because the only
        // thing ordered after it is the flag store of the LAST arriver, and that wavefront's own data was drained by
        // its own s_waitcnt before its own increment — the increments of the others precede it in the counter's
        // modification order, each issued after that wavefront's drain.  The flag itself is a system-scope release
        // store; the waiting side is the command processor (hipStreamWaitValue64), and every kernel launched behind the
        // wait begins with the usual acquire (L2 invalidate / write-back state of a kernel boundary).
        if (lane == 0) {
            const unsigned done = atomicAdd(fs.counter, 1u);
            if (done == fs.nframe - 1) {
                atomicExch(fs.counter, 0u);
                __hip_atomic_store(fs.flag, fs.pass, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ============================================================================================
// launcher of the overlapped-strip sweep (templates; see the split-build note below)
// ============================================================================================
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

template <int DIV, int T>
hipError_t sweepO_div(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                             const SweepCfg& cfg, const Bc2& bc, const FinLines& fin, int part, hipStream_t st,
                             FrameSync fs) {
    constexpr int STRIDE = OverlapGeom<T>::STRIDE;
    const int nstrips = cdiv(nx, STRIDE);
    int ry = cfg.rows_per_chunk;
    if (ry <= 0) {
        if (cfg.tuned_rows > 0) {
            ry = cfg.tuned_rows;
        } else {
            ry = 64;
            while (ry > 16 && static_cast<long>(nstrips) * cdiv(ny, ry) < 8192) ry >>= 1;
            // tiles too small for the on-device trial (< 4 M cells): a launch is at most a round or two of
            // wavefronts and the length of a wavefront's march decides — the shortest chunks win although
            // they double the overhead rows (512^2: +52 %, 1024^2: +37 %, 2048^2: +25 % against 18 rows)
            if (static_cast<long>(nx) * ny < (1L << 22)) ry = 6;
        }
        // the march runs whole groups of six iterations: make ry + 2 (T - 1) a multiple of six so
        // that only a ragged last chunk computes surplus rows
        ry += (6 - (ry + 2 * (T - 1)) % 6) % 6;
    }
    if (ry > ny) ry = ny;
    if (cfg.rows_used && part != 1) *cfg.rows_used = ry;
    // part 0: the whole field.  part 1 / 2 (multi-rank pass): FRAME / BULK.  The frame is the
    // bottom and top bands (hf rows, all strips) plus the first strip and the last one or two
    // strips (>= MAX_FUSE columns) over the rows in between, in chunks of hf rows: thin tiles,
    // one short round of wavefronts, so the faces are ready ~15 us into the pass.
    int hf = 12;  // >= the deepest face (8-row bands were measured slower: more, even thinner tiles)
    hf += (6 - (hf + 2 * (T - 1)) % 6) % 6;
    const int nright = (nx - (nstrips - 1) * STRIDE >= MAX_FUSE) ? 1 : 2;
    const bool split = ny >= 2 * hf + 1 && nstrips >= nright + 2;
    // A band along a PHYSICAL bottom / top edge runs the generic edge body (ghost rows), about twice as slow per
    // iteration as the other frame tiles, and the frame launch lasts as long as its slowest tile (47 us instead of 34 on
    // a 4096 x 8192 tile with one physical side): such a band is only as high as the ghost rows require (T-1 rows,
    // rounded so that its march is whole groups of six: 18 iterations at T = 7 instead of 24).  The tiles above it then
    // start at row T and read the ghost row itself as level-0 input — in a bulk-first pass BEFORE this pass's ghost
    // fill has run: fine for Dirichlet and Periodic sides, whose ghost ring never changes, not for Neumann ones, which
    // keep the band of hf >= T rows.
    int hphys = T - 1;
    hphys += (6 - (hphys + 2 * (T - 1)) % 6) % 6;
    auto thin = [&](int side) {
        return SPECIALISE_EDGES<DIV, T>::value && bc.kind[side] != 3 && bc.kind[side] != CSIM_BC_NEUMANN;
    };
    const int hfb = thin(CSIM_BOTTOM) ? std::min(hf, hphys) : hf;
    const int hft = thin(CSIM_TOP) ? std::min(hf, hphys) : hf;
    Tiling tl{};
    auto add = [&](int strip0, int nstrip, int j0, int j1, int rows) {
        if (nstrip <= 0 || j1 < j0) return;
        TileRegion& r = tl.r[tl.nregions++];
        r.strip0 = strip0, r.nstrip = nstrip, r.j0 = j0, r.j1 = j1, r.ry = rows;
        tl.ntiles += nstrip * cdiv(j1 - j0 + 1, rows);
        r.t_end = tl.ntiles;
    };
    // rows j0..j1 of `nstrip` strips: full-height chunks, or — on launches of two or more rounds of wavefronts —
    // a main region of 7/8 of the chunks (a multiple of four, so that its tiles fill whole blocks whatever the
    // number of strips) followed by a tail region at half the height; returns the tail tiles
    auto add_rows = [&](int strip0, int nstrip, int j0, int j1, int rows) -> int {
        const int nrows = j1 - j0 + 1;
        if (nstrip <= 0 || nrows <= 0) return 0;
        const int nchunks = cdiv(nrows, rows);
        if (!cfg.tail_split || rows < 48 || nchunks < 16 || static_cast<long>(nstrip) * nchunks < 8192) {
            add(strip0, nstrip, j0, j1, rows);
            return 0;
        }
        auto snap = [&](int r) { return r + (6 - (r + 2 * (T - 1)) % 6) % 6; };
        const bool two_level = cfg.tail_split != 2;  // default: 7/8 of the chunks full height + the rest at half height;
                                                     // 2 (experiment): 3/4 + half + quarter height — measured no better
        const int main_chunks = (nchunks * (two_level ? 7 : 3) / (two_level ? 8 : 4)) / 4 * 4;
        const int j_main = j0 + main_chunks * rows - 1;
        const int half = snap(rows / 2), quarter = snap(rows / 4);
        const int rest = j1 - j_main;                       // rows left for the tail regions
        const int j_half = two_level ? j1 : j_main + (rest * 2 / 3) / half * half;  // about two thirds of them at half height
        add(strip0, nstrip, j0, j_main, rows);
        const int before = tl.ntiles;
        add(strip0, nstrip, j_main + 1, j_half, half);
        add(strip0, nstrip, j_half + 1, j1, quarter);
        return tl.ntiles - before;
    };
    int tail_tiles = 0;
    if (part == 0 || ((part == 1 || part == 3) && !split)) {
        // Physical bottom / top edges: the rows whose chunks can produce ghost ROWS of the intermediate levels (the
        // first and last T-1) go into thin bands of their own, so that only those few short tiles run the generic
        // edge body and every other tile of the first / last strips a straight-line column flavour (sweepO_march).
        // The bands come last in the tile order, with the tail region: they are the shortest tiles of the launch.
        int hb = T - 1;
        hb += (6 - (hb + 2 * (T - 1)) % 6) % 6;
        const bool bands = SPECIALISE_EDGES<DIV, T>::value && ny >= 2 * hb + 6;
        const bool band_b = bands && bc.kind[CSIM_BOTTOM] != 3, band_t = bands && bc.kind[CSIM_TOP] != 3;
        tail_tiles = add_rows(0, nstrips, band_b ? hb + 1 : 1, band_t ? ny - hb : ny, ry);
        const int before = tl.ntiles;
        if (band_b) add(0, nstrips, 1, hb, hb);
        if (band_t) add(0, nstrips, ny - hb + 1, ny, hb);
        tail_tiles += tl.ntiles - before;
    } else if (part == 1 || part == 3) {
        add(0, nstrips, 1, hfb, hfb);
        add(0, nstrips, ny - hft + 1, ny, hft);
        int hs = hf;  // side strips: taller chunks waste fewer warm-up rows (2 (T - 1) per chunk) but finish later
        if (cfg.frame_rows >= MAX_FUSE) hs = cfg.frame_rows + (6 - (cfg.frame_rows + 2 * (T - 1)) % 6) % 6;
        add(0, 1, hfb + 1, ny - hft, hs);
        add(nstrips - nright, nright, hfb + 1, ny - hft, hs);
    }
    int nblocks;
    if (part == 3 && split) {  // merged launch: the frame tiles above, then the bulk in the same grid
        tl.frame_tiles = tl.ntiles;
        tl.frame_blocks = cdiv(tl.ntiles, 4);
        fs.nframe = static_cast<unsigned>(tl.frame_tiles);
        const int before = tl.ntiles;
        tail_tiles = add_rows(1, nstrips - 1 - nright, hfb + 1, ny - hft, std::min(ry, ny - hfb - hft));
        nblocks = tl.frame_blocks + cdiv(tl.ntiles - before, 4);
    } else if (part == 3) {  // a tile that is all frame: every tile counts for the flag
        tl.frame_tiles = tl.ntiles;
        tl.frame_blocks = cdiv(tl.ntiles, 4);
        tl.tail_blocks = 0;
        tail_tiles = 0;
        fs.nframe = static_cast<unsigned>(tl.frame_tiles);
        nblocks = tl.frame_blocks;
    } else {
        if (part == 2 && split) tail_tiles = add_rows(1, nstrips - 1 - nright, hfb + 1, ny - hft, std::min(ry, ny - hfb - hft));
        if (tl.ntiles == 0) return hipSuccess;  // part 2 of a field that is all frame
        nblocks = cdiv(tl.ntiles, 4);
        fs = FrameSync{};
    }
    tl.tail_blocks = cdiv(tail_tiles, 4);
    const dim3 grid(nblocks), block(256);
    const int sw = cfg.xcd_swizzle;
    SweepArgs ka;
    ka.nx = nx, ka.ny = ny, ka.pitch = pitch, ka.nstrips = nstrips, ka.swz = sw;
    ka.tl = tl, ka.p = p, ka.bc = bc, ka.fin = fin, ka.fs = fs;
#define CSIM_LAUNCH_O(SXV, SYV) \
    hipLaunchKernelGGL((k_sweepO_dpp<DIV, T, SXV, SYV>), grid, block, cfg.lds_bytes, st, in, out, ka)
#ifdef CSIM_ISA_PROBE
    CSIM_LAUNCH_O(1, 1);
#else
    if (DIV == 3) {  // coefficient form: the upwind directions are folded into the coefficients
        CSIM_LAUNCH_O(1, 1);
    } else {
        // zero velocity components (DIV 0 / 1; the IEEE-division form keeps its four sign flavours): code 2 per axis
        const bool screened = DIV <= 1 && p.fast_thr > 0.0;
        const int cx = screened && p.vx == 0.0 ? 2 : (p.vx >= 0.0 ? 1 : 0), cy = screened && p.vy == 0.0 ? 2 : (p.vy >= 0.0 ? 1 : 0);
        switch (3 * cx + cy) {
            case 8: if constexpr (DIV <= 1) CSIM_LAUNCH_O(2, 2); break;
            case 7: if constexpr (DIV <= 1) CSIM_LAUNCH_O(2, 1); break;
            case 6: if constexpr (DIV <= 1) CSIM_LAUNCH_O(2, 0); break;
            case 5: if constexpr (DIV <= 1) CSIM_LAUNCH_O(1, 2); break;
            case 2: if constexpr (DIV <= 1) CSIM_LAUNCH_O(0, 2); break;
            case 4: CSIM_LAUNCH_O(1, 1); break;
            case 3: CSIM_LAUNCH_O(1, 0); break;
            case 1: CSIM_LAUNCH_O(0, 1); break;
            default: CSIM_LAUNCH_O(0, 0); break;
        }
    }
#endif
#undef CSIM_LAUNCH_O
    return hipGetLastError();
}

template <int T>
hipError_t sweepO_T(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                           const SweepCfg& cfg, const Bc2& bc, const FinLines& fin, int part, hipStream_t st,
                           const FrameSync& fs) {
#ifdef CSIM_ISA_PROBE  // tools: only the instantiation bench.py runs (dx = dy = 1, vx, vy >= 0), for a readable listing
    return sweepO_div<0, T>(in, out, nx, ny, pitch, p, cfg, bc, fin, part, st, fs);
#else
    switch (p.div_mode) {
        case 0: return sweepO_div<0, T>(in, out, nx, ny, pitch, p, cfg, bc, fin, part, st, fs);
        case 1: return sweepO_div<1, T>(in, out, nx, ny, pitch, p, cfg, bc, fin, part, st, fs);
        case 3: return sweepO_div<3, T>(in, out, nx, ny, pitch, p, cfg, bc, fin, part, st, fs);
        default: return sweepO_div<2, T>(in, out, nx, ny, pitch, p, cfg, bc, fin, part, st, fs);
    }
#endif
}


// Split build (csrc/Makefile, -DCSIM_SPLIT_BUILD): the six instantiations sweepO_T<2..7> — 60 kernels with two
// bodies each, 90 % of this file's compile time — are built as six translation units in parallel
// (-DCSIM_INST_T=N: this file up to here plus one explicit instantiation) next to the main one, which only
// declares them.  Without the macros (tools that include this file) everything is instantiated here.
#ifdef CSIM_INST_T
template hipError_t sweepO_T<CSIM_INST_T>(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                                    const SweepCfg& cfg, const Bc2& bc, const FinLines& fin, int part, hipStream_t st,
                                    const FrameSync& fs);
#elif defined(CSIM_SPLIT_BUILD)
extern template hipError_t sweepO_T<2>(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                                    const SweepCfg& cfg, const Bc2& bc, const FinLines& fin, int part, hipStream_t st,
                                    const FrameSync& fs);
extern template hipError_t sweepO_T<3>(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                                    const SweepCfg& cfg, const Bc2& bc, const FinLines& fin, int part, hipStream_t st,
                                    const FrameSync& fs);
extern template hipError_t sweepO_T<4>(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                                    const SweepCfg& cfg, const Bc2& bc, const FinLines& fin, int part, hipStream_t st,
                                    const FrameSync& fs);
extern template hipError_t sweepO_T<5>(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                                    const SweepCfg& cfg, const Bc2& bc, const FinLines& fin, int part, hipStream_t st,
                                    const FrameSync& fs);
extern template hipError_t sweepO_T<6>(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                                    const SweepCfg& cfg, const Bc2& bc, const FinLines& fin, int part, hipStream_t st,
                                    const FrameSync& fs);
extern template hipError_t sweepO_T<7>(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                                    const SweepCfg& cfg, const Bc2& bc, const FinLines& fin, int part, hipStream_t st,
                                    const FrameSync& fs);
#endif

#ifndef CSIM_INST_T  // ---- everything below lives in the main translation unit only ----------------------

// -------------------------------------------------------------------------------------------
// VAR_LDS — LDS-staged marching sweep.  A 256-thread workgroup owns a 512-column strip; every
// row is loaded once (16 B per lane), staged in a double-buffered LDS row (ds_write_b128) with
// its two halo columns, and the W/E neighbours are read back from LDS (ds_read_b64); N/S stay
// in registers.  One workgroup barrier per row.
// LDS row layout (doubles): [1] = left halo, [2 .. 513] = strip, [514] = right halo.
// -------------------------------------------------------------------------------------------
constexpr int LDS_STRIP = 512;
template <int DIV>
__global__ __launch_bounds__(256) void k_sweep_lds(const double* __restrict__ in,
                                                   double* __restrict__ out, int nx, int ny,
                                                   int pitch, int ry, int nwgx, int swz, Phys p) {
    __shared__ __attribute__((aligned(16))) double rows[2][LDS_STRIP + 8];
    const int tid = threadIdx.x;
    const int lin = xcd_remap(blockIdx.x, gridDim.x, swz);
    const int wgx = lin % nwgx, chunk = lin / nwgx;
    const int c0 = wgx * LDS_STRIP;
    const int jb = chunk * ry + 1;
    const int je = min(jb + ry - 1, ny);
    const int col = c0 + 2 * tid;
    const int nvalid = nx - col;
    const size_t xoff = static_cast<size_t>(LPAD + col);
    const bool have = (LPAD + col + 1) < pitch;
    const bool halo_lane = (tid == 0) || (tid == 255);
    const size_t hoff = static_cast<size_t>(tid == 0 ? LPAD + c0 - 1 : LPAD + min(c0 + LDS_STRIP, nx));
    const int hidx = tid == 0 ? 1 : LDS_STRIP + 2;

    auto ld2 = [&](int j) {
        double2 v = make_double2(0.0, 0.0);
        if (have) v = *reinterpret_cast<const double2*>(in + static_cast<size_t>(j) * pitch + xoff);
        return v;
    };
    auto ldh = [&](int j) {
        double e = 0.0;
        if (halo_lane) e = in[static_cast<size_t>(j) * pitch + hoff];
        return e;
    };

    double2 S = ld2(jb - 1);
    double2 C = ld2(jb);
    {
        const double h = ldh(jb);
        *reinterpret_cast<double2*>(&rows[jb & 1][2 + 2 * tid]) = C;
        if (halo_lane) rows[jb & 1][hidx] = h;
    }
    __syncthreads();
    for (int j = jb; j <= je; ++j) {
        const double2 N = ld2(j + 1);
        const double hN = ldh(j + 1);
        const double* cur = rows[j & 1];
        const double Wx = cur[1 + 2 * tid];
        const double Ey = cur[4 + 2 * tid];
        double* nxt = rows[(j + 1) & 1];
        *reinterpret_cast<double2*>(&nxt[2 + 2 * tid]) = N;
        if (halo_lane) nxt[hidx] = hN;
        const double ox = cell<DIV>(C.x, Wx, C.y, S.x, N.x, p);
        const double oy = cell<DIV>(C.y, C.x, Ey, S.y, N.y, p);
        store_pair(out + static_cast<size_t>(j) * pitch + xoff, ox, oy, nvalid);
        S = C;
        C = N;
        __syncthreads();
    }
}

// VAR_NAIVE — one thread per cell pair, all five points straight from global memory (cache
// reuse only).  Kept as the measured baseline the tuned variants are compared against.
template <int DIV>
__global__ __launch_bounds__(256) void k_sweep_naive(const double* __restrict__ in,
                                                     double* __restrict__ out, int nx, int ny,
                                                     int pitch, Phys p) {
    const int col = (blockIdx.x * 256 + threadIdx.x) * 2;
    const int j = blockIdx.y + 1;
    if (col >= nx) return;
    const size_t o = static_cast<size_t>(j) * pitch + LPAD + col;
    const double2 C = *reinterpret_cast<const double2*>(in + o);
    const double2 N = *reinterpret_cast<const double2*>(in + o + pitch);
    const double2 S = *reinterpret_cast<const double2*>(in + o - pitch);
    const double W = in[o - 1];
    const double E = in[o + 2];
    const double ox = cell<DIV>(C.x, W, C.y, S.x, N.x, p);
    const double oy = cell<DIV>(C.y, C.x, E, S.y, N.y, p);
    store_pair(out + o, ox, oy, nx - col);
}

// ---- operators at the reference's own granularity (not the hot path) ------------------------
// MODE 0: out = diffusion(u) on the interior (reference src/diffusion.cpp:9-16)
// MODE 1: out += advection(u)               (reference src/advection.cpp:13-33)
template <int DIV, int MODE>
__global__ __launch_bounds__(256) void k_unit_op(const double* __restrict__ in,
                                                 double* __restrict__ out, int nx, int ny,
                                                 int pitch, Phys p) {
    const int i = blockIdx.x * 256 + threadIdx.x + 1;
    const int j = blockIdx.y + 1;
    if (i > nx) return;
    const size_t o = static_cast<size_t>(j) * pitch + (LPAD - 1) + i;
    const double c = in[o], W = in[o - 1], E = in[o + 1], S = in[o - pitch], N = in[o + pitch];
    if (MODE == 0)
        out[o] = diffuse_term<DIV>(c, W, E, S, N, p);
    else
        out[o] = out[o] + advect_term<DIV>(c, W, E, S, N, p);
}

// outer ring of `in` copied to `out` (reference src/diffusion.cpp:18-25)
__global__ void k_ring_copy(const double* __restrict__ in, double* __restrict__ out, int nx, int ny,
                            int pitch) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int nxt = nx + 2, nyt = ny + 2;
    if (t < nxt) {
        const size_t a = static_cast<size_t>(LPAD - 1 + t);
        out[a] = in[a];
        const size_t b = static_cast<size_t>(nyt - 1) * pitch + a;
        out[b] = in[b];
    }
    if (t < nyt) {
        const size_t a = static_cast<size_t>(t) * pitch + (LPAD - 1);
        out[a] = in[a];
        out[a + nxt - 1] = in[a + nxt - 1];
    }
}

__global__ void k_fill(double* __restrict__ f, int nx, int ny, int pitch, double v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i < nx + 2) f[static_cast<size_t>(j) * pitch + (LPAD - 1) + i] = v;
}

// -------------------------------------------------------------------------------------------
// Ghost fill = unpack of the staged neighbour halos (reference src/halo.cpp:28-43) followed by
// apply_boundary (reference src/boundary.cpp:12-54) in ONE launch, written to the current
// field and, when `b` is given, identically to the ping-pong partner so that after the sweep
// and swap the new field carries the same ghost ring the reference gets from its copy
// (src/main.cpp:104) + ring copy (src/diffusion.cpp:18-25).
// The reference fills sides sequentially (left, right, bottom, top), which only matters at the
// four corners; they are evaluated functionally by one thread from values no other thread of
// this launch writes.
// -------------------------------------------------------------------------------------------
struct GhostDev {
    int bc[4];
    int phys[4];
    double value;
    const double* recv[4];
    const double* adj[4];  // != nullptr: the adjacent interior line of that side is read from here, not from `a`
    int ext_depth;  // > 0: also continue physical edges over that many halo cells (see ghost_extend_cell)
};

__device__ __forceinline__ size_t at(int i, int j, int pitch) {
    return static_cast<size_t>(j) * pitch + (LPAD - 1) + i;
}

// apply_boundary on the HALO part of a physical side: where a Dirichlet/Neumann edge meets a
// neighbour side, the ghost line continues over the H halo cells that came from that neighbour
// (globally, they are the neighbour's own ghost cells of the same physical edge).  t = 8
// segments x H cells.  Inputs are halo cells written by the preceding k_halo2_unpack launch.
__device__ __forceinline__ void ghost_extend_cell(double* __restrict__ f, int nx, int ny, int pitch, int H,
                                                  const int bc[4], const int phys[4], double value, int t) {
    const int seg = t / H, k = t % H;
    if (seg >= 8) return;
    auto at2 = [&](int i, int j) -> double& { return f[static_cast<ptrdiff_t>(j) * pitch + (LPAD - 1) + i]; };
    // segments 0..3: physical bottom/top row over the left/right halo columns
    // segments 4..7: physical left/right column over the bottom/top halo rows
    if (seg < 4) {
        const int row_side = (seg & 1) ? CSIM_TOP : CSIM_BOTTOM, col_side = (seg & 2) ? CSIM_RIGHT : CSIM_LEFT;
        if (!phys[row_side] || phys[col_side] || bc[row_side] == CSIM_BC_PERIODIC) return;
        const int i = col_side == CSIM_LEFT ? -k : nx + 1 + k;
        const int jg = row_side == CSIM_BOTTOM ? 0 : ny + 1, ja = row_side == CSIM_BOTTOM ? 1 : ny;
        at2(i, jg) = bc[row_side] == CSIM_BC_DIRICHLET ? value : at2(i, ja);
    } else {
        const int col_side = (seg & 1) ? CSIM_RIGHT : CSIM_LEFT, row_side = (seg & 2) ? CSIM_TOP : CSIM_BOTTOM;
        if (!phys[col_side] || phys[row_side] || bc[col_side] == CSIM_BC_PERIODIC) return;
        const int j = row_side == CSIM_BOTTOM ? -k : ny + 1 + k;
        const int ig = col_side == CSIM_LEFT ? 0 : nx + 1, ia = col_side == CSIM_LEFT ? 1 : nx;
        at2(ig, j) = bc[col_side] == CSIM_BC_DIRICHLET ? value : at2(ia, j);
    }
}

__global__ __launch_bounds__(256) void k_ghost_fill(double* __restrict__ a, double* __restrict__ b,
                                                    int nx, int ny, int pitch, GhostDev g) {
    __builtin_amdgcn_s_setprio(3);  // short latency-critical kernel, usually sharing the SIMDs with a bulk sweep
    const int t = blockIdx.x * 256 + threadIdx.x;
    auto put = [&](size_t o, double v) {
        a[o] = v;
        if (b) b[o] = v;
    };
    if (t < ny) {  // ghost columns at row j = t + 1
        const int j = t + 1;
        for (int s = CSIM_LEFT; s <= CSIM_RIGHT; ++s) {
            const int ig = s == CSIM_LEFT ? 0 : nx + 1;
            const int ia = s == CSIM_LEFT ? 1 : nx;
            if (g.phys[s]) {
                if (g.bc[s] == CSIM_BC_DIRICHLET)
                    put(at(ig, j, pitch), g.value);
                else if (g.bc[s] == CSIM_BC_NEUMANN)
                    put(at(ig, j, pitch), g.adj[s] ? g.adj[s][t] : a[at(ia, j, pitch)]);
            } else if (g.recv[s]) {
                put(at(ig, j, pitch), g.recv[s][t]);
            }
        }
    }
    if (t < nx) {  // ghost rows at column i = t + 1
        const int i = t + 1;
        for (int s = CSIM_BOTTOM; s <= CSIM_TOP; ++s) {
            const int jg = s == CSIM_BOTTOM ? 0 : ny + 1;
            const int ja = s == CSIM_BOTTOM ? 1 : ny;
            if (g.phys[s]) {
                if (g.bc[s] == CSIM_BC_DIRICHLET)
                    put(at(i, jg, pitch), g.value);
                else if (g.bc[s] == CSIM_BC_NEUMANN)
                    put(at(i, jg, pitch), g.adj[s] ? g.adj[s][t] : a[at(i, ja, pitch)]);
            } else if (g.recv[s]) {
                put(at(i, jg, pitch), g.recv[s][t]);
            }
        }
    }
    const int tc = nx > ny ? nx : ny;
    if (g.ext_depth > 0 && t > tc && t <= tc + 8 * g.ext_depth) {
        // (same values as the corner thread wherever the two overlap, so the order is immaterial)
        ghost_extend_cell(a, nx, ny, pitch, g.ext_depth, g.bc, g.phys, g.value, t - tc - 1);
        return;
    }
    if (t == tc) {  // the four corners
        for (int cs = CSIM_LEFT; cs <= CSIM_RIGHT; ++cs) {
            const int ig = cs == CSIM_LEFT ? 0 : nx + 1;
            const int ia = cs == CSIM_LEFT ? 1 : nx;
            for (int rs = CSIM_BOTTOM; rs <= CSIM_TOP; ++rs) {
                const int jg = rs == CSIM_BOTTOM ? 0 : ny + 1;
                const int ja = rs == CSIM_BOTTOM ? 1 : ny;
                const bool row_d = g.phys[rs] && g.bc[rs] == CSIM_BC_DIRICHLET;
                const bool row_n = g.phys[rs] && g.bc[rs] == CSIM_BC_NEUMANN;
                const bool col_d = g.phys[cs] && g.bc[cs] == CSIM_BC_DIRICHLET;
                const bool col_n = g.phys[cs] && g.bc[cs] == CSIM_BC_NEUMANN;
                if (row_d) {
                    put(at(ig, jg, pitch), g.value);
                } else if (row_n) {
                    // row rule copies the already column-filled ghost cell (ig, ja)
                    double v;
                    if (col_d)
                        v = g.value;
                    else if (col_n)
                        v = g.adj[cs] ? g.adj[cs][ja - 1] : a[at(ia, ja, pitch)];
                    else if (!g.phys[cs] && g.recv[cs])
                        v = g.recv[cs][ja - 1];
                    else
                        v = a[at(ig, ja, pitch)];
                    put(at(ig, jg, pitch), v);
                } else if (col_d) {
                    put(at(ig, jg, pitch), g.value);
                } else if (col_n) {
                    // column rule copies ghost-row cell (ia, jg): stable (periodic) or just received
                    double v;
                    if (!g.phys[rs] && g.recv[rs])
                        v = g.recv[rs][ia - 1];
                    else
                        v = a[at(ia, jg, pitch)];
                    put(at(ig, jg, pitch), v);
                }
            }
        }
    }
}

// The four edge lines of the NEXT field, computed from the current one and written directly
// into the send staging buffers, so the RCCL exchange can start before (and overlap with) the
// full sweep.  Same arithmetic as the sweep => the values sent equal the values later stored.
template <int DIV>
__global__ __launch_bounds__(256) void k_edge_pack(const double* __restrict__ in, int nx, int ny,
                                                   int pitch, Phys p, double* sl, double* sr,
                                                   double* sb, double* st) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    auto upd = [&](int i, int j) {
        const size_t o = at(i, j, pitch);
        return cell<DIV>(in[o], in[o - 1], in[o + 1], in[o - pitch], in[o + pitch], p);
    };
    if (t < ny) {
        if (sl) sl[t] = upd(1, t + 1);
        if (sr) sr[t] = upd(nx, t + 1);
    }
    if (t < nx) {
        if (sb) sb[t] = upd(t + 1, 1);
        if (st) st[t] = upd(t + 1, ny);
    }
}

__global__ __launch_bounds__(256) void k_pack(const double* __restrict__ in, int nx, int ny, int pitch,
                                              double* sl, double* sr, double* sb, double* st) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < ny) {
        if (sl) sl[t] = in[at(1, t + 1, pitch)];
        if (sr) sr[t] = in[at(nx, t + 1, pitch)];
    }
    if (t < nx) {
        if (sb) sb[t] = in[at(t + 1, 1, pitch)];
        if (st) st[t] = in[at(t + 1, ny, pitch)];
    }
}

// ---- deep faces for the multi-step sweep on several ranks (depth H = 2..7) --------------------
// Directions: 0 left, 1 right, 2 bottom, 3 top, 4 bottom-left, 5 bottom-right, 6 top-left,
// 7 top-right.  Faces hold the H outermost interior columns (H x (ny+2)), rows (H x (nx+2); ghost
// entries included so Periodic ghosts travel with them) or the H x H corner block.
struct Halo2Ptrs {
    double* p[8];
};

// H = face depth (= time steps of the fused pass that will consume the faces, 2..7).
// Column faces span rows 0..ny+1 and row faces columns 0..nx+1, i.e. they carry the sender's
// ghost entries along, so that Periodic (never rewritten) ghosts reach the neighbour.
__global__ __launch_bounds__(256) void k_halo2_pack(const double* __restrict__ f, int nx, int ny,
                                                    int pitch, int H, Halo2Ptrs s) {
    __builtin_amdgcn_s_setprio(3);  // short latency-critical kernel, usually sharing the SIMDs with a bulk sweep
    const int t = blockIdx.x * 256 + threadIdx.x;
    auto ld = [&](int i, int j) { return f[static_cast<ptrdiff_t>(j) * pitch + (LPAD - 1) + i]; };
    if (t < H * (ny + 2)) {
        const int c = t / (ny + 2), j = t % (ny + 2);
        if (s.p[0]) s.p[0][t] = ld(1 + c, j);
        if (s.p[1]) s.p[1][t] = ld(nx - H + 1 + c, j);
    }
    if (t < H * (nx + 2)) {
        const int r = t / (nx + 2), i = t % (nx + 2);
        if (s.p[2]) s.p[2][t] = ld(i, 1 + r);
        if (s.p[3]) s.p[3][t] = ld(i, ny - H + 1 + r);
    }
    if (t < H * H) {
        const int r = t / H, c = t % H;
        if (s.p[4]) s.p[4][t] = ld(1 + c, 1 + r);
        if (s.p[5]) s.p[5][t] = ld(nx - H + 1 + c, 1 + r);
        if (s.p[6]) s.p[6][t] = ld(1 + c, ny - H + 1 + r);
        if (s.p[7]) s.p[7][t] = ld(nx - H + 1 + c, ny - H + 1 + r);
    }
}

// r.p[d] = face received FROM direction d (the neighbour's face of the opposite direction).
// The ghost entries a face carries are kept only where the crossing side is a physical edge;
// next to a neighbour side the corner block of the diagonal rank supplies those cells.
__global__ __launch_bounds__(256) void k_halo2_unpack(double* __restrict__ f, int nx, int ny, int pitch,
                                                      int H, Halo2Ptrs r) {
    __builtin_amdgcn_s_setprio(3);  // short latency-critical kernel, usually sharing the SIMDs with a bulk sweep
    const int t = blockIdx.x * 256 + threadIdx.x;
    auto st = [&](int i, int j, double v) { f[static_cast<ptrdiff_t>(j) * pitch + (LPAD - 1) + i] = v; };
    if (t < H * (ny + 2)) {
        const int c = t / (ny + 2), j = t % (ny + 2);
        const bool keep = (j >= 1 && j <= ny) || (j == 0 && !r.p[2]) || (j == ny + 1 && !r.p[3]);
        if (keep) {
            if (r.p[0]) st(1 - H + c, j, r.p[0][t]);
            if (r.p[1]) st(nx + 1 + c, j, r.p[1][t]);
        }
    }
    if (t < H * (nx + 2)) {
        const int q = t / (nx + 2), i = t % (nx + 2);
        const bool keep = (i >= 1 && i <= nx) || (i == 0 && !r.p[0]) || (i == nx + 1 && !r.p[1]);
        if (keep) {
            if (r.p[2]) st(i, 1 - H + q, r.p[2][t]);
            if (r.p[3]) st(i, ny + 1 + q, r.p[3][t]);
        }
    }
    if (t < H * H) {
        const int q = t / H, c = t % H;
        if (r.p[4]) st(1 - H + c, 1 - H + q, r.p[4][t]);
        if (r.p[5]) st(nx + 1 + c, 1 - H + q, r.p[5][t]);
        if (r.p[6]) st(1 - H + c, ny + 1 + q, r.p[6][t]);
        if (r.p[7]) st(nx + 1 + c, ny + 1 + q, r.p[7][t]);
    }
}

// gaussian hotspot on the device (reference src/init.cpp:12-33); exp() may differ from glibc
// in the last ulp, so parity runs upload a host-made field instead.
__global__ __launch_bounds__(256) void k_gaussian(double* __restrict__ f, int nx, int ny, int pitch,
                                                  int x_off, int y_off, double dx, double dy,
                                                  double A, double xc, double yc, double sig) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= nx) return;
    const double x = (x_off + i + 0.5) * dx;
    const double y = (y_off + j + 0.5) * dy;
    const double r2 = (x - xc) * (x - xc) + (y - yc) * (y - yc);
    f[at(i + 1, j + 1, pitch)] = A * exp(-r2 / (2.0 * sig * sig));
}

// ---- wavefront-level reductions --------------------------------------------------------------
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = v + __shfl_xor(v, m, 64);
    return v;
}

// KIND 0: min/max over i0..i1, j0..j1 ; KIND 1: sum ; KIND 2: max |a-b|
template <int KIND>
__global__ __launch_bounds__(256) void k_reduce(const double* __restrict__ a,
                                                const double* __restrict__ b, int i0, int i1,
                                                int j0, int j1, int pitch,
                                                double* __restrict__ partial) {
    __shared__ double sh[2][4];
    double r0 = KIND == 0 ? INFINITY : 0.0;  // min | sum | linf
    double r1 = -INFINITY;                   // max
    for (int j = j0 + blockIdx.x; j <= j1; j += gridDim.x) {
        for (int i = i0 + threadIdx.x; i <= i1; i += 256) {
            const size_t o = at(i, j, pitch);
            const double v = a[o];
            if (KIND == 0) {
                r0 = fmin(r0, v);
                r1 = fmax(r1, v);
            } else if (KIND == 1) {
                r0 = r0 + v;
            } else {
                r0 = fmax(r0, fabs(v - b[o]));
            }
        }
    }
    if (KIND == 0) {
        r0 = wave_min(r0);
        r1 = wave_max(r1);
    } else if (KIND == 1) {
        r0 = wave_sum(r0);
    } else {
        r0 = wave_max(r0);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        sh[0][wave] = r0;
        sh[1][wave] = r1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double x0 = sh[0][0], x1 = sh[1][0];
        for (int w = 1; w < 4; ++w) {
            if (KIND == 0) {
                x0 = fmin(x0, sh[0][w]);
                x1 = fmax(x1, sh[1][w]);
            } else if (KIND == 1) {
                x0 = x0 + sh[0][w];
            } else {
                x0 = fmax(x0, sh[0][w]);
            }
        }
        partial[blockIdx.x] = x0;
        partial[REDUCE_BLOCKS + blockIdx.x] = x1;
    }
}

// Position-weighted 64-bit checksum of the interior: sum over cells of bits(u) * (K + 2 g) mod 2^64, g = the cell's
// GLOBAL linear index.  Every multiplier is odd, so any change of any cell changes the sum, and because the
// weights follow the global position the per-rank sums of a decomposed field add up (mod 2^64) to the checksum
// of the same field held by one rank — the bit-identity check a multi-GPU run can carry in one number.
__global__ __launch_bounds__(256) void k_checksum(const double* __restrict__ f, int nx, int ny, int pitch,
                                                  long x_off, long y_off, long nx_global,
                                                  unsigned long long* __restrict__ partial) {
    __shared__ unsigned long long sh[4];
    unsigned long long acc = 0;
    for (int j = 1 + blockIdx.x; j <= ny; j += gridDim.x) {
        const unsigned long long row = static_cast<unsigned long long>(y_off + j - 1) * static_cast<unsigned long long>(nx_global) +
                                       static_cast<unsigned long long>(x_off);
        for (int i = 1 + threadIdx.x; i <= nx; i += 256) {
            const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(f[at(i, j, pitch)]));
            acc += bits * (0x9E3779B97F4A7C15ull + 2ull * (row + static_cast<unsigned long long>(i - 1)));
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// ============================================================================================
// launchers
// ============================================================================================

template <int DIV>
static hipError_t sweep_div(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                            const SweepCfg& cfg, hipStream_t st) {
    int variant = cfg.variant == VAR_AUTO ? VAR_DPP : cfg.variant;
    int ry = cfg.rows_per_chunk > 0 ? cfg.rows_per_chunk : 64;
    if (ry > ny) ry = ny;
    const int nchunks = cdiv(ny, ry);
    if (variant == VAR_NAIVE) {
        dim3 grid(cdiv(nx, 512), ny);
        hipLaunchKernelGGL(k_sweep_naive<DIV>, grid, dim3(256), 0, st, in, out, nx, ny, pitch, p);
    } else if (variant == VAR_LDS) {
        const int nwgx = cdiv(nx, LDS_STRIP);
        hipLaunchKernelGGL(k_sweep_lds<DIV>, dim3(nwgx * nchunks), dim3(256), 0, st, in, out, nx, ny,
                           pitch, ry, nwgx, cfg.xcd_swizzle, p);
    } else {
        const int nwgx = cdiv(cdiv(nx, WAVE_COLS), 4);
        const dim3 grid(nwgx * nchunks);
        const int pf = cfg.prefetch > 0 ? cfg.prefetch : 2;
        if (pf <= 1)
            hipLaunchKernelGGL((k_sweep_dpp<DIV, 1>), grid, dim3(256), 0, st, in, out, nx, ny, pitch,
                               ry, nwgx, cfg.xcd_swizzle, p);
        else if (pf == 2)
            hipLaunchKernelGGL((k_sweep_dpp<DIV, 2>), grid, dim3(256), 0, st, in, out, nx, ny, pitch,
                               ry, nwgx, cfg.xcd_swizzle, p);
        else if (pf <= 4)
            hipLaunchKernelGGL((k_sweep_dpp<DIV, 4>), grid, dim3(256), 0, st, in, out, nx, ny, pitch,
                               ry, nwgx, cfg.xcd_swizzle, p);
        else
            hipLaunchKernelGGL((k_sweep_dpp<DIV, 8>), grid, dim3(256), 0, st, in, out, nx, ny, pitch,
                               ry, nwgx, cfg.xcd_swizzle, p);
    }
    return hipGetLastError();
}

// overlapped-strip multi-step sweep, T = 2..7 (kind[] / part: see internal.hpp)
hipError_t launch_sweepO(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                         const SweepCfg& cfg, const int kind[4], double value, int T, int part,
                         hipStream_t st, double* const fin_lines[4], const FrameSync* sync) {
    Bc2 bc;
    for (int s = 0; s < 4; ++s) bc.kind[s] = kind[s];
    bc.value = value;
    FinLines fin;
    for (int s = 0; s < 4; ++s) fin.line[s] = fin_lines ? fin_lines[s] : nullptr;
    const FrameSync fs = sync ? *sync : FrameSync{};
    switch (T) {
        case 2: return sweepO_T<2>(in, out, nx, ny, pitch, p, cfg, bc, fin, part, st, fs);
        case 3: return sweepO_T<3>(in, out, nx, ny, pitch, p, cfg, bc, fin, part, st, fs);
        case 4: return sweepO_T<4>(in, out, nx, ny, pitch, p, cfg, bc, fin, part, st, fs);
        case 5: return sweepO_T<5>(in, out, nx, ny, pitch, p, cfg, bc, fin, part, st, fs);
        case 6: return sweepO_T<6>(in, out, nx, ny, pitch, p, cfg, bc, fin, part, st, fs);
        default: return sweepO_T<7>(in, out, nx, ny, pitch, p, cfg, bc, fin, part, st, fs);
    }
}

hipError_t launch_halo2_pack(const double* f, int nx, int ny, int pitch, int depth,
                             double* const send[8], hipStream_t st) {
    Halo2Ptrs s;
    for (int d = 0; d < 8; ++d) s.p[d] = send[d];
    const int n = depth * (std::max(ny, nx) + 2);
    hipLaunchKernelGGL(k_halo2_pack, dim3(cdiv(n, 256)), dim3(256), 0, st, f, nx, ny, pitch, depth, s);
    return hipGetLastError();
}

hipError_t launch_halo2_unpack(double* f, int nx, int ny, int pitch, int depth,
                               double* const recv[8], hipStream_t st) {
    Halo2Ptrs r;
    for (int d = 0; d < 8; ++d) r.p[d] = recv[d];
    const int n = depth * (std::max(ny, nx) + 2);
    hipLaunchKernelGGL(k_halo2_unpack, dim3(cdiv(n, 256)), dim3(256), 0, st, f, nx, ny, pitch, depth, r);
    return hipGetLastError();
}

hipError_t launch_sweep(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                        const SweepCfg& cfg, hipStream_t st) {
    switch (p.div_mode) {
        case 0: return sweep_div<0>(in, out, nx, ny, pitch, p, cfg, st);
        case 1: return sweep_div<1>(in, out, nx, ny, pitch, p, cfg, st);
        case 3: return sweep_div<3>(in, out, nx, ny, pitch, p, cfg, st);
        default: return sweep_div<2>(in, out, nx, ny, pitch, p, cfg, st);
    }
}

template <int MODE>
static hipError_t unit_op(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                          hipStream_t st) {
    dim3 grid(cdiv(nx, 256), ny);
    switch (p.div_mode) {
        case 0: hipLaunchKernelGGL((k_unit_op<0, MODE>), grid, dim3(256), 0, st, in, out, nx, ny, pitch, p); break;
        case 1: hipLaunchKernelGGL((k_unit_op<1, MODE>), grid, dim3(256), 0, st, in, out, nx, ny, pitch, p); break;
        default: hipLaunchKernelGGL((k_unit_op<2, MODE>), grid, dim3(256), 0, st, in, out, nx, ny, pitch, p); break;
    }
    return hipGetLastError();
}

hipError_t launch_diffusion_only(const double* in, double* out, int nx, int ny, int pitch,
                                 const Phys& p, hipStream_t st) {
    return unit_op<0>(in, out, nx, ny, pitch, p, st);
}
hipError_t launch_advection_only(const double* in, double* out, int nx, int ny, int pitch,
                                 const Phys& p, hipStream_t st) {
    return unit_op<1>(in, out, nx, ny, pitch, p, st);
}

hipError_t launch_ring_copy(const double* in, double* out, int nx, int ny, int pitch, hipStream_t st) {
    const int n = (nx > ny ? nx : ny) + 2;
    hipLaunchKernelGGL(k_ring_copy, dim3(cdiv(n, 256)), dim3(256), 0, st, in, out, nx, ny, pitch);
    return hipGetLastError();
}

hipError_t launch_fill(double* f, int nx, int ny, int pitch, double v, hipStream_t st) {
    hipLaunchKernelGGL(k_fill, dim3(cdiv(nx + 2, 256), ny + 2), dim3(256), 0, st, f, nx, ny, pitch, v);
    return hipGetLastError();
}

hipError_t launch_ghost_fill(double* a, double* b, int nx, int ny, int pitch, const GhostArgs& g,
                             hipStream_t st, int ext_depth) {
    GhostDev d;
    d.ext_depth = ext_depth;
    for (int s = 0; s < 4; ++s) {
        d.bc[s] = g.bc[s];
        d.phys[s] = g.phys[s];
        d.recv[s] = g.recv[s];
        d.adj[s] = g.adj[s];
    }
    d.value = g.value;
    const int n = (nx > ny ? nx : ny) + 1 + 8 * ext_depth;
    hipLaunchKernelGGL(k_ghost_fill, dim3(cdiv(n, 256)), dim3(256), 0, st, a, b, nx, ny, pitch, d);
    return hipGetLastError();
}

hipError_t launch_edge_pack(const double* in, int nx, int ny, int pitch, const Phys& p,
                            double* const send[4], hipStream_t st) {
    const int n = nx > ny ? nx : ny;
    const dim3 grid(cdiv(n, 256));
    switch (p.div_mode) {
        case 0: hipLaunchKernelGGL(k_edge_pack<0>, grid, dim3(256), 0, st, in, nx, ny, pitch, p, send[0], send[1], send[2], send[3]); break;
        case 1: hipLaunchKernelGGL(k_edge_pack<1>, grid, dim3(256), 0, st, in, nx, ny, pitch, p, send[0], send[1], send[2], send[3]); break;
        case 3: hipLaunchKernelGGL(k_edge_pack<3>, grid, dim3(256), 0, st, in, nx, ny, pitch, p, send[0], send[1], send[2], send[3]); break;
        default: hipLaunchKernelGGL(k_edge_pack<2>, grid, dim3(256), 0, st, in, nx, ny, pitch, p, send[0], send[1], send[2], send[3]); break;
    }
    return hipGetLastError();
}

hipError_t launch_pack(const double* in, int nx, int ny, int pitch, double* const send[4],
                       hipStream_t st) {
    const int n = nx > ny ? nx : ny;
    hipLaunchKernelGGL(k_pack, dim3(cdiv(n, 256)), dim3(256), 0, st, in, nx, ny, pitch, send[0],
                       send[1], send[2], send[3]);
    return hipGetLastError();
}

hipError_t launch_gaussian(double* f, int nx, int ny, int pitch, int x_off, int y_off, int nxg,
                           int nyg, double dx, double dy, double A, double sigma_frac,
                           double xc_frac, double yc_frac, hipStream_t st) {
    const double Lx = nxg * dx, Ly = nyg * dy;
    const double xc = xc_frac * Lx, yc = yc_frac * Ly;
    const double sig = sigma_frac * (Lx < Ly ? Lx : Ly);
    hipLaunchKernelGGL(k_gaussian, dim3(cdiv(nx, 256), ny), dim3(256), 0, st, f, nx, ny, pitch, x_off,
                       y_off, dx, dy, A, xc, yc, sig);
    return hipGetLastError();
}

static int reduce_grid(int nrows) { return nrows < REDUCE_BLOCKS ? nrows : REDUCE_BLOCKS; }

hipError_t launch_minmax(const double* f, int nx, int ny, int pitch, double* scratch, hipStream_t st) {
    hipLaunchKernelGGL(k_reduce<0>, dim3(reduce_grid(ny + 2)), dim3(256), 0, st, f, nullptr, 0, nx + 1,
                       0, ny + 1, pitch, scratch);
    return hipGetLastError();
}
hipError_t launch_sum(const double* f, int nx, int ny, int pitch, double* scratch, hipStream_t st) {
    hipLaunchKernelGGL(k_reduce<1>, dim3(reduce_grid(ny)), dim3(256), 0, st, f, nullptr, 1, nx, 1, ny,
                       pitch, scratch);
    return hipGetLastError();
}
hipError_t launch_linf(const double* a, const double* b, int nx, int ny, int pitch, double* scratch,
                       hipStream_t st) {
    hipLaunchKernelGGL(k_reduce<2>, dim3(reduce_grid(ny)), dim3(256), 0, st, a, b, 1, nx, 1, ny, pitch,
                       scratch);
    return hipGetLastError();
}

hipError_t launch_checksum(const double* f, int nx, int ny, int pitch, long x_off, long y_off, long nx_global,
                           double* scratch, hipStream_t st) {
    hipLaunchKernelGGL(k_checksum, dim3(reduce_grid(ny)), dim3(256), 0, st, f, nx, ny, pitch, x_off, y_off, nx_global,
                       reinterpret_cast<unsigned long long*>(scratch));
    return hipGetLastError();
}

#endif  // !CSIM_INST_T

}  // namespace csim

// legacy_sweeps.hip — NOT BUILT.  Multi-step sweep families that k_sweepO_dpp (overlapped strips,
// csrc/kernels.hip) superseded in round 1, kept only as the record of what was measured:
//   k_sweep2_dpp / k_sweepT_dpp  2 and 3-4 steps per pass, non-overlapping 128-column strips whose outer
//                                lanes carry the extra columns (1.16 ms per 4-step 16384^2 launch:
//                                15 % slower per step than the overlapped strips)
//   k_sweepTw_dpp                the same with 256-column strips (186 VGPRs -> 2 waves/SIMD, slower)
// They were bit-identical to the oracle when they were retired (round-1 GPU test matrix).  The text
// below is the kernels and their launchers as they stood in csrc/kernels.hip; it needs that file's
// helpers (cell<>, from_prev_lane, xcd_remap, Bc2, ...) to compile.
// -------------------------------------------------------------------------------------------
// Two time steps per HBM pass (temporal blocking).  Same tiling as k_sweep_dpp, but while a
// wavefront marches up its strip it keeps TWO time levels in registers: level n rows (loaded),
// level n+1 rows (never stored) and emits level n+2.  HBM traffic per cell stays one 8-byte
// read + one 8-byte write per PASS, i.e. half of it per step, which is what lifts the sweep
// above the one-step copy ceiling.  The per-cell arithmetic is unchanged, so results stay
// bit-identical to two single steps.
//   - level n+1 is needed one column beyond the strip on each side: lanes 0 / 63 carry that
//     extra column (they load the two outer columns of every row as one 16-byte edge load).
//   - level n+1 is needed one row beyond the chunk: rows jb-1 .. je+1, from level-n rows
//     jb-2 .. je+2 (the device layout keeps two ghost rows/columns for this).
//   - where the strip/chunk touches a PHYSICAL edge, the level n+1 ghost value is not a stencil
//     result but the boundary rule applied to level n+1 (reference src/boundary.cpp:23-53 run
//     at the start of step n+1): Dirichlet -> value, Neumann -> adjacent interior at n+1,
//     Periodic (no-op) -> the stored ghost, unchanged.  kind 3 = side has a
//     neighbour rank: plain stencil on the stored depth-2 halo.
// Requires nx % 128 == 0 (every BASELINE grid and tile); other widths use the one-step kernel.
// -------------------------------------------------------------------------------------------
template <int DIV, int PF>
__global__ __launch_bounds__(256) void k_sweep2_dpp(const double* __restrict__ in,
                                                    double* __restrict__ out, int nx, int ny,
                                                    int pitch, int ry, int nwgx, int nchunks,
                                                    int part, int swz, Phys p, Bc2 bc) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // part 0: every tile.  part 1: only the FRAME tiles (first/last chunk, first/last
    // 128-column strip) — launched first on a multi-rank run so that the depth-2 faces can be
    // packed and sent while part 2 (all the other tiles) is still computing.
    int wgx, chunk, side = -1;
    if (part == 1 && nchunks >= 2) {
        const int b = blockIdx.x;
        if (b < 2 * nwgx) {
            chunk = b < nwgx ? 0 : nchunks - 1;
            wgx = b < nwgx ? b : b - nwgx;
        } else {
            chunk = 1 + ((b - 2 * nwgx) >> 1);
            side = (b - 2 * nwgx) & 1;
            wgx = side ? nwgx - 1 : 0;
        }
    } else {
        const int lin = xcd_remap(blockIdx.x, gridDim.x, swz);
        wgx = lin % nwgx;
        chunk = lin / nwgx;
    }
    const int c0 = (wgx * 4 + wave) * WAVE_COLS;
    if (c0 >= nx) return;  // wave-uniform
    if (side == 0 && c0 != 0) return;
    if (side == 1 && c0 + WAVE_COLS != nx) return;
    if (part == 2 && (nchunks < 2 || chunk == 0 || chunk == nchunks - 1 || c0 == 0 || c0 + WAVE_COLS == nx))
        return;
    const int jb = chunk * ry + 1;
    const int je = min(jb + ry - 1, ny);
    const ptrdiff_t xoff = LPAD + c0 + 2 * lane;
    const bool lane0 = lane == 0;
    const bool edge_lane = lane0 || (lane == 63);
    const ptrdiff_t eoff = LPAD + c0 + (lane0 ? -2 : WAVE_COLS);
    // boundary kind seen by this wave on each side (3 = keep the stencil)
    const int kl = c0 == 0 ? bc.kind[CSIM_LEFT] : 3;
    const int kr = c0 + WAVE_COLS == nx ? bc.kind[CSIM_RIGHT] : 3;
    const int kb = jb == 1 ? bc.kind[CSIM_BOTTOM] : 3;
    const int kt = je == ny ? bc.kind[CSIM_TOP] : 3;
    const int kx = lane0 ? kl : kr;  // rule for this lane's extra column

    auto ld2 = [&](int j) {
        return *reinterpret_cast<const double2*>(in + static_cast<ptrdiff_t>(j) * pitch + xoff);
    };
    auto lde = [&](int j) {
        double2 e = make_double2(0.0, 0.0);
        if (edge_lane) e = *reinterpret_cast<const double2*>(in + static_cast<ptrdiff_t>(j) * pitch + eoff);
        return e;
    };

    double2 aS = ld2(jb - 2), aC = ld2(jb - 1);
    double2 eS = lde(jb - 2), eC = lde(jb - 1);
    double2 q[PF], eq[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        q[u] = make_double2(0.0, 0.0);
        eq[u] = make_double2(0.0, 0.0);
        const int r = jb + u;
        if (r <= je + 2) {
            q[u] = ld2(r);
            eq[u] = lde(r);
        }
    }
    double2 bSS = make_double2(0.0, 0.0), bS = make_double2(0.0, 0.0);
    double xS = 0.0;
    const int rlast = je + 1;
    for (int r0 = jb - 1; r0 <= rlast; r0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int r = r0 + u;  // row of level n+1 produced in this sub-iteration
            if (r <= rlast) {      // wave-uniform
                const double2 aN = q[u];
                const double2 eN = eq[u];
                const int rn = r + 1 + PF;
                if (rn <= je + 2) {
                    q[u] = ld2(rn);
                    eq[u] = lde(rn);
                }
                // ---- level n+1, row r --------------------------------------------------
                double2 b;
                double x;
                const bool gb = (r == 0) && (kb != 3);
                const bool gt = (r == ny + 1) && (kt != 3);
                if (gb || gt) {  // ghost ROW of level n+1: boundary rule, not a stencil
                    const int k = gb ? kb : kt;
                    if (k == CSIM_BC_DIRICHLET) {
                        b = make_double2(bc.value, bc.value);
                        x = bc.value;
                    } else if (k == CSIM_BC_PERIODIC) {
                        b = aC;
                        x = lane0 ? eC.y : eC.x;
                    } else if (gt) {  // Neumann top: copy of row ny at level n+1
                        b = bS;
                        x = xS;
                    } else {  // Neumann bottom: patched below once row 1 exists
                        b = make_double2(0.0, 0.0);
                        x = 0.0;
                    }
                } else {
                    const double Wx = from_prev_lane(aC.y, eC.y);
                    const double Ey = from_next_lane(aC.x, eC.x);
                    b.x = cell<DIV>(aC.x, Wx, aC.y, aS.x, aN.x, p);
                    b.y = cell<DIV>(aC.y, aC.x, Ey, aS.y, aN.y, p);
                    // the extra column: lane 0 owns column c0-1, lane 63 column c0+128
                    const double xc = lane0 ? eC.y : eC.x;
                    const double xw = lane0 ? eC.x : aC.y;
                    const double xe = lane0 ? aC.x : eC.y;
                    const double xs = lane0 ? eS.y : eS.x;
                    const double xn = lane0 ? eN.y : eN.x;
                    const double xst = cell<DIV>(xc, xw, xe, xs, xn, p);
                    x = kx == 3 ? xst
                        : kx == CSIM_BC_DIRICHLET ? bc.value
                        : kx == CSIM_BC_NEUMANN ? (lane0 ? b.x : b.y)
                                                : xc;
                }
                if (r == 1 && kb == CSIM_BC_NEUMANN) {  // level n+1 bottom ghost row := row 1
                    bS = b;
                    xS = x;
                }
                // ---- level n+2, row r-1 ------------------------------------------------
                if (r - 1 >= jb) {
                    const double Wx = from_prev_lane(bS.y, xS);
                    const double Ey = from_next_lane(bS.x, xS);
                    const double ox = cell<DIV>(bS.x, Wx, bS.y, bSS.x, b.x, p);
                    const double oy = cell<DIV>(bS.y, bS.x, Ey, bSS.y, b.y, p);
                    *reinterpret_cast<double2*>(out + static_cast<ptrdiff_t>(r - 1) * pitch + xoff) =
                        make_double2(ox, oy);
                }
                bSS = bS;
                bS = b;
                xS = x;
                aS = aC;
                aC = aN;
                eS = eC;
                eC = eN;
            }
        }
    }
}

// -------------------------------------------------------------------------------------------
// T time steps per HBM pass (T = 3, 4): the general form of k_sweep2_dpp for a single rank.
// A wavefront keeps T time levels in registers while it marches: level 0 rows are loaded, level
// l row (r - l + 1) is produced from level l-1 rows in iteration r, level T is stored.  Level l
// is needed T - l columns beyond the 128-column strip on each side; those "extras" live in ONE
// more register per lane, spread over the lanes (lane k <-> column c0-1-k, lane 63-k <-> column
// c0+128+k), so a whole level of extras costs a single wave-wide cell update:
//     W(extra) = from_next_lane(X, edge = own .y on lane 63),  E(extra) = from_prev_lane(X, edge
//     = own .x on lane 0)   — the same expression serves both sides of the strip.
// Ghost rows / ghost columns of the intermediate levels follow the boundary rule exactly like
// k_sweep2_dpp (Dirichlet value, Neumann = adjacent interior of the same level, Periodic = the
// stored ghost).  Cells outside the ghost ring are computed as garbage and never consumed.
// HBM traffic per cell and pass: 8 B read (+2T/ry halo rows) + 8 B written, for T steps.
// -------------------------------------------------------------------------------------------
struct Row3 {
    double2 m;  // the lane's two strip columns
    double x;   // the lane's extra column (meaningful on the outer lanes only)
};

// Register-resident pipeline of T time levels.  To keep every register index static (no
// rotation moves) the march is unrolled 6-fold: level 0 uses a 6-slot ring (3 rows in use + 3
// more in flight from HBM), levels 1..T-1 use 3-slot rings, and in iteration k every level
// writes slot k mod 3 (level 0: the row that arrives sits in slot (k+2) mod 6).
// EDGE = false is the branch-free body for wavefronts whose strip and (extended) chunk touch no
// physical edge — the vast majority; EDGE = true adds the boundary rules.
template <int DIV, int T, bool EDGE, int SX, int SY>
__device__ __forceinline__ void sweepT_march(const double* __restrict__ in, double* __restrict__ out,
                                             int ny, int pitch, int jb, int je, int c0, int lane,
                                             int kl, int kr, const Phys& p, const Bc2& bc) {
    const ptrdiff_t xoff = LPAD + c0 + 2 * lane;
    const bool lane0 = lane == 0, lane63 = lane == 63;
    const bool xlane = lane < T || lane > 63 - T;
    const ptrdiff_t eoff = lane < 32 ? LPAD + c0 - 1 - lane : LPAD + c0 + WAVE_COLS + (63 - lane);
    const int kb = bc.kind[CSIM_BOTTOM], kt = bc.kind[CSIM_TOP];  // 3 = neighbour rank: plain stencil
    const int kx = lane0 ? kl : (lane63 ? kr : 3);

    auto load = [&](int j) {
        Row3 r;
        const double* row = in + static_cast<ptrdiff_t>(j) * pitch;
        r.m = *reinterpret_cast<const double2*>(row + xoff);
        r.x = 0.0;
        if (xlane) r.x = row[eoff];
        return r;
    };

    const int r_first = jb - (T - 1);
    const int niter = (je - jb + 1) + 2 * (T - 1);
    const int last_row = r_first + niter;  // newest level-0 row ever needed (= r_last + 1)
    Row3 L0[6];
    Row3 L[T][3];  // L[l] used for l = 1 .. T-1
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        L0[q].m = make_double2(0.0, 0.0);
        L0[q].x = 0.0;
        const int row = r_first - 1 + q;
        if (row <= last_row) L0[q] = load(row);
    }
#pragma unroll
    for (int l = 0; l < T; ++l)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            L[l][q].m = make_double2(0.0, 0.0);
            L[l][q].x = 0.0;
        }

    for (int k0 = 0; k0 < niter; k0 += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int k = k0 + u;
            if (k < niter) {  // wave-uniform
                const int r = r_first + k;  // row of level 1 produced in this iteration
#pragma unroll
                for (int l = 1; l <= T; ++l) {
                    const int rho = r - l + 1;
                    // rows rho-1, rho, rho+1 of level l-1
                    const Row3 s = (l == 1) ? L0[u % 6] : L[l - 1][(u + 1) % 3];
                    const Row3 c = (l == 1) ? L0[(u + 1) % 6] : L[l - 1][(u + 2) % 3];
                    const Row3 n = (l == 1) ? L0[(u + 2) % 6] : L[l - 1][u % 3];
                    Row3 o;
                    bool ghost_row = false;
                    if (EDGE && l < T) {
                        const bool gb = rho == 0 && kb != 3, gt = rho == ny + 1 && kt != 3;
                        ghost_row = gb || gt;
                        if (ghost_row) {  // boundary rule instead of the stencil
                            const int kk = gb ? kb : kt;
                            if (kk == CSIM_BC_DIRICHLET) {
                                o.m = make_double2(bc.value, bc.value);
                                o.x = bc.value;
                            } else if (kk == CSIM_BC_PERIODIC) {
                                o = c;
                            } else if (gt) {  // Neumann top: row ny of this level (made last iteration)
                                o = L[l][(u + 2) % 3];
                            } else {  // Neumann bottom: patched when row 1 of this level exists
                                o.m = make_double2(0.0, 0.0);
                                o.x = 0.0;
                            }
                        }
                    }
                    if (!ghost_row) {
                        const double Wx = from_prev_lane(c.m.y, c.x);
                        const double Ey = from_next_lane(c.m.x, c.x);
                        o.m.x = cell<DIV, SX, SY>(c.m.x, Wx, c.m.y, s.m.x, n.m.x, p);
                        o.m.y = cell<DIV, SX, SY>(c.m.y, c.m.x, Ey, s.m.y, n.m.y, p);
                        o.x = 0.0;
                        if (l < T) {
                            const double xw = from_next_lane(c.x, c.m.y);
                            const double xe = from_prev_lane(c.x, c.m.x);
                            o.x = cell<DIV, SX, SY>(c.x, xw, xe, s.x, n.x, p);
                            if (EDGE) {  // ghost column of this level on a physical edge
                                o.x = kx == 3 ? o.x
                                      : kx == CSIM_BC_DIRICHLET ? bc.value
                                      : kx == CSIM_BC_NEUMANN ? (lane0 ? o.m.x : o.m.y)
                                                              : c.x;
                            }
                        }
                    }
                    if (l < T) {
                        if (EDGE && rho == 1 && kb == CSIM_BC_NEUMANN) L[l][(u + 2) % 3] = o;  // ghost row 0 := row 1
                        L[l][u % 3] = o;
                    } else if (rho >= jb) {
                        *reinterpret_cast<double2*>(out + static_cast<ptrdiff_t>(rho) * pitch + xoff) = o.m;
                    }
                }
                // level-0 row r-1 is dead now: fetch row r+5 into its slot
                const int rn = r + 5;
                if (rn <= last_row) L0[u % 6] = load(rn);
            }
        }
    }
}

template <int DIV, int T, int SX, int SY>
__global__ __launch_bounds__(256) void k_sweepT_dpp(const double* __restrict__ in,
                                                    double* __restrict__ out, int nx, int ny,
                                                    int pitch, int ry, int nwgx, int nchunks,
                                                    int part, int swz, int stagger, Phys p, Bc2 bc) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    if (stagger > 0) {
        // Co-resident workgroups of one CU start a fraction of a row-iteration apart, so that the
        // wavefronts sharing a SIMD do not hit their load and arithmetic phases in lockstep
        // (matters when the whole launch is a single round of wavefronts: small per-GPU tiles).
        const int slot = ((blockIdx.x >> 3) >> 5) & 3;
        for (int k = 0; k < slot * stagger; ++k) __builtin_amdgcn_s_sleep(1);
    }
    // part 0: every tile; 1: frame tiles only; 2: all but the frame tiles (see k_sweep2_dpp)
    int wgx, chunk, side = -1;
    if (part == 1 && nchunks >= 2) {
        const int b = blockIdx.x;
        if (b < 2 * nwgx) {
            chunk = b < nwgx ? 0 : nchunks - 1;
            wgx = b < nwgx ? b : b - nwgx;
        } else {
            chunk = 1 + ((b - 2 * nwgx) >> 1);
            side = (b - 2 * nwgx) & 1;
            wgx = side ? nwgx - 1 : 0;
        }
    } else {
        const int lin = xcd_remap(blockIdx.x, gridDim.x, swz);
        wgx = lin % nwgx;
        chunk = lin / nwgx;
    }
    const int c0 = (wgx * 4 + wave) * WAVE_COLS;
    if (c0 >= nx) return;  // wave-uniform
    if (side == 0 && c0 != 0) return;
    if (side == 1 && c0 + WAVE_COLS != nx) return;
    if (part == 2 && (nchunks < 2 || chunk == 0 || chunk == nchunks - 1 || c0 == 0 || c0 + WAVE_COLS == nx))
        return;
    const int jb = chunk * ry + 1;
    const int je = min(jb + ry - 1, ny);
    const int kl = c0 == 0 ? bc.kind[CSIM_LEFT] : 3;
    const int kr = c0 + WAVE_COLS == nx ? bc.kind[CSIM_RIGHT] : 3;
    // intermediate-level rows reach T-1 rows beyond the chunk: any chunk that close to a
    // PHYSICAL bottom/top edge meets the ghost rows 0 / ny+1 (the rule fires on the row index)
    const bool edge = kl != 3 || kr != 3 || (bc.kind[CSIM_BOTTOM] != 3 && jb - (T - 1) < 1) ||
                      (bc.kind[CSIM_TOP] != 3 && je + (T - 1) > ny);
    if (edge)
        sweepT_march<DIV, T, true, SX, SY>(in, out, ny, pitch, jb, je, c0, lane, kl, kr, p, bc);
    else
        sweepT_march<DIV, T, false, SX, SY>(in, out, ny, pitch, jb, je, c0, lane, kl, kr, p, bc);
}

// -------------------------------------------------------------------------------------------
// Wide flavour of the T-step sweep: one wavefront owns 256 columns as two 128-column halves
// (lane l holds columns 2l,2l+1 of each half, so both row loads stay 16-byte-per-lane and fully
// coalesced).  The kernel is VALU-bound at T >= 3, and the per-level cost of the extra columns is
// one wave-wide cell update whatever the strip width: over 256 columns it is 5 updates per 4
// useful ones instead of 3 per 2 (-17 % arithmetic).  The seam between the halves is bridged by
// v_readlane (lane 63 of half A <-> lane 0 of half B).  Needs nx % 256 == 0.
// -------------------------------------------------------------------------------------------
constexpr int WIDE_COLS = 256;

struct Row5 {
    double2 a, b;  // the lane's columns in the two halves of the strip
    double x;      // the lane's extra column (outer lanes only)
};

__device__ __forceinline__ double lane_value(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

template <int DIV, int T, bool EDGE, int SX, int SY>
__device__ __forceinline__ void sweepTw_march(const double* __restrict__ in, double* __restrict__ out,
                                              int ny, int pitch, int jb, int je, int c0, int lane,
                                              int kl, int kr, const Phys& p, const Bc2& bc) {
    const ptrdiff_t xoffA = LPAD + c0 + 2 * lane;
    const ptrdiff_t xoffB = xoffA + WAVE_COLS;
    const bool lane0 = lane == 0, lane63 = lane == 63;
    const bool xlane = lane < T || lane > 63 - T;
    const ptrdiff_t eoff = lane < 32 ? LPAD + c0 - 1 - lane : LPAD + c0 + WIDE_COLS + (63 - lane);
    const int kb = bc.kind[CSIM_BOTTOM], kt = bc.kind[CSIM_TOP];  // 3 = neighbour rank: plain stencil
    const int kx = lane0 ? kl : (lane63 ? kr : 3);

    auto load = [&](int j) {
        Row5 r;
        const double* row = in + static_cast<ptrdiff_t>(j) * pitch;
        r.a = *reinterpret_cast<const double2*>(row + xoffA);
        r.b = *reinterpret_cast<const double2*>(row + xoffB);
        r.x = 0.0;
        if (xlane) r.x = row[eoff];
        return r;
    };
    auto zero = [] {
        Row5 r;
        r.a = make_double2(0.0, 0.0);
        r.b = make_double2(0.0, 0.0);
        r.x = 0.0;
        return r;
    };

    const int r_first = jb - (T - 1);
    const int niter = (je - jb + 1) + 2 * (T - 1);
    const int last_row = r_first + niter;
    Row5 L0[6];
    Row5 L[T][3];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        L0[q] = zero();
        const int row = r_first - 1 + q;
        if (row <= last_row) L0[q] = load(row);
    }
#pragma unroll
    for (int l = 0; l < T; ++l)
#pragma unroll
        for (int q = 0; q < 3; ++q) L[l][q] = zero();

    for (int k0 = 0; k0 < niter; k0 += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int k = k0 + u;
            if (k < niter) {  // wave-uniform
                const int r = r_first + k;
#pragma unroll
                for (int l = 1; l <= T; ++l) {
                    const int rho = r - l + 1;
                    const Row5 s = (l == 1) ? L0[u % 6] : L[l - 1][(u + 1) % 3];
                    const Row5 c = (l == 1) ? L0[(u + 1) % 6] : L[l - 1][(u + 2) % 3];
                    const Row5 n = (l == 1) ? L0[(u + 2) % 6] : L[l - 1][u % 3];
                    Row5 o;
                    bool ghost_row = false;
                    if (EDGE && l < T) {
                        const bool gb = rho == 0 && kb != 3, gt = rho == ny + 1 && kt != 3;
                        ghost_row = gb || gt;
                        if (ghost_row) {
                            const int kk = gb ? kb : kt;
                            if (kk == CSIM_BC_DIRICHLET) {
                                o.a = make_double2(bc.value, bc.value);
                                o.b = o.a;
                                o.x = bc.value;
                            } else if (kk == CSIM_BC_PERIODIC) {
                                o = c;
                            } else if (gt) {
                                o = L[l][(u + 2) % 3];
                            } else {
                                o = zero();
                            }
                        }
                    }
                    if (!ghost_row) {
                        const double seamA = lane_value(c.a.y, 63);  // column c0+127
                        const double seamB = lane_value(c.b.x, 0);   // column c0+128
                        const double Wax = from_prev_lane(c.a.y, c.x);
                        const double Eay = from_next_lane(c.a.x, seamB);
                        const double Wbx = from_prev_lane(c.b.y, seamA);
                        const double Eby = from_next_lane(c.b.x, c.x);
                        o.a.x = cell<DIV, SX, SY>(c.a.x, Wax, c.a.y, s.a.x, n.a.x, p);
                        o.a.y = cell<DIV, SX, SY>(c.a.y, c.a.x, Eay, s.a.y, n.a.y, p);
                        o.b.x = cell<DIV, SX, SY>(c.b.x, Wbx, c.b.y, s.b.x, n.b.x, p);
                        o.b.y = cell<DIV, SX, SY>(c.b.y, c.b.x, Eby, s.b.y, n.b.y, p);
                        o.x = 0.0;
                        if (l < T) {
                            const double xw = from_next_lane(c.x, c.b.y);
                            const double xe = from_prev_lane(c.x, c.a.x);
                            o.x = cell<DIV, SX, SY>(c.x, xw, xe, s.x, n.x, p);
                            if (EDGE) {
                                o.x = kx == 3 ? o.x
                                      : kx == CSIM_BC_DIRICHLET ? bc.value
                                      : kx == CSIM_BC_NEUMANN ? (lane0 ? o.a.x : o.b.y)
                                                              : c.x;
                            }
                        }
                    }
                    if (l < T) {
                        if (EDGE && rho == 1 && kb == CSIM_BC_NEUMANN) L[l][(u + 2) % 3] = o;
                        L[l][u % 3] = o;
                    } else if (rho >= jb) {
                        double* dst = out + static_cast<ptrdiff_t>(rho) * pitch;
                        *reinterpret_cast<double2*>(dst + xoffA) = o.a;
                        *reinterpret_cast<double2*>(dst + xoffB) = o.b;
                    }
                }
                const int rn = r + 5;
                if (rn <= last_row) L0[u % 6] = load(rn);
            }
        }
    }
}

template <int DIV, int T, int SX, int SY>
__global__ __launch_bounds__(256) void k_sweepTw_dpp(const double* __restrict__ in,
                                                     double* __restrict__ out, int nx, int ny,
                                                     int pitch, int ry, int nwgx, int nchunks,
                                                     int part, int swz, Phys p, Bc2 bc) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int wgx, chunk, side = -1;
    if (part == 1 && nchunks >= 2) {
        const int b = blockIdx.x;
        if (b < 2 * nwgx) {
            chunk = b < nwgx ? 0 : nchunks - 1;
            wgx = b < nwgx ? b : b - nwgx;
        } else {
            chunk = 1 + ((b - 2 * nwgx) >> 1);
            side = (b - 2 * nwgx) & 1;
            wgx = side ? nwgx - 1 : 0;
        }
    } else {
        const int lin = xcd_remap(blockIdx.x, gridDim.x, swz);
        wgx = lin % nwgx;
        chunk = lin / nwgx;
    }
    const int c0 = (wgx * 4 + wave) * WIDE_COLS;
    if (c0 >= nx) return;  // wave-uniform
    if (side == 0 && c0 != 0) return;
    if (side == 1 && c0 + WIDE_COLS != nx) return;
    if (part == 2 && (nchunks < 2 || chunk == 0 || chunk == nchunks - 1 || c0 == 0 || c0 + WIDE_COLS == nx))
        return;
    const int jb = chunk * ry + 1;
    const int je = min(jb + ry - 1, ny);
    const int kl = c0 == 0 ? bc.kind[CSIM_LEFT] : 3;
    const int kr = c0 + WIDE_COLS == nx ? bc.kind[CSIM_RIGHT] : 3;
    const bool edge = kl != 3 || kr != 3 || (bc.kind[CSIM_BOTTOM] != 3 && jb - (T - 1) < 1) ||
                      (bc.kind[CSIM_TOP] != 3 && je + (T - 1) > ny);
    if (edge)
        sweepTw_march<DIV, T, true, SX, SY>(in, out, ny, pitch, jb, je, c0, lane, kl, kr, p, bc);
    else
        sweepTw_march<DIV, T, false, SX, SY>(in, out, ny, pitch, jb, je, c0, lane, kl, kr, p, bc);
}

template <int DIV>
static hipError_t sweep2_div(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                             const SweepCfg& cfg, const Bc2& bc, int part, hipStream_t st) {
    int ry = cfg.rows_per_chunk > 0 ? cfg.rows_per_chunk : 64;
    if (ry > ny) ry = ny;
    const int nchunks = cdiv(ny, ry);
    const int nwgx = cdiv(cdiv(nx, WAVE_COLS), 4);
    int nblocks = nwgx * nchunks;
    if (part == 1 && nchunks >= 2) nblocks = 2 * nwgx + 2 * (nchunks - 2);
    if (part == 2 && nchunks < 3) return hipSuccess;  // every tile is a frame tile
    const dim3 grid(nblocks);
    const int pf = cfg.prefetch > 0 ? cfg.prefetch : 2;
    if (pf <= 1)
        hipLaunchKernelGGL((k_sweep2_dpp<DIV, 1>), grid, dim3(256), 0, st, in, out, nx, ny, pitch, ry,
                           nwgx, nchunks, part, cfg.xcd_swizzle, p, bc);
    else if (pf == 2)
        hipLaunchKernelGGL((k_sweep2_dpp<DIV, 2>), grid, dim3(256), 0, st, in, out, nx, ny, pitch, ry,
                           nwgx, nchunks, part, cfg.xcd_swizzle, p, bc);
    else
        hipLaunchKernelGGL((k_sweep2_dpp<DIV, 4>), grid, dim3(256), 0, st, in, out, nx, ny, pitch, ry,
                           nwgx, nchunks, part, cfg.xcd_swizzle, p, bc);
    return hipGetLastError();
}

template <int DIV, int T>
static hipError_t sweepT_div(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                             const SweepCfg& cfg, const Bc2& bc, int part, hipStream_t st) {
    const bool wide = cfg.wide != 0 && nx % WIDE_COLS == 0;
    const int strip = wide ? WIDE_COLS : WAVE_COLS;
    // rows per chunk: 64 on big tiles; smaller tiles (strong scaling across GPUs) trade a little
    // redundant halo work for enough wavefronts to fill the chip (measured on 4096..16384 tiles:
    // >= 8192 wavefronts per launch is the knee for the 128-column strips)
    int ry = cfg.rows_per_chunk;
    if (ry <= 0) {
        const long strips = cdiv(nx, strip);
        const long want = wide ? 4096 : 8192;
        ry = 64;
        while (ry > 16 && strips * cdiv(ny, ry) < want) ry >>= 1;
    }
    if (ry > ny) ry = ny;
    const int nchunks = cdiv(ny, ry);
    const int nwgx = cdiv(cdiv(nx, strip), 4);
    int nblocks = nwgx * nchunks;
    if (part == 1 && nchunks >= 2) nblocks = 2 * nwgx + 2 * (nchunks - 2);
    if (part == 2 && nchunks < 3) return hipSuccess;  // every tile is a frame tile
    const dim3 grid(nblocks), block(256);
    const int sw = cfg.xcd_swizzle;
    const int sign = (p.vx >= 0.0 ? 2 : 0) + (p.vy >= 0.0 ? 1 : 0);
#define CSIM_LAUNCH_T(KERNEL, SXV, SYV)                                                               \
    hipLaunchKernelGGL((KERNEL<DIV, T, SXV, SYV>), grid, block, 0, st, in, out, nx, ny, pitch, ry, nwgx, \
                       nchunks, part, sw, p, bc)
#define CSIM_LAUNCH_TS(KERNEL, SXV, SYV)                                                              \
    hipLaunchKernelGGL((KERNEL<DIV, T, SXV, SYV>), grid, block, 0, st, in, out, nx, ny, pitch, ry, nwgx, \
                       nchunks, part, sw, cfg.stagger, p, bc)
    if (wide) {
        switch (sign) {
            case 3: CSIM_LAUNCH_T(k_sweepTw_dpp, 1, 1); break;
            case 2: CSIM_LAUNCH_T(k_sweepTw_dpp, 1, 0); break;
            case 1: CSIM_LAUNCH_T(k_sweepTw_dpp, 0, 1); break;
            default: CSIM_LAUNCH_T(k_sweepTw_dpp, 0, 0); break;
        }
    } else {
        switch (sign) {
            case 3: CSIM_LAUNCH_TS(k_sweepT_dpp, 1, 1); break;
            case 2: CSIM_LAUNCH_TS(k_sweepT_dpp, 1, 0); break;
            case 1: CSIM_LAUNCH_TS(k_sweepT_dpp, 0, 1); break;
            default: CSIM_LAUNCH_TS(k_sweepT_dpp, 0, 0); break;
        }
    }
#undef CSIM_LAUNCH_T
#undef CSIM_LAUNCH_TS
    return hipGetLastError();
}

// T = 3 or 4 time steps per pass; kind[s] = CSIM_BC_* on physical sides, 3 on neighbour sides
hipError_t launch_sweepT(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                         const SweepCfg& cfg, const int kind[4], double value, int T, int part,
                         hipStream_t st) {
    Bc2 bc;
    for (int s = 0; s < 4; ++s) bc.kind[s] = kind[s];
    bc.value = value;
    if (T == 3) {
        switch (p.div_mode) {
            case 0: return sweepT_div<0, 3>(in, out, nx, ny, pitch, p, cfg, bc, part, st);
            case 1: return sweepT_div<1, 3>(in, out, nx, ny, pitch, p, cfg, bc, part, st);
            default: return sweepT_div<2, 3>(in, out, nx, ny, pitch, p, cfg, bc, part, st);
        }
    }
    switch (p.div_mode) {
        case 0: return sweepT_div<0, 4>(in, out, nx, ny, pitch, p, cfg, bc, part, st);
        case 1: return sweepT_div<1, 4>(in, out, nx, ny, pitch, p, cfg, bc, part, st);
        default: return sweepT_div<2, 4>(in, out, nx, ny, pitch, p, cfg, bc, part, st);
    }
}

bool sweep2_supported(int nx, const SweepCfg& cfg) {
    return nx % WAVE_COLS == 0 && (cfg.variant == VAR_AUTO || cfg.variant == VAR_DPP);
}

hipError_t launch_sweep2(const double* in, double* out, int nx, int ny, int pitch, const Phys& p,
                         const SweepCfg& cfg, const int kind[4], double value, int part,
                         hipStream_t st) {
    Bc2 bc;
    for (int s = 0; s < 4; ++s) bc.kind[s] = kind[s];
    bc.value = value;
    switch (p.div_mode) {
        case 0: return sweep2_div<0>(in, out, nx, ny, pitch, p, cfg, bc, part, st);
        case 1: return sweep2_div<1>(in, out, nx, ny, pitch, p, cfg, bc, part, st);
        default: return sweep2_div<2>(in, out, nx, ny, pitch, p, cfg, bc, part, st);
    }
}


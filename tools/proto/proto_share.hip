// prototype: overlapped-strip fused sweep where the FOUR wavefronts of a workgroup own four adjacent
// 128-column strips and hand each other the one boundary column per time level through LDS (one barrier
// per row iteration) instead of each overlapping its neighbours by 2*TP columns: a workgroup loads 512
// columns and stores 512 - 2*TP of them (125 per wavefront at T = 6 instead of 116: -7 % arithmetic).
// Interior body only (no boundary rules), T = 6; compared bit for bit with the independent-strip kernel.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o proto_share proto_share.hip && ./proto_share
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#pragma clang fp contract(off)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int LPAD = 16;
struct Phys { double kdiff, mdt, vx, vy; };

__device__ __forceinline__ double cell(double c, double W, double E, double S, double N, const Phys& p) {
    const double tc = 2.0 * c;
    const double lx = (E - tc) + W;
    const double ly = (N - tc) + S;
    const double lap = lx + ly;
    const double o = c + p.kdiff * lap;
    const double gx = c - W, gy = c - S;
    const double adv = p.vx * gx + p.vy * gy;
    return o + p.mdt * adv;
}
// lane i <- lane i-1 / i+1; the lane without a source keeps `edge`
__device__ __forceinline__ double from_prev(double src, double edge) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(src), 0x138, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(src), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_next(double src, double edge) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(src), 0x130, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(src), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// SHARE = false: every wavefront an independent strip (the product kernel's geometry)
// SHARE = true : four wavefronts = one 512-column super-strip, boundary columns through LDS
template <int T, bool SHARE>
__global__ __launch_bounds__(256) void k_proto(const double* __restrict__ in, double* __restrict__ out, int nx, int ny,
                                               int pitch, int ry, int nstrips, int ntiles, Phys p) {
    constexpr int TP = 2 * ((T + 1) / 2);
    constexpr int W = SHARE ? 512 : 128;
    constexpr int STRIDE = W - 2 * TP;
    __shared__ double halo[2][T][4][2];  // [buffer][level][wavefront][0: its leftmost column, 1: its rightmost]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int strip, chunk, lc0;  // lc0 = this wavefront's first column inside its (super-)strip
    if (SHARE) {
        const int tile = blockIdx.x;
        if (tile >= ntiles) return;  // block-uniform
        strip = tile % nstrips, chunk = tile / nstrips, lc0 = 128 * wave;
    } else {
        const int tile = blockIdx.x * 4 + wave;
        if (tile >= ntiles) return;
        strip = tile % nstrips, chunk = tile / nstrips, lc0 = 0;
    }
    const int jb = 1 + chunk * ry, je = min(jb + ry - 1, ny);
    const int g0 = strip * STRIDE - TP + lc0;
    const int gx = g0 + 2 * lane;
    const ptrdiff_t xoff = LPAD + gx;
    const int lc = lc0 + 2 * lane;  // column inside the (super-)strip
    const bool out_lane = lc >= TP && lc < TP + STRIDE && gx < nx;
    auto load = [&](int j) {
        // columns beyond the padded row (ragged last super-strip) are clamped: their results are never stored
        const ptrdiff_t xo = min(xoff, static_cast<ptrdiff_t>(pitch - 2));
        return *reinterpret_cast<const double2*>(in + static_cast<ptrdiff_t>(j) * pitch + xo);
    };
    const int r_first = jb - (T - 1);
    const int niter = (je - jb + 1) + 2 * (T - 1);
    const int last_row = r_first + niter;
    double2 L0[6];
    double2 L[T][3];
#pragma unroll
    for (int q = 0; q < 6; ++q) L0[q] = load(min(r_first - 1 + q, last_row));
#pragma unroll
    for (int l = 0; l < T; ++l)
#pragma unroll
        for (int q = 0; q < 3; ++q) L[l][q] = make_double2(0.0, 0.0);
    // which LDS slot this lane reads (lanes < 32: the left neighbour's rightmost column, else the right
    // neighbour's leftmost) and writes (lane 0: own leftmost, lane 63: own rightmost)
    const int rd_wave = lane < 32 ? max(wave - 1, 0) : min(wave + 1, 3);
    const int rd_side = lane < 32 ? 1 : 0;
    const bool writer = lane == 0 || lane == 63;
    const int wr_side = lane == 0 ? 0 : 1;
    if (SHARE) {
        if (threadIdx.x < 2 * T * 4 * 2) (&halo[0][0][0][0])[threadIdx.x] = 0.0;
        __syncthreads();
        if (writer) halo[0][0][wave][wr_side] = lane == 0 ? L0[1].x : L0[1].y;  // level 0, centre row of iteration 0
        __syncthreads();
    }
    int buf = 0;
    for (int k0 = 0; k0 < niter; k0 += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int r = r_first + k0 + u;
            double edge[T];
            if (SHARE) {
#pragma unroll
                for (int l = 0; l < T; ++l) edge[l] = halo[buf][l][rd_wave][rd_side];
            }
#pragma unroll
            for (int l = 1; l <= T; ++l) {
                const int rho = r - l + 1;
                const double2 s = (l == 1) ? L0[u % 6] : L[l - 1][(u + 1) % 3];
                const double2 c = (l == 1) ? L0[(u + 1) % 6] : L[l - 1][(u + 2) % 3];
                const double2 n = (l == 1) ? L0[(u + 2) % 6] : L[l - 1][u % 3];
                const double e = SHARE ? edge[l - 1] : 0.0;
                const double Wx = from_prev(c.y, e);
                const double Ey = from_next(c.x, e);
                double2 o;
                o.x = cell(c.x, Wx, c.y, s.x, n.x, p);
                o.y = cell(c.y, c.x, Ey, s.y, n.y, p);
                if (l < T) {
                    L[l][u % 3] = o;
                    // next iteration, level l+1 needs this row of level l from the neighbours
                    if (SHARE && writer) halo[buf ^ 1][l][wave][wr_side] = lane == 0 ? o.x : o.y;
                } else if (rho >= jb && rho <= je && out_lane) {
                    double* dst = out + static_cast<ptrdiff_t>(rho) * pitch + xoff;
                    if (gx + 1 < nx) *reinterpret_cast<double2*>(dst) = o;
                    else dst[0] = o.x;
                }
            }
            // level 0: the centre row of the NEXT iteration is the `n` row of this one
            if (SHARE && writer) halo[buf ^ 1][0][wave][wr_side] = lane == 0 ? L0[(u + 2) % 6].x : L0[(u + 2) % 6].y;
            L0[u % 6] = load(min(r + 5, last_row));
            if (SHARE) {
                __syncthreads();
                buf ^= 1;
            }
        }
    }
}

template <bool SHARE>
float run(const double* a, double* b, int nx, int ny, int pitch, int ry, Phys p, int reps) {
    constexpr int T = 6;
    constexpr int STRIDE = (SHARE ? 512 : 128) - 12;
    const int nstrips = (nx + STRIDE - 1) / STRIDE, nchunks = (ny + ry - 1) / ry, ntiles = nstrips * nchunks;
    const int nblocks = SHARE ? ntiles : (ntiles + 3) / 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 300; ++r) hipLaunchKernelGGL((k_proto<T, SHARE>), dim3(nblocks), dim3(256), 0, 0, a, b, nx, ny, pitch, ry, nstrips, ntiles, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_proto<T, SHARE>), dim3(nblocks), dim3(256), 0, 0, a, b, nx, ny, pitch, ry, nstrips, ntiles, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int nx = argc > 1 ? atoi(argv[1]) : 16384, ny = argc > 2 ? atoi(argv[2]) : 16384;
    const int pitch = LPAD + ((nx + 1 + 511) / 512) * 512 + 512;
    const size_t elems = static_cast<size_t>(ny + 12) * pitch;
    std::vector<double> h(elems);
    srand(1);
    for (auto& v : h) v = rand() / double(RAND_MAX);
    double *a, *b0, *b1;
    CK(hipMalloc(&a, elems * 8)); CK(hipMalloc(&b0, elems * 8)); CK(hipMalloc(&b1, elems * 8));
    CK(hipMemcpy(a, h.data(), elems * 8, hipMemcpyHostToDevice));
    CK(hipMemset(b0, 0, elems * 8)); CK(hipMemset(b1, 0, elems * 8));
    Phys p{0.1 * 0.05, -0.1, 0.5, 0.25};
    const double* va = a + 5 * static_cast<size_t>(pitch);
    double* v0 = b0 + 5 * static_cast<size_t>(pitch);
    double* v1 = b1 + 5 * static_cast<size_t>(pitch);
    for (int ry : {62, 92, 122, 158}) {
        const float t0 = run<false>(va, v0, nx, ny, pitch, ry, p, 200);
        const float t1 = run<true>(va, v1, nx, ny, pitch, ry, p, 200);
        printf("%dx%d ry=%d  independent strips: %.4f ms (%.0f Mcell/s)   4-wave super-strips: %.4f ms (%.0f Mcell/s)  %+.1f %%\n", nx,
               ny, ry, t0, double(nx) * ny * 6 / t0 / 1e3, t1, double(nx) * ny * 6 / t1 / 1e3, (t0 / t1 - 1) * 100);
    }
    std::vector<double> r0(elems), r1(elems);
    CK(hipMemcpy(r0.data(), b0, elems * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r1.data(), b1, elems * 8, hipMemcpyDeviceToHost));
    size_t bad = 0, checked = 0;
    for (int j = 12; j < ny - 12; ++j)  // interior away from the edges (the prototype has no boundary rules)
        for (int i = 12; i < nx - 12; ++i) {
            const size_t o = static_cast<size_t>(j + 5) * pitch + LPAD + i;
            bad += r0[o] != r1[o];
            ++checked;
        }
    printf("independent vs shared: %zu mismatches in %zu cells away from the edges\n", bad, checked);
    return bad != 0;
}

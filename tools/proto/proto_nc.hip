// prototype: overlapped-strip fused sweep with NC = 2, 3 or 4 columns per lane (interior body only), T = 5, 6, 7
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
__device__ __forceinline__ double shift_from_prev(double src) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), 0x138, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shift_from_next(double src) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), 0x130, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int NC> struct Row { double v[NC]; };

template <int NC, int T>
__global__ __launch_bounds__(256) void k_proto(const double* __restrict__ in, double* __restrict__ out, int nx, int ny,
                                               int pitch, int ry, int nstrips, int ntiles, Phys p) {
    constexpr int W = 64 * NC;
    constexpr int TP = 2 * ((T + 1) / 2);
    constexpr int STRIDE = W - 2 * TP;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;
    const int strip = tile % nstrips, chunk = tile / nstrips;
    const int jb = 1 + chunk * ry, je = min(jb + ry - 1, ny);
    const int g0 = strip * STRIDE - TP;
    const int gx = g0 + NC * lane;
    const ptrdiff_t xoff = LPAD + gx;
    auto load = [&](int j) {
        Row<NC> r;
        const double* src = in + static_cast<ptrdiff_t>(j) * pitch + xoff;
        if (NC % 2 == 0) {
#pragma unroll
            for (int c = 0; c + 1 < NC; c += 2) {
                const double2 t = *reinterpret_cast<const double2*>(src + c);
                r.v[c] = t.x;
                r.v[c + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) r.v[c] = src[c];
        }
        return r;
    };
    const int r_first = jb - (T - 1);
    const int niter = (je - jb + 1) + 2 * (T - 1);
    const int last_row = r_first + niter;
    Row<NC> L0[6];
    Row<NC> L[T][3];
#pragma unroll
    for (int q = 0; q < 6; ++q) L0[q] = load(min(r_first - 1 + q, last_row));
#pragma unroll
    for (int l = 0; l < T; ++l)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int c = 0; c < NC; ++c) L[l][q].v[c] = 0.0;
    for (int k0 = 0; k0 < niter; k0 += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int r = r_first + k0 + u;
#pragma unroll
            for (int l = 1; l <= T; ++l) {
                const int rho = r - l + 1;
                const Row<NC> s = (l == 1) ? L0[u % 6] : L[l - 1][(u + 1) % 3];
                const Row<NC> c = (l == 1) ? L0[(u + 1) % 6] : L[l - 1][(u + 2) % 3];
                const Row<NC> n = (l == 1) ? L0[(u + 2) % 6] : L[l - 1][u % 3];
                Row<NC> o;
                const double Wx = shift_from_prev(c.v[NC - 1]);
                const double Ey = shift_from_next(c.v[0]);
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    const double w = q == 0 ? Wx : c.v[q - 1];
                    const double e = q == NC - 1 ? Ey : c.v[q + 1];
                    o.v[q] = cell(c.v[q], w, e, s.v[q], n.v[q], p);
                }
                if (l < T) {
                    L[l][u % 3] = o;
                } else if (rho >= jb && rho <= je) {
                    double* dst = out + static_cast<ptrdiff_t>(rho) * pitch + xoff;
                    if (NC % 2 == 0) {
#pragma unroll
                        for (int q = 0; q + 1 < NC; q += 2) {
                            const int lc = NC * lane + q;
                            if (lc >= TP && lc < TP + STRIDE && gx + q < nx) {
                                if (gx + q + 1 < nx) *reinterpret_cast<double2*>(dst + q) = make_double2(o.v[q], o.v[q + 1]);
                                else dst[q] = o.v[q];
                            }
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < NC; ++q) {
                            const int lc = NC * lane + q;
                            if (lc >= TP && lc < TP + STRIDE && gx + q < nx) dst[q] = o.v[q];
                        }
                    }
                }
            }
            L0[u % 6] = load(min(r + 5, last_row));
        }
    }
}

template <int NC, int T>
float run(const double* a, double* b, int nx, int ny, int pitch, int ry, Phys p, int reps) {
    constexpr int STRIDE = 64 * NC - 2 * (2 * ((T + 1) / 2));
    const int nstrips = (nx + STRIDE - 1) / STRIDE, nchunks = (ny + ry - 1) / ry, ntiles = nstrips * nchunks;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 300; ++r) hipLaunchKernelGGL((k_proto<NC, T>), dim3((ntiles + 3) / 4), dim3(256), 0, 0, a, b, nx, ny, pitch, ry, nstrips, ntiles, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_proto<NC, T>), dim3((ntiles + 3) / 4), dim3(256), 0, 0, a, b, nx, ny, pitch, ry, nstrips, ntiles, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int nx = argc > 1 ? atoi(argv[1]) : 16384, ny = argc > 2 ? atoi(argv[2]) : 16384;
    const int pitch = LPAD + ((nx + 1 + 255) / 256) * 256 + 256;
    const size_t elems = static_cast<size_t>(ny + 24) * pitch;
    std::vector<double> h(elems);
    srand(1);
    for (auto& v : h) v = rand() / double(RAND_MAX);
    double *a, *b2, *b4;
    CK(hipMalloc(&a, elems * 8)); CK(hipMalloc(&b2, elems * 8)); CK(hipMalloc(&b4, elems * 8));
    CK(hipMemcpy(a, h.data(), elems * 8, hipMemcpyHostToDevice));
    CK(hipMemset(b2, 0, elems * 8)); CK(hipMemset(b4, 0, elems * 8));
    Phys p{0.1 * 0.05, -0.1, 0.5, 0.25};
    const double* va = a + 8 * static_cast<size_t>(pitch);
    double* v2 = b2 + 8 * static_cast<size_t>(pitch);
    double* v4 = b4 + 8 * static_cast<size_t>(pitch);
    double *b3;
    CK(hipMalloc(&b3, elems * 8)); CK(hipMemset(b3, 0, elems * 8));
    double* v3 = b3 + 8 * static_cast<size_t>(pitch);
    for (int ry : {92, 122, 158, 182, 206}) {
        const float t26 = run<2, 6>(va, v2, nx, ny, pitch, ry + 2, p, 100);
        const float t27 = run<2, 7>(va, v2, nx, ny, pitch, ry, p, 100);
        const float t36 = run<3, 6>(va, v3, nx, ny, pitch, ry + 2, p, 100);
        const float t35 = run<3, 5>(va, v3, nx, ny, pitch, ry + 4, p, 100);
        const float t46 = run<4, 6>(va, v4, nx, ny, pitch, ry + 2, p, 100);
        auto rate = [&](float t, int T) { return double(nx) * ny * T / t / 1e3; };
        printf("%dx%d ry~%d  NC2T6 %.0f  NC2T7 %.0f  NC3T6 %.0f  NC3T5 %.0f  NC4T6 %.0f  Mcell/s\n", nx, ny, ry, rate(t26, 6), rate(t27, 7),
               rate(t36, 6), rate(t35, 5), rate(t46, 6));
    }
    run<2, 6>(va, v2, nx, ny, pitch, 122, p, 1); run<3, 6>(va, v3, nx, ny, pitch, 122, p, 1); run<4, 6>(va, v4, nx, ny, pitch, 122, p, 1);
    CK(hipDeviceSynchronize());
    std::vector<double> r2(elems), r4(elems);
    CK(hipMemcpy(r2.data(), b2, elems * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r4.data(), b4, elems * 8, hipMemcpyDeviceToHost));
    std::vector<double> r3(elems);
    CK(hipMemcpy(r3.data(), b3, elems * 8, hipMemcpyDeviceToHost));
    size_t bad3 = 0;
    for (int j = 12; j < ny - 12; ++j)
        for (int i = 12; i < nx - 12; ++i) {
            const size_t o = static_cast<size_t>(j + 8) * pitch + LPAD + i;
            bad3 += r2[o] != r3[o];
        }
    printf("NC=2 vs NC=3 mismatches away from the edges: %zu\n", bad3);
    size_t bad = 0;
    // interior away from the edges (the prototype has no boundary rules)
    for (int j = 12; j < ny - 12; ++j)
        for (int i = 12; i < nx - 12; ++i) {
            const size_t o = static_cast<size_t>(j + 8) * pitch + LPAD + i;
            bad += r2[o] != r4[o];
        }
    printf("NC=2 vs NC=4 mismatches away from the edges: %zu\n", bad);
    return 0;
}

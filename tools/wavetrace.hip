// tools/wavetrace.hip — where does a fused-sweep launch spend its time on small tiles?
// Compiles the product kernels with -DCSIM_TRACE (every wavefront of k_sweepO_dpp records its
// start/end wall clock and its XCC / SE / CU / SIMD) and prints the launch's timeline: how long
// dispatch takes, how many wavefronts each SIMD received, spread of wave durations, idle tail.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DCSIM_TRACE \
//              -I../include -I../climate-sim-mpi-cpp_amd/csrc -o wavetrace wavetrace.hip
// Usage: ./wavetrace NX NY T RY [reps] [boundary kind: 0 dirichlet, 1 neumann, 2 periodic, 3 none]
#include "../climate-sim-mpi-cpp_amd/csrc/kernels.hip"

#include <cstdlib>
#include <map>
#include <string>
#include <vector>

namespace csim {
int fail(int code, const std::string& msg) {
    std::fprintf(stderr, "error %d: %s\n", code, msg.c_str());
    return code;
}
}  // namespace csim

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e = (x);                                                                    \
        if (e != hipSuccess) {                                                                 \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);  \
            std::exit(1);                                                                      \
        }                                                                                      \
    } while (0)

int main(int argc, char** argv) {
    using namespace csim;
    const int nx = argc > 1 ? std::atoi(argv[1]) : 4096;
    const int ny = argc > 2 ? std::atoi(argv[2]) : 8192;
    const int T = argc > 3 ? std::atoi(argv[3]) : 6;
    const int ry = argc > 4 ? std::atoi(argv[4]) : 64;
    const int reps = argc > 5 ? std::atoi(argv[5]) : 5;
    const int pitch = pitch_for(nx);
    const size_t elems = static_cast<size_t>(ny + 2 + 2 * GHOST_EXTRA) * pitch;
    double *a, *b;
    CK(hipMalloc(&a, elems * sizeof(double)));
    CK(hipMalloc(&b, elems * sizeof(double)));
    CK(hipMemset(a, 0, elems * sizeof(double)));
    CK(hipMemset(b, 0, elems * sizeof(double)));
    double* va = a + static_cast<size_t>(GHOST_EXTRA) * pitch;
    double* vb = b + static_cast<size_t>(GHOST_EXTRA) * pitch;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    CK(launch_gaussian(va, nx, ny, pitch, 0, 0, nx, ny, 1.0, 1.0, 1.0, 0.05, 0.5, 0.5, st));

    Phys p{};
    p.kdiff = 0.1 * 0.05;
    p.mdt = -0.1;
    p.vx = 0.5;
    p.vy = 0.25;
    p.dx = p.dy = p.dx2 = p.dy2 = p.rdx = p.rdy = p.rdx2 = p.rdy2 = 1.0;
    p.div_mode = 0;
    SweepCfg cfg;
    cfg.rows_per_chunk = ry;
    const int kd = argc > 6 ? std::atoi(argv[6]) : CSIM_BC_DIRICHLET;  // 3 = no physical edge (no edge-body waves)
    const int kind[4] = {kd, kd, kd, kd};

    const int stride = 128 - 4 * ((T + 1) / 2);
    const int nstrips = (nx + stride - 1) / stride;
    const int nchunks = (ny + ry - 1) / ry;
    const int nblocks = (nstrips * nchunks + 3) / 4;
    // (trace slots: a launch with a tail region of half-height chunks has more tiles than strips x chunks)
    const size_t nwaves = static_cast<size_t>(nblocks) * 4 * 2 + 64;
    unsigned long long* d_tr;
    CK(hipMalloc(&d_tr, nwaves * 3 * sizeof(unsigned long long)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_wave_trace), &d_tr, sizeof(d_tr)));
    std::vector<unsigned long long> tr(nwaves * 3);

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // leave the idle clocks first: ~0.4 s of back-to-back launches
    for (int r = 0; r < 4000; ++r) {
        CK(launch_sweepO(va, vb, nx, ny, pitch, p, cfg, kind, 0.0, T, 0, st));
        std::swap(va, vb);
        if (r % 100 == 99) {
            CK(hipStreamSynchronize(st));
            float ms = 0;
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            if (r >= 199) {
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms > 400.f) break;
            } else if (r == 99) {
                CK(hipEventRecord(e0, st));
            }
        }
    }
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(d_tr, 0, nwaves * 3 * sizeof(unsigned long long), st));
        CK(hipEventRecord(e0, st));
        CK(launch_sweepO(va, vb, nx, ny, pitch, p, cfg, kind, 0.0, T, 0, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::swap(va, vb);
        if (r + 1 < reps) continue;
        CK(hipMemcpy(tr.data(), d_tr, tr.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long tmin = ~0ull, tmax = 0, smax = 0;
        for (size_t w = 0; w < nwaves; ++w) {
            if (!tr[3 * w]) continue;
            tmin = std::min(tmin, tr[3 * w]);
            tmax = std::max(tmax, tr[3 * w + 1]);
            smax = std::max(smax, tr[3 * w]);
        }
        const double us = 0.01;  // 100 MHz ticks
        std::printf("%dx%d T=%d ry=%d: %d blocks (%zu waves, %.2f rounds of 4096), event time %.1f us, "
                    "first start -> last end %.1f us, last start at %.1f us\n",
                    nx, ny, T, ry, nblocks, nwaves, nwaves / 4096.0, ms * 1e3, (tmax - tmin) * us, (smax - tmin) * us);
        // real (non-idle) waves: duration > 2 us
        std::vector<double> dur, start, end;
        std::map<unsigned long long, int> per_simd;
        std::map<unsigned long long, double> simd_busy_end;
        for (size_t w = 0; w < nwaves; ++w) {
            const double d = (tr[3 * w + 1] - tr[3 * w]) * us;
            if (!tr[3 * w] || d < 2.0) continue;
            dur.push_back(d);
            start.push_back((tr[3 * w] - tmin) * us);
            end.push_back((tr[3 * w + 1] - tmin) * us);
            const unsigned long long id = tr[3 * w + 2];
            const unsigned hw = static_cast<unsigned>(id), xcc = static_cast<unsigned>(id >> 32) & 0xf;
            const unsigned long long key = (static_cast<unsigned long long>(xcc) << 16) | (((hw >> 13) & 7) << 12) |
                                           (((hw >> 12) & 1) << 11) | (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3);
            per_simd[key] += 1;
            simd_busy_end[key] = std::max(simd_busy_end[key], (tr[3 * w + 1] - tmin) * us);
        }
        auto pct = [](std::vector<double> v, double q) {
            std::sort(v.begin(), v.end());
            return v.empty() ? 0.0 : v[static_cast<size_t>(q * (v.size() - 1))];
        };
        std::printf("  working waves %zu; duration us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f\n", dur.size(),
                    pct(dur, 0), pct(dur, 0.1), pct(dur, 0.5), pct(dur, 0.9), pct(dur, 1));
        std::printf("  start us: p10 %.1f p50 %.1f p90 %.1f max %.1f;  end us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f\n",
                    pct(start, 0.1), pct(start, 0.5), pct(start, 0.9), pct(start, 1), pct(end, 0), pct(end, 0.1),
                    pct(end, 0.5), pct(end, 0.9), pct(end, 1));
        std::map<int, int> hist;
        for (auto& kv : per_simd) hist[kv.second] += 1;
        std::printf("  SIMDs used %zu; waves per SIMD histogram:", per_simd.size());
        for (auto& kv : hist) std::printf(" %d:%d", kv.first, kv.second);
        std::vector<double> sb;
        for (auto& kv : simd_busy_end) sb.push_back(kv.second);
        std::printf("\n  per-SIMD last end us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f\n", pct(sb, 0), pct(sb, 0.1),
                    pct(sb, 0.5), pct(sb, 0.9), pct(sb, 1));
        // edge-body wavefronts (first/last strip, chunks next to a physical bottom/top edge) vs the rest
        {
            std::vector<double> de, di, ee, ei;
            const int TPv = 2 * ((T + 1) / 2);
            for (size_t w = 0; w < nwaves; ++w) {
                const double d = (tr[3 * w + 1] - tr[3 * w]) * us;
                if (!tr[3 * w] || d < 2.0) continue;
                const int bidx = static_cast<int>(w / 4), wv = static_cast<int>(w % 4);
                int lin = bidx;
                if (nblocks >= 16) {
                    const int per = nblocks >> 3, rem = nblocks & 7, xcd = bidx & 7, q = bidx >> 3;
                    lin = xcd < rem ? xcd * (per + 1) + q : rem * (per + 1) + (xcd - rem) * per + q;
                }
                const int tile = lin * 4 + wv;
                if (tile >= nstrips * nchunks) continue;
                const int strip = tile % nstrips, chunk = tile / nstrips;
                const int jb = chunk * ry + 1, je = std::min(jb + ry - 1, ny), g0 = strip * stride - TPv;
                const bool edge = strip == 0 || g0 + 128 > nx || jb - (T - 1) < 1 || je + (T - 1) > ny;
                (edge ? de : di).push_back(d);
                (edge ? ee : ei).push_back((tr[3 * w + 1] - tmin) * us);
            }
            std::printf("  edge-body waves %zu: duration p50 %.1f p90 %.1f max %.1f, end p50 %.1f max %.1f | interior waves %zu: "
                        "duration p50 %.1f p90 %.1f max %.1f, end p50 %.1f p99 %.1f max %.1f\n",
                        de.size(), pct(de, 0.5), pct(de, 0.9), pct(de, 1), pct(ee, 0.5), pct(ee, 1), di.size(), pct(di, 0.5),
                        pct(di, 0.9), pct(di, 1), pct(ei, 0.5), pct(ei, 0.99), pct(ei, 1));
        }
        // Who finishes first on a SIMD?  Per SIMD with exactly four working waves: their block LAYERS (block id / 256: the
        // k-th block a CU received, if the dispatcher deals blocks round-robin) in the order they END, and in the order
        // they START.  "0123" = the oldest ends first.
        {
            struct W { double start, end; int layer; };
            std::map<unsigned long long, std::vector<W>> by_simd;
            for (size_t w = 0; w < nwaves; ++w) {
                const double d = (tr[3 * w + 1] - tr[3 * w]) * us;
                if (!tr[3 * w] || d < 2.0) continue;
                const unsigned long long id = tr[3 * w + 2];
                const unsigned hw = static_cast<unsigned>(id), xcc = static_cast<unsigned>(id >> 32) & 0xf;
                const unsigned long long key = (static_cast<unsigned long long>(xcc) << 16) | (((hw >> 13) & 7) << 12) |
                                               (((hw >> 12) & 1) << 11) | (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3);
                by_simd[key].push_back({(tr[3 * w] - tmin) * us, (tr[3 * w + 1] - tmin) * us, static_cast<int>(w / 4) >> 8});
            }
            std::map<std::string, int> by_end, by_start;
            for (auto& kv : by_simd) {
                auto v = kv.second;
                if (v.size() != 4) continue;
                std::sort(v.begin(), v.end(), [](const W& a, const W& b) { return a.end < b.end; });
                std::string e, st2;
                for (auto& x : v) e += static_cast<char>('0' + std::min(x.layer, 9));
                std::sort(v.begin(), v.end(), [](const W& a, const W& b) { return a.start < b.start; });
                for (auto& x : v) st2 += static_cast<char>('0' + std::min(x.layer, 9));
                by_end[e] += 1;
                by_start[st2] += 1;
            }
            auto top = [](std::map<std::string, int>& m) {
                std::vector<std::pair<int, std::string>> v;
                for (auto& kv : m) v.push_back({kv.second, kv.first});
                std::sort(v.rbegin(), v.rend());
                for (size_t i = 0; i < v.size() && i < 8; ++i) std::printf(" %s:%d", v[i].second.c_str(), v[i].first);
            };
            std::printf("  block layers of a SIMD's four waves in END order:");
            top(by_end);
            std::printf("\n  ... in START order:");
            top(by_start);
            std::printf("\n");
        }
        // concurrency over time: how many working waves are alive in each 10 % slice of the launch
        const double span = (tmax - tmin) * us;
        std::printf("  alive waves at 5%%..95%% of the span:");
        for (int k = 1; k < 20; k += 2) {
            const double t = span * k / 20.0;
            int alive = 0;
            for (size_t i = 0; i < dur.size(); ++i) alive += start[i] <= t && end[i] > t;
            std::printf(" %d", alive);
        }
        std::printf("\n");
    }
    return 0;
}

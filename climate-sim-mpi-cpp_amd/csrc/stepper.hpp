// stepper.hpp — what the translation units behind include/csim.h share: the stepper handle and the helpers
// that cross file boundaries.  Layout of the host side of the engine (kernels live in kernels.hip):
//   api.cpp      library / device, safe_dt, decomposition, exchange plan, Field mirror, reference-granularity operators
//   stepper.cpp  the stepper handle: create / destroy, communicator, upload / download / snapshots, external halo
//                transport, reductions, options, timers
//   passes.cpp   what a csim_stepper_run enqueues: the RCCL exchange group, single-step and fused passes (the
//                exchange schedules), the chunk-height trial, the run loop
//   planner.cpp  the pass plan of a run (pure host arithmetic)
//   profile.cpp  HIP-event brackets around passes (options "profile", csim_stepper_kernel_time / _comm_time)
#pragma once
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "internal.hpp"


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
    // Relay (bulk-first passes): two streams of equal priority swap roles every pass — the one that carried a pass's
    // exchange and frame launch also takes the NEXT pass's bulk launch — so `tail` names the stream on which the current
    // field state is ordered.  Every entry point that is not a relay pass settles it back onto s_comp first (settle()).
    hipStream_t s_relay[2]{nullptr, nullptr};
    hipStream_t tail = nullptr;
    hipEvent_t ev_tail = nullptr;
    // relay hand-offs between the two streams of THIS device (kernel boundaries carry the agent-scope release / acquire):
    // events without the system-scope fence a default event performs when it is recorded (option "relay_events")
    hipEvent_t ev_relay_ready = nullptr, ev_relay_bulk = nullptr;
    int relay_events = 0;  // 1: the relay records ev_relay_* instead of ev_ready / ev_edge2
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
    size_t cap2[8]{0, 0, 0, 0, 0, 0, 0, 0};  // staging capacity (faces of depth csim::MAX_FUSE)
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
    int diffusion_only_active = 0;  // read-only: the last run had v == 0 and swept with the advection term left out of the screened body
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
    csim::SweepCfg cfg;
    int overlap = 5;        // 0: exchange serial; 1: frame launch, then bulk launch hiding the NEXT pass's exchange;
                            // 3: frame and bulk in ONE launch (needs signal memory, else as 1);
                            // 4: bulk launch first, hiding THIS pass's exchange, then the frame (pass_fused_bulk_first);
                            // 5 (default): as 4 (until round 3: 4 on runs of fewer than 16 passes, 3 otherwise)
    bool phys_ring_filled = false;  // several ranks: the ghost lines of the PHYSICAL sides have been filled in both buffers since the
                                    // last upload / initialisation (bulk-first passes read constant Dirichlet ghosts before their own fill)
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
    int tuned_T[csim::MAX_FUSE + 1]{};  // chunk height found by the trial for passes of that depth (0 = not tried: cfg.tuned_rows re-snapped)
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
    static constexpr int PROF_COMM = csim::MAX_FUSE + 1;  // comm-stream chain of a pass: pack, RCCL group, unpack, ghost fill
    double prof_ms[csim::MAX_FUSE + 2]{};     // indexed by time steps per launch (1..csim::MAX_FUSE), [PROF_COMM]
    long prof_launches[csim::MAX_FUSE + 2]{};
    size_t bytes() const { return sizeof(double) * static_cast<size_t>(ny + 2 + 2 * csim::GHOST_EXTRA) * pitch; }
    // whole-allocation pointer of a view
    double* base(double* view) const { return view - static_cast<size_t>(csim::GHOST_EXTRA) * pitch; }
};

namespace csim {

// api.cpp
int finish_partials(const double* scratch_dev, int nblocks, int kind, double out[2], hipStream_t st);
int reduce_blocks(int nrows);
int upload_2d(double* d, int nx, int ny, int pitch, const double* host);
int download_2d(const double* d, int nx, int ny, int pitch, double* host);
int download_interior_2d(const double* d, int nx, int ny, int pitch, double* host);
void neighbours8(const csim_decomp& dec, int nbr8[8]);
size_t face_doubles(int d, int H, int nx, int ny);
bool valid_bc(const int bc[4]);

// stepper.cpp
int settle(csim_stepper* s);  // the field state back onto the compute stream (see csim_stepper::tail)
bool depth_ok(const csim_stepper* s, int depth);
int fused_depth(const csim_stepper* s);

// passes.cpp
int post_plan(csim_stepper* s, int depth, hipStream_t st);
int refresh_halos(csim_stepper* s);
GhostArgs ghost_args(const csim_stepper* s);
bool ring_is_static(const csim_stepper* s);
int pass_single(csim_stepper* s, const Phys& p, const GhostArgs& g);
int pass_fused(csim_stepper* s, const Phys& p, int T, int next_T, bool final_pass = false);
int tune_rows(csim_stepper* s, const Phys& p, int T, bool preferred_depth = true);

// planner.cpp
// The plan is `lead` passes of depth `lead_depth` followed by the passes listed in `tail` (a run of 10^9
// steps must not materialise 10^8 entries).
struct PassPlan {
    long lead = 0;
    int lead_depth = 1;
    std::vector<int> tail;
    long size() const { return lead + static_cast<long>(tail.size()); }
    int at(long k) const { return k < lead ? lead_depth : tail[static_cast<size_t>(k - lead)]; }
};
void plan_passes(int K, int cap, bool balanced, long tile_cells, PassPlan& plan, bool still = false);

// profile.cpp
int prof_fold(csim_stepper* s);
int prof_start(csim_stepper* s, int kind, hipStream_t st, long* slot);
int prof_stop(csim_stepper* s, long slot, hipStream_t st);
int prof_close(csim_stepper* s);
int prof_begin(csim_stepper* s, int steps, hipStream_t st = nullptr);
int prof_end(csim_stepper* s, hipStream_t st = nullptr);

}  // namespace csim

#define CSIM_SETTLE(s_)            \
    do {                           \
        int rc_ = ::csim::settle(s_);      \
        if (rc_) return rc_;       \
    } while (0)

// planner.cpp — the pass plan of a run: how csim_stepper_run(nsteps) splits K reference steps (src/main.cpp:93, one
// step per loop iteration) into HBM passes of 1..7 time steps.  Pure host arithmetic, identical on every rank.
#include "stepper.hpp"

using namespace csim;


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
// diffusion-only flavour (HBM-bound): 16384^2 2.07 / 1.83 / 1.54 / 1.32 / 0.99 M cell updates per us at depths 7 .. 3,
// 4096^2 1.59 / 1.46 / 1.29 / 1.11 / 0.83 (profiles/r03_diffusion_only_depth.jsonl), relative to T = 7
static const double STEP_COST_STILL[MAX_FUSE + 1] = {0.0, 6.0, 3.2, 2.0, 1.5, 1.28, 1.11, 1.0};
static const double* step_cost_table(long tile_cells, bool still) {
    if (still) return STEP_COST_STILL;
    if (tile_cells >= BIG_TILE_CELLS) return STEP_COST_BIG;
    if (tile_cells <= 0 || tile_cells >= 50000000L) return STEP_COST_MID;  // (0 = size unknown)
    return tile_cells >= SMALL_TILE_CELLS ? STEP_COST_MIDSMALL : STEP_COST_SMALL;
}
static const double PASS_COST = 0.1;   // launch and inter-kernel gap, in time steps of the preferred depth

namespace csim {

void plan_passes(int K, int cap, bool balanced, long tile_cells, PassPlan& plan, bool still) {
    const double* step_cost = step_cost_table(tile_cells, still);
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
    const int pref = std::min(cap, pref_fuse(tile_cells, still));
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

}  // namespace csim

extern "C" {

// the pass schedule as pure host arithmetic (no GPU): what csim_stepper_run(nsteps) will launch on a
// decomposition whose smallest tile is `smallest_tile` cells deep, with option "fuse" = `fuse`
int csim_pass_schedule(int nsteps, int smallest_tile, long tile_cells, int fuse, int* depths, int max_depths,
                       long* npasses) {
    return csim_pass_schedule_for(nsteps, smallest_tile, tile_cells, fuse, 0, depths, max_depths, npasses);
}

int csim_pass_schedule_for(int nsteps, int smallest_tile, long tile_cells, int fuse, int diffusion_only, int* depths,
                           int max_depths, long* npasses) {
    CSIM_REQUIRE(npasses && nsteps >= 0 && smallest_tile >= 1, "bad argument");
    const bool still = diffusion_only != 0;
    CSIM_REQUIRE(fuse >= -1 && fuse <= MAX_FUSE, "fuse must be -1 (auto) or 0..7");
    CSIM_REQUIRE(max_depths == 0 || depths, "depths is null");
    const int fuse_cap = std::max(1, std::min(MAX_FUSE, smallest_tile));
    const int depth = std::min(fuse < 0 ? pref_fuse(tile_cells, still) : fuse, fuse_cap);
    const int cap = depth < 2 ? 1 : fuse < 0 ? std::min(MAX_FUSE, fuse_cap) : depth;
    PassPlan plan;
    plan_passes(nsteps, cap, fuse >= 0, tile_cells, plan, still);
    *npasses = plan.size();
    for (long k = 0; k < plan.size() && k < max_depths; ++k) depths[k] = plan.at(k);
    return CSIM_OK;
}

}  // extern "C"

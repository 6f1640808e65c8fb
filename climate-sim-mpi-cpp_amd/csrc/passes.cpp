// passes.cpp — what csim_stepper_run enqueues (reference src/main.cpp:101-109 per step): the grouped RCCL exchange
// (src/halo.cpp:28-46), the single-step pass, the fused passes with their exchange schedules, the chunk-height
// trial and the run loop.
#include "stepper.hpp"

using namespace csim;

static int post_exchange2(csim_stepper* s, int H, hipStream_t st) { return post_plan(s, H, st); }
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
    // tile (profiles/r03_timeline_torus20_before.txt, _after.txt).
    hipStream_t X, Y;
    if (!s->relay || !s->s_relay[0]) {  // round 2's form: bulk and frame on the compute stream, the chain on the comm stream
        CSIM_SETTLE(s);
        X = s->s_comp;
        Y = s->s_comm;
    } else {
        if (s->tail != s->s_relay[0] && s->tail != s->s_relay[1]) {  // first relay pass since something else ran: hand over
            CSIM_HIP(hipEventRecord(s->ev_tail, s->tail ? s->tail : s->s_comp));
            CSIM_HIP(hipStreamWaitEvent(s->s_relay[0], s->ev_tail, 0));
            s->tail = s->s_relay[0];
        }
        X = s->tail;
        Y = X == s->s_relay[0] ? s->s_relay[1] : s->s_relay[0];
    }
    const bool relay = X != s->s_comp;
    const bool light = relay && s->relay_events;
    hipEvent_t ev_state = light ? s->ev_relay_ready : s->ev_ready, ev_bulk = light ? s->ev_relay_bulk : s->ev_edge2;
    // everything enqueued so far on X produced `cur` (and the partner buffer's ring)
    CSIM_HIP(hipEventRecord(ev_state, X));
    CSIM_HIP(hipStreamWaitEvent(Y, ev_state, 0));
    int rc = prof_begin(s, T, X);
    if (rc) return rc;
    // the bulk goes out first: the GPU starts on it while the host is still enqueuing the exchange
    CSIM_HIP(launch_fused(s, p, kind, T, 2, X));  // nothing to launch on tiles that are all frame
    if (relay) CSIM_HIP(hipEventRecord(ev_bulk, X));  // the bulk's end, for whatever follows the frame on Y
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
    if (!relay) {
        CSIM_HIP(hipEventRecord(s->ev_recv2, Y));
        CSIM_HIP(hipStreamWaitEvent(X, s->ev_recv2, 0));
        F = X;
    }
    CSIM_HIP(launch_fused(s, p, kind, T, 1, F, final_pass));
    rc = prof_end(s, F);
    if (rc) return rc;
    if (relay) {
        CSIM_HIP(hipStreamWaitEvent(Y, ev_bulk, 0));  // the field is complete on Y once the bulk is done too
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

namespace csim {

// one grouped RCCL exchange (replaces the <= 8 MPI requests + MPI_Waitall of reference src/halo.cpp:28-46):
// depth 1 = the staged edge lines of the four sides, depth 2..7 = the deep faces of all eight
// directions (diagonal ranks are direct xGMI peers too), in the order csim_exchange_plan fixes.
int post_plan(csim_stepper* s, int depth, hipStream_t st) {
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

// halos of the CURRENT field: pack its edge lines, exchange, leave them staged in recv[]
int refresh_halos(csim_stepper* s) {
    CSIM_HIP(launch_pack(s->cur, s->nx, s->ny, s->pitch, s->send, s->s_comp));
    int rc = post_plan(s, 1, s->s_comp);
    if (rc) return rc;
    s->halo_fresh = true;
    s->edge_async = false;
    return CSIM_OK;
}

GhostArgs ghost_args(const csim_stepper* s) {
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
bool ring_is_static(const csim_stepper* s) {
    if (s->multi) return false;
    for (int k = 0; k < 4; ++k)
        if (s->bc[k] == CSIM_BC_NEUMANN) return false;
    return true;
}

// ONE reference step: exchange_halos + apply_boundary + fused sweep + swap
int pass_single(csim_stepper* s, const Phys& p, const GhostArgs& g) {
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
        int rc = post_plan(s, 1, s->s_comm);
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

int pass_fused(csim_stepper* s, const Phys& p, int T, int next_T, bool final_pass) {
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
            int rc = post_plan(s, T, s->s_comp);
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
        rc = post_plan(s, next_T, s->s_comm);
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
int tune_rows(csim_stepper* s, const Phys& p, int T, bool preferred_depth) {
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

}  // namespace csim

// Which arithmetic flavour of the multi-step sweep these parameters select (read-only options "fused_2c_active",
// "diffusion_only_active").  v == 0: the screened interior body drops the advection term (kernels.hip, cell) — another
// balance of arithmetic against HBM traffic, so the chunk heights found for the other flavour are not carried over.
static void note_flavour(csim_stepper* s, const Phys& p) {
    s->fused_2c_active = p.fast_thr > 0.0 && p.div_mode != 3;
    const int still = s->fused_2c_active && p.div_mode <= 1 && p.vx == 0.0 && p.vy == 0.0;
    if (still != s->diffusion_only_active) s->forget_tuning();
    s->diffusion_only_active = still;
}

extern "C" {

// the one-off trial of csim_stepper_run's first long call, on request (e.g. before a timed loop)
int csim_stepper_tune(csim_stepper* s, double D, double dt, double vx, double vy) {
    CSIM_REQUIRE(s, "null stepper");
    CSIM_SETTLE(s);
    Phys p = make_phys(s->dx, s->dy, D, dt, vx, vy, s->contract != 0);
    if (!s->fused_2c) p.fast_thr = 0.0;
    note_flavour(s, p);
    const int depth = fused_depth(s);
    if (depth < 2 || s->cfg.rows_per_chunk != 0) return CSIM_OK;
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
    Phys p = make_phys(s->dx, s->dy, D, dt, vx, vy, s->contract != 0);
    if (!s->fused_2c) p.fast_thr = 0.0;
    note_flavour(s, p);
    const int depth = fused_depth(s);
    if (depth < 2) return CSIM_OK;
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
    Phys p = make_phys(s->dx, s->dy, D, dt, vx, vy, s->contract != 0);
    if (!s->fused_2c) p.fast_thr = 0.0;
    note_flavour(s, p);  // before the depth: the diffusion-only flavour prefers 7 steps per pass at every size
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
    plan_passes(nsteps, cap, !auto_depth, s->tile_cells, plan, s->diffusion_only_active != 0);
    // exchange schedule of this run: bulk-first (4, and the default 5).  Until round 3 the default went bulk-first only on
    // runs of fewer than 16 passes and merged (3) otherwise; with the relay (pass_fused_bulk_first) bulk-first is the
    // faster one at every run length on every per-GPU tile of the 16384^2 run (self-linked torus, 1200-step runs:
    // 4096 x 8192 1.26-1.27 M against 1.11-1.19 M merged, 8192 x 16384 1.49-1.50 M against 1.45 M, 8192^2 equal), and
    // it needs nothing but stream order and events: no in-kernel flag, no hipStreamWaitValue64, no write-through stores.
    s->bulk_first_run = s->multi && !s->external && (s->overlap == 4 || s->overlap == 5);
    if (s->bulk_first_run && !s->phys_ring_filled) {
        // A bulk-first pass launches its bulk before its own ghost fill; the bulk tiles next to a thin band along a
        // physical Dirichlet / Periodic side (sweepO_div) read that side's ghost line as level-0 input, which never changes
        // — once it has been written: after an upload or an initialisation that is now (physical sides only, both
        // buffers; the halo-dependent corners and every later refresh are the passes' own ghost fills).
        GhostArgs gp = ghost_args(s);
        for (int k = 0; k < 4; ++k) gp.recv[k] = nullptr;
        CSIM_HIP(launch_ghost_fill(s->cur, s->nxt, s->nx, s->ny, s->pitch, gp, s->tail ? s->tail : s->s_comp));
        s->phys_ring_filled = true;
    }
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

}  // extern "C"

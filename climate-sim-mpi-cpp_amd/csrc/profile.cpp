// profile.cpp — HIP-event brackets around the passes of a run (option "profile"; read back through
// csim_stepper_kernel_time / csim_stepper_comm_time; the reference's only timer is the MPI_Wtime bracket of
// src/main.cpp:94,111 around the whole loop).
#include "stepper.hpp"

namespace csim {

int prof_fold(csim_stepper* s) {
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
int prof_start(csim_stepper* s, int kind, hipStream_t st, long* slot) {
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

int prof_stop(csim_stepper* s, long slot, hipStream_t st) {
    if (slot < 0) return CSIM_OK;
    CSIM_HIP(hipEventRecord(s->ev_pool[static_cast<size_t>(slot) + 1], st));
    return CSIM_OK;
}

// the stop event of an open bracket (see prof_begin)
int prof_close(csim_stepper* s) {
    if (s->prof_open_kind == 0) return CSIM_OK;
    s->prof_open_kind = 0;
    return prof_stop(s, s->prof_open_slot, s->s_comp);
}

int prof_begin(csim_stepper* s, int steps, hipStream_t st) {
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

int prof_end(csim_stepper* s, hipStream_t st) {
    if (!s->prof_active) return CSIM_OK;
    int rc = prof_stop(s, s->prof_slot, st ? st : s->s_comp);
    s->prof_active = false;
    return rc;
}

}  // namespace csim

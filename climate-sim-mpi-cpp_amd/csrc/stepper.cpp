// stepper.cpp — the stepper handle behind include/csim.h (reference src/main.cpp:62-118 minus I/O): creation,
// communicator, upload / download / snapshots, the external halo transport, reductions, options and timers.
// What a run enqueues lives in passes.cpp.
#include "stepper.hpp"

using namespace csim;

size_t csim_stepper::face_len(int d, int H) const { return face_doubles(d, H, nx, ny); }

namespace csim {

// the field state back onto the compute stream (see csim_stepper::tail)
int settle(csim_stepper* s) {
    if (s->tail == nullptr || s->tail == s->s_comp) {
        s->tail = s->s_comp;
        return CSIM_OK;
    }
    CSIM_HIP(hipEventRecord(s->ev_tail, s->tail));
    CSIM_HIP(hipStreamWaitEvent(s->s_comp, s->ev_tail, 0));
    s->tail = s->s_comp;
    return CSIM_OK;
}

// deep-face flavour of the external transport, for csim_stepper_run(.., depth) in external mode
bool depth_ok(const csim_stepper* s, int depth) {
    return depth >= 2 && depth <= MAX_FUSE && depth <= s->nx && depth <= s->ny;
}

// depth of the fused passes of this stepper with the current options (1 = single steps only): what
// the option "fuse" asks for, or pref_fuse(tile) — the depth with the lowest cost per step — in auto mode
int fused_depth(const csim_stepper* s) {
    const int depth = std::min(s->fuse < 0 ? pref_fuse(s->tile_cells, s->diffusion_only_active != 0) : s->fuse, s->fuse_cap);
    const bool dpp_family = s->cfg.variant == VAR_AUTO || s->cfg.variant == VAR_DPP;
    return depth >= 2 && dpp_family ? depth : 1;
}

}  // namespace csim

// Wait for both streams.  With an RCCL communicator the wait polls instead of blocking, so that an asynchronous
// communicator error (a peer that died, a failed transport: ncclCommGetAsyncError) ends it with CSIM_ERR_RCCL
// instead of a silent hang on a stream nobody will ever complete — the reference's MPI_Waitall
// (src/halo.cpp:46) would abort the job through the MPI error handler.  Option "sync_timeout_ms" > 0 bounds the
// wait (CSIM_ERR_TIMEOUT; the streams stay as they are).
static int wait_stream(csim_stepper* s, hipStream_t st, const std::chrono::steady_clock::time_point& t0) {
    if (!s->comm && s->sync_timeout_ms <= 0 && !s->stall_armed) {
        CSIM_HIP(hipStreamSynchronize(st));
        return CSIM_OK;
    }
    for (unsigned long spin = 0;; ++spin) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return CSIM_OK;
        if (q != hipErrorNotReady) return fail(CSIM_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(q));
        (void)hipGetLastError();  // hipErrorNotReady is sticky in the last-error slot
        if ((spin & 63) == 63) {
            if (s->comm) {
                ncclResult_t async = ncclSuccess;
                const ncclResult_t r = ncclCommGetAsyncError(s->comm, &async);
                if (r != ncclSuccess || (async != ncclSuccess && async != ncclInProgress))
                    return fail(CSIM_ERR_RCCL, std::string("halo exchange failed asynchronously: ") +
                                                   ncclGetErrorString(r != ncclSuccess ? r : async));
            }
            if (s->sync_timeout_ms > 0) {
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (ms > static_cast<double>(s->sync_timeout_ms))
                    return fail(CSIM_ERR_TIMEOUT, "csim_stepper_sync: streams still busy after " +
                                                      std::to_string(s->sync_timeout_ms) + " ms (option sync_timeout_ms)");
            }
        }
    }
}

extern "C" {

// ---- stepper -------------------------------------------------------------------------------------
int csim_stepper_create(const csim_decomp* dec, double dx, double dy, const int bc[4],
                        double bc_value, csim_stepper** out) {
    CSIM_REQUIRE(out, "out is null");
    *out = nullptr;
    CSIM_REQUIRE(dec && bc, "null argument");
    CSIM_REQUIRE(dec->nx_local > 0 && dec->ny_local > 0, "empty local tile");
    CSIM_REQUIRE(dx > 0 && dy > 0, "dx/dy must be > 0");
    CSIM_REQUIRE(valid_bc(bc), "unknown boundary type");
    csim_stepper* s = new csim_stepper;
    s->dec = *dec;
    s->dx = dx;
    s->dy = dy;
    s->bc_value = bc_value;
    s->nx = dec->nx_local;
    s->ny = dec->ny_local;
    s->pitch = pitch_for(s->nx);
    for (int k = 0; k < 4; ++k) {
        s->bc[k] = bc[k];
        s->phys[k] = dec->nbr[k] < 0;
        if (!s->phys[k]) s->multi = true;
    }
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) {
        if (e == hipSuccess) e = r;
        return e == hipSuccess;
    };
    ok(hipMalloc(reinterpret_cast<void**>(&s->buf[0]), s->bytes())) &&
        ok(hipMalloc(reinterpret_cast<void**>(&s->buf[1]), s->bytes())) &&
        ok(hipMemset(s->buf[0], 0, s->bytes())) && ok(hipMemset(s->buf[1], 0, s->bytes())) &&
        ok(hipMalloc(reinterpret_cast<void**>(&s->scratch), sizeof(double) * 2 * REDUCE_BLOCKS)) &&
        ok(hipEventCreateWithFlags(&s->ev_tail, hipEventDisableTiming)) &&
        ok(hipEventCreateWithFlags(&s->ev_relay_ready, hipEventDisableTiming | hipEventDisableSystemFence)) &&
        ok(hipEventCreateWithFlags(&s->ev_relay_bulk, hipEventDisableTiming | hipEventDisableSystemFence)) &&
        ok(hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming)) &&
        ok(hipEventCreateWithFlags(&s->ev_edge, hipEventDisableTiming)) &&
        ok(hipEventCreateWithFlags(&s->ev_recv, hipEventDisableTiming));
    for (int k = 0; k < 4 && e == hipSuccess; ++k) {
        const size_t nf = sizeof(double) * static_cast<size_t>(k < 2 ? s->ny : s->nx);
        ok(hipMalloc(reinterpret_cast<void**>(&s->fin[k]), nf)) && ok(hipMemset(s->fin[k], 0, nf));
    }
    for (int k = 0; k < 4 && e == hipSuccess; ++k) {
        if (s->phys[k]) continue;
        const size_t n = sizeof(double) * static_cast<size_t>(k < 2 ? s->ny : s->nx);
        ok(hipMalloc(reinterpret_cast<void**>(&s->send[k]), n)) &&
            ok(hipMalloc(reinterpret_cast<void**>(&s->recv[k]), n)) &&
            ok(hipMemset(s->send[k], 0, n)) && ok(hipMemset(s->recv[k], 0, n));
    }
    // The fused-pass depth must be decided identically on every rank (the face exchange is
    // collective in effect): depth <= the smallest tile of the decomposition.
    {
        const int px = dec->dims[0] > 0 ? dec->dims[0] : 1, py = dec->dims[1] > 0 ? dec->dims[1] : 1;
        const int gx = dec->nx_global > 0 ? dec->nx_global : s->nx, gy = dec->ny_global > 0 ? dec->ny_global : s->ny;
        const int bx = gx / px, by = gy / py;
        // the size classes of the pass planner are single-rank measurements; across ranks every pass carries an
        // exchange whose cost does not shrink with the depth, so shallow passes lose what they gain (20-step
        // runs on the 4096 x 8192 self-torus: 4 x 5 932 k, 7 + 7 + 6 1 176 k): multi-rank steppers plan with the
        // mid-size table (preferred depth 6, depth 7 where it saves a pass)
        s->tile_cells = s->multi ? 0 : static_cast<long>(s->nx) * s->ny;
        const int min_tile = s->multi ? std::min(bx, by) : MAX_FUSE;
        s->fuse_cap = std::max(1, std::min(MAX_FUSE, min_tile));
    }
    neighbours8(*dec, s->nbr8);  // diagonal peers only where both adjacent sides have neighbours
    for (int d = 0; d < 8 && e == hipSuccess; ++d) {
        if (s->nbr8[d] < 0) continue;
        s->cap2[d] = s->face_len(d, MAX_FUSE);
        const size_t n = sizeof(double) * s->cap2[d];
        ok(hipMalloc(reinterpret_cast<void**>(&s->send2[d]), n)) &&
            ok(hipMalloc(reinterpret_cast<void**>(&s->recv2[d]), n)) &&
            ok(hipMemset(s->send2[d], 0, n)) && ok(hipMemset(s->recv2[d], 0, n));
    }
    if (e == hipSuccess) {
        // the exchange goes on a high-priority stream, so its small kernels are dispatched ahead of the
        // bulk sweep that is hiding them
        int lo = 0, hi = 0;  // numerically lower = higher priority
        ok(hipDeviceGetStreamPriorityRange(&lo, &hi)) &&
            ok(hipStreamCreateWithPriority(&s->s_comm, hipStreamNonBlocking, hi)) &&
            ok(hipStreamCreateWithFlags(&s->s_comp, hipStreamNonBlocking));
        // the relay pair of the bulk-first passes: two streams of EQUAL (high) priority — they carry the same kinds of
        // work in turn —, apart from the compute stream, which keeps its normal priority below the comm stream for
        // the frame-first schedules (their exchange kernels must be dispatched ahead of the sweep that hides them)
        if (s->multi)
            ok(hipStreamCreateWithPriority(&s->s_relay[0], hipStreamNonBlocking, hi)) &&
                ok(hipStreamCreateWithPriority(&s->s_relay[1], hipStreamNonBlocking, hi));
        s->tail = s->s_comp;
    }
    if (e == hipSuccess) {
        ok(hipEventCreateWithFlags(&s->ev_edge2, hipEventDisableTiming)) &&
            ok(hipEventCreateWithFlags(&s->ev_recv2, hipEventDisableTiming));
    }
    if (e == hipSuccess && s->multi) {
        ok(hipMalloc(reinterpret_cast<void**>(&s->frame_counter), sizeof(unsigned))) &&
            ok(hipMemset(s->frame_counter, 0, sizeof(unsigned)));
        // signal memory: absent or refused -> mode 3 is simply not offered (set_option reports it)
        if (e == hipSuccess) {
            int can = 0;
            int dev = 0;
            if (hipGetDevice(&dev) == hipSuccess &&
                hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev) == hipSuccess && can) {
                void* p = nullptr;
                if (hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory) == hipSuccess) {
                    s->frame_flag = static_cast<unsigned long long*>(p);
                    *s->frame_flag = 0;  // host-visible
                } else {
                    (void)hipGetLastError();
                }
            }
        }
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) {
        s->cur = s->buf[0] + static_cast<size_t>(GHOST_EXTRA) * s->pitch;
        s->nxt = s->buf[1] + static_cast<size_t>(GHOST_EXTRA) * s->pitch;
    }
    if (e != hipSuccess) {
        csim_stepper_destroy(s);
        return fail(CSIM_ERR_HIP, std::string("csim_stepper_create: ") + hipGetErrorString(e));
    }
    *out = s;
    return CSIM_OK;
}

int csim_stepper_destroy(csim_stepper* s) {
    if (!s) return CSIM_OK;
    if (s->s_comp) (void)hipStreamSynchronize(s->s_comp);
    if (s->s_comm) (void)hipStreamSynchronize(s->s_comm);
    for (hipStream_t r : s->s_relay)
        if (r) (void)hipStreamSynchronize(r);
    if (s->s_io) (void)hipStreamSynchronize(s->s_io);
    if (s->snap_d) (void)hipFree(s->snap_d);
    if (s->snap_h) (void)hipHostFree(s->snap_h);
    if (s->ev_snap_src) (void)hipEventDestroy(s->ev_snap_src);
    if (s->ev_snap_copied) (void)hipEventDestroy(s->ev_snap_copied);
    if (s->s_io) (void)hipStreamDestroy(s->s_io);
    if (s->comm && !s->comm_borrowed) (void)ncclCommDestroy(s->comm);
    for (hipEvent_t ev : s->ev_pool) (void)hipEventDestroy(ev);
    for (int k = 0; k < 4; ++k) {
        if (s->send[k]) (void)hipFree(s->send[k]);
        if (s->recv[k]) (void)hipFree(s->recv[k]);
        if (s->fin[k]) (void)hipFree(s->fin[k]);
    }
    for (int d = 0; d < 8; ++d) {
        if (s->send2[d]) (void)hipFree(s->send2[d]);
        if (s->recv2[d]) (void)hipFree(s->recv2[d]);
    }
    if (s->frame_counter) (void)hipFree(s->frame_counter);
    if (s->frame_flag) (void)hipFree(s->frame_flag);
    if (s->ev_edge2) (void)hipEventDestroy(s->ev_edge2);
    if (s->ev_recv2) (void)hipEventDestroy(s->ev_recv2);
    if (s->ev_edge) (void)hipEventDestroy(s->ev_edge);
    if (s->ev_recv) (void)hipEventDestroy(s->ev_recv);
    if (s->s_comp) (void)hipStreamDestroy(s->s_comp);
    for (hipStream_t r : s->s_relay)
        if (r) (void)hipStreamDestroy(r);
    if (s->s_comm) (void)hipStreamDestroy(s->s_comm);
    if (s->ev_ready) (void)hipEventDestroy(s->ev_ready);
    if (s->ev_tail) (void)hipEventDestroy(s->ev_tail);
    if (s->ev_relay_ready) (void)hipEventDestroy(s->ev_relay_ready);
    if (s->ev_relay_bulk) (void)hipEventDestroy(s->ev_relay_bulk);
    if (s->buf[0]) (void)hipFree(s->buf[0]);
    if (s->buf[1]) (void)hipFree(s->buf[1]);
    if (s->scratch) (void)hipFree(s->scratch);
    delete s;
    return CSIM_OK;
}

int csim_comm_unique_id(void* id, size_t nbytes) {
    static_assert(sizeof(ncclUniqueId) == CSIM_UNIQUE_ID_BYTES, "ncclUniqueId size");
    CSIM_REQUIRE(id && nbytes >= sizeof(ncclUniqueId), "id buffer too small");
    ncclUniqueId u;
    CSIM_NCCL(ncclGetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return CSIM_OK;
}

int csim_stepper_comm_init(csim_stepper* s, const void* id, size_t nbytes) {
    CSIM_REQUIRE(s && id && nbytes >= sizeof(ncclUniqueId), "bad argument");
    if (s->comm) return fail(CSIM_ERR_STATE, "communicator already initialised");
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    CSIM_NCCL(ncclCommInitRank(&s->comm, s->dec.size, u, s->dec.rank));
    return CSIM_OK;
}

// Several steppers of one rank on one communicator (e.g. small parity cases run beside the production tile:
// building a communicator costs ~1 s at 8 ranks).  `s` borrows `owner`'s communicator and never destroys it;
// `owner` must outlive `s`.  The steppers must not have exchanges in flight at the same time (RCCL matches the
// messages of a rank pair in posting order): sync one before running the other.
int csim_stepper_comm_share(csim_stepper* s, csim_stepper* owner) {
    CSIM_REQUIRE(s && owner && s != owner, "bad argument");
    if (s->comm) return fail(CSIM_ERR_STATE, "communicator already initialised");
    if (!owner->comm) return fail(CSIM_ERR_STATE, "owner has no communicator: csim_stepper_comm_init first");
    CSIM_REQUIRE(s->dec.size == owner->dec.size && s->dec.rank == owner->dec.rank, "steppers of different ranks / world sizes");
    s->comm = owner->comm;
    s->comm_borrowed = true;
    return CSIM_OK;
}

int csim_stepper_upload(csim_stepper* s, const double* host) {
    CSIM_REQUIRE(s && host, "null argument");
    CSIM_SETTLE(s);
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    CSIM_HIP(hipStreamSynchronize(s->s_comm));
    int rc = upload_2d(s->cur, s->nx, s->ny, s->pitch, host);
    if (rc) return rc;
    // both ping-pong buffers start with the same ghost ring (reference main.cpp:104 copies u->tmp)
    // (a device-to-device hipMemcpy may return before it has run, and the stepper's streams do
    // not synchronise with the null stream: order the copy on the compute stream and wait)
    CSIM_HIP(hipDeviceSynchronize());
    CSIM_HIP(hipMemcpyAsync(s->base(s->nxt), s->base(s->cur), s->bytes(), hipMemcpyDeviceToDevice,
                            s->s_comp));
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    s->halo_fresh = false;
    s->faces_depth = 0;
    s->ring_ok = false;
    s->phys_ring_filled = false;
    return CSIM_OK;
}

int csim_stepper_download(csim_stepper* s, double* host) {
    CSIM_REQUIRE(s && host, "null argument");
    CSIM_SETTLE(s);
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    return download_2d(s->cur, s->nx, s->ny, s->pitch, host);
}

int csim_stepper_download_interior(csim_stepper* s, double* host) {
    CSIM_REQUIRE(s && host, "null argument");
    CSIM_SETTLE(s);
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    return download_interior_2d(s->cur, s->nx, s->ny, s->pitch, host);
}

// Snapshot without stalling the time loop (the reference packs and writes the interior inside the
// step loop, src/io.cpp:402-424 called from src/main.cpp:96-99).  _begin enqueues a device-side
// copy of the current interior (the ping-pong buffers are free to move on after ~1 ms) followed
// by an asynchronous D2H into a pinned buffer on a third stream, and returns at once; the caller
// keeps enqueuing steps and calls _wait when it wants the data (pointer valid until the next
// _begin).  Layout: ny_local x nx_local, row-major — what write_field_netcdf packs.
int csim_stepper_snapshot_begin(csim_stepper* s) {
    CSIM_REQUIRE(s, "null stepper");
    CSIM_SETTLE(s);
    const size_t bytes = sizeof(double) * static_cast<size_t>(s->nx) * s->ny;
    // each piece is created once; a failed allocation is reported and retried by the next call
    if (!s->s_io) CSIM_HIP(hipStreamCreateWithFlags(&s->s_io, hipStreamNonBlocking));
    if (!s->ev_snap_src) CSIM_HIP(hipEventCreateWithFlags(&s->ev_snap_src, hipEventDisableTiming));
    if (!s->ev_snap_copied) CSIM_HIP(hipEventCreateWithFlags(&s->ev_snap_copied, hipEventDisableTiming));
    if (!s->snap_d) CSIM_HIP(hipMalloc(reinterpret_cast<void**>(&s->snap_d), bytes));
    if (!s->snap_h) CSIM_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->snap_h), bytes, hipHostMallocDefault));
    if (s->snap_pending) CSIM_HIP(hipStreamSynchronize(s->s_io));  // previous snapshot still in flight
    CSIM_HIP(hipEventRecord(s->ev_snap_src, s->s_comp));
    CSIM_HIP(hipStreamWaitEvent(s->s_io, s->ev_snap_src, 0));
    CSIM_HIP(hipMemcpy2DAsync(s->snap_d, sizeof(double) * s->nx, s->cur + s->pitch + LPAD,
                              sizeof(double) * s->pitch, sizeof(double) * s->nx, s->ny,
                              hipMemcpyDeviceToDevice, s->s_io));
    CSIM_HIP(hipEventRecord(s->ev_snap_copied, s->s_io));
    // the sweeps may overwrite the source buffer only after the staging copy has read it
    CSIM_HIP(hipStreamWaitEvent(s->s_comp, s->ev_snap_copied, 0));
    CSIM_HIP(hipMemcpyAsync(s->snap_h, s->snap_d, bytes, hipMemcpyDeviceToHost, s->s_io));
    s->snap_pending = true;
    return CSIM_OK;
}

int csim_stepper_snapshot_wait(csim_stepper* s, const double** host_interior) {
    CSIM_REQUIRE(s && host_interior, "null argument");
    if (!s->snap_pending) return fail(CSIM_ERR_STATE, "no snapshot in flight: csim_stepper_snapshot_begin first");
    CSIM_HIP(hipStreamSynchronize(s->s_io));
    s->snap_pending = false;
    *host_interior = s->snap_h;
    return CSIM_OK;
}

int csim_stepper_init_gaussian(csim_stepper* s, double A, double sigma_frac, double xc_frac,
                               double yc_frac) {
    CSIM_REQUIRE(s, "null stepper");
    CSIM_SETTLE(s);
    CSIM_HIP(hipStreamSynchronize(s->s_comm));
    CSIM_HIP(hipMemsetAsync(s->base(s->cur), 0, s->bytes(), s->s_comp));
    CSIM_HIP(launch_gaussian(s->cur, s->nx, s->ny, s->pitch, s->dec.x_offset, s->dec.y_offset,
                             s->dec.nx_global, s->dec.ny_global, s->dx, s->dy, A, sigma_frac,
                             xc_frac, yc_frac, s->s_comp));
    CSIM_HIP(hipMemcpyAsync(s->base(s->nxt), s->base(s->cur), s->bytes(), hipMemcpyDeviceToDevice,
                            s->s_comp));
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    s->halo_fresh = false;
    s->faces_depth = 0;
    s->ring_ok = false;
    s->phys_ring_filled = false;
    return CSIM_OK;
}

// External transport (e.g. the reference's own MPI): the caller moves the edge lines between
// ranks.  pack: edge lines of the current field -> host buffers (ny doubles for left/right, nx
// for bottom/top; entries of physical sides are ignored).  unpack: the neighbours' lines -> the
// staging buffers the next step's ghost fill reads.
int csim_stepper_halo_pack(csim_stepper* s, double* const host_send[4]) {
    CSIM_REQUIRE(s && host_send, "null argument");
    CSIM_SETTLE(s);
    if (!s->multi) return CSIM_OK;
    CSIM_HIP(launch_pack(s->cur, s->nx, s->ny, s->pitch, s->send, s->s_comp));
    for (int k = 0; k < 4; ++k) {
        if (s->phys[k]) continue;
        CSIM_REQUIRE(host_send[k], "missing host buffer for a neighbour side");
        const size_t n = sizeof(double) * static_cast<size_t>(k < 2 ? s->ny : s->nx);
        CSIM_HIP(hipMemcpyAsync(host_send[k], s->send[k], n, hipMemcpyDeviceToHost, s->s_comp));
    }
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    return CSIM_OK;
}

int csim_stepper_halo_unpack(csim_stepper* s, const double* const host_recv[4]) {
    CSIM_REQUIRE(s && host_recv, "null argument");
    if (!s->multi) return CSIM_OK;
    for (int k = 0; k < 4; ++k) {
        if (s->phys[k]) continue;
        CSIM_REQUIRE(host_recv[k], "missing host buffer for a neighbour side");
        const size_t n = sizeof(double) * static_cast<size_t>(k < 2 ? s->ny : s->nx);
        CSIM_HIP(hipMemcpyAsync(s->recv[k], host_recv[k], n, hipMemcpyHostToDevice, s->s_comp));
    }
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    s->halo_fresh = true;
    return CSIM_OK;
}

// deepest fused pass this stepper can run (1 = single steps only); identical on every rank of a
// decomposition, so external-transport callers can schedule their passes the way run() does
int csim_stepper_fuse_limit(const csim_stepper* s, int* depth) {
    CSIM_REQUIRE(s && depth, "null argument");
    *depth = fused_depth(s);
    return CSIM_OK;
}

int csim_stepper_faces_neighbors(const csim_stepper* s, int depth, int peers[8], int lengths[8]) {
    CSIM_REQUIRE(s && peers && lengths, "null argument");
    CSIM_REQUIRE(depth_ok(s, depth), "face depth must be 2..7 and fit the tile");
    for (int d = 0; d < 8; ++d) {
        peers[d] = s->nbr8[d];
        lengths[d] = s->nbr8[d] >= 0 ? static_cast<int>(s->face_len(d, depth)) : 0;
    }
    return CSIM_OK;
}

int csim_stepper_faces_pack(csim_stepper* s, int depth, double* const host_send[8]) {
    CSIM_REQUIRE(s && host_send, "null argument");
    CSIM_SETTLE(s);
    CSIM_REQUIRE(depth_ok(s, depth), "face depth must be 2..7 and fit the tile");
    if (!s->multi) return CSIM_OK;
    CSIM_HIP(launch_halo2_pack(s->cur, s->nx, s->ny, s->pitch, depth, s->send2, s->s_comp));
    for (int d = 0; d < 8; ++d) {
        if (s->nbr8[d] < 0) continue;
        CSIM_REQUIRE(host_send[d], "missing host buffer for a neighbour direction");
        CSIM_HIP(hipMemcpyAsync(host_send[d], s->send2[d], sizeof(double) * s->face_len(d, depth),
                                hipMemcpyDeviceToHost, s->s_comp));
    }
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    return CSIM_OK;
}

int csim_stepper_faces_unpack(csim_stepper* s, int depth, const double* const host_recv[8]) {
    CSIM_REQUIRE(s && host_recv, "null argument");
    CSIM_REQUIRE(depth_ok(s, depth), "face depth must be 2..7 and fit the tile");
    if (!s->multi) return CSIM_OK;
    for (int d = 0; d < 8; ++d) {
        if (s->nbr8[d] < 0) continue;
        CSIM_REQUIRE(host_recv[d], "missing host buffer for a neighbour direction");
        CSIM_HIP(hipMemcpyAsync(s->recv2[d], host_recv[d], sizeof(double) * s->face_len(d, depth),
                                hipMemcpyHostToDevice, s->s_comp));
    }
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    s->faces_depth = depth;
    return CSIM_OK;
}

int csim_stepper_exchange_halos(csim_stepper* s) {
    CSIM_REQUIRE(s, "null stepper");
    CSIM_SETTLE(s);
    if (!s->multi) return CSIM_OK;
    int rc = refresh_halos(s);
    if (rc) return rc;
    // unpack only (no boundary rule): physical sides are left alone, like reference halo.cpp
    GhostArgs g = ghost_args(s);
    for (int k = 0; k < 4; ++k)
        if (g.phys[k]) g.bc[k] = CSIM_BC_PERIODIC;
    CSIM_HIP(launch_ghost_fill(s->cur, s->nxt, s->nx, s->ny, s->pitch, g, s->s_comp));
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    return CSIM_OK;
}

int csim_stepper_sync(csim_stepper* s) {
    CSIM_REQUIRE(s, "null stepper");
    const auto t0 = std::chrono::steady_clock::now();
    int rc = wait_stream(s, s->s_comp, t0);
    if (rc) return rc;
    for (hipStream_t r : s->s_relay)
        if (r) {
            rc = wait_stream(s, r, t0);
            if (rc) return rc;
        }
    return wait_stream(s, s->s_comm, t0);
}

// Position-weighted 64-bit checksum of the local interior (k_checksum): the per-rank values of a decomposition
// add up modulo 2^64 to the checksum of the same global field on one rank.
int csim_stepper_checksum(csim_stepper* s, unsigned long long* out) {
    CSIM_REQUIRE(s && out, "null argument");
    CSIM_SETTLE(s);
    const long nxg = s->dec.nx_global > 0 ? s->dec.nx_global : s->nx;
    CSIM_HIP(launch_checksum(s->cur, s->nx, s->ny, s->pitch, s->dec.x_offset, s->dec.y_offset, nxg, s->scratch, s->s_comp));
    const int nb = reduce_blocks(s->ny);
    std::vector<unsigned long long> h(static_cast<size_t>(nb));
    CSIM_HIP(hipMemcpyAsync(h.data(), s->scratch, sizeof(unsigned long long) * nb, hipMemcpyDeviceToHost, s->s_comp));
    CSIM_HIP(hipStreamSynchronize(s->s_comp));
    unsigned long long acc = 0;
    for (int k = 0; k < nb; ++k) acc += h[static_cast<size_t>(k)];
    *out = acc;
    return CSIM_OK;
}

int csim_stepper_minmax(csim_stepper* s, double out[2]) {
    CSIM_REQUIRE(s && out, "null argument");
    CSIM_SETTLE(s);
    CSIM_HIP(launch_minmax(s->cur, s->nx, s->ny, s->pitch, s->scratch, s->s_comp));
    return finish_partials(s->scratch, reduce_blocks(s->ny + 2), 0, out, s->s_comp);
}

int csim_stepper_sum(csim_stepper* s, double* out) {
    CSIM_REQUIRE(s && out, "null argument");
    CSIM_SETTLE(s);
    double r[2];
    CSIM_HIP(launch_sum(s->cur, s->nx, s->ny, s->pitch, s->scratch, s->s_comp));
    int rc = finish_partials(s->scratch, reduce_blocks(s->ny), 1, r, s->s_comp);
    *out = r[0];
    return rc;
}

int csim_stepper_set_option(csim_stepper* s, const char* key, long value) {
    CSIM_REQUIRE(s && key, "null argument");
    const std::string k(key);
    if (k == "variant") {
        CSIM_REQUIRE(value >= VAR_AUTO && value <= VAR_NAIVE, "unknown variant");
        s->cfg.variant = static_cast<int>(value);
    } else if (k == "rows_per_chunk") {
        CSIM_REQUIRE(value >= 0, "rows_per_chunk must be >= 0");
        s->cfg.rows_per_chunk = static_cast<int>(value);
    } else if (k == "prefetch") {
        CSIM_REQUIRE(value >= 0 && value <= 8, "prefetch must be 0..8");
        s->cfg.prefetch = static_cast<int>(value);
    } else if (k == "xcd_swizzle") {
        s->cfg.xcd_swizzle = value != 0;
    } else if (k == "tail_split") {
        CSIM_REQUIRE(value >= 0 && value <= 2, "tail_split must be 0, 1 or 2");
        s->cfg.tail_split = static_cast<int>(value);
        s->forget_tuning();  // the best chunk height depends on it
    } else if (k == "overlap") {
        CSIM_REQUIRE(value >= 0 && value <= 5 && value != 2, "overlap must be 0, 1, 3, 4 or 5");
        // the schedules hand state to each other only through "nothing in flight": every csim_stepper_run ends that way
        if (!s->external && (s->faces_depth != 0 || s->pre_unpacked))
            return fail(CSIM_ERR_STATE, "internal: exchange schedule changed with faces of a fused pass in flight");
        if (value == 3 && s->multi && !s->frame_flag)
            return fail(CSIM_ERR_STATE, "overlap 3 needs hipStreamWaitValue64 / signal memory, which this device or runtime refused");
        s->overlap = static_cast<int>(value);
    } else if (k == "frame_fence") {
        CSIM_REQUIRE(value == 0 || value == 1, "frame_fence must be 0 or 1");
        s->frame_fence = static_cast<int>(value);
    } else if (k == "frame_rows") {
        CSIM_REQUIRE(value >= 0 && value <= 4096, "frame_rows must be 0..4096");
        s->cfg.frame_rows = static_cast<int>(value);
    } else if (k == "frame_prio") {
        s->frame_prio = value != 0;
    } else if (k == "relay") {
        CSIM_SETTLE(s);
        s->relay = value != 0;
    } else if (k == "relay_events") {
        CSIM_REQUIRE(value == 0 || value == 1, "relay_events must be 0 or 1");
        s->relay_events = static_cast<int>(value);
    } else if (k == "direct_faces") {
        s->direct_faces = value != 0;
    } else if (k == "fused_2c") {
        s->fused_2c = value != 0;
    } else if (k == "external_halo") {
        s->external = value != 0;
        s->halo_fresh = false;
    } else if (k == "contract") {
        CSIM_REQUIRE(value == 0 || value == 1, "contract must be 0 (reference operation order, default) or 1");
        if (s->contract != static_cast<int>(value)) s->forget_tuning();  // another kernel: its best chunk height is found anew
        s->contract = static_cast<int>(value);
    } else if (k == "fuse") {
        CSIM_REQUIRE(value >= -1 && value <= MAX_FUSE, "fuse must be -1 (auto) or 0..7");
        s->fuse = static_cast<int>(value);
    } else if (k == "lds_bytes") {
        CSIM_REQUIRE(value >= 0 && value <= 65536, "lds_bytes must be 0..65536");
        s->cfg.lds_bytes = static_cast<int>(value);
    } else if (k == "autotune") {
        s->autotune = value != 0;
        s->forget_tuning();
    } else if (k == "tuned_rows" || k == "last_rows") {  // read back through csim_stepper_get_option
        return fail(CSIM_ERR_ARG, k + " is read-only");
    } else if (k == "sync_timeout_ms") {
        CSIM_REQUIRE(value >= 0, "sync_timeout_ms must be >= 0");
        s->sync_timeout_ms = value;
    } else if (k == "test_stall") {
        // Test hook for the stall handling of callers (bench.py's watchdog, csim_stepper_sync's timeout): 1 parks the
        // comm stream on a value of the signal word that no kernel ever publishes — exactly what a lost flag or a
        // dead peer looks like from the host —, 0 releases it from the host and restores the word.
        CSIM_REQUIRE(value == 0 || value == 1, "test_stall must be 0 or 1");
        if (!s->frame_flag) return fail(CSIM_ERR_STATE, "test_stall needs a multi-rank stepper with signal memory");
        constexpr unsigned long long NEVER = 1ull << 62;
        if (value == 1 && !s->stall_armed) {
            CSIM_HIP(hipStreamWaitValue64(s->s_comm, s->frame_flag, NEVER, hipStreamWaitValueGte, ~0ull));
            s->stall_armed = true;
        } else if (value == 0 && s->stall_armed) {
            __atomic_store_n(s->frame_flag, NEVER, __ATOMIC_SEQ_CST);
            CSIM_HIP(hipStreamSynchronize(s->s_comm));
            __atomic_store_n(s->frame_flag, s->pass_no, __ATOMIC_SEQ_CST);
            s->stall_armed = false;
        }
    } else if (k == "profile") {
        CSIM_REQUIRE(value >= 0 && value <= 1024, "profile must be 0..1024");
        s->profile = static_cast<int>(value);
        s->prof_counter = 0;
    } else {
        return fail(CSIM_ERR_ARG, "unknown option: " + k);
    }
    return CSIM_OK;
}

int csim_stepper_get_option(const csim_stepper* s, const char* key, long* value) {
    CSIM_REQUIRE(s && key && value, "null argument");
    const std::string k(key);
    if (k == "variant") *value = s->cfg.variant;
    else if (k == "rows_per_chunk") *value = s->cfg.rows_per_chunk;
    else if (k == "tuned_rows") *value = s->cfg.tuned_rows;
    else if (k.size() == 12 && k.compare(0, 11, "tuned_rows_") == 0 && k[11] >= '2' && k[11] <= '0' + MAX_FUSE)
        *value = s->tuned_T[k[11] - '0'];  // "tuned_rows_2" .. "tuned_rows_7": the trial's result for passes of that depth (0 = none)
    else if (k == "last_rows") *value = s->last_rows;
    else if (k == "prefetch") *value = s->cfg.prefetch;
    else if (k == "xcd_swizzle") *value = s->cfg.xcd_swizzle;
    else if (k == "tail_split") *value = s->cfg.tail_split;
    else if (k == "overlap") *value = s->overlap;
    else if (k == "direct_faces") *value = s->direct_faces;
    else if (k == "relay") *value = s->relay;
    else if (k == "relay_events") *value = s->relay_events;
    else if (k == "fused_2c") *value = s->fused_2c;
    else if (k == "fused_2c_active") *value = s->fused_2c_active;
    else if (k == "diffusion_only_active") *value = s->diffusion_only_active;
    else if (k == "frame_rows") *value = s->cfg.frame_rows;
    else if (k == "external_halo") *value = s->external;
    else if (k == "fuse") *value = s->fuse;
    else if (k == "contract") *value = s->contract;
    else if (k == "autotune") *value = s->autotune;
    else if (k == "profile") *value = s->profile;
    else if (k == "sync_timeout_ms") *value = s->sync_timeout_ms;
    else if (k == "test_stall") *value = s->stall_armed;
    else if (k == "faces_in_flight") *value = s->faces_depth != 0 || s->pre_unpacked;
    else return fail(CSIM_ERR_ARG, "unknown option: " + k);
    return CSIM_OK;
}

int csim_stepper_kernel_time(csim_stepper* s, int steps_per_launch, double* total_ms,
                             long* launches) {
    CSIM_REQUIRE(s && total_ms && launches, "null argument");
    CSIM_REQUIRE(steps_per_launch >= 1 && steps_per_launch <= MAX_FUSE, "steps_per_launch must be 1..7");
    int rc = prof_fold(s);
    if (rc) return rc;
    *total_ms = s->prof_ms[steps_per_launch];
    *launches = s->prof_launches[steps_per_launch];
    return CSIM_OK;
}

int csim_stepper_reset_timers(csim_stepper* s) {
    CSIM_REQUIRE(s, "null stepper");
    int rc = prof_fold(s);
    if (rc) return rc;
    for (int t = 0; t <= csim_stepper::PROF_COMM; ++t) {
        s->prof_ms[t] = 0.0;
        s->prof_launches[t] = 0;
    }
    return CSIM_OK;
}

int csim_stepper_comm_time(csim_stepper* s, double* total_ms, long* passes) {
    CSIM_REQUIRE(s && total_ms && passes, "null argument");
    int rc = prof_fold(s);
    if (rc) return rc;
    *total_ms = s->prof_ms[csim_stepper::PROF_COMM];
    *passes = s->prof_launches[csim_stepper::PROF_COMM];
    return CSIM_OK;
}

}  // extern "C"

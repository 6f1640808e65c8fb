/* include/csim.h — C ABI of the MI355X-native advection–diffusion stepper.
 *
 * This is the drop-in boundary for ONE hot path of antoniorizzoeng/climate-sim-mpi-cpp:
 * the per-time-step sequence
 *     exchange_halos -> apply_boundary -> copy -> diffusion_step -> advection_step -> swap
 * (reference src/main.cpp:101-109).  The reference has no FFI/plugin layer: its boundary is
 * the set of C++ free functions and structs in include/{field,decomp,halo,boundary,diffusion,
 * advection,stability}.hpp.  Each entry point below names the reference interface it replaces.
 * The C++ mirror of those headers (same names and signatures) lives in include/climate/ and
 * calls only this ABI.
 *
 * Conventions
 *   - every function returns 0 (CSIM_OK) or a CSIM_ERR_* code; csim_last_error() gives the text
 *     (thread-local).  No C++ types, no exceptions cross this boundary.
 *   - host arrays use the reference layout (reference src/field.cpp:20-25): row-major, ghost
 *     ring included, element (i,j) at host[j*(nx+2*halo)+i], i contiguous.  Only halo==1 is
 *     supported, like the reference driver (src/main.cpp:65); other values give CSIM_ERR_ARG.
 *   - sides are ordered left(x-), right(x+), bottom(y-), top(y+) everywhere.
 *   - all arithmetic is IEEE fp64 in the reference's association order with no FMA
 *     contraction, so results are bit-identical to the reference CPU path.
 *   - there is NO CPU fallback: without a usable gfx950 device the calls fail with CSIM_ERR_HIP.
 *   - a handle is not thread-safe; use one stepper per GPU (one process per GPU).
 */
#ifndef CSIM_H
#define CSIM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSIM_ABI_VERSION 1

enum {
    CSIM_OK = 0,
    CSIM_ERR_ARG = 1,         /* bad argument (maps to std::out_of_range / std::runtime_error) */
    CSIM_ERR_HIP = 2,         /* a HIP runtime call failed, or no device */
    CSIM_ERR_RCCL = 3,        /* an RCCL call failed */
    CSIM_ERR_STATE = 4,       /* call sequence error (e.g. multi-rank run before comm init) */
    CSIM_ERR_UNSUPPORTED = 5,
    CSIM_ERR_TIMEOUT = 6      /* csim_stepper_sync with option "sync_timeout_ms": the streams did not drain in time */
};

/* reference include/boundary.hpp:5  enum class BCType { Dirichlet, Neumann, Periodic } */
enum { CSIM_BC_DIRICHLET = 0, CSIM_BC_NEUMANN = 1, CSIM_BC_PERIODIC = 2 };
enum { CSIM_LEFT = 0, CSIM_RIGHT = 1, CSIM_BOTTOM = 2, CSIM_TOP = 3 };

#define CSIM_NO_NEIGHBOR (-1)   /* stands for MPI_PROC_NULL */
#define CSIM_UNIQUE_ID_BYTES 128

typedef struct csim_field csim_field;     /* device mirror of reference `struct Field`       */
typedef struct csim_stepper csim_stepper; /* the time-loop engine (reference main.cpp:93-118) */

/* reference include/decomp.hpp:4-17 `struct Decomp2D` without the MPI communicator */
typedef struct csim_decomp {
    int size, rank;
    int dims[2];    /* dims[0] splits x (the contiguous axis), dims[1] splits y */
    int coords[2];
    int nbr[4];     /* left,right = nbr_lr[0..1]; bottom,top = nbr_du[0..1]; CSIM_NO_NEIGHBOR */
    int nx_global, ny_global;
    int nx_local, ny_local;
    int x_offset, y_offset;
} csim_decomp;

/* ---- library / device ------------------------------------------------------------------ */
const char* csim_last_error(void);
int csim_abi_version(void);
int csim_device_count(int* count);
int csim_set_device(int device);           /* one process per GPU: call once with LOCAL_RANK */
int csim_device_name(char* buf, size_t n); /* gcnArchName of the current device */

/* ---- host-side scalars ------------------------------------------------------------------ */
/* reference include/stability.hpp:5-16  double safe_dt(dx,dy,vx,vy,D) */
double csim_safe_dt(double dx, double dy, double vx, double vy, double D);
/* reference src/decomp.cpp:5-34  Decomp2D::init(comm, nx_global, ny_global), MPI-free:
 * MPI_Dims_create(size,2) + MPI_Cart_create(periods 0,0, reorder 0) are re-derived. */
int csim_decomp_init(int size, int rank, int nx_global, int ny_global, csim_decomp* out);
/* The halo exchange of one rank as an ordered message list (pure host arithmetic, no GPU): what
 * the stepper posts inside ONE ncclGroupStart/End — replaces the <= 8 MPI_Isend/Irecv + MPI_Waitall of
 * reference src/halo.cpp:28-46.  depth 1: the four edge lines (ny / nx doubles, interior span);
 * depth 2..7 (fused passes): faces of that depth in 8 directions 0..7 = left, right, bottom, top,
 * bottom-left, bottom-right, top-left, top-right (depth*(ny+2), depth*(nx+2), depth*depth doubles).
 * sends[k].dir = the direction the face leaves in; recvs[k].dir = the direction it arrives FROM.  RCCL
 * matches the messages of a pair of ranks in posting order, so for every pair (a, b) the k-th send
 * of a to b must be the k-th receive b posts for a: tests/test_exchange_plan.py checks exactly that
 * for every rank pair of 2x1 ... 4x2 process grids at every depth. */
typedef struct csim_msg {
    int peer;    /* rank */
    int dir;     /* 0..7 */
    long count;  /* doubles */
} csim_msg;
int csim_exchange_plan(const csim_decomp* dec, int depth, csim_msg sends[8], int* nsend, csim_msg recvs[8],
                       int* nrecv);

/* ---- Field (reference include/field.hpp:5-21, src/field.cpp:6-31) ------------------------ */
int csim_field_create(int nx, int ny, int halo, double dx, double dy, csim_field** out); /* zero-filled */
int csim_field_destroy(csim_field* f);
int csim_field_upload(csim_field* f, const double* host_with_ghosts);
int csim_field_download(const csim_field* f, double* host_with_ghosts);
int csim_field_download_interior(const csim_field* f, double* host_ny_by_nx);
int csim_field_fill(csim_field* f, double value);                 /* Field::fill            */
int csim_field_copy(csim_field* dst, const csim_field* src);      /* std::copy, main.cpp:104 */
int csim_field_swap(csim_field* a, csim_field* b);                /* std::swap, main.cpp:109 */
/* wavefront-level reductions.  minmax spans the whole array, ghosts included, like
 * reference src/main.cpp:73-77; sum/linf span the interior. */
int csim_field_minmax(const csim_field* f, double out_min_max[2]);
int csim_field_sum(const csim_field* f, double* out);
int csim_field_linf_diff(const csim_field* a, const csim_field* b, double* out);

/* ---- the operators at the reference's own granularity ------------------------------------ */
/* reference src/boundary.cpp:12-54  apply_boundary(f, dec, bc, value); is_physical[s] != 0
 * stands for "neighbour on side s is MPI_PROC_NULL". */
int csim_apply_boundary(csim_field* f, const int bc[4], const int is_physical[4], double value);
/* reference src/diffusion.cpp:3-26  diffusion_step(u, out, D, dt) (interior + ring copy) */
int csim_diffusion_step(const csim_field* u, csim_field* out, double D, double dt);
/* reference src/advection.cpp:5-34  advection_step(u, out, vx, vy, dt) (accumulates) */
int csim_advection_step(const csim_field* u, csim_field* out, double vx, double vy, double dt);
/* copy + diffusion_step + advection_step in ONE sweep (reference src/main.cpp:104-107):
 * out := u everywhere, then the fused update on the interior. */
int csim_fused_step(const csim_field* u, csim_field* out, double D, double dt, double vx, double vy);

/* ---- the time loop (reference src/main.cpp:93-118 minus I/O) ------------------------------ */
int csim_stepper_create(const csim_decomp* dec, double dx, double dy, const int bc[4],
                        double bc_value, csim_stepper** out);
int csim_stepper_destroy(csim_stepper* s);
/* multi-GPU: RCCL communicator for the halo exchange (replaces MPI in reference src/halo.cpp).
 * rank 0 calls csim_comm_unique_id, ships the 128 bytes to the other ranks by any means
 * (MPI_Bcast, torch.distributed store, file), then every rank calls csim_stepper_comm_init. */
int csim_comm_unique_id(void* id, size_t nbytes);
int csim_stepper_comm_init(csim_stepper* s, const void* id, size_t nbytes);
/* a second stepper of the SAME rank borrows `owner`'s communicator (small parity cases beside the production tile
 * without paying for another communicator); owner must outlive s, and only one of them may have an exchange in
 * flight at a time (sync one before running the other) */
int csim_stepper_comm_share(csim_stepper* s, csim_stepper* owner);
int csim_stepper_upload(csim_stepper* s, const double* host_with_ghosts);   /* local tile */
int csim_stepper_download(csim_stepper* s, double* host_with_ghosts);
int csim_stepper_download_interior(csim_stepper* s, double* host_ny_by_nx);
/* snapshot without stalling the loop (reference src/io.cpp:402-424 packs + writes inside the step
 * loop, src/main.cpp:96-99): _begin enqueues a device-side copy of the current interior and an
 * asynchronous D2H into a pinned buffer on a third stream and returns at once; keep calling
 * csim_stepper_run, then _wait for the ny_local x nx_local row-major data (valid until the next
 * _begin). */
int csim_stepper_snapshot_begin(csim_stepper* s);
int csim_stepper_snapshot_wait(csim_stepper* s, const double** host_interior);
/* gaussian hotspot written on the device (reference src/init.cpp:12-33) */
int csim_stepper_init_gaussian(csim_stepper* s, double A, double sigma_frac, double xc_frac,
                               double yc_frac);
/* External halo transport (option "external_halo"=1), for callers that keep the reference's
 * MPI exchange (src/halo.cpp:28-46) or any other carrier: pack copies the four edge lines of
 * the current field to host buffers (ny doubles for left/right, nx for bottom/top; entries of
 * physical sides are ignored), unpack stages the lines received from the neighbours for the
 * next csim_stepper_run(.., 1).  One step per run call in this mode. */
int csim_stepper_halo_pack(csim_stepper* s, double* const host_send[4]);
int csim_stepper_halo_unpack(csim_stepper* s, const double* const host_recv[4]);
/* deep-face flavour for csim_stepper_run(.., depth) (ONE fused pass of depth = 2..7 steps) in
 * external mode: directions 0..7 = left, right, bottom, top, bottom-left, bottom-right, top-left,
 * top-right; _neighbors gives the peer rank (or CSIM_NO_NEIGHBOR) and the face length in doubles
 * per direction (depth*(ny+2), depth*(nx+2), depth*depth).  The face packed for direction d must be
 * delivered to peers[d], which unpacks it as coming from the opposite direction (d ^ 1 for
 * sides, 11 - d for corners). */
int csim_stepper_fuse_limit(const csim_stepper* s, int* depth); /* deepest pass available (1 = none) */
int csim_stepper_faces_neighbors(const csim_stepper* s, int depth, int peers[8], int lengths[8]);
int csim_stepper_faces_pack(csim_stepper* s, int depth, double* const host_send[8]);
int csim_stepper_faces_unpack(csim_stepper* s, int depth, const double* const host_recv[8]);
/* reference src/halo.cpp:6-50  exchange_halos(u, dec, comm) on the current field */
int csim_stepper_exchange_halos(csim_stepper* s);
/* nsteps x { exchange_halos; apply_boundary; fused sweep; swap }, enqueued without host syncs;
 * internally up to 7 steps share one pass over HBM; the last pass of a call also leaves the ghost
 * ring the reference would (halos / boundary values of the state before the last step) */
int csim_stepper_run(csim_stepper* s, double D, double dt, double vx, double vy, int nsteps);
/* The pass schedule of csim_stepper_run(nsteps) as pure host arithmetic: depths[k] = time steps the k-th
 * HBM pass advances (the first max_depths of *npasses entries).  smallest_tile = min over the decomposition
 * of the local nx, ny (the face depth cannot exceed it; MAX for one rank), tile_cells = nx * ny of a single-rank
 * stepper (>= 2e8 cells prefer depth 7, below 1.2e7 depth 4, else 6) and 0 for a multi-rank one (every pass
 * carries an exchange: depth 6, or 7 where it saves a pass), fuse = the option "fuse".  It depends on these numbers only, so every rank derives the same
 * schedule without communicating. */
int csim_pass_schedule(int nsteps, int smallest_tile, long tile_cells, int fuse, int* depths, int max_depths,
                       long* npasses);
/* the same for a given arithmetic flavour: diffusion_only != 0 = a run with vx == vy == 0 (see option "fused_2c"), whose
 * sweep does half the arithmetic and is HBM-bound: 7 steps per pass at every tile size (csim_pass_schedule = flavour 0) */
int csim_pass_schedule_for(int nsteps, int smallest_tile, long tile_cells, int fuse, int diffusion_only, int* depths,
                           int max_depths, long* npasses);
/* optional, before a timed loop: the one-off rows-per-chunk trial that the first long
 * csim_stepper_run would otherwise do (option "autotune"), and — with automatic pass depths — a trial for every
 * other depth a pass plan may mix in (4..7; read back as "tuned_rows_4" .. "tuned_rows_7"); does not advance the field */
int csim_stepper_tune(csim_stepper* s, double D, double dt, double vx, double vy);
/* measurement helper: keep this GPU under the stepper's own load for about `seconds` WITHOUT advancing the field
 * and without any communication (whole-tile launches of the multi-step sweep into the scratch buffer, like the
 * trial launches of _tune), returning with the GPU idle not later than `seconds` after the call.  For the wait
 * between a cross-rank barrier and a timed region (reference: the MPI_Wtime bracket of src/main.cpp:94,111 has no
 * such wait): a GPU that idles for milliseconds leaves its sustained power state and runs the first launches of
 * the timed region 5-15 % slower.  On a multi-rank stepper these launches (like the trial launches of _tune) read the
 * deep-halo layers as they are — zero after creation, otherwise the faces of an earlier pass — so their duration is
 * that of a real pass only while those values are finite (a NaN there makes "fused_2c" tiles run twice); the field
 * itself is never affected. */
int csim_stepper_keep_warm(csim_stepper* s, double D, double dt, double vx, double vy, double seconds);
/* waits for everything enqueued.  With an RCCL communicator the wait polls ncclCommGetAsyncError, so a failed
 * exchange returns CSIM_ERR_RCCL instead of hanging (the reference's MPI_Waitall, src/halo.cpp:46, aborts through
 * the MPI error handler); option "sync_timeout_ms" > 0 bounds the wait (CSIM_ERR_TIMEOUT). */
int csim_stepper_sync(csim_stepper* s);
/* bit-identity in one number: sum over the local interior of bits(u) * (0x9E3779B97F4A7C15 + 2 g) mod 2^64, g = the
 * cell's global linear index (y_offset + j) * nx_global + x_offset + i.  The values of all ranks of a decomposition
 * add up (mod 2^64) to the checksum of the same global field on one rank, whatever the process grid. */
int csim_stepper_checksum(csim_stepper* s, unsigned long long* out);
int csim_stepper_minmax(csim_stepper* s, double out_min_max[2]);
int csim_stepper_sum(csim_stepper* s, double* out);
/* tuning / measurement knobs; unknown keys give CSIM_ERR_ARG.  Results never depend on them, with ONE
 * exception that is off by default:
 *   "contract"       0 (default): every cell update is evaluated in the reference's own operation order
 *                    without FMA contraction (src/diffusion.cpp:9-16, src/advection.cpp:13-33) -> results
 *                    are bit-identical to the reference.  1 (opt-in): the same update as the 5-point
 *                    stencil a0 c + aW W + aE E + aS S + aN N in FMA form (5 instead of 15 fp64
 *                    operations per cell); rounding differs by a few ulp per step, L_inf vs the reference
 *                    stays far below the 1e-10 tolerance (tests/test_gpu_contract.py).
 *   "fused_2c"       0/1 (default 1), bit-identical either way: the interior body of the multi-step sweep evaluates
 *                    E - 2c and N - 2c as one fma(-2, c, .) each — 2c is exact, so the result is the reference's —
 *                    and every tile screens the values it loads: if one is so large that some 2c of the pass
 *                    could overflow (or is NaN / Inf), the tile is recomputed with the reference's own operation
 *                    sequence (bit-identical for every non-NaN cell, the same cells NaN; the reference does not define NaN
 *                    payloads).  14 instead of 15 fp64 operations per cell.  "fused_2c_active" (read-only): whether
 *                    the last run's parameters allowed it (growth bound per step, see make_phys)
 *                    With vx == vy == 0 (diffusion only, BASELINE configs[1]) the same screened body also leaves out
 *                    the advection term, whose value is then +-0 (7 instead of 14 operations per cell; a loaded -0
 *                    sends the tile to the reference's sequence, because o + (+0) would turn an o of -0 into +0).
 *                    "diffusion_only_active" (read-only): whether the last run swept that way.  One zero component alone:
 *                    its three operations are left out under the same screen (11 per cell)
 *   "fuse"           time steps per HBM pass: -1 auto (the cheapest split of a run into passes of 2..7 steps, e.g.
 *                    1000 = 166 x 6 + 4, 20 = 7 + 7 + 6), 0/1 off, 2..7 balanced passes of at most that depth
 *   "variant"        single-step kernel family: 0 auto, 1 dpp, 2 lds, 3 naive
 *   "rows_per_chunk" rows one wavefront marches per launch (0 auto), "prefetch" (single-step kernel)
 *   "xcd_swizzle"    0/1 XCD-aware block->tile map; "tail_split" 0/1 (default 1): launches of two or more rounds of
 *                    wavefronts end with the top eighth of the rows in half-height chunks, dispatched last
 *   "overlap"        exchange schedule of a multi-rank run (all bit-identical): 0 serial exchange; 1 frame launch
 *                    first, the NEXT pass's exchange under the bulk launch; 3 frame and bulk in ONE launch: the frame wavefronts publish a flag the comm
 *                    stream waits on (hipStreamWaitValue64), so the exchange starts under the running kernel
 *                    without an event or a second launch (without signal memory: as 1); 4 bulk launch first with
 *                    THIS pass's exchange under it, then the frame launch — no pass of a run, not even the first,
 *                    waits for an unhidden exchange, and only stream order and events are involved; 5 (default) = 4
 *   "relay"          0/1 (default 1), schedule 4: the two streams swap roles every pass, so that the frame launch follows
 *                    the exchange chain, and the next pass's bulk launch the frame launch, on the same stream (no event
 *                    hand-off on the way of the data); "relay_events" 0/1 (default 0, experiment): the relay's cross-stream
 *                    events without the system-scope fence of a default event (+1 % on 20-step calls of the 8-GPU tile;
 *                    left off: what it skips is also what makes the field visible to other agents)
 *   "direct_faces"   0/1 (default 1), schedule 3: the frame wavefronts copy the cells that form the NEXT pass's faces
 *                    straight into the RCCL send buffers before they publish the flag (0: a pack kernel on the
 *                    comm stream does it after the flag)
 *   "frame_rows"     experiment: chunk height of the frame's side strips in a multi-rank pass (0 = default, the 12-14
 *                    rows of the bottom/top bands; measured best)
 *   "external_halo"  0/1 the caller carries the faces (csim_stepper_halo_* / _faces_*)
 *   "sync_timeout_ms" 0 (default): csim_stepper_sync waits as long as it takes; > 0: gives up with CSIM_ERR_TIMEOUT
 *   "test_stall"     test hook: 1 parks the comm stream on a signal value nobody publishes (what a lost flag or a
 *                    dead peer looks like from the host), 0 releases it
 *   "profile"        0 off, k >= 1: HIP events around the sweep launch(es) of every k-th pass
 *                    (csim_stepper_kernel_time)
 *   "autotune"       0/1 (default 1) with rows_per_chunk = 0: the first long run times the candidate
 *                    chunk heights on this GPU (trial launches that do not advance the field) and keeps
 *                    the fastest; "tuned_rows" (read-only) reports it, "last_rows" (read-only) the chunk
 *                    height the most recent fused launch actually used */
int csim_stepper_set_option(csim_stepper* s, const char* key, long value);
int csim_stepper_get_option(const csim_stepper* s, const char* key, long* value);
/* with option "profile"=1: HIP-event time (on the compute stream) and count of the sweep
 * launches since the last reset, per kernel kind: steps_per_launch = 1 selects the single-step
 * kernel, 2..7 the kernels that advance that many time steps per HBM pass */
int csim_stepper_kernel_time(csim_stepper* s, int steps_per_launch, double* total_ms,
                             long* launches);
/* same sampling, multi-rank runs over RCCL with "overlap" = 1: HIP-event time of the comm-stream chain
 * of a pass — packing the next pass's faces, the grouped ncclSend/ncclRecv exchange, unpack and ghost
 * fill — i.e. what the bulk sweep has to hide (replaces the blocking MPI_Waitall of reference
 * src/halo.cpp:46) */
int csim_stepper_comm_time(csim_stepper* s, double* total_ms, long* passes);
int csim_stepper_reset_timers(csim_stepper* s);

#ifdef __cplusplus
}
#endif
#endif /* CSIM_H */

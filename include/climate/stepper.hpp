// include/climate/stepper.hpp — the time loop of reference src/main.cpp:93-118 on the GPU.
//
// Replaces, per step,  exchange_halos(u,dec,comm); apply_boundary(u,dec,bc,0.0);
//                      std::copy(u -> tmp); diffusion_step(u,tmp,D,dt);
//                      advection_step(u,tmp,vx,vy,dt); std::swap(u.data,tmp.data);
// by  stepper.run(D, dt, vx, vy, nsteps)  with the field resident in HBM.  Host `Field`s are
// only touched by upload()/download() (the snapshot point, reference src/io.cpp:402-424).
// Errors surface as std::runtime_error carrying csim_last_error().
#pragma once
#include <algorithm>
#include <stdexcept>
#include <string>
#include <vector>

#include "core.hpp"
#include "csim.h"

namespace climate {

inline void check(int rc) {
    if (rc != CSIM_OK) throw std::runtime_error(std::string("csim: ") + csim_last_error());
}

inline int bc_code(BCType t) {
    return t == BCType::Dirichlet ? CSIM_BC_DIRICHLET : t == BCType::Neumann ? CSIM_BC_NEUMANN : CSIM_BC_PERIODIC;
}

class Stepper {
  public:
    Stepper(const Decomp2D& dec, const BCConfig& bc, double dx, double dy, double bc_value = 0.0) {
        const csim_decomp d = dec.c_abi();
        dec_ = d;
        const int codes[4] = {bc_code(bc.left), bc_code(bc.right), bc_code(bc.bottom), bc_code(bc.top)};
        check(csim_stepper_create(&d, dx, dy, codes, bc_value, &h_));
    }
    ~Stepper() { csim_stepper_destroy(h_); }
    Stepper(const Stepper&) = delete;
    Stepper& operator=(const Stepper&) = delete;

    // multi-GPU: RCCL communicator; `id` = CSIM_UNIQUE_ID_BYTES made by rank 0 (unique_id()) and
    // broadcast by the launcher's means (MPI_Bcast in an MPI build: connect(comm) below)
    static void unique_id(void* id) { check(csim_comm_unique_id(id, CSIM_UNIQUE_ID_BYTES)); }
    void connect_with_id(const void* id) { check(csim_stepper_comm_init(h_, id, CSIM_UNIQUE_ID_BYTES)); }
#ifdef CSIM_WITH_MPI
    void connect(MPI_Comm comm) {
        unsigned char id[CSIM_UNIQUE_ID_BYTES];
        int rank = 0;
        MPI_Comm_rank(comm, &rank);
        if (rank == 0) unique_id(id);
        MPI_Bcast(id, CSIM_UNIQUE_ID_BYTES, MPI_BYTE, 0, comm);
        connect_with_id(id);
    }
#endif

    void upload(const Field& u) { check(csim_stepper_upload(h_, u.data.data())); }
    void download(Field& u) { check(csim_stepper_download(h_, u.data.data())); }
    void download_interior(double* ny_by_nx) { check(csim_stepper_download_interior(h_, ny_by_nx)); }
    // asynchronous snapshot: begin() returns at once, the loop keeps stepping, wait() hands out the
    // ny_local x nx_local interior as of begin() (valid until the next begin())
    void snapshot_begin() { check(csim_stepper_snapshot_begin(h_)); }
    const double* snapshot_wait() {
        const double* p = nullptr;
        check(csim_stepper_snapshot_wait(h_, &p));
        return p;
    }
    void init_gaussian(double A, double sigma_frac, double xc_frac, double yc_frac) {
        check(csim_stepper_init_gaussian(h_, A, sigma_frac, xc_frac, yc_frac));
    }
    void run(double D, double dt, double vx, double vy, int nsteps) {
        check(csim_stepper_run(h_, D, dt, vx, vy, nsteps));
    }
#ifdef CSIM_WITH_MPI
    // `nsteps` reference steps with the faces carried by the caller's MPI (option "external_halo" = 1): what
    // reference src/halo.cpp:28-46 does every step — Irecv / Isend per neighbour, Waitall — but once per
    // fused pass: faces of the pass's depth in 8 directions while at least three steps remain, then single
    // steps with the reference's own 1-cell edge lines, so that the ghost ring left behind is the
    // reference's.  Every rank derives the same schedule from csim_stepper_fuse_limit.
    void advance_mpi(MPI_Comm comm, double D, double dt, double vx, double vy, int nsteps) {
        int depth = 1;
        check(csim_stepper_fuse_limit(h_, &depth));
        int remaining = nsteps;
        std::vector<double> sb[8], rb[8];
        MPI_Request rq[16];
        while (remaining >= 3 && depth >= 2) {
            const int t = std::min(depth, remaining - 1);
            int peers[8], lens[8];
            check(csim_stepper_faces_neighbors(h_, t, peers, lens));
            double* sp[8];
            const double* rp[8];
            for (int d = 0; d < 8; ++d) {
                sb[d].resize(peers[d] >= 0 ? static_cast<size_t>(lens[d]) : 0);
                rb[d].resize(sb[d].size());
                sp[d] = peers[d] >= 0 ? sb[d].data() : nullptr;
                rp[d] = peers[d] >= 0 ? rb[d].data() : nullptr;
            }
            check(csim_stepper_faces_pack(h_, t, sp));
            int nr = 0;
            for (int d = 0; d < 8; ++d)
                if (peers[d] >= 0) {  // the face that left the peer in the opposite direction arrives here
                    const int opp = d < 4 ? (d ^ 1) : 11 - d;
                    MPI_Irecv(rb[d].data(), lens[d], MPI_DOUBLE, peers[d], 200 + opp, comm, &rq[nr++]);
                    MPI_Isend(sb[d].data(), lens[d], MPI_DOUBLE, peers[d], 200 + d, comm, &rq[nr++]);
                }
            MPI_Waitall(nr, rq, MPI_STATUSES_IGNORE);
            check(csim_stepper_faces_unpack(h_, t, rp));
            run(D, dt, vx, vy, t);
            remaining -= t;
        }
        // depth-1 lines: ny doubles left/right, nx bottom/top (src/halo.cpp:12-18 spans), to the four side peers
        int peers1[4] = {CSIM_NO_NEIGHBOR, CSIM_NO_NEIGHBOR, CSIM_NO_NEIGHBOR, CSIM_NO_NEIGHBOR};
        {
            csim_msg sends[8], recvs[8];
            int ns = 0, nrv = 0;
            check(csim_exchange_plan(&dec_, 1, sends, &ns, recvs, &nrv));
            for (int k = 0; k < ns; ++k) peers1[sends[k].dir] = sends[k].peer;
        }
        while (remaining > 0) {
            double* sp[4];
            const double* rp[4];
            for (int d = 0; d < 4; ++d) {
                const size_t n = peers1[d] >= 0 ? static_cast<size_t>(d < 2 ? dec_.ny_local : dec_.nx_local) : 0;
                sb[d].resize(n);
                rb[d].resize(n);
                sp[d] = n ? sb[d].data() : nullptr;
                rp[d] = n ? rb[d].data() : nullptr;
            }
            check(csim_stepper_halo_pack(h_, sp));
            int nr = 0;
            for (int d = 0; d < 4; ++d)
                if (peers1[d] >= 0) {
                    MPI_Irecv(rb[d].data(), static_cast<int>(rb[d].size()), MPI_DOUBLE, peers1[d], 100 + (d ^ 1), comm, &rq[nr++]);
                    MPI_Isend(sb[d].data(), static_cast<int>(sb[d].size()), MPI_DOUBLE, peers1[d], 100 + d, comm, &rq[nr++]);
                }
            MPI_Waitall(nr, rq, MPI_STATUSES_IGNORE);
            check(csim_stepper_halo_unpack(h_, rp));
            run(D, dt, vx, vy, 1);
            --remaining;
        }
    }
#endif

    // one-off chunking trial on this GPU (otherwise done inside the first long run())
    void tune(double D, double dt, double vx, double vy) { check(csim_stepper_tune(h_, D, dt, vx, vy)); }
    void sync() { check(csim_stepper_sync(h_)); }
    void set_option(const char* key, long v) { check(csim_stepper_set_option(h_, key, v)); }
    void minmax(double& mn, double& mx) {
        double o[2];
        check(csim_stepper_minmax(h_, o));
        mn = o[0];
        mx = o[1];
    }
    double sum() {
        double s = 0;
        check(csim_stepper_sum(h_, &s));
        return s;
    }
    // bit-identity in one number: position-weighted 64-bit checksum of the local interior; the values of all ranks of a
    // decomposition add up (mod 2^64) to the checksum of the same global field on one rank (csim_stepper_checksum)
    unsigned long long checksum() {
        unsigned long long v = 0;
        check(csim_stepper_checksum(h_, &v));
        return v;
    }
    csim_stepper* handle() { return h_; }

  private:
    csim_stepper* h_ = nullptr;
    csim_decomp dec_{};
};

}  // namespace climate

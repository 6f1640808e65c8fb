// main.cpp — `climate_sim_hip`: the reference driver's run (src/main.cpp:23-138) with the time
// loop on the GPU.  Same config surface (--config file.yaml + --key=value overrides), same dt
// clamp and warning, same banner / "IC min/max" / "timing:" lines (scripts/run_benchmark.sh
// parses `timing: total_max=`), same snapshot cadence and file (outputs/snapshots.nc holds the
// state BEFORE step n for every n % out_every == 0; the final state is not written — SURVEY Q6).
// Between snapshots the field never leaves HBM: stepper.run(k) advances k steps without host
// syncs.  Extensions (flags the reference ignores): --no-output, --device-ic, --halo=mpi|rccl, --checksum (prints the
// position-weighted 64-bit checksum of the final global field: the same number on any process grid), and ic.mode=file
// (the reference throws for it).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <iostream>
#include <optional>
#include <string>
#include <vector>

#include "climate/decomp.hpp"
#include "climate/field.hpp"
#include "climate/io.hpp"
#include "climate/snapshot.hpp"
#include "climate/stability.hpp"
#include "climate/stepper.hpp"

namespace fs = std::filesystem;

namespace {

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool has_flag(const std::vector<std::string>& a, const std::string& f) {
    return std::find(a.begin(), a.end(), f) != a.end();
}

// gaussian hotspot / zero presets and the error strings of reference src/init.cpp:12-47
void initial_condition(const Decomp2D& dec, Field& u, const SimConfig& cfg) {
    if (cfg.ic.mode == "preset") {
        if (cfg.ic.preset == "gaussian_hotspot") {
            const double Lx = cfg.nx * cfg.dx, Ly = cfg.ny * cfg.dy;
            const double xc = cfg.ic.xc_frac * Lx, yc = cfg.ic.yc_frac * Ly;
            const double sig = cfg.ic.sigma_frac * std::min(Lx, Ly);
            for (int j = 0; j < u.ny_local; ++j) {
                const double y = (dec.y_offset + j + 0.5) * cfg.dy;
                double* row = &u.data[static_cast<size_t>(j + 1) * u.nx_total() + 1];
                for (int i = 0; i < u.nx_local; ++i) {
                    const double x = (dec.x_offset + i + 0.5) * cfg.dx;
                    const double r2 = (x - xc) * (x - xc) + (y - yc) * (y - yc);
                    row[i] = cfg.ic.A * std::exp(-r2 / (2.0 * sig * sig));
                }
            }
        } else if (cfg.ic.preset != "constant_zero") {
            throw std::runtime_error("Unknown IC preset: " + cfg.ic.preset);
        }
    } else if (cfg.ic.mode == "file") {  // extension: classic NetCDF (y,x) or (time,y,x) double
        // every rank reads only its own block (rows y_offset.., columns x_offset..) row by row into
        // the interior of its Field: per-rank start/count like the reference's writer, src/io.cpp:402-418
        const std::string var = cfg.ic.var.empty() ? "u" : cfg.ic.var;
        int ny = 0, nx = 0;
        netcdf_dims_2d(cfg.ic.path, var, ny, nx);
        if (ny != cfg.ny || nx != cfg.nx) throw std::runtime_error("IC file grid does not match nx/ny");
        read_netcdf_window(cfg.ic.path, var, 0, dec.y_offset, dec.x_offset, u.ny_local, u.nx_local,
                           &u.data[static_cast<size_t>(u.nx_total()) + 1], static_cast<size_t>(u.nx_total()),
                           ny, nx);
    } else {
        throw std::runtime_error("Unknown IC mode: " + cfg.ic.mode);
    }
}

double reduce_max(double v) {
#ifdef CSIM_WITH_MPI
    double r = 0.0;
    MPI_Reduce(&v, &r, 1, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD);
    return r;
#else
    return v;
#endif
}

// wrap-around sum over the ranks: the per-rank checksums of a decomposed field add up to the single-rank value
unsigned long long reduce_sum_u64(unsigned long long v) {
#ifdef CSIM_WITH_MPI
    unsigned long long r = 0;
    MPI_Reduce(&v, &r, 1, MPI_UNSIGNED_LONG_LONG, MPI_SUM, 0, MPI_COMM_WORLD);
    return r;
#else
    return v;
#endif
}

}  // namespace

int main(int argc, char** argv) {
#ifdef CSIM_WITH_MPI
    MPI_Init(&argc, &argv);
#endif
    int world_rank = 0, world_size = 1;
    csim::get_world(MPI_COMM_WORLD, world_size, world_rank);

    std::vector<std::string> args(argv + 1, argv + argc);
    std::optional<std::string> cfg_path;
    for (size_t i = 0; i < args.size(); ++i) {
        if (args[i].rfind("--config=", 0) == 0)
            cfg_path = args[i].substr(9);
        else if (args[i] == "--config" && i + 1 < args.size())
            cfg_path = args[i + 1];
    }
    SimConfig cfg = merged_config(cfg_path, args);

    const double dt_limit = safe_dt(cfg.dx, cfg.dy, cfg.vx, cfg.vy, cfg.D);
    if (cfg.dt > dt_limit) {
        if (world_rank == 0)
            std::cerr << "[warn] dt=" << cfg.dt << " exceeds stability limit " << dt_limit
                      << " -> clamping to dt=" << dt_limit << "\n";
        cfg.dt = dt_limit;
    }
    if (world_rank == 0) {
        std::cout << "climate-sim-mpi-cpp \n"
                  << "  grid: " << cfg.nx << " x " << cfg.ny << "  dt: " << cfg.dt
                  << "  steps: " << cfg.steps << "  D: " << cfg.D << "  v=(" << cfg.vx << "," << cfg.vy << ")\n"
                  << "  bc: left=" << bc_to_string(cfg.bc.left) << " right=" << bc_to_string(cfg.bc.right)
                  << " bottom=" << bc_to_string(cfg.bc.bottom) << " top=" << bc_to_string(cfg.bc.top) << "\n";
    }

    Decomp2D dec;
    dec.init(MPI_COMM_WORLD, cfg.nx, cfg.ny);

    int ndev = 0;
    climate::check(csim_device_count(&ndev));
    if (ndev <= 0) throw std::runtime_error("no HIP device visible (check HIP_VISIBLE_DEVICES): this driver has no CPU path");
    const char* lr = std::getenv("LOCAL_RANK");
    climate::check(csim_set_device(lr ? std::atoi(lr) % ndev : world_rank % ndev));

    const bool no_output = has_flag(args, "--no-output");
    const bool device_ic = has_flag(args, "--device-ic");
    const bool halo_mpi = has_flag(args, "--halo=mpi");
    const bool want_checksum = has_flag(args, "--checksum");
    (void)halo_mpi;

    climate::Stepper st(dec, cfg.bc, cfg.dx, cfg.dy, 0.0);
    if (world_size > 1) {
#ifdef CSIM_WITH_MPI
        if (halo_mpi)
            st.set_option("external_halo", 1);
        else
            st.connect(MPI_COMM_WORLD);
#else
        throw std::runtime_error("multi-rank runs of climate_sim_hip need the -DCSIM_WITH_MPI build");
#endif
    }

    const int halo = 1;
    Field u(dec.nx_local, dec.ny_local, halo, cfg.dx, cfg.dy);
    if (device_ic && cfg.ic.mode == "preset" && cfg.ic.preset == "gaussian_hotspot") {
        st.init_gaussian(cfg.ic.A, cfg.ic.sigma_frac, cfg.ic.xc_frac, cfg.ic.yc_frac);
        double mn = 0, mx = 0;
        st.minmax(mn, mx);
        if (world_rank == 0) std::cout << "IC min/max: " << mn << " / " << mx << "\n";
    } else {
        initial_condition(dec, u, cfg);
        if (world_rank == 0) {
            const double mn = *std::min_element(u.data.begin(), u.data.end());
            const double mx = *std::max_element(u.data.begin(), u.data.end());
            std::cout << "IC min/max: " << mn << " / " << mx << "\n";
        }
        st.upload(u);
    }

    int ncid = 0, varid = 0;
    if (!no_output) {
        if (world_rank == 0) fs::create_directories("outputs");
        if (world_rank == 0) std::cout << "Opening NetCDF file for parallel output\n";
        open_netcdf_parallel("outputs/snapshots.nc", dec, cfg, MPI_COMM_WORLD, ncid, varid);
    }

    st.tune(cfg.D, cfg.dt, cfg.vx, cfg.vy);  // set-up, like the reference's: not part of the timed loop
    st.sync();
    const double t0 = now_s();
    double sum_step = 0.0;
    int time_index = 0;
    int n = 0;
    while (n < cfg.steps) {
        const double ts = now_s();
        // snapshot of the state BEFORE step n: captured on the device now, copied to the host and
        // written to the file while the GPU is already running the next steps
        const bool snap = !no_output && n % cfg.out_every == 0;
        if (snap) st.snapshot_begin();
        int k = no_output ? cfg.steps - n : std::min(cfg.out_every - n % cfg.out_every, cfg.steps - n);
#ifdef CSIM_WITH_MPI
        if (world_size > 1 && halo_mpi)  // reference-style MPI faces, once per fused pass (stepper.hpp)
            st.advance_mpi(MPI_COMM_WORLD, cfg.D, cfg.dt, cfg.vx, cfg.vy, k);
        else
#endif
        st.run(cfg.D, cfg.dt, cfg.vx, cfg.vy, k);  // enqueued, not waited for
        if (snap) write_interior_netcdf(ncid, varid, st.snapshot_wait(), dec, time_index++);
        st.sync();
        n += k;
        sum_step += now_s() - ts;
    }
    if (!no_output) close_netcdf_parallel(ncid);
    const double total = now_s() - t0;

    const double total_max = reduce_max(total);
    const double step_worst = reduce_max(sum_step / std::max(1, cfg.steps));
    if (world_rank == 0) {
        std::cout << "timing: total_max=" << total_max << " s, worst_avg_step=" << step_worst << " s\n";
        const double cells = static_cast<double>(cfg.nx) * cfg.ny * cfg.steps;
        std::cout << "throughput: " << cells / total_max / 1e6 << " Mcell-updates/s, "
                  << cells * 16.0 / total_max / 1e9 << " GB/s algorithmic (16 B/cell-update), "
                  << cells * 16.0 / total_max / 8e12 * 100.0 << " % of 8 TB/s HBM peak, ranks=" << world_size
                  << " dims=" << dec.dims[0] << "x" << dec.dims[1] << "\n";
    }
    if (want_checksum) {
        const unsigned long long cs = reduce_sum_u64(st.checksum());
        if (world_rank == 0) {
            char buf[32];
            std::snprintf(buf, sizeof buf, "0x%016llx", cs);
            std::cout << "checksum: " << buf << " (final field, interior; independent of the process grid)\n";
        }
    }
    dec.finalize();
#ifdef CSIM_WITH_MPI
    MPI_Finalize();
#endif
    return 0;
}

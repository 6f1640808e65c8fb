// hosttool.cpp — `csim_hosttool`: host-only front end of the config and snapshot code, used by
// the CPU test-suite (no GPU needed).
//   csim_hosttool print-config [--config f.yaml] [--key=value ...]   -> one JSON object
//   csim_hosttool nc-write out.nc raw.bin nrec [--config ...] [--nx= --ny= ...]
//                 raw.bin = nrec x ny x nx fp64 (native endian), written through the
//                 reference-named API open_netcdf_parallel / write_field_netcdf / close
//   csim_hosttool nc-read in.nc var step out.bin
//   csim_hosttool nc-attrs in.nc
#include <sys/resource.h>

#include <cstdio>
#include <fstream>
#include <iostream>
#include <optional>
#include <string>
#include <vector>

#include "climate/decomp.hpp"
#include "climate/io.hpp"
#include "climate/snapshot.hpp"

static std::optional<std::string> config_path(const std::vector<std::string>& args) {
    std::optional<std::string> p;
    for (size_t i = 0; i < args.size(); ++i) {
        if (args[i].rfind("--config=", 0) == 0)
            p = args[i].substr(9);
        else if (args[i] == "--config" && i + 1 < args.size())
            p = args[i + 1];
    }
    return p;
}

static std::string jstr(const std::string& s) {
    std::string o = "\"";
    for (char c : s) {
        if (c == '"' || c == '\\') o.push_back('\\');
        o.push_back(c);
    }
    return o + "\"";
}

int main(int argc, char** argv) {
    try {
        if (argc < 2) throw std::runtime_error("usage: csim_hosttool print-config|nc-write|nc-read|nc-read-window|nc-attrs ...");
        const std::string cmd = argv[1];
        std::vector<std::string> args(argv + 2, argv + argc);
        if (cmd == "print-config") {
            const SimConfig c = merged_config(config_path(args), args);
            std::printf("{\"nx\": %d, \"ny\": %d, \"dx\": %.17g, \"dy\": %.17g, \"D\": %.17g, \"vx\": %.17g, "
                        "\"vy\": %.17g, \"dt\": %.17g, \"steps\": %d, \"out_every\": %d, \"bc\": [%s, %s, %s, %s], "
                        "\"output_prefix\": %s, \"ic\": {\"mode\": %s, \"preset\": %s, \"A\": %.17g, "
                        "\"sigma_frac\": %.17g, \"xc_frac\": %.17g, \"yc_frac\": %.17g, \"path\": %s, \"var\": %s}}\n",
                        c.nx, c.ny, c.dx, c.dy, c.D, c.vx, c.vy, c.dt, c.steps, c.out_every,
                        jstr(bc_to_string(c.bc.left)).c_str(), jstr(bc_to_string(c.bc.right)).c_str(),
                        jstr(bc_to_string(c.bc.bottom)).c_str(), jstr(bc_to_string(c.bc.top)).c_str(),
                        jstr(c.output_prefix).c_str(), jstr(c.ic.mode).c_str(), jstr(c.ic.preset).c_str(),
                        c.ic.A, c.ic.sigma_frac, c.ic.xc_frac, c.ic.yc_frac, jstr(c.ic.path).c_str(),
                        jstr(c.ic.var).c_str());
        } else if (cmd == "nc-write") {
            if (args.size() < 3) throw std::runtime_error("nc-write out.nc raw.bin nrec [config args]");
            const SimConfig c = merged_config(config_path(args), args);
            const int nrec = std::stoi(args[2]);
            csim::set_world(1, 0);
            Decomp2D dec;
            dec.init(MPI_COMM_WORLD, c.nx, c.ny);
            std::ifstream in(args[1], std::ios::binary);
            int ncid = 0, varid = 0;
            open_netcdf_parallel(args[0], dec, c, MPI_COMM_WORLD, ncid, varid);
            Field f(c.nx, c.ny, 1, c.dx, c.dy);
            std::vector<double> rec(static_cast<size_t>(c.nx) * c.ny);
            for (int k = 0; k < nrec; ++k) {
                in.read(reinterpret_cast<char*>(rec.data()), static_cast<std::streamsize>(rec.size() * 8));
                if (!in) throw std::runtime_error("short raw input");
                for (int j = 0; j < c.ny; ++j)
                    for (int i = 0; i < c.nx; ++i) f.at(i + 1, j + 1) = rec[static_cast<size_t>(j) * c.nx + i];
                if (!write_field_netcdf(ncid, varid, f, dec, k)) return 1;
            }
            close_netcdf_parallel(ncid);
        } else if (cmd == "nc-read") {
            if (args.size() < 4) throw std::runtime_error("nc-read in.nc var step out.bin");
            int ny = 0, nx = 0;
            std::vector<double> v;
            read_netcdf_2d(args[0], args[1], std::stoi(args[2]), ny, nx, v);
            std::ofstream out(args[3], std::ios::binary);
            out.write(reinterpret_cast<const char*>(v.data()), static_cast<std::streamsize>(v.size() * 8));
            std::printf("%d %d\n", ny, nx);
        } else if (cmd == "nc-read-window") {
            // one rank's block of an IC file, as the driver loads it: into the interior of a (wy+2) x (wx+2)
            // array with a ghost ring; prints the file's extent and by how many KiB the read raised this process's peak RSS
            if (args.size() < 8) throw std::runtime_error("nc-read-window in.nc var step y0 x0 wy wx out.bin");
            const int y0 = std::stoi(args[3]), x0 = std::stoi(args[4]), wy = std::stoi(args[5]), wx = std::stoi(args[6]);
            int ny = 0, nx = 0;
            std::vector<double> v(static_cast<size_t>(wy + 2) * (wx + 2), -7.0);
            struct rusage ru0 {};
            getrusage(RUSAGE_SELF, &ru0);  // peak so far: the process image with its shared libraries
            read_netcdf_window(args[0], args[1], std::stoi(args[2]), y0, x0, wy, wx, &v[static_cast<size_t>(wx + 2) + 1],
                               static_cast<size_t>(wx + 2), ny, nx);
            std::ofstream out(args[7], std::ios::binary);
            out.write(reinterpret_cast<const char*>(v.data()), static_cast<std::streamsize>(v.size() * 8));
            struct rusage ru {};
            getrusage(RUSAGE_SELF, &ru);
            std::printf("%d %d %ld\n", ny, nx, ru.ru_maxrss - ru0.ru_maxrss);
        } else if (cmd == "nc-attrs") {
            for (auto& kv : read_netcdf_attrs(args.at(0))) std::printf("%s=%s\n", kv.first.c_str(), kv.second.c_str());
        } else {
            throw std::runtime_error("unknown command " + cmd);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 3;
    }
    return 0;
}

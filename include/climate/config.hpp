// include/climate/config.hpp — the run configuration with the reference's field names
// (reference include/io.hpp:10-68: ICConfig, SimConfig, CLIOverrides and the loader functions),
// so callers written against the reference keep compiling.  No yaml-cpp: driver/config.cpp has a
// small block/flow YAML subset parser that covers the documents the reference loader accepts
// (configs/dev.yaml and the flat-key form used in its unit tests).
// Precedence: struct defaults < YAML file < command-line overrides; validate() runs after each.
#pragma once
#include <optional>
#include <string>
#include <vector>

#include "core.hpp"

// initial condition: mode "preset" (gaussian_hotspot | constant_zero) or "file" (path, var)
struct ICConfig {
    std::string mode = "preset", preset = "gaussian_hotspot";
    // gaussian hotspot: amplitude, width as a fraction of min(Lx, Ly), centre as fractions of Lx, Ly
    double A = 1.0, sigma_frac = 0.05;
    double xc_frac = 0.5, yc_frac = 0.5;
    std::string path, var;
};

struct SimConfig {
    // grid: global interior cells and spacings
    int nx = 256, ny = 256;
    double dx = 1.0, dy = 1.0;
    // time stepping: dt is clamped to safe_dt() by the driver; snapshot every out_every steps
    double dt = 0.1;
    int steps = 100, out_every = 50;
    // physics: diffusivity and advection velocity
    double D = 0.0, vx = 0.0, vy = 0.0;
    BCConfig bc;
    ICConfig ic{};
    std::string output_prefix = "snap";  // parsed, unused by the reference driver as well

    // throws std::runtime_error("nx/ny must be > 0" | "dx/dy must be > 0" | "dt must be > 0" |
    // "steps must be > 0" | "out_every must be >= 1"), the reference's messages
    void validate() const;
};

// what the command line asked to override; empty optionals leave the YAML/default value alone
struct CLIOverrides {
    std::optional<int> nx, ny, steps, out_every;
    std::optional<double> dx, dy, dt, D, vx, vy;
    std::optional<BCType> bc_left, bc_right, bc_bottom, bc_top;
    std::optional<std::string> output_prefix;
    struct {
        std::optional<std::string> mode, preset, path, format, var;
        std::optional<double> A, sigma_frac, xc_frac, yc_frac;
    } ic;
};

// "dirichlet"|"fixed", "neumann"|"noflux"|"zero-flux", "periodic"|"period" (case-insensitive);
// anything else throws std::runtime_error("Unknown BC type: ...")
BCType bc_from_string(const std::string& s);
std::string bc_to_string(BCType bc);

SimConfig load_yaml_file(const std::string& path);
SimConfig load_yaml_text(const std::string& text);  // extension: the same loader on a string
// `--key=value` and `--key value`; unknown flags are skipped (that includes `--bc=...`)
CLIOverrides parse_cli_overrides(const std::vector<std::string>& args);
SimConfig merged_config(const std::optional<std::string>& yaml_path,
                        const std::vector<std::string>& cli_args);

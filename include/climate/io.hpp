// include/climate/io.hpp — configuration surface, source-compatible with reference
// include/io.hpp:10-68 (SimConfig / ICConfig / CLIOverrides, load_yaml_file,
// parse_cli_overrides, merged_config, bc_from_string / bc_to_string).  yaml-cpp is not needed:
// a small block/flow YAML subset parser covers the documents the reference loader accepts
// (configs/dev.yaml and the flat-key form of tests/simulation/unit/test_io.cpp).
// Snapshot output (NetCDF CDF-5 without PnetCDF): see snapshot.hpp.
#pragma once
#include <optional>
#include <string>
#include <vector>

#include "boundary.hpp"

struct ICConfig {
    std::string mode = "preset";
    std::string preset = "gaussian_hotspot";
    double A = 1.0;
    double sigma_frac = 0.05;
    double xc_frac = 0.5;
    double yc_frac = 0.5;
    std::string path;
    std::string var;
};

struct SimConfig {
    int nx = 256, ny = 256;
    double dx = 1.0, dy = 1.0;
    double D = 0.0;
    double vx = 0.0, vy = 0.0;
    double dt = 0.1;
    int steps = 100;
    int out_every = 50;
    BCConfig bc;
    std::string output_prefix = "snap";
    ICConfig ic{};

    void validate() const;  // throws std::runtime_error with the reference's messages
};

struct CLIOverrides {
    std::optional<int> nx, ny;
    std::optional<double> dx, dy;
    std::optional<double> D, vx, vy;
    std::optional<double> dt;
    std::optional<int> steps, out_every;
    std::optional<BCType> bc_left, bc_right, bc_bottom, bc_top;
    std::optional<std::string> output_prefix;
    struct {
        std::optional<std::string> mode, preset, path, format, var;
        std::optional<double> A, sigma_frac, xc_frac, yc_frac;
    } ic;
};

SimConfig load_yaml_file(const std::string& path);
SimConfig load_yaml_text(const std::string& text);  // extension: same loader on a string
CLIOverrides parse_cli_overrides(const std::vector<std::string>& args);
SimConfig merged_config(const std::optional<std::string>& yaml_path,
                        const std::vector<std::string>& cli_args);

BCType bc_from_string(const std::string& s);
std::string bc_to_string(BCType bc);

// config.cpp — YAML/CLI configuration with the reference's keys, nesting, aliases, precedence
// (defaults < YAML < CLI) and validation messages (reference src/io.cpp:30-376), without
// yaml-cpp.  Known reference quirks are kept on purpose:
//   Q2  `--bc=...` is not a recognised flag and is silently ignored (only --bc.left/right/
//       bottom/top exist); a YAML scalar `bc: periodic` IS honoured.
//   Q4  IC parameters are read directly under `ic:` (ic.A, ic.sigma_frac, ...); a nested
//       `ic.params:` block and `ic.file` are ignored.
#include <algorithm>
#include <cctype>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

#include "climate/config.hpp"

namespace {

std::string lower(std::string s) {
    for (auto& c : s) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    return s;
}

std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace(static_cast<unsigned char>(s[a]))) ++a;
    while (b > a && std::isspace(static_cast<unsigned char>(s[b - 1]))) --b;
    return s.substr(a, b - a);
}

std::string unquote(const std::string& s) {
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\'')))
        return s.substr(1, s.size() - 2);
    return s;
}

// ---- YAML subset: nested block mappings by indentation + one-line flow mappings { a: 1, b: 2 }.
// The document becomes a flat map "grid.nx" -> "512"; a key holding a mapping also gets the
// marker value "\x01map".
using Flat = std::map<std::string, std::string>;
const char* const MAP_MARK = "\x01map";

std::string strip_comment(const std::string& line) {
    bool sq = false, dq = false;
    for (size_t i = 0; i < line.size(); ++i) {
        const char c = line[i];
        if (c == '\'' && !dq) sq = !sq;
        if (c == '"' && !sq) dq = !dq;
        if (c == '#' && !sq && !dq && (i == 0 || std::isspace(static_cast<unsigned char>(line[i - 1]))))
            return line.substr(0, i);
    }
    return line;
}

void parse_flow(const std::string& prefix, const std::string& body, Flat& out) {
    // body is the text between '{' and '}' (no nesting needed for the reference's documents)
    size_t pos = 0;
    while (pos < body.size()) {
        size_t comma = body.find(',', pos);
        if (comma == std::string::npos) comma = body.size();
        const std::string item = trim(body.substr(pos, comma - pos));
        pos = comma + 1;
        if (item.empty()) continue;
        const size_t colon = item.find(':');
        if (colon == std::string::npos) throw std::runtime_error("YAML: expected key: value in flow map");
        out[prefix + trim(item.substr(0, colon))] = unquote(trim(item.substr(colon + 1)));
    }
}

Flat parse_yaml(const std::string& text) {
    Flat out;
    std::vector<std::pair<int, std::string>> stack;  // (indent, prefix incl. trailing '.')
    std::istringstream in(text);
    std::string raw;
    while (std::getline(in, raw)) {
        std::string line = strip_comment(raw);
        if (trim(line).empty() || trim(line) == "---") continue;
        int indent = 0;
        while (indent < static_cast<int>(line.size()) && line[indent] == ' ') ++indent;
        const std::string body = trim(line);
        const size_t colon = body.find(':');
        if (colon == std::string::npos) throw std::runtime_error("YAML: expected 'key: value': " + body);
        while (!stack.empty() && stack.back().first >= indent) stack.pop_back();
        const std::string prefix = stack.empty() ? "" : stack.back().second;
        const std::string key = unquote(trim(body.substr(0, colon)));
        const std::string val = trim(body.substr(colon + 1));
        if (val.empty()) {  // block mapping follows
            out[prefix + key] = MAP_MARK;
            stack.emplace_back(indent, prefix + key + ".");
        } else if (val.front() == '{') {
            const size_t close = val.rfind('}');
            if (close == std::string::npos) throw std::runtime_error("YAML: unterminated flow map");
            out[prefix + key] = MAP_MARK;
            parse_flow(prefix + key + ".", val.substr(1, close - 1), out);
        } else {
            out[prefix + key] = unquote(val);
        }
    }
    return out;
}

int to_int(const std::string& s) {
    size_t n = 0;
    const int v = std::stoi(s, &n);
    if (n != s.size()) throw std::runtime_error("bad integer: " + s);
    return v;
}
double to_dbl(const std::string& s) {
    size_t n = 0;
    const double v = std::stod(s, &n);
    if (n != s.size()) throw std::runtime_error("bad number: " + s);
    return v;
}

struct Doc {
    Flat kv;
    bool has(const std::string& k) const { return kv.count(k) != 0; }
    bool is_map(const std::string& k) const {
        auto it = kv.find(k);
        return it != kv.end() && it->second == MAP_MARK;
    }
    const std::string& get(const std::string& k) const { return kv.at(k); }
    // section.key when `section` exists, otherwise the flat top-level key (reference io.cpp:88-125)
    std::optional<std::string> in(const std::string& section, const std::string& key) const {
        const std::string k = has(section) ? section + "." + key : key;
        auto it = kv.find(k);
        if (it == kv.end() || it->second == MAP_MARK) return std::nullopt;
        return it->second;
    }
};

SimConfig from_doc(const Doc& d) {
    SimConfig c;
    auto geti = [&](const char* sec, const char* key, int& dst) {
        if (auto v = d.in(sec, key)) dst = to_int(*v);
    };
    auto getd = [&](const char* sec, const char* key, double& dst) {
        if (auto v = d.in(sec, key)) dst = to_dbl(*v);
    };
    geti("grid", "nx", c.nx);
    geti("grid", "ny", c.ny);
    getd("grid", "dx", c.dx);
    getd("grid", "dy", c.dy);
    getd("physics", "D", c.D);
    getd("physics", "vx", c.vx);
    getd("physics", "vy", c.vy);
    getd("time", "dt", c.dt);
    geti("time", "steps", c.steps);
    geti("time", "out_every", c.out_every);
    if (d.has("bc")) {
        if (!d.is_map("bc")) {  // scalar: all four sides
            c.bc.left = c.bc.right = c.bc.bottom = c.bc.top = bc_from_string(d.get("bc"));
        } else {
            if (d.has("bc.left")) c.bc.left = bc_from_string(d.get("bc.left"));
            if (d.has("bc.right")) c.bc.right = bc_from_string(d.get("bc.right"));
            if (d.has("bc.bottom")) c.bc.bottom = bc_from_string(d.get("bc.bottom"));
            if (d.has("bc.top")) c.bc.top = bc_from_string(d.get("bc.top"));
        }
    }
    if (d.has("output")) {
        if (d.has("output.prefix")) c.output_prefix = d.get("output.prefix");
    } else if (d.has("output_prefix")) {
        c.output_prefix = d.get("output_prefix");
    }
    if (d.has("ic")) {
        auto s = [&](const char* k, std::string& dst) {
            const std::string key = std::string("ic.") + k;
            if (d.has(key) && !d.is_map(key)) dst = d.get(key);
        };
        auto f = [&](const char* k, double& dst) {
            const std::string key = std::string("ic.") + k;
            if (d.has(key) && !d.is_map(key)) dst = to_dbl(d.get(key));
        };
        s("mode", c.ic.mode);
        s("preset", c.ic.preset);
        f("A", c.ic.A);
        f("sigma_frac", c.ic.sigma_frac);
        f("xc_frac", c.ic.xc_frac);
        f("yc_frac", c.ic.yc_frac);
        s("path", c.ic.path);
        s("var", c.ic.var);
    }
    c.validate();
    return c;
}

}  // namespace

BCType bc_from_string(const std::string& s) {
    const std::string t = lower(s);
    if (t == "dirichlet" || t == "fixed") return BCType::Dirichlet;
    if (t == "neumann" || t == "noflux" || t == "zero-flux") return BCType::Neumann;
    if (t == "periodic" || t == "period") return BCType::Periodic;
    throw std::runtime_error("Unknown BC type: " + s);
}

std::string bc_to_string(BCType bc) {
    return bc == BCType::Neumann ? "neumann" : bc == BCType::Periodic ? "periodic" : "dirichlet";
}

void SimConfig::validate() const {
    if (nx <= 0 || ny <= 0) throw std::runtime_error("nx/ny must be > 0");
    if (dx <= 0 || dy <= 0) throw std::runtime_error("dx/dy must be > 0");
    if (dt <= 0) throw std::runtime_error("dt must be > 0");
    if (steps <= 0) throw std::runtime_error("steps must be > 0");
    if (out_every < 1) throw std::runtime_error("out_every must be >= 1");
}

SimConfig load_yaml_text(const std::string& text) { return from_doc(Doc{parse_yaml(text)}); }

SimConfig load_yaml_file(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("bad file: " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return load_yaml_text(ss.str());
}

// `--key=value` and `--key value`; unknown flags (including `--bc=...`, Q2) are skipped
CLIOverrides parse_cli_overrides(const std::vector<std::string>& args) {
    CLIOverrides o;
    struct Slot {
        const char* key;
        std::optional<int>* i;
        std::optional<double>* d;
        std::optional<std::string>* s;
        std::optional<BCType>* b;
    };
    const Slot slots[] = {
        {"nx", &o.nx, nullptr, nullptr, nullptr},
        {"ny", &o.ny, nullptr, nullptr, nullptr},
        {"dx", nullptr, &o.dx, nullptr, nullptr},
        {"dy", nullptr, &o.dy, nullptr, nullptr},
        {"D", nullptr, &o.D, nullptr, nullptr},
        {"vx", nullptr, &o.vx, nullptr, nullptr},
        {"vy", nullptr, &o.vy, nullptr, nullptr},
        {"dt", nullptr, &o.dt, nullptr, nullptr},
        {"steps", &o.steps, nullptr, nullptr, nullptr},
        {"out_every", &o.out_every, nullptr, nullptr, nullptr},
        {"bc.left", nullptr, nullptr, nullptr, &o.bc_left},
        {"bc.right", nullptr, nullptr, nullptr, &o.bc_right},
        {"bc.bottom", nullptr, nullptr, nullptr, &o.bc_bottom},
        {"bc.top", nullptr, nullptr, nullptr, &o.bc_top},
        {"output.prefix", nullptr, nullptr, &o.output_prefix, nullptr},
        {"output_prefix", nullptr, nullptr, &o.output_prefix, nullptr},
        {"ic.mode", nullptr, nullptr, &o.ic.mode, nullptr},
        {"ic.preset", nullptr, nullptr, &o.ic.preset, nullptr},
        {"ic.A", nullptr, &o.ic.A, nullptr, nullptr},
        {"ic.sigma_frac", nullptr, &o.ic.sigma_frac, nullptr, nullptr},
        {"ic.xc_frac", nullptr, &o.ic.xc_frac, nullptr, nullptr},
        {"ic.yc_frac", nullptr, &o.ic.yc_frac, nullptr, nullptr},
        {"ic.path", nullptr, nullptr, &o.ic.path, nullptr},
        {"ic.var", nullptr, nullptr, &o.ic.var, nullptr},
    };
    for (size_t k = 0; k < args.size(); ++k) {
        const std::string& a = args[k];
        if (a.rfind("--", 0) != 0) continue;
        for (const Slot& sl : slots) {
            const std::string flag = std::string("--") + sl.key;
            std::optional<std::string> val;
            if (a.rfind(flag + "=", 0) == 0)
                val = a.substr(flag.size() + 1);
            else if (a == flag && k + 1 < args.size())
                val = args[k + 1];
            if (!val) continue;
            if (sl.i) *sl.i = std::stoi(*val);
            if (sl.d) *sl.d = std::stod(*val);
            if (sl.s) *sl.s = *val;
            if (sl.b && !val->empty()) *sl.b = bc_from_string(*val);
            break;
        }
    }
    return o;
}

SimConfig merged_config(const std::optional<std::string>& yaml_path,
                        const std::vector<std::string>& cli_args) {
    SimConfig c;
    if (yaml_path && !yaml_path->empty()) c = load_yaml_file(*yaml_path);
    const CLIOverrides o = parse_cli_overrides(cli_args);
    auto put = [](auto& dst, const auto& src) {
        if (src) dst = *src;
    };
    put(c.nx, o.nx);
    put(c.ny, o.ny);
    put(c.dx, o.dx);
    put(c.dy, o.dy);
    put(c.D, o.D);
    put(c.vx, o.vx);
    put(c.vy, o.vy);
    put(c.dt, o.dt);
    put(c.steps, o.steps);
    put(c.out_every, o.out_every);
    put(c.bc.left, o.bc_left);
    put(c.bc.right, o.bc_right);
    put(c.bc.bottom, o.bc_bottom);
    put(c.bc.top, o.bc_top);
    put(c.output_prefix, o.output_prefix);
    put(c.ic.mode, o.ic.mode);
    put(c.ic.preset, o.ic.preset);
    put(c.ic.A, o.ic.A);
    put(c.ic.sigma_frac, o.ic.sigma_frac);
    put(c.ic.xc_frac, o.ic.xc_frac);
    put(c.ic.yc_frac, o.ic.yc_frac);
    put(c.ic.path, o.ic.path);
    c.validate();
    return c;
}

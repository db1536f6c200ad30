// config_reader.hpp — config.json lookup with the reference's exact semantics (main.cpp:29-56, 348-398).
//
// The reference does not parse JSON: it concatenates the file's lines WITHOUT newlines (main.cpp:352-355)
// and, per key, finds the first `"key"` substring, the next ':', skips " \t\n\r", then takes either a
// quoted string or everything up to the first ',' or '}' (main.cpp:357-372).  Consequences kept here on
// purpose so the same config.json drives both programs identically:
//   * nesting ("simulation": {...}) is irrelevant;
//   * booleans are compared with == "true" on the UNTRIMMED token, so `true }` (blank before the brace)
//     reads as false (main.cpp:384-385);
//   * numbers go through stoi/stod, which tolerate trailing blanks (main.cpp:386-391);
//   * a missing file keeps the defaults n=50000, spread=400, mass=1000, theta=1, dt=0.03 (main.cpp:337-345).
#pragma once
#include <filesystem>
#include <fstream>
#include <iostream>
#include <string>

namespace nbody_host {

struct SimConfig {
    bool useBarnesHut = true;
    bool useParallel = true;
    int numBodies = 50000;
    float spread = 400.0f;
    float mass = 1000.0f;
    float theta = 1.0f;
    float dt = 0.03f;
    int windowSize = 1000;
    std::string windowTitle = "N-Body Simulation (BH + AVX/OpenMP)";
    bool loaded = false;
    std::string path;
};

// cwd, its parent, its grandparent — in that order (main.cpp:35-42).
inline std::string find_config_file(const std::string& filename)
{
    namespace fs = std::filesystem;
    fs::path dir = fs::current_path();
    for (int up = 0; up < 3; ++up) {
        const fs::path cand = dir / filename;
        if (fs::exists(cand)) return cand.string();
        dir = dir.parent_path();
    }
    return "";
}

class FlatKeyLookup {
public:
    explicit FlatKeyLookup(std::istream& in)
    {
        std::string line;
        while (std::getline(in, line)) text_ += line;
    }
    explicit FlatKeyLookup(std::string text) : text_(std::move(text)) {}

    // "" when the key (or its ':') is absent
    std::string value(const std::string& key) const
    {
        size_t at = text_.find('"' + key + '"');
        if (at == std::string::npos) return "";
        at = text_.find(':', at);
        if (at == std::string::npos) return "";
        at = text_.find_first_not_of(" \t\n\r", at + 1);
        if (at == std::string::npos) return "";
        if (text_[at] == '"') {
            const size_t close = text_.find('"', at + 1);
            return text_.substr(at + 1, close == std::string::npos ? std::string::npos : close - at - 1);
        }
        const size_t stop = text_.find_first_of(",}", at);
        return text_.substr(at, stop == std::string::npos ? std::string::npos : stop - at);
    }

private:
    std::string text_;
};

inline void apply_lookup(const FlatKeyLookup& kv, SimConfig& c)
{
    std::string s;
    if (!(s = kv.value("useBarnesHut")).empty()) c.useBarnesHut = (s == "true");
    if (!(s = kv.value("useParallel")).empty()) c.useParallel = (s == "true");
    if (!(s = kv.value("numBodies")).empty()) c.numBodies = std::stoi(s);
    if (!(s = kv.value("spread")).empty()) c.spread = (float)std::stod(s);
    if (!(s = kv.value("mass")).empty()) c.mass = (float)std::stod(s);
    if (!(s = kv.value("theta")).empty()) c.theta = (float)std::stod(s);
    if (!(s = kv.value("dt")).empty()) c.dt = (float)std::stod(s);
    if (!(s = kv.value("windowSize")).empty()) c.windowSize = std::stoi(s);
    if (!(s = kv.value("windowTitle")).empty()) c.windowTitle = s;
}

inline SimConfig load_config(const std::string& filename = "config.json", bool verbose = true)
{
    SimConfig c;
    c.path = find_config_file(filename);
    std::ifstream f(c.path);
    if (!c.path.empty() && f.is_open()) {
        apply_lookup(FlatKeyLookup(f), c);
        c.loaded = true;
        if (verbose)
            std::cout << "Config loaded: useBarnesHut=" << c.useBarnesHut << ", useParallel=" << c.useParallel
                      << ", numBodies=" << c.numBodies << std::endl;
    } else if (verbose) {
        std::cerr << "config.json not found. Using defaults." << std::endl;
    }
    return c;
}

}  // namespace nbody_host

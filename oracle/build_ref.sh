#!/usr/bin/env bash
# TEST INFRASTRUCTURE — builds oracle/_ref/libnbody_ref.so from the reference's OWN source.
#
# /root/reference/benchmark.cpp is compiled where it lies: it is streamed through awk
# (patches below) straight into g++'s stdin; no copy of the reference text, patched or not,
# is written anywhere.  Only the compiled library lands in oracle/_ref/ (git-ignored).
# The reference keeps its whole physics inline in an infinite-loop main() with a random seed
# (benchmark.cpp:297-557), so a callable, deterministic entry point needs these line edits;
# every edit asserts the text it expects and the whole file is sha256-pinned, so a different
# reference fails loudly instead of silently mis-patching.
#
# Flags follow the reference's CMakeLists.txt:50-53 / run.txt:8
# (-O3 -ffast-math -mavx2 -fopenmp -DNDEBUG) except -march=native -> -march=x86-64-v3:
# the library is built here and executed on the GPU box's host CPU, which may be a different
# micro-architecture; x86-64-v3 is AVX2+FMA, exactly what the reference's intrinsics use.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF_SRC="${REF_SRC:-/root/reference/benchmark.cpp}"
OUT_DIR="$HERE/_ref"
PIN="f8bd52ad288bfa65f13f7f0df7b926aabb828f8525e24f4ef2484798ee510a8c"

if [ ! -f "$REF_SRC" ]; then
    echo "build_ref: $REF_SRC not present (GPU box?) - keeping prebuilt oracle/_ref" >&2
    exit 0
fi
GOT="$(sha256sum "$REF_SRC" | cut -d' ' -f1)"
if [ "$GOT" != "$PIN" ]; then
    echo "build_ref: reference sha256 $GOT != pinned $PIN" >&2
    exit 1
fi
mkdir -p "$OUT_DIR"

# $1 = output stem, $2 = extra compiler flag ("" or -DREF_COUNT_VOTES: see ref_hooks.h)
build_reference() {
local stem="$1" extra="$2"
awk '
function expect(re) { if ($0 !~ re) { printf("build_ref: line %d does not match /%s/: %s\n", NR, re, $0) > "/dev/stderr"; bad = 1 } }
NR==72  { expect("private:");                 print "public:"; next }
NR==245 { expect("node_stack.pop_back\\(\\)");  print; print "            REF_VISIT(start_index);"; next }
NR==267 { expect("^            \\{");           print; print "                REF_ACCEPT(start_index, current_node_idx);"; next }
NR==297 { expect("^int main\\(\\)");          print "extern \"C\" int ref_benchmark_run(RefIO *p)"; next }
NR==300 { expect("bool useBarnesHut = true"); print "    bool useBarnesHut = p->use_barnes_hut != 0;"; next }
NR==301 { expect("bool useParallel = true");  print "    bool useParallel = p->use_parallel != 0;"; next }
NR==302 { expect("int n = 10000");            print "    int n = p->n;"; next }
NR==303 { expect("float spread = 400");       print "    float spread = p->spread;"; next }
NR==304 { expect("float mass = 1000");        print "    float mass = p->mass;"; next }
NR==305 { expect("float theta = 1");          print "    float theta = p->theta;"; next }
NR==306 { expect("float dt = 0.05");          print "    float dt = p->dt;"; next }
NR==312 { expect("random_device");            print "    std::mt19937 rng(p->seed);"; next }
NR==314 { expect("log.txt");                  print "    std::ofstream logFile(p->log_enabled ? \"log.txt\" : \"/dev/null\", std::ios::app);"; next }
NR==339 { expect("^    }");                   print; print "    REF_AFTER_INIT(data);"; next }
NR==349 { expect("while \\(true\\)");         print "    for (int ref_step = 0; ref_step < p->nsteps; ++ref_step)"; next }
NR==357 { expect("tree.build\\(data, spread\\)"); print; print "            REF_AFTER_BUILD(tree);"; next }
NR==380 { expect("tree.computeForce_vec");    print; print "                    REF_BH_ACC(i, accX_vec, accY_vec);"; next }
NR==478 { expect("^            }");           print; print "            REF_NAIVE_ACC(tmp_accX, tmp_accY);"; next }
NR==513 { expect("physicsEnd = Clock::now");  print; print "        REF_STEP_TIME(physicsStart, physicsEnd);"; next }
NR==534 { expect("elapsedTime >= logInterval"); print "        if (p->log_enabled && elapsedTime >= logInterval)"; next }
NR==559 { expect("logFile.is_open");          print "    REF_AFTER_LOOP(data);"; print; next }
{ print }
END { if (bad) exit 3 }
' "$REF_SRC" | g++ -std=c++17 -O3 -march=x86-64-v3 -ffast-math -mavx2 -mfma -fopenmp -DNDEBUG \
    -fPIC -c $extra -include "$HERE/ref_hooks.h" -x c++ - -o "$OUT_DIR/$stem.o"
# Link WITHOUT -ffast-math: with it GCC links crtfastmath.o, whose constructor flips the whole
# host process (numpy, the test runner) into FTZ/DAZ the moment the library is loaded.
g++ -shared -fopenmp "$OUT_DIR/$stem.o" -o "$OUT_DIR/lib$stem.so" -lpthread
rm -f "$OUT_DIR/$stem.o"
echo "build_ref: built $OUT_DIR/lib$stem.so"

}
# libnbody_ref.so: the reference, unprobed inside its walk (benchmark.cpp:242-276 untouched) - what every timing uses.
# libnbody_ref_counted.so: the same with the per-group vote counters at benchmark.cpp:245/267 - what the vote-parity tests use.
build_reference nbody_ref ""
build_reference nbody_ref_counted "-DREF_COUNT_VOTES"

# ---- drop-in build: the same file with INTEGRATION.md's edit applied (physics -> include/nbody.h on the GPU) ----
# Same compiler flags as above, so that the generator and the harness are the same machine code in both builds.
LIBDIR="$HERE/../n-body_amd/lib"
if [ -f "$LIBDIR/libnbody_amd.so" ]; then
awk '
function expect(re) { if ($0 !~ re) { printf("build_ref(dropin): line %d does not match /%s/: %s\n", NR, re, $0) > "/dev/stderr"; bad = 1 } }
NR==72  { expect("private:");                 print "public:"; next }
NR==297 { expect("^int main\\(\\)");          print "extern \"C\" int ref_benchmark_run_dropin(RefIO *p)"; next }
NR==300 { expect("bool useBarnesHut = true"); print "    bool useBarnesHut = p->use_barnes_hut != 0;"; next }
NR==301 { expect("bool useParallel = true");  print "    bool useParallel = p->use_parallel != 0;"; next }
NR==302 { expect("int n = 10000");            print "    int n = p->n;"; next }
NR==303 { expect("float spread = 400");       print "    float spread = p->spread;"; next }
NR==304 { expect("float mass = 1000");        print "    float mass = p->mass;"; next }
NR==305 { expect("float theta = 1");          print "    float theta = p->theta;"; next }
NR==306 { expect("float dt = 0.05");          print "    float dt = p->dt;"; next }
NR==312 { expect("random_device");            print "    std::mt19937 rng(p->seed);"; next }
NR==314 { expect("log.txt");                  print "    std::ofstream logFile(p->log_enabled ? \"log.txt\" : \"/dev/null\", std::ios::app);"; next }
NR==339 { expect("^    }");                   print; print "    REF_AFTER_INIT(data);"; next }
NR==341 { expect("QuadTree tree;");           print "    REF_DROPIN_CREATE(data)"; next }
NR==342 { expect("tree.init\\(n, spread\\)");  next }
NR==349 { expect("while \\(true\\)");         print "    for (int ref_step = 0; ref_step < p->nsteps; ++ref_step)"; next }
NR==354 { expect("if \\(useBarnesHut\\)");    print "        REF_DROPIN_STEP();"; next }
NR>354 && NR<511 { next }
NR==511 { expect("^        }");               next }
NR==513 { expect("physicsEnd = Clock::now");  print; print "        REF_STEP_TIME(physicsStart, physicsEnd);"; next }
NR==534 { expect("elapsedTime >= logInterval"); print "        if (p->log_enabled && elapsedTime >= logInterval)"; next }
NR==559 { expect("logFile.is_open");          print "    REF_DROPIN_FINISH(data);"; print "    REF_AFTER_LOOP(data);"; print; next }
{ print }
END { if (bad) exit 3 }
' "$REF_SRC" | g++ -std=c++17 -O3 -march=x86-64-v3 -ffast-math -mavx2 -mfma -fopenmp -DNDEBUG \
    -fPIC -c -include "$HERE/dropin_hooks.h" -x c++ - -o "$OUT_DIR/nbody_ref_dropin.o"
g++ -shared -fopenmp "$OUT_DIR/nbody_ref_dropin.o" -o "$OUT_DIR/libnbody_ref_dropin.so" -L"$LIBDIR" -lnbody_amd \
    -Wl,-rpath,'$ORIGIN/../../n-body_amd/lib' -lpthread
rm -f "$OUT_DIR/nbody_ref_dropin.o"
echo "build_ref: built $OUT_DIR/libnbody_ref_dropin.so (reference harness + GPU physics)"
fi

# ---- config reader: main.cpp's own findConfigFile (26-56) and parameter block (337-402), nothing else of that file ----
MAIN_SRC="$(dirname "$REF_SRC")/main.cpp"
MAIN_PIN="e44086afbcd232ec9e00df38a96be942083f7510f039cdf9d363923d04807610"
if [ -f "$MAIN_SRC" ]; then
    GOT="$(sha256sum "$MAIN_SRC" | cut -d' ' -f1)"
    if [ "$GOT" != "$MAIN_PIN" ]; then echo "build_ref: main.cpp sha256 $GOT != pinned $MAIN_PIN" >&2; exit 1; fi
    awk '
    function expect(re) { if ($0 !~ re) { printf("build_ref(config): line %d does not match /%s/: %s\n", NR, re, $0) > "/dev/stderr"; bad = 1 } }
    NR==26  { expect("namespace fs = std::filesystem"); }
    NR==29  { expect("std::string findConfigFile"); }
    NR>=26 && NR<=56 { print; next }
    NR==334 { expect("^int main\\(\\)"); print "extern \"C\" int ref_main_parse_config(const char *filename, RefConfig *out)"; print "{ out->threw = 0; try {"; next }
    NR==337 { expect("bool useBarnesHut = true"); }
    NR==348 { expect("findConfigFile\\(\"config.json\"\\)"); print "    std::string configPath = findConfigFile(filename);"; next }
    NR==402 { expect("^    }"); print; print "    REF_CONFIG_OUT();"; print "    return 0; } catch (...) { out->threw = 1; return 1; } }"; next }
    NR>=336 && NR<402 { print; next }
    END { if (bad) exit 3 }
    ' "$MAIN_SRC" | g++ -std=c++17 -O2 -fPIC -shared -include "$HERE/config_hooks.h" -x c++ - -o "$OUT_DIR/libnbody_ref_config.so"
    echo "build_ref: built $OUT_DIR/libnbody_ref_config.so (main.cpp config reader)"
fi

# ---- main.cpp physics: its own SoA/QuadTree (2-315) and physics block (485-642) in a callable loop ----
if [ -f "$MAIN_SRC" ]; then
    awk '
    function expect(re) { if ($0 !~ re) { printf("build_ref(mainphys): line %d does not match /%s/: %s\n", NR, re, $0) > "/dev/stderr"; bad = 1 } }
    NR==1   { expect("SFML/Graphics.hpp"); next }
    NR==107 { expect("private:"); print "public:"; next }
    NR>=2 && NR<=315 { print; next }
    NR==334 { expect("^int main\\(\\)");
              print "extern \"C\" int ref_main_physics_run(RefIO *p)";
              print "{";
              print "    bool useBarnesHut = p->use_barnes_hut != 0, useParallel = p->use_parallel != 0;";
              print "    int n = p->n; float spread = p->spread, theta = p->theta, dt = p->dt;";
              print "    BodyDataSoA data; data.resize(n);";
              print "    REF_AFTER_INIT(data);";
              print "    QuadTree tree; tree.init(n, spread);";
              print "    for (int ref_step = 0; ref_step < p->nsteps; ++ref_step) {";
              next }
    NR==485 { expect("if \\(useBarnesHut\\)"); }
    NR==488 { expect("tree.build\\(data, spread\\)"); print; print "            REF_AFTER_BUILD(tree);"; next }
    NR==642 { expect("^        }"); print; print "    }"; print "    REF_AFTER_LOOP(data);"; print "    return 0;"; print "}"; next }
    NR>=485 && NR<642 { print; next }
    END { if (bad) exit 3 }
    ' "$MAIN_SRC" | g++ -std=c++17 -O3 -march=x86-64-v3 -ffast-math -mavx2 -mfma -fopenmp -DNDEBUG \
        -fPIC -c -include "$HERE/mainphys_hooks.h" -x c++ - -o "$OUT_DIR/nbody_ref_mainphys.o"
    g++ -shared -fopenmp "$OUT_DIR/nbody_ref_mainphys.o" -o "$OUT_DIR/libnbody_ref_mainphys.so" -lpthread
    rm -f "$OUT_DIR/nbody_ref_mainphys.o"
    echo "build_ref: built $OUT_DIR/libnbody_ref_mainphys.so (main.cpp physics block)"
fi


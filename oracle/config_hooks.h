/* TEST INFRASTRUCTURE — not product code.
 *
 * Frame for the config-reader build of oracle/build_ref.sh: the reference's OWN config.json code —
 * findConfigFile (main.cpp:26-56) and the block that fills the parameter locals (main.cpp:337-402) — streamed
 * from /root/reference/main.cpp through awk into g++ between the two halves of a wrapper function, so that
 * n-body_amd/host/config_reader.hpp can be checked against the reference's machine code on arbitrary files
 * (tests/test_capi_host.py).  The rest of main.cpp needs SFML and is not compiled.
 */
#pragma once
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

extern "C" {
typedef struct RefConfig {
    int useBarnesHut, useParallel, numBodies;
    float spread, mass, theta, dt;
    int windowSize;
    char windowTitle[256];
    int threw;            /* 1: the reference's std::stoi / std::stod threw (it would terminate) */
} RefConfig;
int ref_main_parse_config(const char *filename, RefConfig *out);
}

#define REF_CONFIG_OUT()                                                                               \
    do {                                                                                               \
        out->useBarnesHut = useBarnesHut; out->useParallel = useParallel; out->numBodies = n;          \
        out->spread = spread; out->mass = mass; out->theta = theta; out->dt = dt;                      \
        out->windowSize = windowSize;                                                                  \
        std::strncpy(out->windowTitle, windowTitle.c_str(), sizeof(out->windowTitle) - 1);             \
        out->windowTitle[sizeof(out->windowTitle) - 1] = 0;                                            \
    } while (0)

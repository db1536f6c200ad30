/* TEST INFRASTRUCTURE — not product code.
 *
 * Frame for the main.cpp-physics build of oracle/build_ref.sh.  main.cpp cannot be built as a program here (SFML is
 * absent), but its physics is plain C++/AVX: the recipe streams main.cpp's own data structures and QuadTree
 * (main.cpp:2-315, the SFML include dropped) and its physics block (main.cpp:485-642) from /root/reference through awk
 * into g++, the block wrapped into a callable loop over caller-supplied bodies.  It pins ref_compat = 2 (main.cpp's
 * Barnes-Hut branch lets the N%8 tail bodies drift, main.cpp:535-544) to the reference's machine code.
 */
#pragma once
#include "ref_hooks.h"
extern "C" int ref_main_physics_run(RefIO *p);

#!/usr/bin/env python3
"""Developer tool: price the vector instructions of one kernel's loop from the compiler's assembly, block by block, with
the per-instruction issue costs measured on gfx950 (tools/ubench.hip, tools/ubench_walk.hip: SIMD cycles per wave64
instruction with 8 waves per SIMD).  Usage:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/capi.s n-body_amd/csrc/nbody_capi.hip
    tools/isa_cost.py /tmp/capi.s _ZN5nbody14bh_walk_kernelILb0ELb0EEEvNS_10BhWalkArgsE
Prints, per basic block inside loops, the VALU / SALU / memory instruction counts and the VALU issue cycles."""
import re
import sys

FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32"}
FMA = {"v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32"}
TRANS = {"v_sqrt_f32", "v_rcp_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32"}


def cost(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base in FAST:
        return 2.3
    if base in FMA:
        return 3.6
    if base in TRANS:
        return 8.2
    return 4.2


def main():
    path, sym = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
    block, rows, in_loop = "entry", [], False
    cur = dict(name=block, valu=0, cyc=0.0, salu=0, mem=0, lds=0, loop=False)
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith("s_endpgm"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m or s.startswith("; %bb."):
            rows.append(cur)
            name = m.group(1) if m else s.split()[1]
            cur = dict(name=name, valu=0, cyc=0.0, salu=0, mem=0, lds=0, loop="in Loop" in l or "Loop Header" in l)
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        if op.startswith("v_"):
            cur["valu"] += 1
            cur["cyc"] += cost(op)
        elif op.startswith("s_"):
            cur["salu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cur["mem"] += 1
        elif op.startswith("ds_"):
            cur["lds"] += 1
    rows.append(cur)
    tv = tc = ts = 0
    for r in rows:
        if not r["loop"]:
            continue
        print(f"{r['name']:14s} VALU {r['valu']:3d} = {r['cyc']:6.1f} cyc   SALU {r['salu']:3d}   mem {r['mem']}  lds {r['lds']}")
        tv += r["valu"]; tc += r["cyc"]; ts += r["salu"]
    print(f"all loop blocks: VALU {tv} = {tc:.1f} cycles, SALU {ts}  (rare blocks included: subtract them by hand)")


if __name__ == "__main__":
    main()

"""Developer tool: per-kernel means of one `rocprofv3 --pmc ... --kernel-trace --output-format csv` pass.

    python tools/pmc_summary.py DIR [KERNEL_SUBSTRING ...]

Reads every *counter_collection.csv under DIR and prints, per kernel name (shortened), the number of dispatches, the
mean duration and the mean of every counter per dispatch (rocprofv3 reports one row per dispatch x counter; the value
is the sum over all XCDs/SEs)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)                     # argument list
    name = re.sub(r"^void ", "", name)
    name = name.replace("nbody::", "")
    return name[:96]


def main():
    d = sys.argv[1]
    want = sys.argv[2:]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(f"no counter_collection.csv under {d}")
        return 1
    val = defaultdict(lambda: defaultdict(float))          # kernel -> counter -> sum
    disp = defaultdict(set)
    dur = defaultdict(dict)
    meta = {}
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                if want and not any(w in k for w in want):
                    continue
                did = (f, r["Dispatch_Id"])
                disp[k].add(did)
                val[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if "Start_Timestamp" in r and r.get("End_Timestamp"):
                    dur[k][did] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
                meta[k] = (r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("LDS_Block_Size"), r.get("VGPR_Count"),
                           r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size"))
    for k in sorted(val, key=lambda k: -sum(dur[k].values()) if dur[k] else 0):
        n = len(disp[k])
        us = sum(dur[k].values()) / max(1, len(dur[k])) if dur[k] else float("nan")
        g = meta[k]
        print(f"{k}\n    dispatches {n}  mean {us:.2f} us  grid {g[0]} wg {g[1]} lds {g[2]} vgpr {g[3]} agpr {g[4]} sgpr {g[5]} scratch {g[6]}")
        for c in sorted(val[k]):
            print(f"    {c:28s} {val[k][c] / n:16.1f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())

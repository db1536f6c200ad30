"""Census of the Barnes-Hut accept/open vote: the COMPILED REFERENCE (oracle/_ref, counters exported at
benchmark.cpp:245 / :267 by oracle/ref_hooks.h) against the oracle's IEEE vote, group by group.  CPU only.

    python tools/vote_census.py [N theta]...      (default: the sizes of profiles/r2_vote_census.txt)
"""
import os
import platform
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

O = entry.load_oracle()
args = sys.argv[1:]
cases = [(int(args[i]), float(args[i + 1])) for i in range(0, len(args), 2)] or \
    [(1024, 1.0), (10000, 1.0), (10000, 0.5), (65536, 1.0), (100000, 0.5), (1000000, 1.0), (1000000, 0.5)]
cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][:1]
print(f"host CPU: {cpu[0] if cpu else platform.processor()}")
print("N theta groups ref_visits ref_accepts oracle_visits oracle_accepts groups_differing(visits|accepts|accepted-node-hash) seconds")
for n, theta in cases:
    b = O.init_disc(n, seed=12345)
    t0 = time.time()
    r = O.ref_run(n, use_barnes_hut=1, nsteps=1, bodies=b, theta=theta, count_votes=True)
    v, a, h = O.Tree(n).build(b, 400.0).group_counters(b, theta)
    differ = int(((v != r["grp_visits"]) | (a != r["grp_accepts"]) | (h != r["grp_hash"])).sum())
    print(n, theta, n // 8, int(r["step_visits"][0]), int(r["step_accepts"][0]), int(v.sum()), int(a.sum()), differ,
          f"{time.time() - t0:.1f}", flush=True)

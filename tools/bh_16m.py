"""Developer tool: rehearse BASELINE config 5 (BH theta=0.5, N=2^24 clustered Plummer, 8 GPUs) on ONE GPU by running
rank 0's share (1/8 of the targets; the tree build is replicated on every rank anyway)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
nb = e.load_package(); O = e.load_oracle()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
theta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
srt = "--sorted" in sys.argv          # bh_sorted: compact groups, vote at 0.6 theta (ref_compat 0)
t0 = time.time(); b = O.init_plummer(n, scale=50.0, seed=12345); print(f"plummer init {time.time()-t0:.1f}s", flush=True)
inside = (np.abs(b.posX) <= 400) & (np.abs(b.posY) <= 400)
print(f"bodies inside the root box: {inside.mean()*100:.1f}%", flush=True)
span = nb.shard_range(n, world, 0)
sim = nb.NBodySim(n, theta=theta, useBarnesHut=True, shard=span, **(dict(bh_sorted=True, ref_compat=0) if srt else {})); sim.upload(*b.arrays())
sim.compute_accel(); sim.sync()
sim.profile(True).profile_reset()
for _ in range(3): sim.compute_accel()
sim.sync()
walk, c0 = sim.profile_read(0); build, c2 = sim.profile_read(2)
cnt = sim.bh_counters()
print(f"{'sorted ' if srt else ''}N={n} theta={theta} shard 1/{world}: build {build/c2:.3f} ms | walk {walk/c0:.3f} ms | visits/group {cnt['visits']/(span[1]//8):.0f} accepts/group {cnt['accepts']/(span[1]//8):.0f}", flush=True)
# accuracy vs f64 naive on a few targets of the shard
idx = np.arange(span[0], span[0] + span[1], max(1, span[1] // 64), dtype=np.int32)[:64]
ax, ay = sim.accel()
t0 = time.time(); nx, ny = O.naive_accel_f64_subset(b, idx); print(f"f64 naive on {len(idx)} targets {time.time()-t0:.1f}s", flush=True)
err = np.hypot(ax[idx]-nx, ay[idx]-ny)/np.hypot(nx, ny)
print(f"BH vs f64 naive: median {np.median(err):.4f} p90 {np.percentile(err,90):.4f} max {err.max():.4f}", flush=True)
sim.close()

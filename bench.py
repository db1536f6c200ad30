#!/usr/bin/env python3
"""bench.py — headline benchmark: G body-body interactions/s of the naive O(N^2) step at N = 2^20
(BASELINE.json metric), on 1..8 MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n BODIES] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path: pair-force accumulation of every owned target against all N sources
(self pairs counted, as the reference evaluates them, benchmark.cpp:434) + forward-Euler update + (N>1) the
all-gather of posX/posY.  Inputs are resident in HBM before the timed region.  Total work is fixed as N grows
(strong scaling): the 2^20 bodies are sharded over the ranks.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

PEAK_FP32_TFLOPS = 157.3      # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
FLOP_PER_INTERACTION = 20     # BASELINE.json convention


class ClockSampler:
    """Shader clock and socket power under the benchmark's load, read through the amdsmi library that ships with ROCm
    (gpu_metrics: one gfx clock per XCD).  The chip runs the pair kernel at its ~1.4 kW power cap, so the clock it holds
    — not the kernel — is what moves ms/step between boxes and runs; with it in the JSON line a slow run explains itself.
    On this pool the driver refreshes gpu_metrics ONCE PER SECOND (tools/clock_probe2.py): the thread polls every 10 ms
    and keeps one row per refresh (a change of system_clock_counter), stamped with the time it was first seen.
    Best effort: any failure leaves the clock null."""

    def __init__(self, gpu_index, period_s=0.01):
        self.rows, self.err, self._stop = [], None, False
        self.period = period_s
        try:
            import threading
            for p in ("/opt/rocm/share/amd_smi",):
                if os.path.isdir(p) and p not in sys.path:
                    sys.path.append(p)
            import amdsmi
            self.smi = amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            self.h = hs[gpu_index if gpu_index < len(hs) else 0]
            self._last = self._read()[1]                  # fail here, not in the thread
            self.th = threading.Thread(target=self._run, daemon=True)
        except Exception as e:                            # noqa: BLE001 (instrumentation must never stop the bench)
            self.err, self.th = f"{type(e).__name__}: {e}"[:200], None

    def _read(self):
        m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        clk = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and 0 < c < 10000]
        pw = m.get("current_socket_power")
        act = m.get("average_gfx_activity")
        return (time.perf_counter(), m.get("system_clock_counter"), float(np.mean(clk)) if clk else float("nan"),
                float(pw) if isinstance(pw, (int, float)) else float("nan"),
                float(act) if isinstance(act, (int, float)) else float("nan"))

    def _run(self):
        while not self._stop:
            try:
                r = self._read()
            except Exception as e:                        # noqa: BLE001
                self.err = f"{type(e).__name__}: {e}"[:200]
                return
            if r[1] != self._last:                        # a fresh table
                self._last = r[1]
                self.rows.append(r)
            time.sleep(self.period)

    def start(self):
        if self.th:
            self.th.start()
        return self

    def stop(self, t0, t1, linger_s=0.0):
        """Summary of the refreshes first seen inside [t0, t1] (perf_counter times; the GPU was under this load
        without a break from t0 on)."""
        if linger_s > 0 and self.th and not [r for r in self.rows if t0 + 1.0 <= r[0] <= t1]:
            time.sleep(linger_s)                              # short run: wait for the refresh that covers its end
        self._stop = True
        if self.th:
            self.th.join(timeout=2.0)
        # a refresh describes the second BEFORE it (its average_gfx_activity says how much of that second was busy):
        # keep the ones whose whole second lay under the load
        rows = [r for r in self.rows if t0 + 1.0 <= r[0] <= t1 and r[2] == r[2] and not (r[4] < 99.0)]
        partial = False
        if not rows:
            # a run shorter than two refreshes: take the refresh with the highest clock seen under the load (its second
            # was only partly loaded, so the figure is a lower bound) and say so
            # (the refresh with the highest socket POWER, not the highest clock: the moment the load ends the chip jumps
            # to its 2.4 GHz boost at idle power, which says nothing about the run)
            cand = [r for r in self.rows if t0 <= r[0] <= t1 + 1.05 and r[2] == r[2] and r[2] > 1000.0 and r[3] == r[3] and r[4] >= 30.0]
            if cand:
                rows, partial = [max(cand, key=lambda r: r[3])], True
        if not rows:
            return {"gfx_mhz": None, "note": self.err or "no gpu_metrics refresh under the load (it refreshes once per second): run more steps"}
        clk = np.array([r[2] for r in rows]); pw = np.array([r[3] for r in rows])
        out = {"gfx_mhz": float(np.median(clk)), "gfx_mhz_min": float(clk.min()), "gfx_mhz_max": float(clk.max()),
               "snapshots": len(rows), "partial_second": partial,
               "source": "amdsmi gpu_metrics current_gfxclks, mean over the XCDs; one snapshot per 1-s refresh, kept when its whole second was under this load (gfx activity >= 99 %)"}
        if np.isfinite(pw).any():
            out["socket_power_w"] = float(np.nanmedian(pw))
        return out


def cpu_baseline(budget_s=12.0):
    """The reference's own OpenMP+AVX2 naive loop (oracle/_ref, built from /root/reference/benchmark.cpp by
    oracle/build_ref.sh) timed on this host: N=65536, as many steps as fit the budget (>= 3)."""
    O = entry.load_oracle()
    import ctypes
    try:
        cores = ctypes.CDLL("libgomp.so.1").omp_get_max_threads()
    except OSError:
        cores = os.cpu_count()
    n = 65536
    if O.ref_available():
        b = O.init_disc(n, seed=12345)
        O.ref_run(n, use_barnes_hut=0, nsteps=1, bodies=b)            # warm-up (page faults, thread pool)
        ms = []
        t0 = time.time()
        while len(ms) < 3 or (time.time() - t0 < budget_s and len(ms) < 200):
            r = O.ref_run(n, use_barnes_hut=0, nsteps=2, bodies=b)
            ms.extend(r["step_ms"].tolist())
        med = float(np.median(ms))
        out = dict(value=n * n / (med * 1e-3) / 1e9, unit="G interactions/s", cores=int(cores), kind="reference",
                   sample=f"reference benchmark.cpp naive loop (OpenMP+AVX2, -O3 -ffast-math, x86-64-v3), N={n}, "
                          f"median of {len(ms)} steps, {med:.1f} ms/step; rate is N-independent (compute-bound, "
                          f"SURVEY.md §6.2) so N=2^20 would take ~{med * 256 / 1e3:.0f} s/step")
        try:                                   # SURVEY.md §8d also asks for the one-core figure
            gomp = ctypes.CDLL("libgomp.so.1")
            gomp.omp_set_num_threads(1)
            one = O.ref_run(n, use_barnes_hut=0, nsteps=2, bodies=b)["step_ms"]
            gomp.omp_set_num_threads(int(cores))
            out["single_core_value"] = n * n / (float(np.median(one)) * 1e-3) / 1e9
        except OSError:
            pass
        return out
    # no compiled reference on this box: time the plain-C restatement instead
    b = O.init_disc(16384, seed=12345)
    t0 = time.time()
    O.naive_accel_f32(b)
    dt = time.time() - t0
    return dict(value=16384 ** 2 / dt / 1e9, unit="G interactions/s", cores=int(cores), kind="port",
                sample="oracle/nbody_oracle.c f32 restatement, N=16384, one force pass")


def cpu_baseline_bh(n, theta):
    """The reference's own Barnes-Hut step (oracle/_ref) on this host's threads: same bodies, steps 1-3."""
    O = entry.load_oracle()
    import ctypes
    try:
        cores = ctypes.CDLL("libgomp.so.1").omp_get_max_threads()
    except OSError:
        cores = os.cpu_count()
    if not O.ref_available():
        return None
    b = O.init_disc(n, seed=12345)
    r = O.ref_run(n, use_barnes_hut=1, nsteps=3, bodies=b, theta=theta)
    med = float(np.median(r["step_ms"]))
    return dict(value=med, unit="ms/step", cores=int(cores), kind="reference",
                sample=f"reference benchmark.cpp Barnes-Hut loop (serial build + OpenMP walk), N={n}, theta={theta}, "
                       f"median of steps 1-3")


def self_launch(gpus):
    """`python bench.py --gpus N` from a plain interpreter: run `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a child in its own
    process group, relay its stdout (rank 0's one JSON line) and stderr, return its exit code; on a time limit
    (NBODY_LAUNCH_TIMEOUT, default 900 s) kill exactly that process group.  The ranks arm their own watchdogs."""
    import signal
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    limit = float(os.environ.get("NBODY_LAUNCH_TIMEOUT", "900"))
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)       # stdout/stderr inherited: the line goes straight out
    try:
        return proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        print(f"bench.py: the {gpus}-rank job did not finish within {limit:.0f} s; killing its process group", file=sys.stderr, flush=True)
        try:
            os.killpg(proc.pid, signal.SIGTERM)
            proc.wait(timeout=15)
        except (ProcessLookupError, subprocess.TimeoutExpired):
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        return 124
    except KeyboardInterrupt:
        os.killpg(proc.pid, signal.SIGTERM)
        raise


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)      # 11 steps of 0.18 s: two seconds under load, so that at least one
    ap.add_argument("--warmup", type=int, default=3)     # 1-s gpu_metrics refresh describes a fully loaded second
    ap.add_argument("--bodies", "--n", dest="n", type=int, default=1 << 20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["naive", "bh"], default="naive",
                    help="naive = the BASELINE.json metric (default); bh = Barnes-Hut ms/step (BASELINE configs 4-5)")
    ap.add_argument("--theta", type=float, default=1.0)
    ap.add_argument("--bh-fast-math", action="store_true",
                    help="bh workload: 1-ulp hardware sqrt/rcp in the walk (same votes, terms differ by a few ulp)")
    ap.add_argument("--bh-sorted", action="store_true",
                    help="bh workload: groups of 8 in the tree's own space-filling order, vote at 0.6 theta (ref_compat 0; not the reference's grouping)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # called as plain `python bench.py --gpus N`: start the N ranks ourselves.  Nothing in THIS process has touched
        # the GPU (no torch import yet), and the ranks are fresh children, never an exec of an initialised process.
        raise SystemExit(self_launch(args.gpus))
    import faulthandler
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or os.environ.get("NBODY_DEBUG"):
        # A stalled rank (first contact with RCCL over xGMI, a rendezvous that never completes, the gloo rehearsal)
        # dumps every thread's stack and exits instead of hanging its launcher: armed for EVERY backend when world > 1.
        # Generous while `import torch` pages in on a fresh box (1-2 minutes); re-armed per phase below.
        faulthandler.dump_traceback_later(int(os.environ.get("NBODY_IMPORT_TIMEOUT", "420")), exit=True)

    def watchdog(phase_s=None):
        if world > 1 or os.environ.get("NBODY_DEBUG"):
            faulthandler.cancel_dump_traceback_later()
            if phase_s is not None:
                faulthandler.dump_traceback_later(int(os.environ.get("NBODY_DEBUG_TIMEOUT", str(phase_s))), exit=True)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    watchdog(300)                                  # process group + context + upload (a first RCCL bootstrap on 8 GPUs can take a minute)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    # one rank per GPU; NBODY_DIST_BACKEND=gloo + fewer GPUs than ranks is the single-GPU rehearsal of the N>1 path
    backend = os.environ.get("NBODY_DIST_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    pkg = entry.load_package()
    from nbody_amd.sharded import ShardedSim, owned_targets
    from nbody_amd.initial_conditions import disc
    n = args.n
    bodies = disc(n, seed=12345)             # benchmark.cpp:317-339, seed fixed; (m, posX, posY, velX, velY, radius)

    bh = args.workload == "bh"
    sim = ShardedSim(pkg, n, rank, world, device=local_rank, dt=0.05, dist=dist, useBarnesHut=bh, theta=args.theta,
                     bh_fast_math=bh and args.bh_fast_math, **(dict(bh_sorted=True, ref_compat=0) if bh and args.bh_sorted else {}))
    sim.upload(*bodies)
    my_targets = owned_targets(n, sim.spans, rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sampler = ClockSampler(local_rank).start() if rank == 0 else None
    # warm-up + timed steps: ~0.2 s per naive step at N = 2^20 on one GPU
    watchdog(300 + int(2.0 * (args.steps + args.warmup) * max(1.0, (n / float(1 << 20)) ** 2) / world))
    barrier()
    t_load = time.perf_counter()
    sim.step(args.warmup)
    barrier()
    if bh:
        # The reference's disc is unbound (velocity factor 0.3): within ~10 steps most bodies have left the fixed
        # root box and the walk shrinks (SURVEY.md §6.2, §8d "time steps 1-5 only").  Warm-up therefore runs on the
        # same context (graph captured, clocks up) and the INITIAL state is uploaded again before the timed steps.
        sim.upload_state(*bodies[1:5])
        barrier()
    # naive: per-kernel hipEvent timing stays on inside the timed region (2 launches per step, ~180 ms each).
    # Barnes-Hut: the timed region runs the captured hipGraph (no event pairs inside a capture); the per-kernel
    # figures come from one eager step (step 1 of the same state) AFTER it.
    sim.sim.profile(not bh).profile_reset()
    barrier()
    t0 = time.perf_counter()
    sim.step(args.steps)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    clock = sampler.stop(t_load, t1, linger_s=1.05) if sampler else None
    if bh:
        sim.upload_state(*bodies[1:5])
        sim.sim.profile(True).profile_reset()
        sim.step(1)                    # per-kernel times and the visit counters both describe step 1
        barrier()
    force_ms, force_launches = sim.sim.profile_read(0)
    integ_ms, integ_launches = sim.sim.profile_read(1)
    elapsed = sim.reduce_max(elapsed)

    if rank == 0 and bh:
        build_ms, build_launches = sim.sim.profile_read(2)
        cnt = sim.sim.bh_counters()
        # which walk kernel the library's host code picks (nbody_capi.hip, enqueue_bh_walk): four waves per group up to
        # 1792 ... 5120 groups of 8 targets depending on theta, 8 groups per wave above
        groups = (sim.spans[rank][1] + 7) // 8
        walk_kernel = "bh_walk_quad_kernel" if groups <= 1792 + int((min(max(args.theta, 0.5), 1.0) - 0.5) * 2 * (5120 - 1792)) else "bh_walk_kernel"
        walk_ms = force_ms / max(1, force_launches)
        # algorithmic bytes of one walk launch: one 24-B node record (float4 + int2) per group-visit
        walk_gbs = 24.0 * cnt["visits"] / (walk_ms * 1e-3) / 1e9
        line = {
            "metric": f"ms/step, Barnes-Hut theta={args.theta} step", "value": elapsed * 1e3 / args.steps, "unit": "ms/step",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Barnes-Hut step (tree build + COM + group-of-8 walk + Euler), N={n}, theta={args.theta}, "
                                   f"seeded disc (benchmark.cpp:317-339, seed 12345); timed: steps 1-{args.steps} of that state "
                                   f"(warm-up ran on the same context, then the initial state was uploaded again)",
                       "num_bodies": n, "parallelism": f"tree replicated, walk i-sharded x{world}",
                       "arithmetic": "hardware sqrt/rcp, exact vote" if args.bh_fast_math else "IEEE (bit-identical to the oracle)",
                       "grouping": "bh_sorted: 8 consecutive bodies of the tree's quadrant-path order, vote at 0.6 theta" if args.bh_sorted
                                   else "reference: 8 consecutive body indices (benchmark.cpp:366-380)"},
            "roofline": {"bound": "hbm", "achieved": walk_gbs, "peak": 8000.0, "unit": "GB/s", "frac": walk_gbs / 8000.0,
                         "traffic": None, "kernel": walk_kernel, "kernel_ms": walk_ms, "clock": clock,
                         "build_ms": build_ms / max(1, build_launches), "launches": int(force_launches),
                         "node_visits_step1": int(cnt["visits"]), "accepted_nodes_step1": int(cnt["accepts"]),
                         "note": "tree walk priced in bytes (24 B per group-visit of rank 0's launch, SURVEY.md §8d: no flop "
                                 "convention); counters show it bound by the number of vector instructions at N=1e6 (45 per wave "
                                 "iteration, the vector unit busy 95 % of the cycles, profiles/r3_walk_pmc.txt), so the HBM "
                                 "fraction is a floor, not a target"},
        }
        # The unit that actually bounds the walk: VALU issue.  Counters cannot be read from inside the process, so the
        # instruction count of this launch is its LIVE group-visit count x the VALU instructions per group-visit and the
        # issue cycles per VALU instruction that the committed counter passes measured for that kernel
        # (SQ_INSTS_VALU / visits, 4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU):
        #   bh_walk_kernel       6.08 VALU per group-visit, 4.35 cycles each (profiles/r3_walk_pmc.txt, N = 1e6; 5.98 and
        #                        4.35 at N = 2^24 theta 0.5, same file; before round 3's rewrite of the kernel: 10.3 and 4.2)
        #   bh_walk_quad_kernel  24.9 VALU per group-visit, 4.1 cycles each (profiles/r3_pmc_bh10k_kernels.txt, N = 1e4)
        # against 1024 SIMDs x the clock (amdsmi reading of this run when there is one, else 2.4 GHz).
        vpv, cpi = (24.9, 4.13) if walk_kernel == "bh_walk_quad_kernel" else (6.08, 4.35)
        ghz = (clock.get("gfx_mhz") or 2400.0) / 1e3 if clock else 2.4
        used = cnt["visits"] * vpv * cpi / (walk_ms * 1e-3) / 1e9 if walk_ms > 0 else 0.0
        line["roofline_valu"] = {"bound": "valu-issue", "achieved": used, "peak": 1024 * ghz, "unit": "G SIMD-cycles/s",
                                 "frac": used / (1024 * ghz), "kernel": walk_kernel, "kernel_ms": walk_ms,
                                 "valu_insts_per_group_visit": vpv, "issue_cycles_per_valu_inst": cpi,
                                 "note": "fraction of all SIMD issue cycles spent issuing VALU instructions; per-visit figures are "
                                         "calibrated from committed rocprofv3 --pmc passes, the visit count and the duration are live"}
        if world == 1 and n == 1_000_000 and args.theta == 1.0 and not args.bh_fast_math:
            line["roofline"]["traffic"] = (2 * 1349562.7 + 9565.5) * 1024   # FETCH_SIZE KiB x2 (gfx950) + WRITE_SIZE KiB, per walk launch
            line["roofline"]["traffic_source"] = "profiles/r3_walk_pmc.txt (separate rocprofv3 --pmc passes; not live)"
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_bh(n, args.theta)
        print(json.dumps(line), flush=True)
    elif rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = n * float(n) * args.steps / elapsed / 1e9          # whole job: all N^2 pairs of every step
        k_ms = force_ms / max(1, force_launches)
        achieved = FLOP_PER_INTERACTION * float(my_targets) * n / (k_ms * 1e-3) / 1e12
        info = sim.sim.info()
        k_all = sim.sim.profile_samples(0)
        # The peak assumes 2.4 GHz; the chip holds less at its power cap.  frac_at_clock prices the same launch
        # against the peak AT THE CLOCK IT RAN AT, which separates "slow box" from "slow kernel".
        mhz = clock.get("gfx_mhz") if clock else None
        frac_at_clock = achieved / (PEAK_FP32_TFLOPS * mhz / 2400.0) if mhz and not clock.get("partial_second") else None
        line = {
            "metric": "G body-body interactions/s, N=1M naive step",
            "value": value, "unit": "G interactions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"naive O(N^2) step, N={n}, seeded uniform-(angle,r) disc (benchmark.cpp:317-339, "
                                   f"seed 12345), spread=400 mass=1000 radius=8 dt=0.05",
                       "num_bodies": n, "parallelism": f"i-shard x{world}, all-gather posX/posY per step",
                       "exchange": ("none" if world == 1 else sim.exchange_mode + ":" + backend),
                       "force_variant": info["force_variant"], "j_slices": info["j_slices"]},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS, "traffic": None,
                         "kernel": "force_naive_kernel", "kernel_ms": k_ms, "launches": int(force_launches),
                         "kernel_ms_min": float(k_all.min()) if len(k_all) else None,
                         "kernel_ms_median": float(np.median(k_all)) if len(k_all) else None,
                         "kernel_ms_max": float(k_all.max()) if len(k_all) else None,
                         "clock": clock, "frac_at_measured_clock": frac_at_clock,
                         "note": "FP32 vector (no MFMA: elementwise-transcendental pair loop); algorithmic flop = "
                                 "20 x owned targets x N sources per launch (rank 0); integrate4_kernel "
                                 f"{integ_ms / max(1, integ_launches) * 1e3:.1f} us/launch"},
        }
        # second kernel of the step: j-slice reduction + Euler, HBM-bound.  Algorithmic bytes per owned target:
        # 8 B per slice partial + 16 B (pos, vel) read, 24 B (acc, pos, vel) written (40 B unfused)
        i_ms = integ_ms / max(1, integ_launches)
        i_gbs = (8.0 * info["j_slices"] + 40.0) * my_targets / (i_ms * 1e-3) / 1e9 if i_ms > 0 else 0.0
        line["roofline_integrate"] = {"bound": "hbm", "achieved": i_gbs, "peak": 8000.0, "unit": "GB/s", "frac": i_gbs / 8000.0,
                                      "traffic": None, "kernel": "integrate4_kernel", "kernel_ms": i_ms,
                                      "launches": int(integ_launches),
                                      "note": "hipEvent pairs add ~3 us to a kernel this short: rocprofv3 shows 18.3 us = 6.0 TB/s "
                                              "for the same launch (profiles/r3_naive_n1m_integrate4_kernel_stats.txt; "
                                              "the one-body-per-thread kernel it replaced: 21.1 us)"}
        if world == 1 and n == 1 << 20 and info["j_slices"] == 8:
            # HBM bytes per force launch from the committed PMC passes of this very command (FETCH_SIZE KiB x2 on
            # gfx950 + WRITE_SIZE KiB); counters cannot be read from inside the process, so this is not live
            line["roofline"]["traffic"] = (2 * 32806.3 + 65536.0) * 1024
            line["roofline"]["traffic_source"] = "profiles/r3_naive_n1m_pmc_summary.txt (bytes per launch; algorithmic: 8.4 MB read + 67.1 MB of j-slice partials written)"
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    watchdog(60)
    sim.close()
    if world > 1:
        dist.destroy_process_group()
    watchdog(None)


if __name__ == "__main__":
    main()

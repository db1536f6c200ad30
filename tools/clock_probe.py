"""Which shader-clock read-outs work on the GPU box as an ordinary user, and what do they show while the N = 2^20
force kernel runs?  Exploration tool (round 2): prints every source it can read, sampled every ~20 ms."""
import glob
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402


def sysfs_sources():
    out = {}
    for pat in ("/sys/class/drm/card*/device/pp_dpm_sclk", "/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input",
                "/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input",
                "/sys/class/drm/card*/device/gpu_busy_percent", "/sys/class/drm/card*/device/hwmon/hwmon*/temp1_input"):
        for f in glob.glob(pat):
            try:
                out[f] = open(f).read().strip().replace("\n", " | ")
            except OSError as e:
                out[f] = f"<{e}>"
    return out


def main():
    print("== static ==")
    for k, v in sysfs_sources().items():
        print(k, "->", v[:200])
    for cmd in (["rocm-smi", "--showclocks"], ["amd-smi", "metric", "--clock"], ["amd-smi", "metric", "--power"]):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=30)
            print("$", " ".join(cmd), "rc", r.returncode)
            print((r.stdout + r.stderr)[-1500:])
        except Exception as e:
            print("$", " ".join(cmd), "failed:", e)
    pkg = entry.load_package()
    from nbody_amd.initial_conditions import disc
    n = 1 << 20
    sim = pkg.NBodySim(n, useBarnesHut=False)
    sim.upload(*disc(n, seed=12345))
    samples = []
    stop = [False]
    files = [f for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")] + \
            [f for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")] + \
            [f for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")]

    def sampler():
        while not stop[0]:
            rec = [time.perf_counter()]
            for f in files:
                try:
                    rec.append(int(open(f).read()))
                except Exception:
                    rec.append(-1)
            samples.append(rec)
            time.sleep(0.02)
    th = threading.Thread(target=sampler)
    th.start()
    time.sleep(0.2)
    t0 = time.perf_counter()
    sim.profile(True)
    sim.step(10)
    sim.sync()
    t1 = time.perf_counter()
    time.sleep(0.2)
    stop[0] = True
    th.join()
    ms, k = sim.profile_read(0)
    print(f"== under load: 10 force launches, {ms / k:.2f} ms each; window {t0:.3f}..{t1:.3f}")
    print("files:", files)
    for rec in samples[::4]:
        print(f"{rec[0] - t0:+.3f}", *rec[1:])
    # amd-smi while loaded (one shot)
    sim.step(10)
    try:
        r = subprocess.run(["amd-smi", "metric", "--clock", "--power"], capture_output=True, text=True, timeout=30)
        print(r.stdout[-2500:], r.stderr[-500:])
    except Exception as e:
        print("amd-smi under load failed:", e)
    sim.sync()
    sim.close()


if __name__ == "__main__":
    main()

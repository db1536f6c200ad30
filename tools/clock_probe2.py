"""gpu_metrics cadence under load: dump the fields that change, every 50 ms, while 15 force launches run."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.append("/opt/rocm/share/amd_smi")
import __graft_entry__ as entry
import amdsmi
amdsmi.amdsmi_init()
h = amdsmi.amdsmi_get_processor_handles()[0]
m = amdsmi.amdsmi_get_gpu_metrics_info(h)
print({k: (v if not isinstance(v, list) else v[:8]) for k, v in m.items()})
pkg = entry.load_package()
from nbody_amd.initial_conditions import disc
n = 1 << 20
sim = pkg.NBodySim(n, useBarnesHut=False)
sim.upload(*disc(n, seed=12345))
rows = []; stop = [False]
def run():
    while not stop[0]:
        t = time.perf_counter()
        m = amdsmi.amdsmi_get_gpu_metrics_info(h)
        try:
            ci = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
        except Exception as e:
            ci = str(e)[:60]
        rows.append((t, m.get("system_clock_counter"), m.get("current_gfxclks", [])[:8], m.get("average_gfxclk_frequency"), m.get("current_gfxclk"),
                     m.get("current_socket_power"), m.get("average_socket_power"), m.get("average_gfx_activity"), m.get("firmware_timestamp"), ci))
        time.sleep(0.05)
th = threading.Thread(target=run); th.start()
time.sleep(0.3)
t0 = time.perf_counter()
sim.step(15); sim.sync()
t1 = time.perf_counter()
time.sleep(0.3); stop[0] = True; th.join()
print("load window", 0.0, t1 - t0)
for r in rows:
    print(f"{r[0]-t0:+.3f}", *r[1:])
sim.close()

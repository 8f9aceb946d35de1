"""Does a pre-armed run slow down because the device lowers its clocks (developer tool)?  Blocks of 300
captured config-2 steps, armed / un-armed (argv[1] = 1 / 0), with the shader clock and the socket power
sampled from sysfs by a second thread."""
import glob
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyro_amd as pyro  # noqa: E402
from pyro_amd import examples  # noqa: E402
from pyro_amd.infer import SVI, Trace_ELBO  # noqa: E402
from pyro_amd.infer.autoguide import AutoNormal  # noqa: E402

armed = len(sys.argv) > 1 and sys.argv[1] == "1"
pause = len(sys.argv) > 2 and sys.argv[2] == "1"
dev = torch.device("cuda:0")
X, y = examples.synthetic_logreg_data(1_000_000, 32, dev, seed=1)
pyro.set_rng_seed(0)
pyro.enable_validation(False)
svi = SVI(examples.logreg_model, AutoNormal(examples.logreg_model, init_scale=0.1), pyro.optim.Adam({"lr": 0.01}),
          Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1), hip_graph=True,
          graph_warmup=2, prearm=armed)
for _ in range(10):
    svi.step(X, y)
torch.cuda.synchronize()


def read(path):
    try:
        return open(path).read()
    except OSError:
        return ""


sclk_files = glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")
pow_files = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + \
    glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")
samples, stop = [], False


def sampler():
    while not stop:
        clk = ""
        for f in sclk_files[:1]:
            for ln in read(f).splitlines():
                if "*" in ln:
                    clk = ln.split(":")[1].strip().rstrip("*").strip()
        pw = ""
        for f in pow_files[:1]:
            v = read(f).strip()
            pw = "%.0fW" % (int(v) / 1e6) if v.isdigit() else v
        samples.append((time.perf_counter(), clk, pw))
        time.sleep(0.02)


th = threading.Thread(target=sampler, daemon=True)
th.start()
t_start = time.perf_counter()
blocks = []
for _ in range(24):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        svi.step(X, y)
    if pause and armed:
        svi.pause()
    torch.cuda.synchronize()
    blocks.append((t0 - t_start, (time.perf_counter() - t0) / 300 * 1e6))
stop = True
th.join()
print("armed" if armed else "un-armed", "pause" if pause else "", "sysfs:", sclk_files[:1], pow_files[:1])
for t0, us in blocks:
    near = [s for s in samples if t0 <= s[0] - t_start <= t0 + 0.03]
    print("t=%.3fs  %.1f us/step  %s" % (t0, us, " ".join("%s/%s" % (c, p) for _, c, p in near[:2])))
(e,) = svi._graphs.values()
print("gate", None if e.gate is None else (e.gate.late, e.gate.next), "penalty", e.arm_penalty, "backoff", e.arm_backoff)

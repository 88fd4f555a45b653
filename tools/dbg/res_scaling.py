import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "traffic-env_amd")]
import torch
from gym_traffic import workload as wl
for cfg, envs in (("cfg0", 1), ("cfg1", 1), ("cfg1", 16), ("cfg1", 1024)):
    for n in (1, 10, 50):
        eng = wl.setup_engine(cfg, envs=envs)
        eng.step(n); torch.cuda.synchronize()
        reps = max(3, 400 // n)
        t0 = time.perf_counter()
        for _ in range(reps): eng.step(n)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps * 1e6
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        eng2 = wl.setup_engine(cfg, envs=envs); eng2.step(n); torch.cuda.synchronize()
        ev0.record(); eng2.step(n); ev1.record(); torch.cuda.synchronize()
        print("%s x %d: step(%d) %.1f us wall/call (%.2f us/tick), one call by events %.1f us" % (cfg, envs, n, dt, dt / n, ev0.elapsed_time(ev1) * 1e3), flush=True)

"""Experiment: one handle of E envs against two handles of E/2 envs stepped on two HIP streams (the small per-road
kernels of one half overlap the car pass of the other)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "traffic-env_amd")]
import torch
from gym_traffic import workload as wl

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
E = int(sys.argv[2]) if len(sys.argv) > 2 else wl.CONFIGS[cfg]["envs"]
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
T = 200
dev = torch.device("cuda", 0)
engs = [wl.setup_engine(cfg, device=dev, envs=E // parts, env_id_offset=k * (E // parts)) for k in range(parts)]
streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
def run(n, chunk):
    done = 0
    while done < n:
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.step(chunk)
        done += chunk
for e in engs:
    e.reset_counters()
for chunk in (200, 10, 2):
    run(20, min(chunk, 20)); torch.cuda.synchronize()
    u0 = sum(e.vehicle_updates() for e in engs)
    t0 = time.perf_counter(); run(T, chunk); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    u1 = sum(e.vehicle_updates() for e in engs)
    print("%s: %d handles x %d envs, calls of %d ticks: %.4f ms per tick, %.3e vehicle-updates/s" % (cfg, parts, E // parts, chunk, dt / T * 1e3, (u1 - u0) / dt), flush=True)

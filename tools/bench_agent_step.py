"""Time of one fused agent decision (tfx_agent_step: 10 ticks + remi + observation) at the headline size:
4096 envs of the 16x16 grid with the bench's inputs.  TFX_PAIRS=0 for the tick-by-tick form."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "traffic-env_amd")]
import torch
from gym_traffic import workload as wl

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
E = int(sys.argv[2]) if len(sys.argv) > 2 else wl.CONFIGS[cfg]["envs"]
eng = wl.setup_engine(cfg, device=torch.device("cuda", 0), envs=E)
for _ in range(wl.SETTLE_TICKS.get(cfg, 0) // 50):        # the bench's settle ticks: past the prefill's transient
    eng.step(50)
# (the workload's own inputs: the on-device fixed light cycle and periodic arrivals - no ring overflows, so no env
# stops early and every decision runs all its ticks)
for T in (10,):
    for _ in range(3):
        eng.agent_step(T, remi=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        eng.agent_step(T, remi=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%s x %d envs: %d-tick decision %.3f ms (%.3f ms per tick), %.3e env-decisions/s, %d envs done, "
          "%d of %d env-pairs one tick at a time (k_risk's bound)"
          % (cfg, E, T, dt * 1e3, dt * 1e3 / T, E / dt, int(eng.done.sum()), eng.slow_pairs(), 13 * E * (T // 2)), flush=True)

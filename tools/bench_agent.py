"""Agent-step throughput on the small configurations (launch-bound regime): fused HIP-graph agent
step vs the same sequence launched eagerly vs plain tfx_step(10)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "traffic-env_amd")]
import torch
LAST_DONE = 0.0
from gym_traffic import workload as wl

def run(cfg, envs, mode, steps=18):
    os.environ["TFX_GRAPH"] = "0" if mode == "eager" else "1"
    eng = wl.setup_engine(cfg, envs=envs)
    f = (lambda: eng.step(10)) if mode == "ticks" else (lambda: eng.agent_step(10, remi=True))
    # (the workloads jam after a few hundred ticks and a jammed env stands still for most of a fused
    # decision, so only the first 200 ticks after the prefill are timed - several fresh engines)
    for _ in range(2): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): f()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # (an env that overflows stands still for the rest of its decision, so the fused step gets
    # cheaper as the workload jams: report how many envs were done in the last decision)
    global LAST_DONE
    LAST_DONE = float(eng._adone.float().mean()) if mode != "ticks" else float("nan")
    return dt / steps * 1e6


def best(cfg, envs, mode):
    return min(run(cfg, envs, mode) for _ in range(5))

for cfg, envs in (("cfg0", 1), ("cfg1", 16), ("cfg1", 1024), ("cfg2", 64)):
    r = {}
    for m in ("graph", "eager", "ticks"):
        r[m] = best(cfg, envs, m)
        if m == "graph":
            DONE_G = LAST_DONE
    print("%s envs=%d: agent step (10 ticks) graph %.0f us | eager %.0f us | tfx_step(10) %.0f us | envs done in the last decision %.0f %%"
          % (cfg, envs, r["graph"], r["eager"], r["ticks"], 100 * DONE_G))

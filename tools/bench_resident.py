"""k_res against the per-tick kernels on the small configurations: one fused 10-tick agent decision
(HIP graph) and plain tfx_step(10), same box, first 200 ticks after the prefill (before the
workloads jam).  TFX_RESIDENT / TFX_RES_EPB are read when an engine binds its buffers."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "traffic-env_amd")]
import torch
from gym_traffic import workload as wl


def run(cfg, envs, mode, steps=18):
    eng = wl.setup_engine(cfg, envs=envs)
    f = (lambda: eng.step(10)) if mode == "ticks" else (lambda: eng.agent_step(10, remi=True))
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6, eng.fused_ticks()[0]


def best(cfg, envs, mode):
    rs = [run(cfg, envs, mode) for _ in range(4)]
    return min(r[0] for r in rs), rs[0][1]


cases = [("cfg0", 1), ("cfg1", 16), ("cfg1", 256), ("cfg1", 1024), ("cfg1", 4096)]
if len(sys.argv) > 1:
    cases = [(a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[1:]]
for cfg, envs in cases:
    out = []
    for res, lpr, epb in (("0", "0", "0"), ("1", "3", "0"), ("1", "2", "0"), ("1", "4", "0"), ("1", "1", "0"), ("1", "3", "3"), ("1", "2", "1"), ("1", "4", "1"),
                          ("1", "2", "2"), ("1", "2", "3"), ("1", "1", "4"), ("1", "1", "7")):
        os.environ["TFX_RESIDENT"] = res
        os.environ.pop("TFX_RES_EPB", None)
        os.environ["TFX_RES_LPR"] = lpr if lpr != "0" else "2"
        if epb != "0":
            if int(epb) > envs:
                continue
            os.environ["TFX_RES_EPB"] = epb
        g, fused = best(cfg, envs, "graph")
        t, _ = best(cfg, envs, "ticks")
        out.append("%s: decision %.0f us, step(10) %.0f us" % ("per-tick" if res == "0" else (
            "k_res %s lane%s/road epb=%s" % ("2+4" if lpr == "3" else lpr, "s" if lpr != "1" else "", epb if epb != "0" else "auto")), g, t))
    print("%s x %d envs | " % (cfg, envs) + " | ".join(out), flush=True)

import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "traffic-env_amd")]
import torch
from gym_traffic import workload as wl
cfg, envs, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
eng = wl.setup_engine(cfg, envs=envs)
for _ in range(6):
    eng.step(n)
torch.cuda.synchronize()
print("ok", eng.fused_ticks())

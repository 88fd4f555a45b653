"""A complete RL rollout loop on the device: E batched traffic envs (on-device Poisson arrivals, episodes
restarted on overflow without a host round trip), a small torch policy reading the fused decision's
observation and writing the light actions, one `agent_step` (10 ticks + remi reward) per decision.
Nothing but the loop's Python runs on the host.  Prints decisions/s and env-ticks/s.

    python tools/rollout_demo.py [envs] [m] [n] [capacity] [decisions]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "traffic-env_amd")]
import torch
from gym_traffic.envs.vec_env import TrafficVecEnv

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cap = int(sys.argv[4]) if len(sys.argv) > 4 else 34
N = int(sys.argv[5]) if len(sys.argv) > 5 else 300
venv = TrafficVecEnv(E, m, n, 200.0, capacity=cap, spawn='device', local_cars_per_sec=0.12, seed=0)
eng = venv.engine
venv.reset()
dev = eng.device
torch.manual_seed(0)
policy = torch.nn.Sequential(torch.nn.Linear(2 * eng.r + eng.I, 128), torch.nn.Tanh(), torch.nn.Linear(128, eng.I)).to(dev)
actions = torch.zeros((E, eng.I), dtype=torch.int32, device=dev)
ret = torch.zeros((E,), device=dev)
episodes = torch.zeros((), dtype=torch.int64, device=dev)


def decide(k):
    global ret
    aobs, arew, adone = venv.agent_step(actions, n_ticks=10)
    with torch.no_grad():
        actions.copy_((policy(aobs) > 0).to(torch.int32))
    ret += arew.mean(dim=1)
    episodes.add_(adone.sum())
    venv.reset_done(adone)                  # masked restart on the device; no synchronisation


for k in range(20):
    decide(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(N):
    decide(k)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%d envs %dx%d C=%d, step kernel %s: %d decisions in %.3f s = %.0f env-decisions/s, %.3e env-ticks/s "
      "(%.0f us per batched decision incl. the policy); %d episodes ended by an overflow, mean return %.2f"
      % (E, m, n, cap, eng.step_kernel(), N, dt, E * N / dt, E * N * 10 / dt, dt / N * 1e6, int(episodes), float(ret.mean())))

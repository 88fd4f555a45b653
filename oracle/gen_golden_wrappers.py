#!/usr/bin/env python3
"""Capture golden vectors for the WRAPPER stack from the reference (build container only).

TEST INFRASTRUCTURE.  Two parts, both written to tests/golden/wrappers/ as data (inputs + outputs only):

  wrappers_counter.npz  the reference's HistoryWrapper / StrobeWrapper / LastWrapper / WarmupWrapper
                        (gym_traffic/wrappers/*.py) driven over oracle/fake_env.py's CounterEnv;
  stack_*.npz           the reference's full agent-facing stack - Repeater, WarmupWrapper, Remi,
                        LocalizeWrapper, SquishReward, HistoryWrapper assembled as make_env does
                        (traffic_test.py:27-93) - over the reference TrafficEnv: per decision the
                        observation, reward, done and info['light_times'], plus the episode's
                        trip_times / cars_on_roads ("unfinished", util.py:92).

Usage:  python oracle/gen_golden_wrappers.py
"""
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_loader import load_reference  # noqa: E402
from fake_env import make_counter_env  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "wrappers")

# name -> (wrapper spec, env kwargs, number of steps).  tests/test_wrappers.py rebuilds the same
# cases from this table (stored in the fixture as JSON).
COUNTER_CASES = {
    "hist3": dict(wrap=["history", 3], env=dict(), steps=6),
    "hist2_alias": dict(wrap=["history", 2], env=dict(alias=True), steps=4),
    "warm4": dict(wrap=["warmup", 4], env=dict(), steps=3),
    "last5": dict(wrap=["last", 5], env=dict(done_at=7), steps=3),
    "strobe_6_3_sum": dict(wrap=["strobe", 6, 3, [0, 2]], env=dict(array_limit=True), steps=4),
    "strobe_4_1_nosum": dict(wrap=["strobe", 4, 1, []], env=dict(array_limit=True), steps=3),
    "strobe_6_2_done": dict(wrap=["strobe", 6, 2, [1]], env=dict(array_limit=True, done_at=17), steps=3),
    "strobe_6_2_done_mid": dict(wrap=["strobe", 6, 2, [1]], env=dict(array_limit=True, done_at=8), steps=1),
    "warm2_hist2": dict(wrap=["warmup+history", 2, 2], env=dict(), steps=4),
}

STACKS = {
    # Repeater(10) -> Warmup(2) -> Remi -> History(3): what --warmup_lights 2 --history 3 builds
    "stack_remi_warm_hist": dict(m=3, n=3, L=250.0, C=20, seed=0, lcps=0.12, poisson=True, mode='train',
                                 light_secs=5, warmup_lights=2, remi=True, local_weight=1,
                                 squish_rewards=False, history=3, decisions=7),
    # no Remi: summed env rewards (overflow penalties) + LocalizeWrapper; C=10 overflows early, so
    # the Repeater's `if done: break` is exercised
    "stack_localize_overflow": dict(m=2, n=2, L=250.0, C=10, seed=4, lcps=0.3, poisson=True, mode='train',
                                    light_secs=5, warmup_lights=0, remi=False, local_weight=3,
                                    squish_rewards=False, history=1, decisions=12, actions='const0'),
    "stack_squish_regular": dict(m=2, n=2, L=250.0, C=12, seed=1, lcps=0.25, poisson=False, mode='train',
                                 light_secs=4, warmup_lights=0, remi=True, local_weight=1,
                                 squish_rewards=True, history=1, decisions=14),
    "stack_validate": dict(m=2, n=2, L=250.0, C=20, seed=2, lcps=0.12, poisson=True, mode='validate',
                           light_secs=5, warmup_lights=0, remi=True, local_weight=1,
                           squish_rewards=False, history=1, decisions=12),
}


def build_counter(case, gym, GSpace, W):
    env = make_counter_env(gym, GSpace, **case["env"])
    w = case["wrap"]
    if w[0] == "history":
        return W["history"].HistoryWrapper(w[1])(env)
    if w[0] == "warmup":
        return W["warmup"].WarmupWrapper(w[1])(env)
    if w[0] == "last":
        return W["strobe"].LastWrapper(w[1])(env)
    if w[0] == "strobe":
        return W["strobe"].StrobeWrapper(w[1], w[2], w[3])(env)
    if w[0] == "warmup+history":
        return W["history"].HistoryWrapper(w[2])(W["warmup"].WarmupWrapper(w[1])(env))
    raise KeyError(w[0])


def run_counter(case, gym, GSpace, W, seed=123):
    """-> dict of arrays: reset obs, then per step obs / reward / done (ragged obs kept per step)."""
    np.random.seed(seed)
    env = build_counter(case, gym, GSpace, W)
    out = {"reset": np.array(env.reset())}
    arng = np.random.RandomState(seed + 1)
    for k in range(case["steps"]):
        a = arng.randint(2, size=3).astype(np.int32)
        obs, rew, done, _ = env.step(a)
        out["a%d" % k] = a
        out["obs%d" % k] = np.array(obs)
        out["rew%d" % k] = np.asarray(rew, np.float64)
        out["done%d" % k] = np.array(bool(done))
    return out


def run_stack(sc, mods):
    te = mods["gym_traffic.envs.traffic_env"]
    rg = mods["gym_traffic.envs.roadgraph"]
    gym = mods["gym"]
    args = mods["args"]
    tt = importlib.import_module("traffic_test")          # the reference's driver: Repeater, Remi, ...
    H = importlib.import_module("gym_traffic.wrappers.history")
    Wm = importlib.import_module("gym_traffic.wrappers.warmup")
    te.CAPACITY = int(sc["C"])
    args.update_flags(poisson=bool(sc["poisson"]), rate=0.5, local_cars_per_sec=float(sc["lcps"]),
                      entry='all', learn_switch=False, mode=sc["mode"], light_secs=sc["light_secs"],
                      warmup_lights=sc["warmup_lights"], remi=sc["remi"], local_weight=sc["local_weight"],
                      squish_rewards=sc["squish_rewards"], history=sc["history"], render=False)
    FLAGS = args.FLAGS
    env = gym.make('traffic-v0')
    env.set_graph(rg.GridRoad(sc["m"], sc["n"], sc["L"]))
    env.seed_generator(sc["seed"])
    env.reset_entrypoints()
    # np.empty garbage is not a golden value (see gen_golden.py)
    env.state[:] = 0
    env.rewards[:] = 0
    env.waiting[:] = 0
    env.obs[:] = 0
    env.passed_dst[:] = False
    base = env
    # assembled exactly in make_env's order (traffic_test.py:84-92)
    env = tt.Repeater(FLAGS.light_iterations)(env)
    if FLAGS.warmup_lights > 0:
        env = Wm.WarmupWrapper(FLAGS.warmup_lights)(env)
    if FLAGS.remi:
        env = tt.Remi(env)
    if FLAGS.local_weight > 1:
        env = tt.LocalizeWrapper(env)
    if FLAGS.squish_rewards:
        env = tt.SquishReward(env)
    if FLAGS.history > 1:
        env = H.HistoryWrapper(FLAGS.history)(env)
    np.random.seed(sc["seed"])
    out = {"reset": np.array(env.reset()), "light_iterations": np.int64(FLAGS.light_iterations)}
    arng = np.random.RandomState(sc["seed"] + 1)
    Iq = base.graph.intersections
    lt_all, lt_off = [], [0]
    obs_l, rew_l, done_l, act_l, steps_l, gen_l = [], [], [], [], [], []
    for k in range(sc["decisions"]):
        a = (np.zeros(Iq, np.int32) if sc.get("actions") == 'const0'
             else arng.randint(2, size=Iq).astype(np.int32))
        obs, rew, done, info = env.step(a)
        act_l.append(a)
        obs_l.append(np.array(obs))
        rew_l.append(np.atleast_1d(np.asarray(rew, np.float64)))
        done_l.append(bool(done))
        steps_l.append(float(base.steps))
        gen_l.append(int(base.generated_cars))
        if info:
            lt_all.extend(np.asarray(info['light_times'], np.float64).tolist())
        lt_off.append(len(lt_all))
    out.update(actions=np.stack(act_l), obs=np.stack(obs_l), rewards=np.stack(rew_l),
               done=np.asarray(done_l), env_steps=np.asarray(steps_l), generated_cars=np.asarray(gen_l),
               light_times=np.asarray(lt_all, np.float64), light_off=np.asarray(lt_off, np.int64),
               trip_times=np.asarray(base.trip_times, np.float64),
               unfinished=np.int64(np.sum(base.cars_on_roads())),
               final_leading=np.array(base.leading), final_lastcar=np.array(base.lastcar),
               scenario=np.array(json.dumps(sc)))
    return out


def main():
    mods = load_reference()
    gym = mods["gym"]
    GSpace = mods["gym_traffic.spaces.gspace"].GSpace
    W = {k: importlib.import_module("gym_traffic.wrappers." + k) for k in ("history", "warmup", "strobe")}
    flat = {"cases": np.array(json.dumps(COUNTER_CASES))}
    for name, case in COUNTER_CASES.items():
        for k, v in run_counter(case, gym, GSpace, W).items():
            flat["%s/%s" % (name, k)] = v
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "wrappers_counter.npz"), **flat)
    print("wrappers_counter.npz: %d arrays" % len(flat))
    for name, sc in STACKS.items():
        out = run_stack(sc, mods)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print("%-26s decisions=%d ticks=%d done=%d light_times=%d trips=%d unfinished=%d" % (
            name, sc["decisions"], int(out["env_steps"][-1]), int(out["done"].sum()),
            len(out["light_times"]), len(out["trip_times"]), int(out["unfinished"])))


if __name__ == "__main__":
    main()

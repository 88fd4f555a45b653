#!/usr/bin/env python3
"""Capture golden vectors from the reference TrafficEnv (build container only).

TEST INFRASTRUCTURE.  Runs the reference's own `gym_traffic.envs.TrafficEnv`
(/root/reference/gym_traffic/envs/traffic_env.py:221-394, imported in place through
oracle/ref_loader.py) on a fixed list of scenarios and writes small `.npz` fixtures to
tests/golden/.  A fixture is DATA only: the inputs that drive a run (tables, initial
phase, per-tick actions, per-tick spawn roads) and the outputs the reference produced
(per-tick ring indices, obs, rewards, done, waiting, passed_dst, car x/v/w by slot,
remi rewards, cars_on_roads, trip times).  No reference source text is stored.

Instrumentation (wrappers around reference callables; no reference logic is changed):
  * `add_car` is wrapped to log the entry road of every car spawned by
    `TrafficEnv.add_new_cars` (traffic_env.py:274-283) -> the spawn schedule.
  * `advance_finished_cars` / `advance_hack` are wrapped to snapshot the state
    between `move_cars` and the advance (the "mid" state) for kernel-level parity.

Usage:  python oracle/gen_golden.py [--only NAME]
"""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_loader import load_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

# name -> scenario.  C = CAPACITY (slots per road); cars/road = C-2.
SCENARIOS = {}


def scen(name, **kw):
    d = dict(m=2, n=2, L=250.0, C=20, seed=0, T=300, poisson=True, rate=0.5, lcps=0.12,
             entry='all', learn_switch=False, mode='train', actions='random10',
             remi_every=10, state_every=1, state_next=False, mid=False, archetypes=None)
    d.update(kw)
    SCENARIOS[name] = d


for _s in (0, 1, 2, 3):
    for _p in (True, False):
        for _c in (10, 20):
            scen("g2x2_s%d_%s_c%d" % (_s, "poi" if _p else "reg", _c), seed=_s, poisson=_p, C=_c,
                 mid=(_s == 0))
scen("g2x2_fixedcycle", seed=2, actions='cycle3')
scen("g2x2_learnswitch", seed=3, learn_switch=True)
scen("g2x2_validate", seed=2, mode='validate', T=400)
scen("g2x2_entry_one", seed=1, entry='one', lcps=0.3)
scen("g2x2_const0_jam", seed=4, actions='const0', lcps=0.3, C=12, T=200)
scen("g3x3_default", m=3, n=3, seed=0, T=400)
scen("g3x2_rect", m=3, n=2, seed=5, T=240, C=16, lcps=0.2)
# the two benchmark shapes: car states of ticks 10 j and 10 j + 1 (4x4) / 30 j and 30 j + 1 (16x16), so the reference's
# floats are compared there too (teacher-forced), not only its integers
scen("g4x4_cfg1", m=4, n=4, L=200.0, C=34, seed=0, T=300, lcps=0.3, state_every=10, state_next=True)
scen("g16x16_cfg2_ints", m=16, n=16, L=400.0, C=66, seed=0, T=150, lcps=0.25, state_every=0)
scen("g16x16_cfg2", m=16, n=16, L=400.0, C=66, seed=1, T=181, lcps=0.12, state_every=30, state_next=True)
# CAPACITY = 130 (BASELINE config 5's 128-car roads, traffic_env.py:46-47,202-212 with rings longer than one
# wavefront): one car per entry road every two ticks - more than a signalised road discharges - so all 32 entry roads
# grow past 64 cars (from tick ~130), reach 128, overflow (from tick 206) and wrap their rings (14 000 wrapped
# road-ticks) while they keep handing cars over; car states of ticks 20 j and 20 j + 1 (teacher-forced floats)
scen("g8x8_c130", m=8, n=8, L=800.0, C=130, seed=6, T=700, poisson=False, lcps=1.0, state_every=20, state_next=True)
# More than the one archetype the reference ships (traffic_env.py:35-43 is a TABLE, :164 draws a row per car): the
# default car, a long slow truck and a short quick car with delta = 2.  Rows: v, l, a, delta, v0, b, T, s0.
scen("g2x2_three_archetypes", seed=7, T=300, lcps=0.25,
     archetypes=[[11.11, 4, 3, 4, 13.89, 6, 2, 1], [8.0, 8, 1.5, 4, 10.0, 4, 2.5, 2], [12.0, 3.5, 4, 2, 16.0, 7, 1.5, 1]])
scen("g3x3_two_archetypes_c10", m=3, n=3, C=10, seed=8, T=300, lcps=0.2,
     archetypes=[[11.11, 4, 3, 4, 13.89, 6, 2, 1], [8.0, 8, 1.5, 4, 10.0, 4, 2.5, 2]])
# exponents that are no integers (the reference's `**` takes any float, traffic_env.py:56): delta = 2.5, 4 and 0.75
scen("g2x2_archetypes_fractional_delta", seed=9, T=300, lcps=0.25,
     archetypes=[[11.11, 4, 3, 2.5, 13.89, 6, 2, 1], [8.0, 8, 1.5, 4, 10.0, 4, 2.5, 2], [12.0, 3.5, 4, 0.75, 16.0, 7, 1.5, 1]])
# config 5 itself for the first 130 ticks (integers only): 64x64, Poisson arrivals at the default rate
scen("g64x64_c130_ints", m=64, n=64, L=800.0, C=130, seed=0, T=130, state_every=0)


def run(name, sc, mods):
    te = mods["gym_traffic.envs.traffic_env"]
    rg = mods["gym_traffic.envs.roadgraph"]
    gym = mods["gym"]
    args = mods["args"]
    FLAGS = args.FLAGS

    te.CAPACITY = int(sc["C"])
    args.update_flags(poisson=bool(sc["poisson"]), rate=float(sc["rate"]),
                      local_cars_per_sec=float(sc["lcps"]), entry=sc["entry"],
                      learn_switch=bool(sc["learn_switch"]), mode=sc["mode"])
    C = te.CAPACITY
    xi, vi, wi = te.xi, te.vi, te.wi
    keep_archetypes = te.archetypes
    arch_cols = [te.vi, te.li, te.ai, te.deltai, te.v0i, te.bi, te.ti, te.s0i]
    if sc.get("archetypes"):
        tab = np.zeros((len(sc["archetypes"]), te.params), np.float32)
        tab[:, arch_cols] = np.asarray(sc["archetypes"], np.float32)
        te.archetypes = tab
    spawn_arch_log = []     # archetype row of every car spawned in the current tick

    def arch_of(state):
        """Row of te.archetypes each slot's car was copied from (by its length, accel, ... - everything but x, v, w);
        -1 where no row matches (dead slots, fake leaders)."""
        key = state[:, [te.li, te.ai, te.deltai, te.v0i, te.bi, te.ti, te.s0i], :]          # [R, 7, C]
        rows = te.archetypes[:, [te.li, te.ai, te.deltai, te.v0i, te.bi, te.ti, te.s0i]]   # [n, 7]
        out = np.full((state.shape[0], state.shape[2]), -1, np.int8)
        for a in range(rows.shape[0]):
            out[(key == rows[a][None, :, None]).all(axis=1)] = a
        return out

    spawn_log = []          # roads of cars spawned in the current tick
    in_spawn = [False]
    mid_snap = {}

    orig_add_car = te.add_car
    orig_adv = te.advance_finished_cars
    orig_hack = te.advance_hack

    def logged_add_car(road, car, *a):
        if in_spawn[0]:
            spawn_log.append(int(road))
            rows = te.archetypes[:, arch_cols[1:]]
            spawn_arch_log.append(int(np.argmax((rows == np.asarray(car)[arch_cols[1:]][None, :]).all(axis=1))))
        return orig_add_car(road, car, *a)

    def snap_mid(state, leading, lastcar):
        mid_snap["x"], mid_snap["v"] = live_planes(state, leading, lastcar, C, (xi, vi))

    def logged_adv(dests, length, nexts, state, leading, lastcar, *a):
        snap_mid(state, leading, lastcar)
        return orig_adv(dests, length, nexts, state, leading, lastcar, *a)

    def logged_hack(dests, length, nexts, state, leading, lastcar, *a):
        snap_mid(state, leading, lastcar)
        return orig_hack(dests, length, nexts, state, leading, lastcar, *a)

    te.add_car = logged_add_car
    te.advance_finished_cars = logged_adv
    te.advance_hack = logged_hack
    try:
        env = gym.make('traffic-v0')
        graph = rg.GridRoad(sc["m"], sc["n"], sc["L"])
        env.set_graph(graph)
        env.seed_generator(sc["seed"])
        env.reset_entrypoints()
        orig_add_new = env.add_new_cars

        def add_new(tick):
            in_spawn[0] = True
            try:
                return orig_add_new(tick)
            finally:
                in_spawn[0] = False
        env.add_new_cars = add_new

        np.random.seed(sc["seed"])
        # np.empty garbage in never-written arrays is not a golden value: pin it to 0 so the
        # fixture is reproducible (detected/rewards/waiting are stale-on-purpose in the
        # reference, traffic_env.py:259-272; their *initial* content is undefined there).
        env.state[:] = 0
        env.rewards[:] = 0
        env.reset()
        Iq, r, R = graph.intersections, graph.train_roads, graph.roads
        T = sc["T"]
        arng = np.random.RandomState(sc["seed"] + 1)

        out = dict(
            phases=graph.phases.copy(), dest=graph.dest.copy(), nexts=graph.nexts.copy(),
            entrypoints=graph.entrypoints.copy(), cars_per_sec=np.float64(FLAGS.cars_per_sec),
            init_phase=env.current_phase.copy(),
        )
        actions = np.zeros((T, Iq), np.int32)
        leading = np.zeros((T + 1, R), np.int32)
        lastcar = np.zeros((T + 1, R), np.int32)
        obs = np.zeros((T + 1, 2 * r + 2 * Iq), np.int32)
        rewards = np.zeros((T + 1, Iq), np.float32)
        waiting = np.zeros((T + 1, r), np.int32)
        passed_dst = np.zeros((T + 1, Iq), np.uint8)
        done = np.zeros(T + 1, np.uint8)
        leader_x = np.zeros((T + 1, R), np.float32)
        spawn_off = np.zeros(T + 1, np.int64)
        spawn_road = []
        spawn_arch = []
        se = sc["state_every"]
        st_ticks, st_x, st_v, st_w, st_a, mid_x, mid_v = [], [], [], [], [], [], []
        remi_ticks, remi_rew, cor = [], [], []
        trip_count = np.zeros(T + 1, np.int64)

        def record(k):
            leading[k] = env.leading
            lastcar[k] = env.lastcar
            obs[k] = env.obs
            rewards[k] = env.rewards
            waiting[k] = env.waiting
            passed_dst[k] = env.passed_dst
            leader_x[k] = env.state[np.arange(R), xi, env.leading]
            if se and (k % se == 0 or (sc.get("state_next") and k % se == 1)):
                x, v, w = live_planes(env.state, env.leading, env.lastcar, C, (xi, vi, wi))
                st_ticks.append(k)
                st_x.append(x)
                st_v.append(v)
                st_w.append(w)
                if sc.get("archetypes"):
                    live = (x != 0) | (v != 0) | (w != 0)
                    a = arch_of(env.state)
                    st_a.append(np.where(a >= 0, a, 0).astype(np.int8))

        env.waiting[:] = 0
        record(0)
        cur = None
        for t in range(T):
            if sc["actions"] == 'random10':
                if t % 10 == 0:
                    cur = arng.randint(2, size=Iq).astype(np.int32)
            elif sc["actions"] == 'cycle3':
                cur = np.full(Iq, int((t % 6) >= 3), np.int32)
            elif sc["actions"] == 'const0':
                cur = np.zeros(Iq, np.int32)
            actions[t] = cur
            del spawn_log[:]
            del spawn_arch_log[:]
            # the reference accepts any dtype here (bool from a3c, float64 from const0)
            _, _, d, _ = env.step(cur.copy())
            spawn_road.extend(spawn_log)
            spawn_arch.extend(spawn_arch_log)
            spawn_off[t + 1] = len(spawn_road)
            done[t + 1] = bool(d)
            trip_count[t + 1] = len(env.trip_times)
            if sc["mid"]:
                mid_x.append(mid_snap["x"])
                mid_v.append(mid_snap["v"])
            record(t + 1)
            if sc["remi_every"] and (t + 1) % sc["remi_every"] == 0:
                cor.append(env.cars_on_roads().copy())
                remi_rew.append(env.remi_reward().copy())
                remi_ticks.append(t + 1)
        out.update(actions=actions, leading=leading, lastcar=lastcar, obs=obs, rewards=rewards,
                   waiting=waiting, passed_dst=passed_dst, done=done, leader_x=leader_x,
                   spawn_off=spawn_off, spawn_road=np.asarray(spawn_road, np.int32),
                   generated_cars=np.int64(env.generated_cars),
                   remi_ticks=np.asarray(remi_ticks, np.int64),
                   remi_rewards=np.asarray(remi_rew, np.float32).reshape(len(remi_ticks), Iq),
                   cars_on_roads=np.asarray(cor, np.int32).reshape(len(remi_ticks), sc["m"], sc["n"], 4),
                   trip_times=np.asarray(env.trip_times, np.float64), trip_count=trip_count)
        if se:
            out.update(state_ticks=np.asarray(st_ticks, np.int64), state_x=np.stack(st_x),
                       state_v=np.stack(st_v), state_w=np.stack(st_w))
        if sc.get("archetypes"):
            out.update(archetypes=te.archetypes.copy(), spawn_arch=np.asarray(spawn_arch, np.int32),
                       state_a=np.stack(st_a))
        if sc["mid"]:
            out.update(mid_x=np.stack(mid_x), mid_v=np.stack(mid_v))
        out["scenario"] = np.array(json.dumps(sc))
        out["numpy_version"] = np.array(np.__version__)
        return out
    finally:
        te.add_car = orig_add_car
        te.advance_finished_cars = orig_adv
        te.advance_hack = orig_hack
        te.archetypes = keep_archetypes


def live_planes(state, leading, lastcar, C, planes):
    """Copies of the requested parameter planes [R, C] with every slot that is neither a live
    car nor the fake leader set to 0 (dead slots / slot 0 hold garbage in the reference)."""
    R = state.shape[0]
    slots = np.arange(C)[None, :]
    ld = leading[:, None]
    lc = lastcar[:, None]
    unwrapped = (slots > ld) & (slots <= lc)
    wrapped = (slots > ld) | ((slots >= 1) & (slots <= lc))
    live = np.where(ld <= lc, unwrapped, wrapped) & (ld != lc)
    keep = live | (slots == ld)
    outs = []
    for p in planes:
        a = state[:, p, :].copy()
        a[~keep] = 0
        outs.append(a)
    return outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    mods = load_reference()
    os.makedirs(OUT, exist_ok=True)
    for name, sc in SCENARIOS.items():
        if a.only and a.only != name:
            continue
        out = run(name, sc, mods)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-24s T=%d spawned=%d overflow_ticks=%d trips=%d  %.1f KB" % (
            name, sc["T"], int(out["generated_cars"]), int(out["done"].sum()),
            len(out["trip_times"]), os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()

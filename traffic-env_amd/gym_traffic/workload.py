"""Synthetic fixed-spawn traffic for benchmarks and full-size parity properties.

The recipe is a pure function of (config, env id, tick), so the GPU (on-device rules), the CPU
oracle (explicit schedules built from the same rules) and every rank of a sharded run agree on
the inputs without exchanging anything:
  prefill   every road holds `prefill` cars, head first at x_k = L - gap*(k+1); a car within 40 m of
            the end of a road that is red at t = 0 (phase 0: the N-S blocks) stands still, every
            other car moves at 8 m/s;
  spawns    entry road e receives one car in every tick t with t % spawn_period == e % spawn_period;
  lights    fixed cycle: action = ((t + env_id % light_period) // light_period) & 1 for every
            intersection of env env_id (algorithms/fixed.py:6-7 pattern, staggered over envs).
"""
import numpy as np

CONFIGS = {
    # name: m, n, L, CAPACITY, envs per GPU, prefill, gap
    "cfg0": dict(m=2, n=2, length=250.0, capacity=10, envs=1, prefill=4, gap=8.0),
    "cfg1": dict(m=4, n=4, length=200.0, capacity=34, envs=1024, prefill=24, gap=8.0),
    "cfg2": dict(m=16, n=16, length=400.0, capacity=66, envs=4096, prefill=48, gap=8.0),
    "cfg4": dict(m=64, n=64, length=800.0, capacity=130, envs=16, prefill=96, gap=8.0),
}
SPAWN_PERIOD = 8
LIGHT_PERIOD = 20
# bench.py runs this many untimed ticks right after the prefill, before its warm-up: the prefill's identical platoons
# all brake in the same ticks, and that start-up transient (~25 ticks) is not the traffic the benchmark is about
SETTLE_TICKS = {"cfg2": 100, "cfg4": 100}


def describe(name):
    c = CONFIGS[name]
    return ("%s: %d envs/GPU, %dx%d grid, L=%gm, CAPACITY=%d (%d cars/road max), %d-car prefill, "
            "1 spawn/entry road/%d ticks, %d-tick light cycle" %
            (name, c["envs"], c["m"], c["n"], c["length"], c["capacity"], c["capacity"] - 2,
             c["prefill"], SPAWN_PERIOD, LIGHT_PERIOD))


def prefill_one_env(m, n, length, capacity, prefill, gap):
    """(x, v, leading, lastcar) for ONE env: x, v float32 [R, C]; leading/lastcar int32 [R]."""
    v_cells = m * n
    r = 4 * v_cells
    R = r + 2 * m + 2 * n
    C = int(capacity)
    assert 0 <= prefill <= C - 2
    x = np.zeros((R, C), np.float32)
    v = np.zeros((R, C), np.float32)
    leading = np.ones(R, np.int32)
    lastcar = np.full(R, 1 + prefill, np.int32)
    k = np.arange(prefill, dtype=np.float32)
    pos = (np.float32(length) - np.float32(gap) * (k + 1)).astype(np.float32)
    x[:, 2:2 + prefill] = pos[None, :]
    x[:, 1] = np.inf                                   # fake leader slot (reset value)
    road = np.arange(R)
    red_at_start = (road >= 2 * v_cells) & (road < r)  # phases[e] == current_phase == 0
    near = pos > np.float32(length - 40.0)
    speed = np.full((R, prefill), np.float32(8.0))
    speed[np.ix_(red_at_start, near)] = 0.0
    v[:, 2:2 + prefill] = speed
    return x, v, leading, lastcar


def spawn_roads_for_tick(entrypoints, t, period=SPAWN_PERIOD):
    return [int(e) for e in entrypoints if t % period == int(e) % period]


def cycle_actions(env_ids, n_intersections, t, period=LIGHT_PERIOD):
    a = ((t + (np.asarray(env_ids) % period)) // period) & 1
    return np.repeat(a.astype(np.int32)[:, None], n_intersections, axis=1)


def algorithmic_bytes_per_tick(live_cars, roads, intersections):
    """16 B per vehicle-update (x, v read + write) + 48 B per road-tick + 32 B per
    intersection-tick, summed over the envs of one launch."""
    return 16 * live_cars + 48 * roads + 32 * intersections


def setup_engine(name, device=None, envs=None, env_id_offset=0, planes=2, layout=None, validate=False, trip_cap=4096):
    """A TfxEngine for config `name`, prefilled and switched to the on-device spawn/light rules.
    validate: the reference's `--mode validate` (cars carry their spawn tick, trip times are logged)."""
    import torch
    from gym_traffic.core import TfxEngine
    c = CONFIGS[name]
    E = int(envs if envs is not None else c["envs"])
    eng = TfxEngine(c["m"], c["n"], c["length"], c["capacity"], n_envs=E, rate=0.5, planes=3 if validate else planes,
                    device=device, env_id_offset=env_id_offset, layout=layout, validate=validate, trip_cap=trip_cap)
    eng.reset(np.zeros((1, eng.I), np.int32))
    x, v, leading, lastcar = prefill_one_env(c["m"], c["n"], c["length"], c["capacity"], c["prefill"], c["gap"])
    dev = eng.device
    ring = eng.xv                                   # ring-layout view / staging copy
    ring[..., 0].copy_(torch.as_tensor(x).to(dev)[None].expand_as(ring[..., 0]))
    ring[..., 1].copy_(torch.as_tensor(v).to(dev)[None].expand_as(ring[..., 1]))
    eng.leading[:] = torch.as_tensor(leading).to(dev)[None]
    eng.lastcar[:] = torch.as_tensor(lastcar).to(dev)[None]
    eng.refresh()
    eng.drop_staging()                              # the rollout itself never looks at the cars
    eng.set_spawns(period=SPAWN_PERIOD)
    eng.set_actions(cycle_period=LIGHT_PERIOD)
    return eng

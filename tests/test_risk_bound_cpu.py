"""CPU check of the bound k_risk relies on (csrc/tfx_move_tt.hpp): in one tick a car moves at most
rate * v + a * rate^2 / 2, evaluated in float32 in the kernel's operation order, because the IDM acceleration never
exceeds a (traffic_env.py:56-57).  The oracle's move_cars on pathological states (unsorted, beyond the road end,
enormous / denormal speeds, zero gap denominators) must never put a car beyond `reach`, so "can this car leave the
road this tick" answered from `reach` is never wrong on the unsafe side: agent steps over two-tick passes stay
exact (an env in which the first tick of a pair could overflow is taken one tick at a time)."""
import numpy as np
import pytest

from oracle.oracle import OracleEnv, live_mask

torch = pytest.importorskip("torch")
from test_gpu_parity import random_state  # noqa: E402  (the state generator only: no GPU is touched)

import sys  # noqa: E402
import os  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "traffic-env_amd"))
from gym_traffic.envs.roadgraph import GridRoad  # noqa: E402


@pytest.mark.parametrize("rate", [0.25, 0.5, 1.0])
@pytest.mark.parametrize("sorted_x", [True, False])
def test_no_car_moves_beyond_the_reach_k_risk_computes(rate, sorted_x):
    rng = np.random.RandomState(int(rate * 100) + int(sorted_x))
    m, n, C, length, E = 3, 2, 20, 150.0, 6
    g = GridRoad(m, n, length)
    orc = OracleEnv(m, n, length, C, g.dest, g.phases, g.nexts, n_envs=E, rate=rate)
    a_max = np.float32(3.0)                       # the archetype's a (traffic_env.py:36)
    r32 = np.float32(rate)
    half_ar2 = (np.float32(0.5) * (a_max * r32)) * r32
    checked = left = 0
    for trial in range(12):
        x, v, w, leading, lastcar = random_state(rng, E, orc.R, C, length, crowd=rng.choice([0.3, 0.9]),
                                                 beyond=rng.choice([0.0, 0.1, 1.6]), sorted_x=sorted_x)
        if trial % 3 == 2:
            v[rng.rand(*v.shape) < 0.05] = 3e7
            v[rng.rand(*v.shape) < 0.05] = 1e-30
            pick = rng.rand(*x[:, :, 2:].shape) < 0.1
            x[:, :, 2:][pick] = (x[:, :, 1:-1] - np.float32(4.0))[pick]
            np.put_along_axis(x, leading[:, :, None].astype(np.int64), np.inf, axis=2)
        orc.reset(rng.randint(2, size=orc.I).astype(np.int32))
        for k in range(E):
            orc.load_planes(k, x[k], v[k], w[k], leading[k], lastcar[k])
        orc.obs[:, 2 * orc.r + orc.I:] = rng.randint(0, 12, size=(E, orc.I))      # some lights past yellow
        orc.move_cars()
        for k in range(E):
            live = live_mask(leading[k], lastcar[k], C)
            x0, v0 = x[k][live].astype(np.float32), v[k][live].astype(np.float32)
            with np.errstate(all="ignore"):
                reach = x0 + np.maximum(r32 * v0 + half_ar2, np.float32(0.0))
            x1 = orc.x[k][live]
            ok = np.isnan(x1) | (x1 <= reach)     # (a NaN position never leaves the road: NaN > length is false)
            assert ok.all(), (trial, k, x0[~ok][:3], v0[~ok][:3], x1[~ok][:3], reach[~ok][:3])
            checked += int(live.sum())
            left += int((x1 > length).sum())
    assert checked > 5000 and left > 50


@pytest.mark.parametrize("sorted_x", [True, False])
def test_the_bound_with_the_largest_acceleration_of_a_table_of_archetypes(sorted_x):
    """Heterogeneous cars: k_risk uses the table's LARGEST a (Dev.risk_a).  Every row's acceleration is
    a_row * (1 - q^delta - u^2) <= a_row <= a_max for any integer exponent, so the same reach holds for mixed rows."""
    rng = np.random.RandomState(77 + int(sorted_x))
    m, n, C, length, E, rate = 3, 2, 20, 150.0, 6, 0.5
    g = GridRoad(m, n, length)
    orc = OracleEnv(m, n, length, C, g.dest, g.phases, g.nexts, n_envs=E, rate=rate)
    tab10 = np.zeros((4, 10), np.float32)
    tab10[:, 1:9] = [[11.11, 4, 3, 4, 13.89, 6, 2, 1], [8.0, 8, 1.5, 1, 10.0, 4, 2.5, 2],
                     [12.0, 3.5, 4, 2, 16.0, 7, 1.5, 1], [9.0, 12, 1.0, 8, 11.0, 3, 3.0, 3]]
    r32 = np.float32(rate)
    half_ar2 = (np.float32(0.5) * (np.float32(tab10[:, 3].max()) * r32)) * r32
    checked = left = 0
    for trial in range(10):
        x, v, w, leading, lastcar = random_state(rng, E, orc.R, C, length, crowd=rng.choice([0.3, 0.9]),
                                                 beyond=rng.choice([0.0, 0.1, 1.6]), sorted_x=sorted_x)
        arch = rng.randint(0, 4, size=x.shape).astype(np.uint8)
        orc.reset(rng.randint(2, size=orc.I).astype(np.int32))
        for k in range(E):
            orc.load_planes(k, x[k], v[k], w[k], leading[k], lastcar[k], arch=arch[k], archetypes=tab10)
        orc.obs[:, 2 * orc.r + orc.I:] = rng.randint(0, 12, size=(E, orc.I))
        orc.move_cars()
        for k in range(E):
            live = live_mask(leading[k], lastcar[k], C)
            x0, v0 = x[k][live].astype(np.float32), v[k][live].astype(np.float32)
            with np.errstate(all="ignore"):
                reach = x0 + np.maximum(r32 * v0 + half_ar2, np.float32(0.0))
            x1 = orc.x[k][live]
            ok = np.isnan(x1) | (x1 <= reach)
            assert ok.all(), (trial, k, x0[~ok][:3], v0[~ok][:3], x1[~ok][:3], reach[~ok][:3])
            checked += int(live.sum())
            left += int((x1 > length).sum())
    assert checked > 4000 and left > 40

"""oracle/numpy_env.py (the NumPy-batched CPU baseline of bench.py) against oracle/idm_oracle.c: two
independent restatements of the reference's tick must agree bit for bit - which also cross-checks the oracle."""
import numpy as np
import pytest

from oracle.numpy_env import NumpyBatchedEnv
from oracle.oracle import OracleEnv, live_mask
from gym_traffic.envs.roadgraph import GridRoad


@pytest.mark.parametrize("m,n,C,L,T", [(3, 3, 12, 120.0, 90), (2, 4, 20, 200.0, 140)])
def test_numpy_batched_env_equals_c_oracle(m, n, C, L, T):
    E = 3
    g = GridRoad(m, n, L)
    g.generate_entrypoints(0)
    entry = np.asarray(g.entrypoints)
    orc = OracleEnv(m, n, L, C, g.dest, g.phases, g.nexts, n_envs=E)
    npe = NumpyBatchedEnv(m, n, L, C, g.dest, g.phases, g.nexts, n_envs=E)
    rng = np.random.RandomState(5 + C)
    ph = rng.randint(2, size=(E, orc.I)).astype(np.int32)
    orc.reset(ph)
    npe.reset(ph)
    overflow_ticks = wrapped = 0
    for t in range(T):
        act = rng.randint(2, size=(E, orc.I)).astype(np.int32) if t % 7 == 0 else act
        cnt = (rng.rand(E, len(entry)) < 0.35).astype(np.int32) + (rng.rand(E, len(entry)) < 0.1)
        roads = [np.repeat(entry, cnt[k]).tolist() for k in range(E)]
        _, _, od = orc.step(act, roads)
        _, _, nd = npe.step(act, cnt, entrypoints=entry)
        assert np.array_equal(nd, od.astype(bool)), t
        assert np.array_equal(npe.leading, orc.leading) and np.array_equal(npe.lastcar, orc.lastcar), t
        assert np.array_equal(npe.obs, orc.obs), t
        assert np.array_equal(npe.rewards, orc.rewards), t
        assert np.array_equal(npe.waiting, orc.waiting), t
        assert np.array_equal(npe.passed_dst, orc.passed_dst.astype(bool)), t
        for k in range(E):
            live = live_mask(orc.leading[k], orc.lastcar[k], C)
            assert np.array_equal(npe.x[k][live].view(np.int32), orc.x[k][live].view(np.int32)), (t, k)
            assert np.array_equal(npe.v[k][live].view(np.int32), orc.v[k][live].view(np.int32)), (t, k)
        overflow_ticks += int(od.any())
        wrapped += int((orc.leading > orc.lastcar).sum())
    assert npe.vehicle_updates == orc.vehicle_updates
    assert wrapped > 0 and (overflow_ticks > 0 or C > 12)


def test_numpy_baseline_leg_runs():
    from oracle.numpy_env import time_config
    out = time_config("cfg0", budget_s=0.3, envs=1)
    assert out["value"] > 0 and out["cores"] == 1

"""Property tests of the oracle itself (hypothesis) and a sanitizer pass over it: random ring states
must keep the ring invariants the reference relies on, conserve cars, and be memory-clean under
ASan/UBSan (GPU sanitizers are not available on the pool, so the CPU build carries that check)."""
import os
import subprocess
import sys

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from conftest import ROOT
from oracle.oracle import OracleEnv, live_mask, ring_order
from gym_traffic.envs.roadgraph import GridRoad


def random_env(seed, m, n, C, length, crowd):
    rng = np.random.RandomState(seed)
    g = GridRoad(m, n, length)
    g.generate_entrypoints(0)
    env = OracleEnv(m, n, length, C, g.dest, g.phases, g.nexts)
    R = env.R
    x = np.zeros((R, C), np.float32)
    v = np.zeros((R, C), np.float32)
    leading = rng.randint(1, C, size=R).astype(np.int32)
    lastcar = leading.copy()
    for e in range(R):
        cnt = min(C - 2, rng.binomial(C - 2, crowd))
        pos = np.sort(rng.uniform(-10, length * 1.05, size=cnt))[::-1]
        s = int(leading[e])
        for j in range(cnt):
            s = s + 1 if s + 1 < C else 1
            x[e, s] = pos[j]
            v[e, s] = rng.uniform(0, 14)
        lastcar[e] = s
        x[e, leading[e]] = np.inf
    env.reset(rng.randint(2, size=env.I))
    env.load_planes(0, x, v, np.zeros_like(x), leading, lastcar)
    return env, g, rng


@settings(max_examples=40, deadline=None)
@given(seed=st.integers(0, 10 ** 6), m=st.integers(1, 3), n=st.integers(1, 3),
       C=st.sampled_from([4, 7, 10, 20]), crowd=st.sampled_from([0.1, 0.5, 0.95]))
def test_ring_invariants_and_conservation(seed, m, n, C, crowd):
    env, g, rng = random_env(seed, m, n, C, 80.0, crowd)
    for t in range(12):
        before = int(env.cars_on_roads_flat().sum())
        exit_before = env.cars_on_roads_flat()[0, env.r:].copy()
        roads = rng.choice(g.entrypoints, size=rng.randint(0, 3)).tolist()
        obs, rew, done = env.step(rng.randint(2, size=env.I), [roads])
        ld, lc = env.leading[0], env.lastcar[0]
        assert ld.min() >= 1 and lc.min() >= 1 and ld.max() <= C - 1 and lc.max() <= C - 1
        after = int(env.cars_on_roads_flat().sum())
        assert after <= before + len(roads)                       # cars only enter through spawns
        if not done[0]:
            # without overflow every car is accounted for: gone cars left through exit roads
            assert before + len(roads) - after >= 0
        assert set(np.unique(rew)).issubset({-10.0 * k for k in range(0, 40)})
        assert (obs[0, :env.r] >= 0).all() and (obs[0, env.r:2 * env.r] >= 0).all()
        # fake-leader params are zero except x; live cars carry the archetype's length
        st_ = env.state[0]
        rows = np.arange(env.R)
        assert (st_[rows, 1:, ld] == 0).all()
        live = live_mask(ld, lc, C)
        assert (st_[:, 2, :][live] == 4.0).all()


def test_no_car_is_lost_or_duplicated_when_nothing_overflows():
    env, g, rng = random_env(5, 2, 2, 20, 120.0, 0.3)
    total_in = int(env.cars_on_roads_flat().sum())
    left = 0
    for t in range(60):
        roads = rng.choice(g.entrypoints, size=1).tolist()
        exits_before = env.cars_on_roads_flat()[0, env.r:].sum()
        n0 = int(env.cars_on_roads_flat().sum())
        _, _, done = env.step(rng.randint(2, size=env.I), [roads])
        assert not done[0]
        n1 = int(env.cars_on_roads_flat().sum())
        total_in += 1
        left += n0 + 1 - n1
    assert left >= 0 and int(env.cars_on_roads_flat().sum()) == total_in - left


def test_oracle_is_clean_under_asan_ubsan(tmp_path):
    """Build the oracle with -fsanitize=address,undefined and replay a golden run in a child
    process (LD_PRELOAD of the sanitizer runtime); any report makes the child exit non-zero."""
    so = os.path.join(ROOT, "oracle", "liboracle_asan.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    if not os.path.exists(libasan):
        pytest.skip("libasan not available")
    code = r'''
import sys, json, numpy as np
sys.path[:0] = [%r, %r]
import oracle.oracle as oo
oo.LIB = %r
oo.build = lambda force=False: oo.LIB
g = np.load(%r)
sc = json.loads(str(g["scenario"]))
env = oo.OracleEnv(sc["m"], sc["n"], sc["L"], sc["C"], g["dest"], g["phases"], g["nexts"], n_envs=2)
env.reset(g["init_phase"])
off = g["spawn_off"]
for t in range(150):
    roads = g["spawn_road"][off[t]:off[t + 1]]
    env.step(g["actions"][t], [roads, roads[:1]], nthreads=2)
    if (t + 1) %% 10 == 0:
        env.remi_reward(); env.cars_on_roads()
assert np.array_equal(env.leading[0], g["leading"][150])
print("ok")
''' % (ROOT, os.path.join(ROOT, "traffic-env_amd"), so, os.path.join(ROOT, "tests", "golden", "g2x2_s0_poi_c10.npz"))
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_shared_power_function_is_the_rounded_binary64_power():
    """include/tfx_pow.h (the (v/v0)**delta of exponents that are no integers in 1..8, compiled by the oracle AND by the HIP
    kernels): against RN32(pow64(q, delta)) - the correctly rounded result but for values within ~1e-8 ulp of a boundary -
    over a sample of bases and exponents, plus the special values.  (HIP == oracle bit for bit on this function is what
    the fractional-delta archetype tests of the GPU suite check.)"""
    import ctypes
    import subprocess
    import tempfile
    from conftest import ROOT
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "p.c")
        open(src, "w").write('#include "tfx_pow.h"\nfloat f(float q, float d) { return tfx_pow_det(q, d); }\n')
        so = os.path.join(tmp, "p.so")
        subprocess.check_call(["gcc", "-O3", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), "-o", so, src])
        lib = ctypes.CDLL(so)
        lib.f.restype = ctypes.c_float
        lib.f.argtypes = [ctypes.c_float, ctypes.c_float]
        rng = np.random.RandomState(3)
        qs = np.concatenate([rng.uniform(0, 2, 15000), np.exp(rng.uniform(-30, 5, 15000))]).astype(np.float32)
        ds = rng.choice([0.5, 0.75, 1.5, 2.5, 3.7, 4.5, 7.3, 0.01, 12.0, 33.3, 64.0], size=qs.size).astype(np.float32)
        with np.errstate(over="ignore"):
            want = (qs.astype(np.float64) ** ds.astype(np.float64)).astype(np.float32)
        got = np.array([lib.f(float(q), float(d)) for q, d in zip(qs, ds)], np.float32)
        assert np.array_equal(got.view(np.int32), want.view(np.int32))
        assert lib.f(0.0, 2.5) == 0.0 and lib.f(1.0, 2.5) == 1.0 and lib.f(float("inf"), 0.3) == float("inf")
        assert np.isnan(lib.f(float("nan"), 2.5)) and lib.f(1e-40, 2.5) == 0.0 and lib.f(3e38, 2.5) == float("inf")

"""CPU-side tests of the host logic that surrounds the GPU step: road tables, entry points, the
seeded spawn schedules (must equal the reference's, car for car), GSpace, the gym protocol shim,
flags, the synthetic workload recipe and render geometry.  No GPU, no oracle on the product path."""
import numpy as np
import pytest

from conftest import golden_names

import gym_traffic  # noqa: F401  (installs the gym shim when gym is absent, registers traffic-v0)
import gym
from gym_traffic.envs.roadgraph import GridRoad
from gym_traffic.spaces.gspace import GSpace
from gym_traffic.spawner import SpawnSchedule, counts_from_roads
from gym_traffic import workload as wl


@pytest.mark.parametrize("name", golden_names(None))
def test_gridroad_tables_and_spawn_schedule_match_reference(name, golden_cache):
    g = golden_cache(name)
    sc = g.sc
    gr = GridRoad(sc["m"], sc["n"], sc["L"])
    spec = 0b1110 if sc["entry"] == "one" else 0
    gr.generate_entrypoints(spec)
    assert np.array_equal(gr.nexts, g["nexts"])
    assert np.array_equal(gr.dest, g["dest"])
    assert np.array_equal(gr.phases, g["phases"])
    assert np.array_equal(gr.entrypoints, g["entrypoints"])
    assert gr.roads == len(g["nexts"]) and gr.train_roads == 4 * sc["m"] * sc["n"]
    # reset_entrypoints: cars_per_sec = local * m * open sides (traffic_env.py:394)
    cps = sc["lcps"] * sc["m"] * (4 - bin(spec).count("1"))
    assert cps == float(g["cars_per_sec"])
    n_arch = 1 if g.archetypes is None else len(g.archetypes)
    s = SpawnSchedule(np.random.RandomState(sc["seed"]), sc["poisson"], gr.entrypoints, lambda: (cps, sc["rate"]),
                      n_archetypes=n_arch)
    n = 0
    for t in range(sc["T"]):
        got = s.next_tick()
        assert got == g.spawns(t).tolist(), (name, t)
        if n_arch > 1:      # ... and the archetype row add_new_cars drew for each of them (traffic_env.py:164)
            assert s.rows == g.spawn_archs(t).tolist(), (name, t)
        n += len(got)
    assert n == int(g["generated_cars"])


def test_gridroad_structure():
    gr = GridRoad(3, 4, 100)
    R, r = gr.roads, gr.train_roads
    assert (R, r, gr.intersections) == (4 * 12 + 2 * 3 + 2 * 4, 48, 12)
    # every non-entry road has exactly one predecessor; exits have none downstream
    nx = gr.nexts
    assert np.all(nx[r:] == -1) and np.all(nx[:r] >= 0)
    targets = nx[:r]
    assert len(set(targets.tolist())) == r                      # one producer per road
    gr.generate_entrypoints(0)
    assert set(range(R)) - set(targets.tolist()) == set(gr.entrypoints.tolist())
    assert gr.locs.shape == (R, 2, 2) and gr.locs.dtype == np.float32
    # a road's end is its successor's start (up to the lane offset eps)
    for e in range(r):
        assert np.abs(gr.locs[e, 1] - gr.locs[nx[e], 0]).max() <= 0.05 * 100


def test_entry_spec_bits():
    gr = GridRoad(2, 3, 50)
    v = 6
    assert gr.generate_entrypoints(0b1110).tolist() == [0, 3]                       # west side only
    assert gr.generate_entrypoints(0b1101).tolist() == [v + 2, v + 5]               # east side only
    assert gr.generate_entrypoints(0b1011).tolist() == [2 * v, 2 * v + 1, 2 * v + 2]
    assert gr.generate_entrypoints(0b0111).tolist() == [3 * v + 3, 3 * v + 4, 3 * v + 5]
    assert gr.generate_entrypoints(0b1111).size == 0


def test_gspace_surface():
    sp = GSpace([4], np.int32(2))
    assert sp.size == 4 and sp.shape == [4] and sp.limit.dtype == np.int32
    np.random.seed(3)
    a = sp.sample()
    np.random.seed(3)
    assert np.array_equal(a, np.random.randint(np.int32(2), size=[4], dtype=np.int32))
    assert sp.empty().shape == (4,) and sp.empty().dtype == np.int32
    assert sp.to_action(np.array([[True, False], [False, True]])).tolist() == [1, 0, 0, 1]
    rep = sp.replicated(3)
    assert rep.shape == [3, 4] and rep.size == 12
    assert GSpace([2, 3], np.float32(1)).empty().dtype == np.float32


def test_gym_protocol_and_registration():
    class Dummy(gym.Env):
        def __init__(self):
            self.action_space = GSpace([2], np.int32(2))
            self.observation_space = GSpace([3], np.int32(1))
            self.reward_size = 2
            self.renders = 0

        def _step(self, a):
            return np.zeros(3), np.ones(2), False, None

        def _reset(self):
            return np.zeros(3)

        def _render(self, mode='human', close=False):
            self.renders += 1

    env = Dummy()
    assert env.unwrapped is env
    env.step(0)
    assert env.renders == 0
    env.rendering = True                     # render-every-inner-step switch (reference __init__.py:6-10)
    env.step(0)
    assert env.renders == 1

    class Twice(gym.Wrapper):
        def _step(self, a):
            o, r, d, i = self.env.step(a)
            return o, r * 2, d, i

    w = Twice(env)
    assert w.reward_size == 2 and w.unwrapped is env and w.action_space is env.action_space
    assert w.step(0)[1].tolist() == [2, 2]
    from gym.envs.registration import register  # noqa: F401
    e = gym.make('traffic-v0')
    assert type(e).__name__ == 'TrafficEnv' and e.graph is None


def test_counts_from_roads():
    idx = {5: 0, 9: 1, 2: 2}
    assert counts_from_roads([9, 9, 2], idx, 3).tolist() == [0, 2, 1]


def test_flags_defaults_and_update():
    from gym_traffic.flags import FLAGS, flag, update_flags
    assert flag('rate') == 0.5 and flag('poisson') is True and flag('entry') == 'all'
    assert flag('no_such_flag', 7) == 7
    update_flags(rate=0.25)
    assert FLAGS.rate == 0.25
    update_flags(rate=0.5)


def test_workload_recipe():
    c = wl.CONFIGS["cfg2"]
    x, v, leading, lastcar = wl.prefill_one_env(c["m"], c["n"], c["length"], c["capacity"], c["prefill"], c["gap"])
    R = 4 * 256 + 64
    assert x.shape == (R, 66) and (lastcar - leading == 48).all()
    assert x[0, 2] == np.float32(392.0) and x[0, 49] == np.float32(400 - 8 * 48)
    assert np.isinf(x[:, 1]).all()
    assert (np.diff(x[:, 2:50], axis=1) < 0).all()               # head first, decreasing x
    ns = slice(2 * 256, 4 * 256)
    assert (v[ns, 2:6] == 0).all() and (v[ns, 7:50] == 8).all()    # red at t=0: stopped within 40 m
    assert (v[:512, 2:50] == 8).all()
    gr = GridRoad(16, 16, 400)
    gr.generate_entrypoints(0)
    per = [len(wl.spawn_roads_for_tick(gr.entrypoints, t)) for t in range(8)]
    assert sum(per) == 64                                       # every entry road once per period
    a = wl.cycle_actions(np.arange(40), 3, 5)
    assert a.shape == (40, 3) and a[0].tolist() == [0, 0, 0] and a[15].tolist() == [1, 1, 1]
    assert wl.algorithmic_bytes_per_tick(10, 2, 1) == 16 * 10 + 48 * 2 + 32


def test_render_geometry():
    from gym_traffic.render import light_colours, car_segments
    gr = GridRoad(2, 2, 100)
    cols = light_colours(gr, np.array([1, 0, 1, 0]), np.array([0, 9, 9, 0]), 6)
    assert cols[0].tolist() == [1, 1, 0]        # road 0: phase 1 == current 1 (red side), fresh -> yellow
    assert cols[1].tolist() == [0, 1, 0]        # phase 1 != 0, old -> green
    assert cols[2].tolist() == [1, 0, 0]        # red and not fresh
    assert cols[3].tolist() == [1, 0, 0]        # green side but fresh -> still red
    state = np.zeros((gr.roads, 3, 6), np.float32)
    leading = np.ones(gr.roads, np.int32)
    lastcar = np.ones(gr.roads, np.int32)
    state[0, 0, 2:4] = [60, 30]
    lastcar[0] = 3
    seg = car_segments(gr, state, leading, lastcar, 4.0)
    assert seg.shape == (2, 4)
    assert np.allclose(seg[0, 0] - seg[0, 2], 4.0) and np.allclose(seg[:, 1], seg[:, 3])


def test_philox_known_answer_and_gap_table():
    """Host mirror of the device arrival generator: Random123 known answers for philox4x32-10 and
    the gap table against sampled round(Exp(mean))."""
    from gym_traffic.devrng import philox4x32, gap_table, PoissonMirror
    assert philox4x32(0, 0, 0, 0, 0, 0) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    assert philox4x32(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF) == \
        (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)
    cdf = gap_table(0.8)
    assert cdf[-1] == 0xFFFFFFFF and (np.diff(cdf.astype(np.int64)) >= 0).all()
    p = cdf.astype(np.float64) / 2 ** 32
    gaps = np.array([round(x) for x in np.random.RandomState(0).exponential(1 / 0.8, size=200000)])
    for k in range(5):
        assert abs((gaps <= k).mean() - p[k]) < 4e-3
    # long-run arrival rate of the mirrored stream = 1 / E[gap] (whole-tick gaps: the reference's
    # generator is burstier than its nominal cars_per_tick at high rates); streams differ per env id
    m = PoissonMirror(2.5, 99, 8, [0, 1])
    tot = sum(m.next_tick() for _ in range(600))
    p25 = gap_table(2.5).astype(np.float64) / 2 ** 32
    rate = 1.0 / (1.0 - p25).sum()
    assert abs(tot[0].sum() / 600.0 - rate) < 0.25 * rate and not np.array_equal(tot[0], tot[1])


def test_spawn_schedule_shortcuts_consume_the_stream_like_the_literal_calls():
    """SpawnSchedule draws `randint(0, len(a))` for the reference's `rand.choice(a)` and skips
    `rand.randint(1)`: both must leave a legacy RandomState exactly where the literal calls do."""
    ent = np.array([0, 4, 8, 12, 19, 23, 27, 31, 32, 33, 34, 35, 60, 61, 62, 63], np.int32)
    for seed in range(20):
        a, b = np.random.RandomState(seed), np.random.RandomState(seed)
        for _ in range(100):
            gap_a, gap_b = a.exponential(0.37), b.exponential(0.37)
            a.randint(1)                                   # the reference's archetypes[randint(1)]
            road_a = a.choice(ent)                         # add_new_cars: rand.choice(entrypoints)
            road_b = ent[b.randint(0, len(ent))]
            assert gap_a == gap_b and road_a == road_b
        sa, sb = a.get_state(), b.get_state()
        assert np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]


def test_arrival_streams_in_c_equal_the_python_replay_and_numpy():
    """tfx_arrivals_replay (csrc/tfx_arrivals.cpp) against SpawnSchedule - which itself equals the
    reference's generators call for call - for both generators, many seeds, several rates: same cars
    on the same entry roads in the same ticks, and the MT19937 stream left in the same state."""
    from gym_traffic.spawner import ArrivalStreams, SpawnSchedule, counts_from_roads
    ent = np.array([0, 4, 8, 12, 19, 23, 27, 31, 32, 33, 34, 35, 60, 61, 62, 63], np.int32)
    col = {int(rd): j for j, rd in enumerate(ent)}
    for poisson in (True, False):
        for cpt in (0.07, 0.48, 1.0, 3.84):
            seeds = list(range(11, 11 + 9))
            c_side = ArrivalStreams(seeds, poisson, ent, col, len(ent), cpt)
            py = [SpawnSchedule(np.random.RandomState(s), poisson, ent, lambda: (cpt, 1.0)) for s in seeds]
            total = 0
            for chunk in (1, 7, 40, 3):
                counts, made = c_side.next_ticks(chunk)
                counts, made = counts.copy(), made.copy()
                for t in range(chunk):
                    for k, sch in enumerate(py):
                        roads = sch.next_tick()
                        want = counts_from_roads(roads, col, len(ent))
                        assert np.array_equal(counts[t, k], want), (poisson, cpt, k, t)
                        assert made[t, k] == len(roads)
                        total += len(roads)
            assert total > 0
            for k, sch in enumerate(py):
                a, b = sch.rand.get_state(), c_side.random_state(k).get_state()
                assert np.array_equal(a[1], b[1]) and a[2] == b[2], (poisson, cpt, k)


def test_arrival_streams_threads_do_not_change_the_streams(monkeypatch):
    """Large calls are split over host threads (streams are independent): same result as one thread."""
    from gym_traffic.spawner import ArrivalStreams
    ent = np.arange(12, dtype=np.int32)
    col = {int(rd): j for j, rd in enumerate(ent)}
    seeds = list(range(700))
    a = ArrivalStreams(seeds, True, ent, col, 12, 0.9)
    monkeypatch.setenv("TFX_HOST_THREADS", "1")
    ca, ma = [x.copy() for x in a.next_ticks(64)]
    monkeypatch.setenv("TFX_HOST_THREADS", "6")
    b = ArrivalStreams(seeds, True, ent, col, 12, 0.9)
    cb, mb = b.next_ticks(64)
    assert np.array_equal(ca, cb) and np.array_equal(ma, mb) and int(ma.sum()) > 10000
    assert a.random_state(699).get_state()[2] == b.random_state(699).get_state()[2]


def test_pmc_traffic_is_reported_only_for_the_profiled_code(tmp_path, monkeypatch):
    """bench.py's roofline.traffic comes from a committed rocprofv3 summary; it must vanish (null) as soon as the
    machine code of that kernel in the library differs from the code that was profiled (tools/kernel_hash.py), or the
    kernel / workload is another.  pmc_summary.py refuses to write a summary for counters that did not come with the
    hash recorded on the box - there is no hand stamp."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, ROOT)
    import bench
    import kernel_hash
    hs = kernel_hash.kernel_hashes()
    assert {"k_move_tt", "k_tail", "k_res", "k_move_t", "k_advance"} <= set(hs) and all(len(v) == 16 for v in hs.values())
    assert hs == kernel_hash.kernel_hashes()
    assert kernel_hash.family("_ZN3tfx9k_move_ttILb1ELb0ELb0ELb0EEEvNS_3DevEii") == "k_move_tt"
    prof = tmp_path / "profiles"
    prof.mkdir()
    good = {"kernel": "k_move_t", "kernel_hash": hs["k_move_t"], "hbm_bytes_per_tick": 123.0}
    (prof / "pmc_cfgX.json").write_text(json.dumps(good))
    (prof / "pmc_cfgY.json").write_text(json.dumps(dict(good, kernel_hash="0" * 16)))
    (prof / "pmc_cfgW.json").write_text(json.dumps({"kernel": "k_move_t", "csrc_hash": "x" * 16, "hbm_bytes_per_tick": 1.0}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.load_pmc_traffic("cfgX", "k_move_t") == 123.0
    assert bench.load_pmc_traffic("cfgX", "k_res") is None          # another kernel moved the cars
    assert bench.load_pmc_traffic("cfgY", "k_move_t") is None       # other code was profiled
    assert bench.load_pmc_traffic("cfgW", "k_move_t") is None       # a summary without the code's hash (earlier rounds)
    assert bench.load_pmc_traffic("cfgZ", "k_move_t") is None       # never profiled
    # counters without the hash taken on the box: no summary
    pm = tmp_path / "run" / "cfg_fetch"
    pm.mkdir(parents=True)
    (pm / "p_counter_collection.csv").write_text(
        "Kernel_Name,Counter_Name,Counter_Value\n"
        "\"void tfx::k_move_t<4, 3, false>(tfx::Dev, int)\",FETCH_SIZE,10\n"
        "\"void tfx::k_move_t<4, 3, false>(tfx::Dev, int)\",WRITE_SIZE,10\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), "--round", "rXX", "--config", "cfgT",
                        "--pmc", str(pm), "--kernel", "k_move_t", "--out", str(tmp_path / "out")],
                       cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode != 0 and "kernel_hashes.json" in (r.stderr + r.stdout)
    assert not os.path.exists(str(tmp_path / "out"))
    # ... and with it: the summary carries THAT hash, whatever the library in this tree looks like
    (tmp_path / "run" / "kernel_hashes.json").write_text(json.dumps({"k_move_t": "feedfacefeedface"}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), "--round", "rXX", "--config", "cfgT",
                        "--pmc", str(pm), "--kernel", "k_move_t", "--out", str(tmp_path / "out")],
                       cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert json.load(open(str(tmp_path / "out" / "pmc_cfgT.json")))["kernel_hash"] == "feedfacefeedface"


def test_pmc_summary_tells_the_move_kernels_apart():
    """tools/pmc_summary.py keys its tables by a short kernel name: the two-tick pass, its one-tick form and the
    older movers must not fold into each other (k_move_tt would otherwise be read as k_move_t + "t")."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary as ps
    assert ps.short("void tfx::k_move_tt<true, false>(tfx::Dev, int, int)") == "k_move_tt"
    assert ps.short("void tfx::k_move_tt<false, false>(tfx::Dev, int, int)") == "k_move_tt1"
    assert ps.short("void tfx::k_move_t<4, 3, false>(tfx::Dev, int)") == "k_move_t"
    assert ps.short("void tfx::k_tail<false>(tfx::Dev, int)") == "k_tail"
    assert ps.short("void tfx::k_move_ts<16, false>(tfx::Dev, int)") == "k_move_ts"
    assert ps.short("void tfx::k_move_tts<false, false>(tfx::Dev, int)") == "k_move_tts"
    assert ps.short("void tfx::k_edge<false>(tfx::Dev, int)") == "k_edge"
    assert ps.short("void tfx::k_advance<true>(tfx::Dev, int)") == "k_advance"
    assert ps.short("void tfx::k_res<2, false>(tfx::Dev, tfx::ResArgs)") == "k_res"
    assert ps.short("void at::native::vectorized_elementwise_kernel<4>(int)") is None


@pytest.mark.parametrize("name", [n for n in golden_names() if "_reg_" in n])
def test_regular_mirror_counts_equal_the_reference_schedule(name, golden_cache):
    """The host mirror of the on-device `regular` generator (tfx_set_regular) makes, tick for tick, as many cars as the
    reference's generator did in the captured run (traffic_env.py:167-176); the entry roads come from the Philox stream
    and are a uniform choice over the entry points."""
    from gym_traffic.devrng import RegularMirror
    g = golden_cache(name)
    per_tick = np.diff(g["spawn_off"])
    cpt = float(g["cars_per_sec"]) * g.sc["rate"]
    n_entry = len(g["entrypoints"])
    mir = RegularMirror(cpt, 99, n_entry, [0, 1, 7])
    hist = np.zeros(n_entry, np.int64)
    for t in range(len(per_tick)):
        cnt = mir.next_tick()
        assert (cnt.sum(axis=1) == per_tick[t]).all(), (name, t)
        hist += cnt.sum(axis=0)
    assert hist.min() > 0 and hist.max() < 3 * hist.sum() / n_entry


def test_no_kernel_carries_a_stack_frame_or_flat_accesses():
    """tools/kernel_regs.py over the built library: the kernels whose per-road pointers are moved onto LDS copies must
    reach them with ds_ instructions, and no kernel may carry the stack frame of a function the inliner left as a real
    call (the parameter block then travels through scratch memory and every access through it becomes a flat one:
    k_tail<AGENT, HET> ran a decision 26 % slower that way: a 1.1 KB frame).  Register spills of the capped kernels (the
    side-word forms at five wavefronts per SIMD: up to ~130 bytes) are tolerated."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_regs
    if not os.path.exists(kernel_regs.LIB):
        pytest.skip("library not built")
    table = kernel_regs.kernel_table()
    assert len(table) > 60                                     # every instantiation is listed
    bad = [(k["name"], k["scratch"], k.get("flat_", 0)) for k in table if k["scratch"] >= 256 or k.get("flat_", 0) > 16]
    assert not bad, bad
    tails = [k for k in table if k["name"].startswith("void k_tail<")]
    assert len(tails) == 12 and all(k["ds_"] > 100 for k in tails)
    # the register cap the split call counts on: the plain k_tail fits the slots one wavefront of the pass frees
    plain_tail = [k for k in tails if k["name"].startswith("void k_tail<false, false, false, false>")][0]
    assert plain_tail["vgpr"] <= 80

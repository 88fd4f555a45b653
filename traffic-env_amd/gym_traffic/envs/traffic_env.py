"""TrafficEnv: the reference's gym environment (traffic_env.py:221-394), stepped on the GPU.

Drop-in surface: `set_graph, seed_generator, reset_entrypoints, reset/_reset, step/_step,
render/_render, cars_on_roads, remi_reward` and the attributes agents and wrappers read (`obs,
rewards, passed, detected, current_phase, elapsed, waiting, passed_dst, leading, lastcar, state,
trip_times, steps, generated_cars, graph, action_space, observation_space, reward_size`).  NumPy in,
NumPy out, one env - exactly what algorithms/*.py and traffic_test.py's wrappers expect.  All
simulation state lives in device tensors owned by a TfxEngine (gym_traffic/core.py); a tick is one
HIP kernel launch (grids that fit a compute unit's LDS) or two.  There is no CPU step path.

Differences from the reference that a caller can observe:
  * `CAPACITY` is a constructor/`set_graph` parameter (module default 20) instead of a frozen
    module constant, so the 8/32/64/128-cars-per-road configurations exist;
  * `obs`/`rewards` are still live buffers re-used every step, but `state`, `leading`, `lastcar`,
    `waiting`, `passed_dst` are device-backed views: reading copies from the GPU, item assignment
    writes through (Remi's `passed_dst[:] = False`, traffic_test.py:63, keeps working);
  * arrays the reference leaves to `np.empty` garbage start at 0.
"""
import time

import gym
import numpy as np

from gym_traffic.spaces.gspace import GSpace
from gym_traffic.flags import FLAGS, flag
from gym_traffic.spawner import SpawnSchedule, counts_from_roads

# module constants, same names as the reference (traffic_env.py:17-25)
THRESH = 0.2
PASSING_REWARD = 0
YELLOW_TICKS = 6
DECEL_PENALTY = False
OVERFLOW_PENALTY = 10
CAPACITY = 20
EPS = 1e-8

# car-parameter indices of the reference's state array (traffic_env.py:33-34); this package stores
# the x, v (and w) planes per car, the other seven are the archetype's constants
params = 10
xi, vi, li, ai, deltai, v0i, bi, ti, s0i, wi = range(params)
archetypes = np.zeros((1, params), dtype=np.float32)
archetypes[0, [vi, ai, deltai, v0i, li, bi, ti, s0i]] = [11.11, 3, 4, 13.89, 4, 6, 2, 1]


def inv_popcount(spec):
    """Open sides of the grid for entry spec `spec` (traffic_env.py:180-185): 4 - popcount(spec & 15)."""
    return 4 - bin(int(spec) & 0b1111).count('1')


def cars_on_roads(leading, lastcar, capacity=None):
    """traffic_env.py:214-218 on host arrays (used by callers that import it, e.g. greedy.py:4)."""
    cap = CAPACITY if capacity is None else capacity
    leading, lastcar = np.asarray(leading), np.asarray(lastcar)
    return (lastcar - leading + (leading > lastcar) * np.int32(cap - 1)).astype(np.int32)


class DeviceView(object):
    """NumPy-looking window on one env's slice of a device tensor: reads download, item
    assignment uploads (and lets the engine rebuild what it caches)."""

    def __init__(self, tensor_fn, after_write=None, dtype=None, before_write=None):
        self._t = tensor_fn
        self._after = after_write
        self._before = before_write
        self._dtype = dtype

    def numpy(self):
        a = self._t().detach().cpu().numpy()
        return a.astype(self._dtype) if self._dtype is not None else a

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    def __getitem__(self, idx):
        return self.numpy()[idx]

    def __setitem__(self, idx, value):
        import torch
        if self._before:
            self._before()
        t = self._t()
        whole = (isinstance(idx, slice) and idx == slice(None)) or idx is Ellipsis
        if whole and np.ndim(value) == 0:
            # `passed_dst[:] = False` (Remi, traffic_test.py:63) and the like: one device-side fill,
            # no download / upload of the tensor
            t.fill_(value)
        elif whole:
            t.copy_(torch.as_tensor(np.ascontiguousarray(np.broadcast_to(
                np.asarray(value), tuple(t.shape)))).to(t.dtype))
        else:
            a = t.detach().cpu().numpy().copy()
            a[idx] = value
            t.copy_(torch.as_tensor(a))
        if self._after:
            self._after()

    def __len__(self):
        return self._t().shape[0]

    @property
    def shape(self):
        return tuple(self._t().shape)

    def copy(self):
        return self.numpy().copy()

    def tolist(self):
        return self.numpy().tolist()

    def __eq__(self, other):
        return self.numpy() == other

    def __repr__(self):
        return "DeviceView(%r)" % (self.numpy(),)


ARCH_COLS = [vi, li, ai, deltai, v0i, bi, ti, s0i]     # engine row order: v, l, a, delta, v0, b, T, s0


class StateView(DeviceView):
    """env.state as the reference shapes it (traffic_env.py:364): float32 [R, 10, C], `state[e, p, s]` = parameter
    p (xi, vi, li, ai, deltai, v0i, bi, ti, s0i, wi) of the car in ring slot s of road e - so readers written against
    the reference (`state[i, [xi, li], ...]` in update_locs, :348-358) work unchanged.  The device keeps x, v, w (and
    the archetype row) per car; a read assembles the ten planes: live slots carry their car's x, v, w and the seven
    constants of its archetype row, the fake leader's slot carries its x and zeros (as :262-263 leaves it), dead
    slots are zero (the reference leaves np.empty garbage there).  `state[...] = value` writes x, v, w - and, with
    several archetypes, the row whose constants the slot now holds - back to the device."""

    def __init__(self, eng, table):
        self._eng = eng
        self._table = np.asarray(table, np.float32)            # [n, 10]

    def numpy(self):
        eng = self._eng
        x, v, w = eng.planes_numpy()
        ld, lc = eng.leading[0].cpu().numpy(), eng.lastcar[0].cpu().numpy()
        slots = np.arange(eng.C)[None, :]
        l2, c2 = ld[:, None], lc[:, None]
        live = np.where(l2 <= c2, (slots > l2) & (slots <= c2), (slots > l2) | ((slots >= 1) & (slots <= c2))) & (l2 != c2)
        rows = eng.arch[0].cpu().numpy().astype(np.int64) if eng.het else np.zeros((eng.R, eng.C), np.int64)
        st = np.transpose(self._table[rows], (0, 2, 1)) * live[:, None, :]          # [R, 10, C]
        st[:, xi, :] = np.where(live | (slots == l2), x[0], 0)
        st[:, vi, :] = np.where(live, v[0], 0)
        st[:, wi, :] = np.where(live, w[0], 0)
        return np.ascontiguousarray(st, np.float32)

    def __setitem__(self, idx, value):
        a = self.numpy()
        a[idx] = value
        eng = self._eng
        arch = None
        if eng.het:
            key = [li, ai, deltai, v0i, bi, ti, s0i]
            arch = np.zeros((eng.R, eng.C), np.uint8)
            for row in range(self._table.shape[0]):
                arch[(a[:, key, :] == self._table[row, key][None, :, None]).all(axis=1)] = row
            arch = arch[None]
        eng.load_state(a[None, :, xi, :], a[None, :, vi, :], eng.leading.cpu().numpy(),
                       eng.lastcar.cpu().numpy(), w=a[None, :, wi, :], arch=arch)

    def __len__(self):
        return self._eng.R

    @property
    def shape(self):
        return (self._eng.R, params, self._eng.C)


class TrafficEnv(gym.Env):
    metadata = {'render.modes': ['human', 'rgb_array']}

    def __init__(self, capacity=None, device=None):
        self.capacity = capacity
        self.device = device
        self.engine = None
        self.viewer = None
        self.graph = None

    # ---- construction (traffic_env.py:361-382) ------------------------------------------------
    def set_graph(self, graph, capacity=None):
        from gym_traffic.core import TfxEngine
        self.viewer = None
        self.graph = graph
        self.capacity = int(capacity or self.capacity or CAPACITY)
        self._validate = flag('mode', 'train') == 'validate'
        self._spec = getattr(self, '_spec', 0)
        self._build_engine()
        r, i = graph.train_roads, graph.intersections
        self.action_space = GSpace([i], np.int32(2))
        self.observation_space = GSpace([2 * r + 2 * i], np.int32(1))
        self.obs = np.zeros([2 * r + 2 * i], dtype=np.int32)
        self.passed = self.obs[:r]
        self.detected = self.obs[r:r + r]
        self.current_phase = self.obs[r + r:r + r + i]
        self.elapsed = self.obs[-i:]
        self.rewards = np.zeros(i, dtype=np.float32)
        self.reward_size = self.rewards.size
        self.trip_times = []
        self.steps = np.float32(0)
        self.generated_cars = 0
        self.reset_entrypoints()

    def _build_engine(self):
        from gym_traffic.core import TfxEngine
        g = self.graph
        # the module's `archetypes` table, read when the engine is built like the reference's generators read it
        # when a car is made (traffic_env.py:164): rows beyond the first, or another exponent, switch the engine to
        # per-car parameters
        table = np.array(archetypes, np.float32).reshape(-1, params)
        self.engine = TfxEngine(g.m, g.n, float(g.len), self.capacity, n_envs=1,
                                rate=float(FLAGS.rate), learn_switch=bool(flag('learn_switch', False)),
                                validate=self._validate, entry_spec=self._spec, planes=3,
                                device=self.device, archetypes=table[:, ARCH_COLS])
        self._built = (float(FLAGS.rate), bool(flag('learn_switch', False)), self._validate, self._spec,
                       table[:, ARCH_COLS].tobytes())
        eng = self.engine
        # [R, 10, C] as the reference shapes it; a read assembles it from the device
        self.state = StateView(eng, table)
        # ring semantics for `env.leading[i] = k`: the cars keep their SLOTS, so the slot image is
        # exported under the old indices before the write and imported under the new ones after it
        self.leading = DeviceView(lambda: eng.leading[0], eng.refresh, before_write=eng._export)
        self.lastcar = DeviceView(lambda: eng.lastcar[0], eng.refresh, before_write=eng._export)
        self.waiting = DeviceView(lambda: eng.waiting[0])
        self.passed_dst = DeviceView(lambda: eng.passed_dst[0], dtype=np.bool_)
        self._spawn_counts = np.zeros((1, max(1, eng.n_entry)), np.int32)
        # read-backs go through pinned mirrors of packed buffers: ONE device-to-host copy and one stream
        # synchronisation per step (two per fused decision), one host-to-device copy for the inputs
        self._mirror = eng.out_mirror()
        self._mirror_rep = None
        self._trips_seen = 0

    def _sync_flags(self):
        """The reference re-reads FLAGS.rate / learn_switch / mode on every tick (traffic_env.py:
        225,237,240).  They are baked into the device config, so a change rebuilds the engine
        around the current state."""
        now = (float(FLAGS.rate), bool(flag('learn_switch', False)),
               flag('mode', 'train') == 'validate', self._spec,
               np.array(archetypes, np.float32).reshape(-1, params)[:, ARCH_COLS].tobytes())
        if now == self._built:
            return
        old = self.engine
        snap = [t.clone() for t in (old.xv, old.w, old.leading, old.lastcar, old.obs, old.rewards,
                                    old.waiting, old.passed_dst)]
        tick = old.tick
        self._validate = now[2]
        self._build_engine()
        eng = self.engine
        for dst, src in zip((eng.xv, eng.w, eng.leading, eng.lastcar, eng.obs, eng.rewards, eng.waiting,
                             eng.passed_dst), snap):
            dst.copy_(src)
        eng.refresh()
        eng.set_tick(tick)

    def seed_generator(self, seed=None):
        self.rand = np.random.RandomState(seed)
        self._schedule = None   # created lazily: entrypoints / cars_per_sec may still change

    def reset_entrypoints(self):
        entry = flag('entry', 'all')
        if entry == "random":
            spec = int(np.random.randint(0b1111, dtype='uint32'))
        elif entry == "one":
            spec = 0b1110
        else:
            spec = 0
        self._spec = spec
        self.graph.generate_entrypoints(spec)
        self.cars_per_sec = FLAGS.local_cars_per_sec * self.graph.m * inv_popcount(spec)
        # the reference publishes this through the global flags (traffic_env.py:394); keep doing so
        FLAGS.cars_per_sec = self.cars_per_sec
        if self.engine is not None and self._built[3] != spec:
            self._sync_flags()
        if getattr(self, '_schedule', None) is not None:
            self._schedule.entrypoints = self.graph.entrypoints

    def _spawns(self):
        if getattr(self, '_schedule', None) is None:
            if not hasattr(self, 'rand'):
                self.seed_generator()
            self._schedule = SpawnSchedule(self.rand, flag('poisson', True), self.graph.entrypoints,
                                           lambda: (FLAGS.cars_per_sec, FLAGS.rate),
                                           n_archetypes=archetypes.shape[0])
        return self._schedule.next_tick()

    # ---- gym protocol -------------------------------------------------------------------------
    def _reset(self):
        self._sync_flags()
        self.steps = np.float32(0)
        self.generated_cars = 0
        self.engine.reset(self.action_space.sample())
        self._trips_seen = 0
        self._pull()
        return self.obs

    def _step(self, action):
        self._sync_flags()
        eng = self.engine
        roads = self._spawns()
        self.generated_cars += len(roads)
        counts_from_roads(roads, eng.entry_index, eng.n_entry, out=self._spawn_counts[0])
        # `current_phase[:] = action` / logical_xor semantics: any dtype, truthiness for the change
        act = np.asarray(action)
        if flag('learn_switch', False):
            act = (act != 0)
        eng.stage_inputs(act.astype(np.int32), self._spawn_counts)
        if eng.het:
            eng.set_spawn_rows(self._rows_of([roads], [self._schedule.rows]))
        first = eng.tick
        eng.step(1, update_done=False)
        self.steps += 1
        overflowed = self._pull(first)
        return self.obs, self.rewards, overflowed, None

    def repeat(self, action, n_ticks):
        """`n_ticks` x `_step(action)` exactly as the Repeater wrapper drives it (reference
        traffic_test.py:36-53: passed summed, detected of the last tick, signed elapsed/100, rewards
        summed, `if done: break`) as ONE device submission (tfx_agent_step, replayed as a HIP graph)
        instead of n_ticks host round trips.  Returns (total_obs float32[2r+I], total_reward
        float32[I], done).  Afterwards the env is in the state the tick-by-tick loop leaves it in,
        except that `self.passed` / `self.rewards` hold the step's sums rather than the last
        tick's values (`Remi` overwrites the latter anyway)."""
        self._sync_flags()
        eng, n = self.engine, int(n_ticks)
        if getattr(self, '_rep_counts', None) is None or self._rep_counts.shape[0] != n:
            self._rep_counts = np.zeros((n, 1, max(1, eng.n_entry)), np.int32)
        mark = self._mark_spawner()
        made, all_roads, all_rows = [], [], []
        for t in range(n):
            roads = self._spawns()
            made.append(len(roads))
            all_roads.append(roads)
            all_rows.append(self._schedule.rows)
            counts_from_roads(roads, eng.entry_index, eng.n_entry, out=self._rep_counts[t, 0])
        act = np.asarray(action)
        if flag('learn_switch', False):
            act = (act != 0)
        eng.stage_inputs(act.astype(np.int32), self._rep_counts, per_tick=True)
        if eng.het:
            eng.set_spawn_rows(self._rows_of(all_roads, all_rows), per_tick=True)
        first = eng.tick
        eng.agent_step(n, remi=False)
        if self._mirror_rep is None:
            self._mirror_rep = eng.agent_mirror()
        arep = self._mirror_rep.start()
        got = self._mirror.pull()                 # (one synchronisation completes both copies)
        total_obs, total_reward = arep["aobs"][0].copy(), arep["areward"][0].copy()
        done = bool(got["done"][0])
        ran = n
        if done:
            # the loop broke after the overflowing tick: un-draw the arrivals of the ticks that
            # never ran and put the device clock where `steps` is
            ran = int(got["done_tick"][0]) - first
            if ran < n:
                self._rewind_spawner(mark)
                for _ in range(ran):
                    self._spawns()
                eng.set_tick(first + ran)
        self.steps += ran
        self.generated_cars += sum(made[:ran])
        self.obs[:] = got["obs"][0]
        self.rewards[:] = got["rewards"][0]
        if self._validate:
            self._collect_trips(int(got["n_trips"][0]))
        return total_obs, total_reward, done

    def _rows_of(self, roads_per_tick, rows_per_tick):
        """uint8 [n_ticks, 1, n_entry, S]: archetype row of the j-th car each entry road receives in each tick."""
        eng = self.engine
        per = [{} for _ in roads_per_tick]
        for t, (roads, rows) in enumerate(zip(roads_per_tick, rows_per_tick)):
            for rd, a in zip(roads, rows):
                per[t].setdefault(eng.entry_index[rd], []).append(a)
        S = max([len(v) for d in per for v in d.values()] + [1])
        out = np.zeros((len(per), 1, max(1, eng.n_entry), S), np.uint8)
        for t, d in enumerate(per):
            for j, v in d.items():
                out[t, 0, j, :len(v)] = v
        return out if len(per) > 1 else out[0]

    def _mark_spawner(self):
        s = getattr(self, '_schedule', None)
        if s is None:
            if not hasattr(self, 'rand'):
                self.seed_generator()
            return (self.rand.get_state(), None)
        return (s.rand.get_state(), (s._started, s._gap, s._i))

    def _rewind_spawner(self, mark):
        state, inner = mark
        self.rand.set_state(state)
        s = getattr(self, '_schedule', None)
        if s is not None:
            if inner is None:
                self._schedule = None
            else:
                s._started, s._gap, s._i = inner

    def _pull(self, since_tick=None):
        """Refresh the live host buffers (obs, rewards) from the device; returns the done flag (with
        `since_tick`: an overflow in a tick >= since_tick, from the env's overflow stamp).  In
        validate mode also collects the trip times recorded since the last pull."""
        got = self._mirror.pull()
        self.obs[:] = got["obs"][0]
        self.rewards[:] = got["rewards"][0]
        if self._validate:
            self._collect_trips(int(got["n_trips"][0]))
        if since_tick is not None:
            return bool(got["done_tick"][0] > since_tick)
        return bool(got["done"][0])

    def _collect_trips(self, n_now):
        seen = getattr(self, '_trips_seen', 0)
        if n_now > seen:
            self.trip_times.extend(self.engine.trip_times[0, seen:n_now].cpu().numpy())
        self._trips_seen = n_now

    def cars_on_roads(self):
        return self.engine.cars_on_roads()[0].cpu().numpy()

    def remi_reward(self):
        self.rewards[:] = self.engine.remi_reward()[0].cpu().numpy()
        return self.rewards

    # ---- rendering (traffic_env.py:285-359; pyglet is not available -> matplotlib) -------------
    def _render(self, mode='human', close=False):
        from gym_traffic.render import MatplotlibViewer
        if close:
            if self.viewer is not None:
                self.viewer.close()
                self.viewer = None
            return
        if self.viewer is None:
            self.viewer = MatplotlibViewer(self.graph)
        eng = self.engine
        frame = self.viewer.draw(self.graph, self.state.numpy(), eng.leading[0].cpu().numpy(),
                                 eng.lastcar[0].cpu().numpy(), self.current_phase, self.elapsed,
                                 YELLOW_TICKS, float(archetypes[0, li]), mode)
        if mode == 'human':
            time.sleep(FLAGS.rate / 2)
        return frame

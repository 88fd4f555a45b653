"""ctypes front-end of the CPU oracle (oracle/idm_oracle.c).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (traffic-env_amd/gym_traffic) never does: its step path is the HIP library
and it raises if that library is missing.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")

NPARAMS = 10
XI, VI, LI, AI, DELTAI, V0I, BI, TI, S0I, WI = range(NPARAMS)

# the single archetype of traffic_env.py:35-43 (x=0, w set at spawn)
ARCHETYPE = np.zeros(NPARAMS, np.float32)
ARCHETYPE[VI] = 11.11
ARCHETYPE[AI] = 3
ARCHETYPE[DELTAI] = 4
ARCHETYPE[V0I] = 13.89
ARCHETYPE[LI] = 4
ARCHETYPE[BI] = 6
ARCHETYPE[TI] = 2
ARCHETYPE[S0I] = 1


class OrcCfg(C.Structure):
    _fields_ = [("m", C.c_int32), ("n", C.c_int32), ("I", C.c_int32), ("r", C.c_int32),
                ("R", C.c_int32), ("C", C.c_int32), ("E", C.c_int32),
                ("length", C.c_float), ("rate", C.c_float),
                ("archetype", C.c_float * NPARAMS),
                ("yellow_ticks", C.c_int32), ("thresh", C.c_float), ("detect_dist", C.c_float),
                ("overflow_penalty", C.c_float), ("eps", C.c_float),
                ("learn_switch", C.c_int32), ("validate", C.c_int32)]


class OrcBufs(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("state", "leading", "lastcar", "obs", "rewards",
                                          "waiting", "passed_dst", "done")]


def build(force=False):
    """Compile liboracle.so if missing or stale.  Building the checker is not using it."""
    src = os.path.join(HERE, "idm_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB)
        assert _lib.orc_sizeof_cfg() == C.sizeof(OrcCfg)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleEnv(object):
    """E independent copies of the reference's TrafficEnv state, stepped by the C oracle.

    Tables (`dest`, `phases`, `nexts`) are int32[R] as produced by the reference's GridRoad
    (roadgraph.py:26-39); tests take them from the golden fixtures.
    """

    def __init__(self, m, n, length, capacity, dest, phases, nexts, n_envs=1, rate=0.5,
                 learn_switch=False, validate=False, archetype=None, trip_cap=4096):
        self.m, self.n, self.E, self.C = int(m), int(n), int(n_envs), int(capacity)
        self.I = self.m * self.n
        self.r = 4 * self.I
        self.R = self.r + 2 * self.m + 2 * self.n
        self.dest = np.ascontiguousarray(dest, np.int32)
        self.phases = np.ascontiguousarray(phases, np.int32)
        self.nexts = np.ascontiguousarray(nexts, np.int32)
        assert self.dest.shape == (self.R,)
        c = OrcCfg()
        c.m, c.n, c.I, c.r, c.R, c.C, c.E = self.m, self.n, self.I, self.r, self.R, self.C, self.E
        c.length = float(length)
        c.rate = float(rate)
        arch = ARCHETYPE if archetype is None else np.asarray(archetype, np.float32)
        for k in range(NPARAMS):
            c.archetype[k] = float(arch[k])
        c.yellow_ticks = 6
        c.thresh = 0.2
        c.detect_dist = 10.0
        c.overflow_penalty = 10.0
        c.eps = 1e-8
        c.learn_switch = int(bool(learn_switch))
        c.validate = int(bool(validate))
        self.cfg = c
        E, R, Cc, I, r = self.E, self.R, self.C, self.I, self.r
        self.state = np.zeros((E, R, NPARAMS, Cc), np.float32)
        self.leading = np.ones((E, R), np.int32)
        self.lastcar = np.ones((E, R), np.int32)
        self.obs = np.zeros((E, 2 * r + 2 * I), np.int32)
        self.rewards = np.zeros((E, I), np.float32)
        self.waiting = np.zeros((E, r), np.int32)
        self.passed_dst = np.zeros((E, I), np.uint8)
        self.done = np.zeros(E, np.uint8)
        self.steps = np.zeros(E, np.float32)
        self.trip_cap = int(trip_cap)
        self.trip_times = np.zeros((E, self.trip_cap), np.float32)
        self.n_trips = np.zeros(E, np.int64)
        self._updates = C.c_int64(0)   # live cars advanced by move_cars so far (benchmark bookkeeping)
        b = OrcBufs()
        for k in ("state", "leading", "lastcar", "obs", "rewards", "waiting", "passed_dst", "done"):
            setattr(b, k, _p(getattr(self, k)))
        self.bufs = b

    # views with the reference's names
    @property
    def passed(self):
        return self.obs[:, :self.r]

    @property
    def detected(self):
        return self.obs[:, self.r:2 * self.r]

    @property
    def current_phase(self):
        return self.obs[:, 2 * self.r:2 * self.r + self.I]

    @property
    def elapsed(self):
        return self.obs[:, 2 * self.r + self.I:]

    @property
    def x(self):
        return self.state[:, :, XI, :]

    @property
    def v(self):
        return self.state[:, :, VI, :]

    @property
    def w(self):
        return self.state[:, :, WI, :]

    def reset(self, phase_init):
        ph = np.ascontiguousarray(np.broadcast_to(np.asarray(phase_init, np.int32), (self.E, self.I)))
        lib().orc_reset_batch(C.byref(self.cfg), C.byref(self.bufs), _p(ph))
        self.steps[:] = 0
        self.n_trips[:] = 0
        return self.obs

    def step(self, action, spawn_roads=None, nthreads=1, spawn_arch=None, archetypes=None):
        """action int[E][I] (or [I], broadcast); spawn_roads: list (len E) of int sequences, or a
        (spawn_off int64[E+1], roads int32[]) CSR pair, or None.  archetypes float32 [n, 10] (the reference's
        `archetypes` table, traffic_env.py:35-43) with spawn_arch - list (len E) of the row drawn for every spawned car,
        parallel to spawn_roads - for runs with more than the config's single archetype."""
        act = np.ascontiguousarray(np.broadcast_to(np.asarray(action, np.int32), (self.E, self.I)))
        if spawn_roads is None:
            off = np.zeros(self.E + 1, np.int64)
            roads = np.zeros(1, np.int32)
        elif isinstance(spawn_roads, tuple):
            off, roads = spawn_roads
            off = np.ascontiguousarray(off, np.int64)
            roads = np.ascontiguousarray(roads, np.int32)
            if roads.size == 0:
                roads = np.zeros(1, np.int32)
        else:
            assert len(spawn_roads) == self.E
            off = np.zeros(self.E + 1, np.int64)
            off[1:] = np.cumsum([len(s) for s in spawn_roads])
            roads = np.zeros(max(1, int(off[-1])), np.int32)
            for k, s in enumerate(spawn_roads):
                roads[off[k]:off[k + 1]] = s
        arch_tab = arch_ids = None
        if archetypes is not None:
            arch_tab = np.ascontiguousarray(archetypes, np.float32).reshape(-1, NPARAMS)
            arch_ids = np.zeros(max(1, int(off[-1])), np.int32)
            if spawn_arch is not None:
                for k, s in enumerate(spawn_arch):
                    arch_ids[off[k]:off[k + 1]] = s
        lib().orc_step_batch_arch(C.byref(self.cfg), _p(self.dest), _p(self.phases), _p(self.nexts),
                                  C.byref(self.bufs), _p(act), _p(off), _p(roads), _p(self.steps),
                                  _p(self.trip_times), _p(self.n_trips), C.c_int64(self.trip_cap),
                                  C.c_int(int(nthreads)), C.byref(self._updates), _p(arch_tab), _p(arch_ids))
        self.steps += np.float32(1)
        return self.obs, self.rewards, self.done

    @property
    def vehicle_updates(self):
        return int(self._updates.value)

    def move_cars(self):
        lib().orc_move_cars_batch(C.byref(self.cfg), _p(self.dest), _p(self.phases), _p(self.nexts),
                                  C.byref(self.bufs))

    def advance(self):
        lib().orc_advance_batch(C.byref(self.cfg), _p(self.dest), _p(self.nexts), C.byref(self.bufs),
                                _p(self.steps), _p(self.trip_times), _p(self.n_trips),
                                C.c_int64(self.trip_cap))
        return self.done

    def remi_reward(self):
        lib().orc_remi_batch(C.byref(self.cfg), _p(self.dest), _p(self.phases), C.byref(self.bufs))
        return self.rewards

    def cars_on_roads_flat(self):
        out = np.zeros((self.E, self.R), np.int32)
        lib().orc_cars_on_roads_batch(C.byref(self.cfg), C.byref(self.bufs), _p(out))
        return out

    def cars_on_roads(self):
        """[E, m, n, 4] as TrafficEnv.cars_on_roads (traffic_env.py:255-257)."""
        flat = self.cars_on_roads_flat()[:, :self.r]
        return np.transpose(flat.reshape(self.E, 4, self.m, self.n), (0, 2, 3, 1))

    # ---- state import/export in (x, v, w) planes, the layout the HIP path and fixtures use ----
    def load_planes(self, k, x, v, w, leading, lastcar, arch=None, archetypes=None):
        """Set env k from [R, C] planes: live slots become archetype cars carrying (x, v, w); the
        fake-leader slot gets all-zero params and the given x (traffic_env.py:262-263,133).  arch int [R, C] +
        archetypes float32 [n, 10]: the table row each live slot's car was copied from (default: the config's)."""
        self.leading[k] = leading
        self.lastcar[k] = lastcar
        live = live_mask(self.leading[k], self.lastcar[k], self.C)
        st = self.state[k]
        if arch is None:
            rows = np.asarray(self.cfg.archetype[:], np.float32)[None, :, None]
        else:
            rows = np.transpose(np.asarray(archetypes, np.float32)[np.asarray(arch, np.int64)], (0, 2, 1))   # [R, 10, C]
        st[:] = rows * live[:, None, :].astype(np.float32)
        st[:, XI, :] = np.where(live, x, 0)
        st[:, VI, :] = np.where(live, v, 0)
        st[:, WI, :] = np.where(live, w, 0)
        rows = np.arange(self.R)
        st[rows, XI, self.leading[k]] = np.asarray(x)[rows, self.leading[k]]

    def arch_plane(self, k, archetypes):
        """int8 [R, C]: row of `archetypes` each slot's car equals in everything but x, v, w (0 where none does)."""
        cols = [LI, AI, DELTAI, V0I, BI, TI, S0I]
        key = self.state[k][:, cols, :]
        tab = np.asarray(archetypes, np.float32)[:, cols]
        out = np.zeros((self.R, self.C), np.int8)
        for a in range(tab.shape[0]):
            out[(key == tab[a][None, :, None]).all(axis=1)] = a
        return out

    def planes(self, k=0):
        """(x, v, w) [R, C] copies with dead slots zeroed (leader slot keeps its x)."""
        live = live_mask(self.leading[k], self.lastcar[k], self.C)
        keep = live | (np.arange(self.C)[None, :] == self.leading[k][:, None])
        out = []
        for p in (XI, VI, WI):
            a = self.state[k, :, p, :].copy()
            a[~(keep if p == XI else live)] = 0
            out.append(a)
        return out


def live_mask(leading, lastcar, C):
    """bool[R, C]: slots holding a real car, in the reference's ring convention
    (README.md:14-23 of the reference: cars occupy wrap(leading+1) .. lastcar)."""
    slots = np.arange(C)[None, :]
    ld = np.asarray(leading)[:, None]
    lc = np.asarray(lastcar)[:, None]
    unwrapped = (slots > ld) & (slots <= lc)
    wrapped = (slots > ld) | ((slots >= 1) & (slots <= lc))
    return np.where(ld <= lc, unwrapped, wrapped) & (ld != lc)


def ring_order(leading, lastcar, C):
    """Slot indices of road cars from head (first behind the fake leader) to tail."""
    out = []
    s = leading
    while s != lastcar:
        s = s + 1
        if s >= C:
            s = 1
        out.append(s)
    return out

"""TfxEngine: E batched traffic envs resident on one MI355X.

Owns the device buffers (PyTorch-ROCm tensors - used for memory and streams only) and drives the
HIP kernels through the C ABI (include/tfx.h, csrc/tfx_hip.hip).  Both front-ends sit on top of it:
`TrafficEnv` (the reference's single-env gym object, NumPy in / NumPy out) and `TrafficVecEnv`
(tensors in / tensors out for batched RL rollouts).

State layout = the reference's arrays with a leading env dimension (traffic_env.py:361-382):
    xv [E,R,C,2] f32 ((x, v) per ring slot; `x`/`v` are views)   w [E,R,C] f32 (planes == 3 only)
    With layout='transposed' (the default when w is not carried) the kernels keep the cars in a
    position-major array instead (csrc/tfx_move_t.hpp); `xv`/`x`/`v` then are a ring-layout staging
    copy that is refreshed from the device on access and pushed back by refresh()/load_state().
    leading/lastcar [E,R] i32   obs [E,2r+2I] i32   rewards [E,I] f32   waiting [E,r] i32
    passed_dst [E,I] u8   done_tick [E] i32
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _native as nat

# the single archetype of the reference (traffic_env.py:35-43)
ARCHETYPE = dict(car_v=11.11, car_l=4.0, car_a=3.0, car_delta=4.0, car_v0=13.89, car_b=6.0,
                 car_T=2.0, car_s0=1.0)
# module constants of the reference (traffic_env.py:17-25)
CONSTANTS = dict(yellow_ticks=6, thresh=0.2, detect_dist=10.0, overflow_penalty=10.0, eps=1e-8)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class HostMirror(object):
    def __init__(self, device, tensors):
        self.device = device
        self.dev = list(tensors)
        self.host = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in self.dev]
        self.views = [h.numpy() for h in self.host]

    def pull(self):
        for h, t in zip(self.host, self.dev):
            h.copy_(t, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return self.views


class PackedMirror(object):
    """Pinned host copy of ONE contiguous device byte buffer that several tensors are views of: one
    device-to-host copy and one stream synchronisation bring all of them over.  `views` maps a name
    to (byte offset, dtype, shape); pull() returns {name: NumPy view} (valid until the next pull)."""

    def __init__(self, device, buf, views):
        self.device = device
        self.dev = buf
        self.host = torch.empty(buf.shape, dtype=torch.uint8).pin_memory()
        raw = self.host.numpy()
        self.views = {}
        for name, (off, dtype, shape) in views.items():
            n = int(np.prod(shape)) * np.dtype(dtype).itemsize
            self.views[name] = raw[off:off + n].view(dtype).reshape(shape)

    def start(self):
        """Queue the copy only (a later pull() / synchronisation on the same stream completes it)."""
        self.host.copy_(self.dev, non_blocking=True)
        return self.views

    def pull(self):
        self.host.copy_(self.dev, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return self.views


class TfxEngine(object):
    def __init__(self, m, n, length, capacity, n_envs=1, rate=0.5, learn_switch=False,
                 validate=False, entry_spec=0, planes=None, trip_cap=4096, device=None, env_id_offset=0,
                 layout=None, archetypes=None):
        """archetypes: rows (v, l, a, delta, v0, b, T, s0) of the reference's `archetypes` table
        (traffic_env.py:35-43) when there is more than its single default row, or a delta other than 4
        ("heterogeneous cars": needs planes = 3 and the transposed layout; every car keeps its row, see
        set_spawns(..., rows=) and the `arch` property)."""
        if not torch.cuda.is_available():
            raise nat.TfxError("no GPU visible: the traffic env step runs on MI355X only (no CPU fallback)")
        self.lib = nat.lib()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.m, self.n, self.C, self.E = int(m), int(n), int(capacity), int(n_envs)
        arch = None if archetypes is None else np.asarray(archetypes, np.float32).reshape(-1, 8)
        self.het = arch is not None and (len(arch) > 1 or float(arch[0, 3]) != 4.0)
        self.archetypes = arch
        self.P = int(planes) if planes else (3 if (validate or self.het) else 2)
        cfg = nat.TfxConfig()
        cfg.m, cfg.n, cfg.capacity, cfg.n_envs, cfg.planes = self.m, self.n, self.C, self.E, self.P
        cfg.length, cfg.rate = float(length), float(rate)
        for k, v in ARCHETYPE.items():
            setattr(cfg, k, v)
        if arch is not None:
            if len(arch) > nat.MAX_ARCH:
                raise ValueError("at most %d archetype rows" % nat.MAX_ARCH)
            cfg.n_archetypes = len(arch)
            for a, row in enumerate(arch):
                for j in range(8):
                    cfg.arch[a][j] = float(row[j])
        for k, v in CONSTANTS.items():
            setattr(cfg, k, v)
        cfg.learn_switch, cfg.validate = int(bool(learn_switch)), int(bool(validate))
        cfg.entry_spec = int(entry_spec)
        cfg.env_id_offset = int(env_id_offset)
        if layout is None:
            layout = os.environ.get("TFX_LAYOUT") or "transposed"
        if layout not in ("ring", "transposed"):
            raise ValueError("layout must be 'ring' or 'transposed'")
        self.layout = layout
        cfg.layout = 1 if layout == "transposed" else 0
        self.cfg = cfg
        self.validate = bool(validate)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            nat.check(self.lib.tfx_create(C.byref(cfg), C.byref(h)))
        self.h = h
        dims = [C.c_int32() for _ in range(4)]
        nat.check(self.lib.tfx_dims(h, *[C.byref(d) for d in dims]))
        self.I, self.r, self.R, self.n_entry = [int(d.value) for d in dims]
        self.obs_len = 2 * self.r + 2 * self.I
        self.dest = np.zeros(self.R, np.int32)
        self.phases = np.zeros(self.R, np.int32)
        self.nexts = np.zeros(self.R, np.int32)
        self.entrypoints = np.zeros(max(1, self.n_entry), np.int32)
        nat.check(self.lib.tfx_tables(h, *[a.ctypes.data_as(C.c_void_p) for a in
                                           (self.dest, self.phases, self.nexts, self.entrypoints)]))
        self.entrypoints = self.entrypoints[:self.n_entry]
        self.entry_index = {int(rd): j for j, rd in enumerate(self.entrypoints)}

        E, R, P, Cc, I, r = self.E, self.R, self.P, self.C, self.I, self.r
        dev = self.device
        # zeros, not empty: the reference leaves these to np.empty garbage; a defined start keeps
        # runs reproducible (dead slots are never read)
        if self.layout == "transposed":
            pairs = C.c_int64()
            nat.check(self.lib.tfx_xv_pairs(h, C.byref(pairs)))
            self._t = torch.zeros((int(pairs.value), 2), dtype=torch.float32, device=dev)
            # the ring-shaped staging copy behind `xv` / `x` / `v` / `w` / load_state is made on first
            # use: a rollout that never looks at the cars does not pay a second copy of them
            self._ring = None
            self._ringw = None
            self._ringa = None
            self._tw = torch.zeros((self._t.shape[0],), dtype=torch.float32, device=dev) if P == 3 else None
        else:
            self._t = None
            self._ring = torch.zeros((E, R, Cc, 2), dtype=torch.float32, device=dev)
            # spawn ticks (validate mode): ring-shaped like the cars
            self._ringw = torch.zeros((E, R, Cc), dtype=torch.float32, device=dev) if P == 3 else None
            self._ringa = None
            self._tw = None
        # transposed handles: `_epoch` counts calls that move the cars; the staging copy mirrors the
        # device state only while `_stage_epoch` equals it
        self._epoch = 0
        self._stage_epoch = -1
        self.leading = torch.ones((E, R), dtype=torch.int32, device=dev)
        self.lastcar = torch.ones((E, R), dtype=torch.int32, device=dev)
        # what a caller reads back after a step lives in ONE buffer (obs | rewards | done_tick | n_trips |
        # done) so that the single-env surface needs one device-to-host copy per step
        self.trip_cap = int(trip_cap)
        lay, off = {}, 0
        for name, dtype, shape in (("obs", np.int32, (E, self.obs_len)), ("rewards", np.float32, (E, I)),
                                   ("done_tick", np.int32, (E,)), ("n_trips", np.int32, (E,)),
                                   ("done", np.uint8, (E,))):
            lay[name] = (off, dtype, shape)
            off += int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._out = torch.zeros(((off + 15) // 16 * 16,), dtype=torch.uint8, device=dev)
        self._out_layout = lay

        def view(name, tdtype):
            o, dtype, shape = lay[name]
            n = int(np.prod(shape)) * np.dtype(dtype).itemsize
            return self._out[o:o + n].view(tdtype).view(*shape)
        self.obs = view("obs", torch.int32)
        self.rewards = view("rewards", torch.float32)
        self.done_tick = view("done_tick", torch.int32)
        self.done = view("done", torch.uint8)
        self.n_trips = view("n_trips", torch.int32) if validate else None
        self.waiting = torch.zeros((E, r), dtype=torch.int32, device=dev)
        self.passed_dst = torch.zeros((E, I), dtype=torch.uint8, device=dev)
        self.trip_times = torch.zeros((E, self.trip_cap), dtype=torch.float32, device=dev) if validate else None
        self._cars = torch.zeros((E, R), dtype=torch.int32, device=dev)
        b = nat.TfxBuffers()
        b.xv = _ptr(self._t if self._t is not None else self._ring)
        b.w = _ptr(self._tw if self._tw is not None else self._ringw)
        b.leading, b.lastcar = _ptr(self.leading), _ptr(self.lastcar)
        b.obs, b.rewards, b.waiting = _ptr(self.obs), _ptr(self.rewards), _ptr(self.waiting)
        b.passed_dst, b.done_tick = _ptr(self.passed_dst), _ptr(self.done_tick)
        b.trip_times, b.n_trips, b.trip_cap = _ptr(self.trip_times), _ptr(self.n_trips), self.trip_cap
        nat.check(self.lib.tfx_bind_buffers(h, C.byref(b)))
        self._action_buf = None
        self._spawn_buf = None
        self._held_action = None
        self._held_spawn = None
        self._action_bound = None
        self._spawn_bound = None
        self._stages = {}
        self.tick = 0
        # views with the reference's attribute names (traffic_env.py:372-376)
        self.passed = self.obs[:, :r]
        self.detected = self.obs[:, r:2 * r]
        self.current_phase = self.obs[:, 2 * r:2 * r + I]
        self.elapsed = self.obs[:, 2 * r + I:]

    # ---- car state in the reference's ring layout -----------------------------------------------
    def _staging(self):
        if self._ring is None:
            self._ring = torch.zeros((self.E, self.R, self.C, 2), dtype=torch.float32, device=self.device)
            if self.P == 3:
                self._ringw = torch.zeros((self.E, self.R, self.C), dtype=torch.float32, device=self.device)
            if self.het:
                self._ringa = torch.zeros((self.E, self.R, self.C), dtype=torch.uint8, device=self.device)

    def drop_staging(self):
        """Free the ring-shaped staging copy of a transposed handle (it comes back on the next access)."""
        if self._t is not None:
            self._ring = None
            self._ringw = None
            self._ringa = None
            self._stage_epoch = -1

    def _export(self):
        if self._t is not None:
            self._staging()
            with torch.cuda.device(self.device):
                nat.check(self.lib.tfx_export_ring(self.h, _ptr(self._ring), _ptr(self._ringw), _ptr(self._ringa),
                                                   self._stream()))
            self._stage_epoch = self._epoch

    @property
    def xv(self):
        """[E,R,C,2] (x, v) by ring slot.  Ring layout: the live device array.  Transposed layout: a
        staging copy brought up to date by this access; write to it, then call refresh()."""
        self._export()
        return self._ring

    @property
    def w(self):
        """[E,R,C] spawn tick by ring slot (None unless planes == 3); same staging rule as `xv`."""
        if self.P != 3:
            return None
        self._export()
        return self._ringw

    @property
    def arch(self):
        """[E,R,C] uint8: row of the archetype table each ring slot's car was spawned from (None unless the
        engine has heterogeneous cars); same staging rule as `xv`."""
        if not self.het:
            return None
        self._export()
        return self._ringa

    @property
    def x(self):
        return self.xv[..., 0]

    @property
    def v(self):
        return self.xv[..., 1]

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                self.lib.tfx_destroy(h)
            except Exception:
                pass

    # ---- stream plumbing: kernels go on torch's current stream --------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- the reference's operations ----------------------------------------------------------
    def reset(self, phase_init):
        """TrafficEnv._reset (traffic_env.py:259-272); phase_init int[E,I] or [I]."""
        ph = torch.as_tensor(np.array(np.broadcast_to(
            np.asarray(phase_init, np.int32), (self.E, self.I)))).to(self.device)
        with torch.cuda.device(self.device):
            nat.check(self.lib.tfx_reset(self.h, _ptr(ph), self._stream()))
        self._epoch += 1
        self._keep = ph
        self.tick = 0
        self.done.zero_()

    def reset_envs(self, mask, phase_init):
        """_reset for the envs selected by `mask` (bool/uint8 [E], tensor or array) only - the episode
        boundary of a batched rollout; phase_init int [E,I] (rows of unselected envs are ignored)."""
        m = mask if isinstance(mask, torch.Tensor) else torch.as_tensor(np.asarray(mask))
        m = m.to(device=self.device, dtype=torch.uint8).contiguous()
        ph = phase_init if isinstance(phase_init, torch.Tensor) else torch.as_tensor(
            np.array(np.broadcast_to(np.asarray(phase_init, np.int32), (self.E, self.I))))
        ph = ph.to(device=self.device, dtype=torch.int32).contiguous()
        with torch.cuda.device(self.device):
            nat.check(self.lib.tfx_reset_envs(self.h, _ptr(ph), _ptr(m), self._stream()))
        self._epoch += 1
        self._keep = (ph, m)
        self.done.masked_fill_(m.bool(), 0)

    def refresh(self, cars=None):
        """After writing xv / leading / lastcar from outside: push the ring-layout staging copy to
        the device layout (transposed handles) and rebuild the tail cache.

        The staging copy mirrors the device only until the cars move again.  A caller that kept a
        reference to `xv` / `x` / `v` / `w` across a step and wrote into it holds a STALE image: pushing
        it would overwrite the live cars, dropping it would silently lose the edit - so that raises.
        Take `eng.xv` again after the step (the access refreshes it), then write, then refresh().
        cars=False: rebuild the tail cache only (the caller changed leading / lastcar alone and wants
        the k-th car of a road to stay its k-th car); ring semantics for such a write need `_export()`
        BEFORE it (the env's DeviceViews do that)."""
        with torch.cuda.device(self.device):
            if self._t is not None and self._ring is not None and cars is not False:
                if self._stage_epoch != self._epoch:
                    raise nat.TfxError("refresh(): the ring-layout staging copy behind xv / x / v / w is stale (the cars "
                                       "moved since it was exported); read eng.xv again before editing it, call "
                                       "drop_staging() to discard it, or refresh(cars=False) to rebuild the tails only")
                nat.check(self.lib.tfx_import_ring(self.h, _ptr(self._ring), _ptr(self._ringw), _ptr(self._ringa),
                                                   self._stream()))
            nat.check(self.lib.tfx_refresh(self.h, self._stream()))

    def set_poisson(self, cars_per_tick, seed=0):
        """On-device Poisson arrivals (the reference's generator, traffic_env.py:160-164, with a
        counter-based RNG): cars_per_tick = cars_per_sec * rate for the whole env."""
        from gym_traffic.devrng import gap_table
        cdf = gap_table(cars_per_tick)
        nat.check(self.lib.tfx_set_poisson(self.h, float(cars_per_tick), int(seed),
                                           cdf.ctypes.data_as(C.c_void_p), int(cdf.size)))
        self._spawn_bound = None

    def set_regular(self, cars_per_tick, seed=0):
        """On-device form of the reference's `regular` generator (traffic_env.py:167-176): ceil(cars_per_tick) cars
        every round(1 / cars_per_tick) ticks (Python's round, as the reference computes it), each on an entry road drawn
        from the env's Philox stream; cars_per_tick = cars_per_sec * rate for the whole env."""
        import math
        every, burst = round(1 / cars_per_tick), math.ceil(cars_per_tick)
        nat.check(self.lib.tfx_set_regular(self.h, int(every), int(burst), int(seed)))
        self._spawn_bound = None

    def set_greedy(self, spacing=3):
        """On-device greedy controller (algorithms/greedy.py:14-16), a decision every `spacing` ticks."""
        key = ("greedy", int(spacing))
        if self._action_bound != key:
            nat.check(self.lib.tfx_set_actions(self.h, nat.ACTION_GREEDY, None, int(spacing), 0))
            self._action_bound = key

    def set_actions(self, actions=None, cycle_period=None, per_tick=False):
        """actions: int tensor/array [E,I] (or [I] broadcast; with per_tick a leading n_ticks
        dim), or cycle_period for the on-device fixed cycle.  Host arrays go through a pinned staging
        buffer into a device buffer the engine keeps per shape, held device tensors are copied into
        one, so the bound pointer - and any captured graph - stays valid from call to call; a
        per-tick device tensor is bound as it is (zero copy)."""
        if cycle_period is not None:
            key = ("cycle", int(cycle_period))
            if self._action_bound != key:      # an unchanged rule keeps the captured agent-step graph
                nat.check(self.lib.tfx_set_actions(self.h, nat.ACTION_CYCLE, None, int(cycle_period), 0))
                self._action_bound = key
            return
        a, owned = self._as_dev_i32(actions, "act_pt" if per_tick else "act")
        if per_tick:
            mode = nat.ACTION_BROADCAST if a.dim() == 2 else nat.ACTION_BUFFER
            key = ("per_tick", mode, a.data_ptr())
        else:
            mode = nat.ACTION_BROADCAST if a.dim() == 1 else nat.ACTION_BUFFER
            if not owned:
                if self._held_action is None or self._held_action.shape != a.shape:
                    self._held_action = torch.empty_like(a)
                self._held_action.copy_(a)
                a = self._held_action
            key = ("held", mode, a.data_ptr())
        self._action_buf = a
        if self._action_bound != key:
            nat.check(self.lib.tfx_set_actions(self.h, mode, _ptr(a), 0, 1 if per_tick else 0))
            self._action_bound = key

    def set_spawns(self, counts=None, period=None, per_tick=False, rows=None):
        """counts: int [E,n_entry] (with per_tick: [n_ticks,E,n_entry]); period: on-device
        one-car-every-`period`-ticks per entry road; neither: no spawns.  Buffers as in set_actions.
        rows (heterogeneous cars): uint8 [E,n_entry,S] (per_tick: [n_ticks,E,n_entry,S]) - the archetype-table
        row of the j-th car each entry road receives this tick (cars past S, or all without `rows`: row 0)."""
        if self.het and counts is not None:
            if rows is None:
                nat.check(self.lib.tfx_set_spawn_archetypes(self.h, None, 0, 0))
                self._rows_buf = None
            else:
                r8 = torch.as_tensor(np.ascontiguousarray(rows, np.uint8)).to(self.device) if not isinstance(rows, torch.Tensor) \
                    else rows.to(device=self.device, dtype=torch.uint8).contiguous()
                self._rows_buf = r8
                nat.check(self.lib.tfx_set_spawn_archetypes(self.h, _ptr(r8), int(r8.shape[-1]), 1 if per_tick else 0))
        if period is not None:
            key = ("periodic", int(period))
            if self._spawn_bound != key:
                nat.check(self.lib.tfx_set_spawns(self.h, nat.SPAWN_PERIODIC, None, int(period), 0))
                self._spawn_bound = key
        elif counts is not None:
            c, owned = self._as_dev_i32(counts, "spawn_pt" if per_tick else "spawn")
            if per_tick:
                key = ("per_tick", c.data_ptr())
            else:
                if not owned:
                    if self._held_spawn is None or self._held_spawn.shape != c.shape:
                        self._held_spawn = torch.empty_like(c)
                    self._held_spawn.copy_(c)
                    c = self._held_spawn
                key = ("held", c.data_ptr())
            self._spawn_buf = c
            if self._spawn_bound != key:
                nat.check(self.lib.tfx_set_spawns(self.h, nat.SPAWN_COUNTS, _ptr(c), 0, 1 if per_tick else 0))
                self._spawn_bound = key
        elif self._spawn_bound != ("none",):
            nat.check(self.lib.tfx_set_spawns(self.h, nat.SPAWN_NONE, None, 0, 0))
            self._spawn_bound = ("none",)

    def set_spawn_rows(self, rows, per_tick=False):
        """Heterogeneous cars: uint8 [E,n_entry,S] (per_tick: [n_ticks,E,n_entry,S]) archetype-table row of the j-th
        car each entry road receives (tfx_set_spawn_archetypes); pairs with the count buffer already bound."""
        r8 = torch.as_tensor(np.ascontiguousarray(rows, np.uint8)).to(self.device) if not isinstance(rows, torch.Tensor) \
            else rows.to(device=self.device, dtype=torch.uint8).contiguous()
        self._rows_buf = r8
        nat.check(self.lib.tfx_set_spawn_archetypes(self.h, _ptr(r8), int(r8.shape[-1]), 1 if per_tick else 0))

    def _as_dev_i32(self, a, key):
        """-> (int32 device tensor, owned).  Device tensors pass through (owned False).  Host arrays
        are written to a pinned staging buffer and copied asynchronously into a device buffer kept
        per (key, shape): `owned` True - the engine may bind that buffer itself."""
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=torch.int32).contiguous(), False
        a = np.ascontiguousarray(a, dtype=np.int32)
        slot = self._stages.get((key, a.shape))
        if slot is None:
            slot = [torch.empty(a.shape, dtype=torch.int32).pin_memory(),
                    torch.empty(a.shape, dtype=torch.int32, device=self.device), torch.cuda.Event(), False]
            self._stages[(key, a.shape)] = slot
        pin, dev, ev, used = slot
        if used:
            ev.synchronize()            # the previous upload from this staging buffer has left the host
        pin.numpy()[...] = a
        dev.copy_(pin, non_blocking=True)
        ev.record(torch.cuda.current_stream(self.device))
        slot[3] = True
        return dev, True

    def out_mirror(self):
        """Pinned mirror of the read-back buffer: pull() -> {'obs', 'rewards', 'done_tick', 'n_trips',
        'done'} as NumPy views, one copy + one synchronisation."""
        return PackedMirror(self.device, self._out, self._out_layout)

    def agent_mirror(self):
        """Pinned mirror of the fused decision's outputs: pull() -> {'aobs', 'areward'}."""
        if getattr(self, "_aobs", None) is None:
            raise nat.TfxError("agent_mirror() needs one agent_step() first")
        na = self.E * (2 * self.r + self.I)
        lay = {"aobs": (0, np.float32, (self.E, 2 * self.r + self.I)),
               "areward": (4 * na, np.float32, (self.E, self.I))}
        return PackedMirror(self.device, self._aout.view(torch.uint8), lay)

    def stage_inputs(self, actions, counts, per_tick=False):
        """Light actions int [E,I] and arrival counts int [E,n_entry] (per_tick: [n,E,n_entry]) from host
        arrays through ONE pinned staging buffer and ONE host-to-device copy; binds both inputs."""
        actions = np.asarray(actions, np.int32).reshape(self.E, self.I)
        counts = np.asarray(counts, np.int32)
        na = self.E * self.I
        key = ("inp", counts.shape)
        slot = self._stages.get(key)
        if slot is None:
            slot = [torch.empty((na + counts.size,), dtype=torch.int32).pin_memory(),
                    torch.empty((na + counts.size,), dtype=torch.int32, device=self.device), torch.cuda.Event(), False]
            self._stages[key] = slot
        pin, dev, ev, used = slot
        if used:
            ev.synchronize()
        h = pin.numpy()
        h[:na] = actions.ravel()
        h[na:] = counts.ravel()
        dev.copy_(pin, non_blocking=True)
        ev.record(torch.cuda.current_stream(self.device))
        slot[3] = True
        a, c = dev[:na], dev[na:]
        ka = ("held", nat.ACTION_BUFFER, a.data_ptr())
        if self._action_bound != ka:
            nat.check(self.lib.tfx_set_actions(self.h, nat.ACTION_BUFFER, _ptr(a), 0, 0))
            self._action_bound = ka
        ks = ("per_tick" if per_tick else "held", c.data_ptr())
        if self._spawn_bound != ks:
            nat.check(self.lib.tfx_set_spawns(self.h, nat.SPAWN_COUNTS, _ptr(c), 0, 1 if per_tick else 0))
            self._spawn_bound = ks
        self._action_buf, self._spawn_buf = a, c

    def host_mirror(self, *tensors):
        """Pinned host copies of small device tensors, refreshed together with ONE stream
        synchronisation: mirror.pull() -> list of NumPy views (valid until the next pull)."""
        return HostMirror(self.device, tensors)

    def step(self, n_ticks=1, update_done=True):
        """n_ticks x TrafficEnv._step (traffic_env.py:224-248) with the inputs set by
        set_actions / set_spawns.  Updates `done` (overflow in any of these ticks) unless the caller
        derives it from `done_tick` itself (update_done=False saves a launch)."""
        first = self.tick
        self._epoch += 1
        with torch.cuda.device(self.device):
            nat.check(self.lib.tfx_step(self.h, int(n_ticks), self._stream()))
            self.tick += int(n_ticks)
            if update_done:
                nat.check(self.lib.tfx_done(self.h, _ptr(self.done), first, self._stream()))

    def agent_step(self, n_ticks=10, remi=True):
        """One agent decision fused on the device (Repeater + Remi of traffic_test.py:27-64) with the
        inputs set by set_actions / set_spawns.  Returns (aobs f32 [E,2r+I], areward f32 [E,I],
        adone u8 [E]) - buffers owned by the engine and overwritten by the next call."""
        if getattr(self, "_aobs", None) is None:
            na, nr = self.E * (2 * self.r + self.I), self.E * self.I
            self._aout = torch.zeros((na + nr,), dtype=torch.float32, device=self.device)   # aobs | areward
            self._aobs = self._aout[:na].view(self.E, 2 * self.r + self.I)
            self._arew = self._aout[na:].view(self.E, self.I)
            # the decision's done flags ARE the engine's `done` (what reset_done() defaults to)
            self._adone = self.done
        self._epoch += 1
        with torch.cuda.device(self.device):
            nat.check(self.lib.tfx_agent_step(self.h, int(n_ticks), int(bool(remi)), _ptr(self._aobs),
                                              _ptr(self._arew), _ptr(self._adone), self._stream()))
        self.tick += int(n_ticks)
        return self._aobs, self._arew, self._adone

    def move_cars(self):
        self._epoch += 1
        with torch.cuda.device(self.device):
            nat.check(self.lib.tfx_move_cars(self.h, self._stream()))

    def advance_finished_cars(self):
        first = self.tick
        self._epoch += 1
        with torch.cuda.device(self.device):
            nat.check(self.lib.tfx_advance_finished_cars(self.h, self._stream()))
            self.tick += 1
            nat.check(self.lib.tfx_done(self.h, _ptr(self.done), first, self._stream()))

    def remi_reward(self):
        with torch.cuda.device(self.device):
            nat.check(self.lib.tfx_remi(self.h, self._stream()))
        return self.rewards

    def cars_on_roads_flat(self):
        with torch.cuda.device(self.device):
            nat.check(self.lib.tfx_cars_on_roads(self.h, _ptr(self._cars), self._stream()))
        return self._cars

    def cars_on_roads(self):
        """[E, m, n, 4] as TrafficEnv.cars_on_roads (traffic_env.py:255-257)."""
        flat = self.cars_on_roads_flat()[:, :self.r]
        return flat.reshape(self.E, 4, self.m, self.n).permute(0, 2, 3, 1)

    def set_tick(self, tick):
        nat.check(self.lib.tfx_set_tick(self.h, int(tick)))
        self.tick = int(tick)

    def vehicle_updates(self):
        out = C.c_uint64()
        nat.check(self.lib.tfx_vehicle_updates(self.h, C.byref(out), self._stream()))
        return int(out.value)

    def reset_counters(self):
        nat.check(self.lib.tfx_reset_counters(self.h, self._stream()))

    def profile(self, max_ticks):
        """Record HIP events around the kernels of the next `max_ticks` ticks (0 = off)."""
        nat.check(self.lib.tfx_profile(self.h, int(max_ticks)))

    def profile_read(self):
        mv, ad, n = C.c_double(), C.c_double(), C.c_int32()
        nat.check(self.lib.tfx_profile_read(self.h, C.byref(mv), C.byref(ad), C.byref(n)))
        return dict(move_ms=mv.value, advance_ms=ad.value, ticks=n.value)

    def fastdiv_status(self):
        en, bad = C.c_int32(), C.c_uint64()
        nat.check(self.lib.tfx_fastdiv_status(self.h, C.byref(en), C.byref(bad)))
        return dict(enabled=bool(en.value), mismatches=int(bad.value))

    def fused_ticks(self):
        """(ticks run so far by the LDS-resident multi-tick kernel k_res, whether this handle's envs fit it)."""
        n, cap = C.c_int64(), C.c_int32()
        nat.check(self.lib.tfx_fused_ticks(self.h, C.byref(n), C.byref(cap)))
        return int(n.value), bool(cap.value)

    def pair_ticks(self):
        """Ticks run so far as two-tick passes (k_move_tt + k_edge; tfx_pair_ticks)."""
        n = C.c_int64()
        nat.check(self.lib.tfx_pair_ticks(self.h, C.byref(n)))
        return int(n.value)

    def tail_ticks(self):
        """Ticks of those whose pair was finished by ONE launch, a workgroup per env (k_tail; tfx_tail_ticks)."""
        n = C.c_int64()
        nat.check(self.lib.tfx_tail_ticks(self.h, C.byref(n)))
        return int(n.value)

    def slow_pairs(self):
        """Env-pairs of fused decisions that ran one tick at a time because the first tick could overflow (tfx_slow_pairs)."""
        n = C.c_uint64()
        nat.check(self.lib.tfx_slow_pairs(self.h, C.byref(n), self._stream()))
        return int(n.value)

    def split_ticks(self):
        """Ticks of step() calls that ran as two halves of the env range on two streams (tfx_split_ticks)."""
        n = C.c_int64()
        nat.check(self.lib.tfx_split_ticks(self.h, C.byref(n)))
        return int(n.value)

    def step_kernel(self):
        """Name of the kernel that moved the cars in the last tick ('k_move_t', 'k_res', ...)."""
        return self.lib.tfx_step_kernel(self.h).decode()

    def launch_info(self):
        v = [C.c_int32() for _ in range(3)]
        nat.check(self.lib.tfx_launch_info(self.h, *[C.byref(x) for x in v]))
        return dict(grid=v[0].value, block=v[1].value, waves_per_road=v[2].value)

    def planes_numpy(self):
        """(x, v, w) as NumPy [E,R,C] copies (w is zeros when it is not carried)."""
        ring = self.xv.cpu().numpy()
        x, v = ring[..., 0], ring[..., 1]
        w = self._ringw.cpu().numpy() if self.P == 3 else np.zeros_like(x)   # (exported with xv)
        return x, v, w

    # ---- bulk state import (tests, checkpoint restore) ----------------------------------------
    def load_state(self, x, v, leading, lastcar, w=None, arch=None):
        """x, v[, w][, arch]: [E,R,C]; leading/lastcar: [E,R].  Rebuilds the kernel's tail cache."""
        self._staging()
        self._stage_epoch = self._epoch
        if self._ringa is not None and arch is not None:
            self._ringa.copy_(torch.as_tensor(np.asarray(arch, np.uint8)).to(self.device))
        self._ring[..., 0].copy_(torch.as_tensor(np.asarray(x, np.float32)).to(self.device))
        self._ring[..., 1].copy_(torch.as_tensor(np.asarray(v, np.float32)).to(self.device))
        if self._ringw is not None and w is not None:
            self._ringw.copy_(torch.as_tensor(np.asarray(w, np.float32)).to(self.device))
        self.leading.copy_(torch.as_tensor(np.asarray(leading, np.int32)).to(self.device))
        self.lastcar.copy_(torch.as_tensor(np.asarray(lastcar, np.int32)).to(self.device))
        self.refresh()

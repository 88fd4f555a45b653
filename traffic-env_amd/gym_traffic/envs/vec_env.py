"""TrafficVecEnv: E independent traffic envs stepped together on one MI355X, tensors in/out.

This is the batched form of the reference's TrafficEnv (traffic_env.py:221-394) for RL rollouts:
the same tick semantics per env, every env's arrays stacked along a leading dimension, no host
round trip inside `step`.  Spawns come either from per-env replicas of the reference's seeded
generators (`spawn='poisson'|'regular'`: host RandomState schedules, bit-identical per env to a
reference env seeded `seed + env_id`), from the on-device form of the reference's Poisson generator
(`spawn='device'`: Philox streams keyed by (seed, global env id), no host work per tick - the one to
use for throughput), from the on-device form of its `regular` generator (`spawn='regular_device'`: the reference's car
counts per tick, entry roads from the same Philox streams) or from the on-device fixed-rate rule (`spawn='periodic'`).
Sharding across GPUs is by env id (gym_traffic/distributed.py); envs share nothing.
"""
import numpy as np
import torch

from gym_traffic.core import TfxEngine
from gym_traffic.envs.roadgraph import GridRoad
from gym_traffic.spawner import ArrivalStreams


class TrafficVecEnv(object):
    def __init__(self, num_envs, m, n, length, capacity=20, rate=0.5, local_cars_per_sec=0.12,
                 spawn='poisson', spawn_period=8, entry_spec=0, learn_switch=False, validate=False,
                 seed=0, env_id_offset=0, device=None):
        self.num_envs = int(num_envs)
        self.graph = GridRoad(m, n, length)
        self.graph.generate_entrypoints(entry_spec)
        self.engine = TfxEngine(m, n, length, capacity, n_envs=num_envs, rate=rate,
                                learn_switch=learn_switch, validate=validate,
                                entry_spec=entry_spec, device=device, env_id_offset=env_id_offset)
        self.rate = float(rate)
        open_sides = 4 - bin(int(entry_spec) & 15).count('1')
        self.cars_per_sec = local_cars_per_sec * m * open_sides
        self.spawn = spawn
        self.env_id_offset = int(env_id_offset)
        eng = self.engine
        if spawn in ('poisson', 'regular'):
            # env k of this shard is global env (env_id_offset + k): its stream does not depend on
            # how the envs are sharded over GPUs
            # (replayed in C for all envs at once: reference-identical arrivals at any batch size)
            self._arrivals = ArrivalStreams([seed + self.env_id_offset + k for k in range(self.num_envs)],
                                            spawn == 'poisson', self.graph.entrypoints, eng.entry_index,
                                            max(1, eng.n_entry), self.cars_per_sec * self.rate)
        elif spawn == 'device':
            eng.set_poisson(self.cars_per_sec * self.rate, seed=seed)
        elif spawn == 'regular_device':
            eng.set_regular(self.cars_per_sec * self.rate, seed=seed)
        elif spawn == 'periodic':
            eng.set_spawns(period=spawn_period)
        elif spawn in (None, 'none'):
            eng.set_spawns()
        else:
            raise ValueError("spawn must be poisson|regular|device|regular_device|periodic|none")
        self._phase_rng = np.random.RandomState(seed + 7919 + self.env_id_offset)
        self.obs, self.rewards, self.done = eng.obs, eng.rewards, eng.done

    @property
    def observation_shape(self):
        return (self.num_envs, self.engine.obs_len)

    def reset(self, phase_init=None):
        eng = self.engine
        if phase_init is None:
            phase_init = self._phase_rng.randint(2, size=(eng.E, eng.I)).astype(np.int32)
        eng.reset(phase_init)
        return eng.obs

    def reset_done(self, done=None, phase_init=None):
        """Start a new episode in the envs that are done (default: the `done` flags of the last step
        or decision), leaving the others running; returns the mask that was reset."""
        eng = self.engine
        mask = eng.done.clone() if done is None else done      # (reset_envs clears eng.done of those envs)
        if phase_init is None:
            phase_init = self._phase_rng.randint(2, size=(eng.E, eng.I)).astype(np.int32)
        eng.reset_envs(mask, phase_init)
        return mask

    def step(self, actions=None, n_ticks=1, cycle_period=None):
        """actions: int tensor [E, I] on the device (held for n_ticks, like the Repeater wrapper,
        traffic_test.py:48-49) or None with cycle_period for the on-device fixed-cycle controller.
        Returns live device tensors (obs int32 [E,2r+2I], rewards f32 [E,I], done u8 [E])."""
        eng = self.engine
        if cycle_period is not None:
            eng.set_actions(cycle_period=cycle_period)
        elif actions is not None:
            eng.set_actions(actions)
        if self.spawn in ('poisson', 'regular'):
            counts, _ = self._arrivals.next_ticks(int(n_ticks))
            eng.set_spawns(counts=counts, per_tick=True)
        eng.step(int(n_ticks))
        return eng.obs, eng.rewards, eng.done

    def agent_step(self, actions=None, n_ticks=10, remi=True, cycle_period=None):
        """One agent decision for every env as ONE device submission: the Repeater (+ Remi) wrappers
        of the reference (traffic_test.py:27-64) fused in tfx_agent_step.  Returns device tensors
        (aobs f32 [E,2r+I], areward f32 [E,I], adone u8 [E]) owned by the engine.  An env that
        overflows stands still for the rest of the decision (`if done: break`); with host-side
        arrival schedules the arrivals drawn for its remaining ticks are dropped."""
        eng, n = self.engine, int(n_ticks)
        if cycle_period is not None:
            eng.set_actions(cycle_period=cycle_period)
        elif actions is not None:
            eng.set_actions(actions)
        if self.spawn in ('poisson', 'regular'):
            counts, _ = self._arrivals.next_ticks(n)
            eng.set_spawns(counts=counts, per_tick=True)
        return eng.agent_step(n, remi=remi)

    def remi_reward(self):
        return self.engine.remi_reward()

    # ---- validate-mode metrics, batched (reference: traffic_test.py:41-46, traffic_env.py:139-157, util.py:91-92) ----
    def light_times(self, actions):
        """float32 [E, I]: for every light the action flips, the seconds it had been in its phase -
        (elapsed + 1) * xor(current_phase, action) / 2, what the Repeater reports per decision as
        info['light_times'] in validate mode (traffic_test.py:41-46); 0 where the action flips nothing.  Call it
        BEFORE the decision's step, like the reference does.  The single-env list is `t[k][t[k] != 0]`."""
        eng = self.engine
        a = actions if isinstance(actions, torch.Tensor) else torch.as_tensor(np.asarray(actions))
        a = a.to(eng.device)
        flips = (eng.current_phase != 0) != (a != 0)
        return ((eng.elapsed + 1) * flips).to(torch.float32) / 2

    def unfinished(self):
        """int64 [E]: cars still on the train roads, `np.sum(env.cars_on_roads())` per env (util.py:92)."""
        eng = self.engine
        return eng.cars_on_roads_flat()[:, :eng.r].sum(dim=1)

    def trip_times(self, env=None):
        """Trip times (seconds) of the cars that left the map since the last reset, in the order advance_hack logs
        them (traffic_env.py:153-154): a list of float32 arrays, one per env - or env `env`'s array.  Needs
        validate=True."""
        eng = self.engine
        if eng.n_trips is None:
            raise RuntimeError("trip times are recorded in validate mode only (TrafficVecEnv(..., validate=True))")
        n = eng.n_trips.cpu().numpy()
        tt = eng.trip_times.cpu().numpy()
        if env is not None:
            return tt[env, :min(int(n[env]), eng.trip_cap)].copy()
        return [tt[k, :min(int(n[k]), eng.trip_cap)].copy() for k in range(eng.E)]

    def cars_on_roads(self):
        return self.engine.cars_on_roads()

"""Batched forms of the wrappers for E envs at once on torch tensors.

They wrap anything with the TrafficVecEnv surface - `num_envs`, `reset() -> obs [E, L]`,
`step(actions, n_ticks=1) -> (obs [E, L], rewards [E, I], done [E])` returning live tensors - and
keep every intermediate on the env's device.  Per env the numbers are those of the single-env
wrappers (gym_traffic/wrappers/{agent,history,strobe,warmup}.py); where those stop an episode early
(`if done: break`) the batched forms freeze that env's outputs instead, because the other envs go on.

    venv = TrafficVecEnv(E, m, n, length, spawn='periodic')
    venv = VecHistory(VecRemiRepeater(venv, 10), 4)        # obs [E, 4, 2r+I] per agent decision
"""
import torch


def _sample_actions(venv):
    eng = getattr(venv, 'engine', None)
    shape = (venv.num_envs, eng.I) if eng is not None else venv.action_shape
    dev = eng.device if eng is not None else getattr(venv, 'device', 'cpu')
    return torch.randint(0, 2, shape, dtype=torch.int32, device=dev)


class VecWrapper(object):
    def __init__(self, venv):
        self.venv = venv
        self.num_envs = venv.num_envs

    def __getattr__(self, name):        # engine, graph, cars_on_roads, ... of the wrapped env
        return getattr(self.venv, name)

    def reset(self, *a, **k):
        return self.venv.reset(*a, **k)

    def step(self, actions):
        return self.venv.step(actions)

    def sample_actions(self):
        inner = self.venv
        return inner.sample_actions() if hasattr(inner, 'sample_actions') else _sample_actions(inner)


class VecRemiRepeater(VecWrapper):
    """Repeater(repeat_count) followed (remi=True) by Remi: one agent decision per `step`.  On a
    TrafficVecEnv this is the fused device path (`agent_step`); on any other vec env the ticks are
    looped here with the same freezing rule."""

    def __init__(self, venv, repeat_count, remi=True):
        super(VecRemiRepeater, self).__init__(venv)
        self.repeat_count, self.remi = int(repeat_count), bool(remi)

    def reset(self, *a, **k):
        self.venv.reset(*a, **k)
        return self.step(self.sample_actions())[0]

    def step(self, actions):
        inner = self.venv
        # validate mode: what the reference's Repeater puts into info (traffic_test.py:41-46), for every env at once:
        # `self.info['light_times']` float32 [E, I], 0 where the action flips nothing
        eng = getattr(inner, 'engine', None)
        if eng is not None and eng.validate and hasattr(inner, 'light_times'):
            self.info = {'light_times': inner.light_times(actions)}
        else:
            self.info = None
        if hasattr(inner, 'agent_step'):
            return inner.agent_step(actions, self.repeat_count, remi=self.remi)
        r, i = inner.r, inner.I
        total_obs = total_reward = alive = None
        for _ in range(self.repeat_count):
            obs, rew, done = inner.step(actions)
            if total_obs is None:
                total_obs = torch.zeros((obs.shape[0], 2 * r + i), dtype=torch.float32, device=obs.device)
                total_reward = torch.zeros_like(rew, dtype=torch.float32)
                alive = torch.ones(obs.shape[0], dtype=torch.bool, device=obs.device)
            m = alive[:, None]
            sign = 2 * obs[:, -2 * i:-i] - 1
            frame = torch.cat([total_obs[:, :r] + obs[:, :r], obs[:, r:2 * r].float(),
                               (obs[:, -i:].double() / 100 * sign).float()], dim=1)
            total_obs = torch.where(m, frame, total_obs)
            total_reward = torch.where(m, total_reward + rew, total_reward)
            alive = alive & (done == 0)
        if self.remi:
            total_reward = inner.remi_reward().clone()
        return total_obs, total_reward, (~alive).to(torch.uint8)


class VecWarmup(VecWrapper):
    """reset() + `ignore_count` steps under sampled actions (wrappers/warmup.py)."""

    def __init__(self, venv, ignore_count):
        super(VecWarmup, self).__init__(venv)
        self.ignore_count = int(ignore_count)

    def reset(self, *a, **k):
        obs = self.venv.reset(*a, **k)
        for _ in range(self.ignore_count):
            obs, _, done = self.venv.step(self.sample_actions())[:3]
            assert not bool(done.any()), "Episode completed during warmup"
        return obs


class VecHistory(VecWrapper):
    """Observation [E, history_count, L]: the last frames, oldest first (wrappers/history.py).
    Frames are copied into the window, so live observation buffers are safe to wrap."""

    def __init__(self, venv, history_count):
        super(VecHistory, self).__init__(venv)
        self.history_count = int(history_count)
        self.frames = None

    def reset(self, *a, **k):
        first = self.venv.reset(*a, **k)
        self.frames = first.new_zeros((first.shape[0], self.history_count) + tuple(first.shape[1:]))
        self.frames[:, 0] = first
        for h in range(1, self.history_count):
            self.frames[:, h] = self.venv.step(self.sample_actions())[0]
        return self.frames

    def step(self, actions):
        out = self.venv.step(actions)
        self.frames = torch.roll(self.frames, -1, dims=1)
        self.frames[:, -1] = out[0]
        return (self.frames,) + tuple(out[1:])


class VecStrobe(VecWrapper):
    """Hold the action for `repeat_count` inner steps, return `num_samples` rows per env: within a
    window the columns in `sum_indices` are summed, the others keep the window's last value
    (wrappers/strobe.py).  Extra output `rows_valid [E]`: windows completed before the env was done
    (the single-env wrapper returns only those rows)."""

    def __init__(self, venv, repeat_count, num_samples, sum_indices=()):
        super(VecStrobe, self).__init__(venv)
        self.repeat_count, self.num_samples = int(repeat_count), int(num_samples)
        self.sample_size = self.repeat_count // self.num_samples
        assert self.sample_size * self.num_samples == self.repeat_count
        self.sum_indices = list(sum_indices)

    def reset(self, *a, **k):
        self.venv.reset(*a, **k)
        return self.step(self.sample_actions())[0]

    def step(self, actions):
        rows = total = alive = valid = mask = None
        for kk in range(self.repeat_count):
            obs, rew, done = self.venv.step(actions)[:3]
            if rows is None:
                rows = obs.new_zeros((obs.shape[0], self.num_samples, obs.shape[1]))
                total = torch.zeros_like(rew, dtype=torch.float32)
                alive = torch.ones(obs.shape[0], dtype=torch.bool, device=obs.device)
                valid = torch.zeros(obs.shape[0], dtype=torch.int32, device=obs.device)
                mask = obs.new_zeros((obs.shape[1],))
                mask[self.sum_indices] = 1
            w = kk // self.sample_size
            new = obs if kk % self.sample_size == 0 else rows[:, w] * mask + obs
            rows[:, w] = torch.where(alive[:, None], new, rows[:, w])
            total = torch.where(alive[:, None], total + rew, total)
            valid = torch.where(alive, torch.full_like(valid, (kk + 1) // self.sample_size), valid)
            alive = alive & (done == 0)
        return rows, total, (~alive).to(torch.uint8), valid


class VecLast(VecWrapper):
    """`repeat_count` inner steps, last observation, summed reward, no early stop (LastWrapper)."""

    def __init__(self, venv, repeat_count):
        super(VecLast, self).__init__(venv)
        self.repeat_count = int(repeat_count)

    def step(self, actions):
        total = None
        for _ in range(self.repeat_count):
            obs, rew, done = self.venv.step(actions)[:3]
            total = rew.clone().float() if total is None else total + rew
        return obs, total, done


class VecLocalize(VecWrapper):
    """reward[e, i] <- mean_j(reward[e, j] + (w - 1) * [i == j] * reward[e, i]) / w (LocalizeWrapper)."""

    def __init__(self, venv, local_weight):
        super(VecLocalize, self).__init__(venv)
        self.local_weight = local_weight

    def step(self, actions):
        out = self.venv.step(actions)
        rew, w = out[1], self.local_weight
        n = rew.shape[1]
        shaped = (rew.mean(dim=1, keepdim=True) + rew * (w - 1) / n) / w
        return (out[0], shaped) + tuple(out[2:])


class VecSquish(VecWrapper):
    """reward[e] <- mean_i reward[e, i] (SquishReward)."""

    def step(self, actions):
        out = self.venv.step(actions)
        return (out[0], out[1].mean(dim=1)) + tuple(out[2:])

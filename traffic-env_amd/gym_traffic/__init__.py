"""gym_traffic on MI355X: the reference's package surface over hand-written HIP kernels.

Importing this package does what the reference's gym_traffic/__init__.py:1-23 does - adds
`rendering` / `reward_size` to `gym.Env`, makes `step` render-then-`_step`, propagates
`reward_size` through wrappers, registers 'traffic-v0' - and first makes a `gym` available when
none is installed (gym_traffic/_gymcompat.py).
"""
import sys

try:  # a real (old-API) gym, if the user has one
    import gym  # noqa: F401
    _HAVE_REAL_GYM = not getattr(gym, '__tfx_compat__', False)
except ImportError:
    from . import _gymcompat
    gym = _gymcompat.install()
    _HAVE_REAL_GYM = False

if _HAVE_REAL_GYM and not getattr(gym.Env, '_tfx_patched', False):
    def _step_with_render(self, action):
        if self.rendering:
            self.render()
        return self._step(action)

    gym.Env.rendering = False
    gym.Env.step = _step_with_render
    gym.Env.reward_size = 1
    _wrapper_init = gym.Wrapper.__init__

    def _init_keeps_reward_size(self, env, *a, **k):
        _wrapper_init(self, env, *a, **k)
        self.reward_size = env.reward_size

    gym.Wrapper.__init__ = _init_keeps_reward_size
    gym.Env._tfx_patched = True

from gym.envs.registration import register  # noqa: E402

try:
    register(id='traffic-v0', entry_point='gym_traffic.envs:TrafficEnv')
except Exception:  # already registered (module reloaded)
    pass

__all__ = ['gym']

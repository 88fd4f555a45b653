"""Wrappers around the traffic env: the callers' side of the hot path.

Single-env wrappers speak the 2017 gym protocol (`_step/_reset`) and keep the reference's names and
factory style - `HistoryWrapper(n)(env)`, `StrobeWrapper(...)(env)`, `WarmupWrapper(n)(env)`
(reference gym_traffic/wrappers/*.py) and `Repeater(n)(env)`, `Remi(env)`, `LocalizeWrapper`,
`SquishReward` (reference traffic_test.py:27-75).  `vec` holds the same transformations for
`TrafficVecEnv`: E envs at once on device tensors, no host round trip.
"""


def preset(cls, name, **fixed):
    """A subclass of wrapper class `cls` whose constructor takes only the env; the remaining
    constructor arguments are fixed here.  This is how the reference's wrapper factories are used:
    `HistoryWrapper(4)` is a class, `HistoryWrapper(4)(env)` an env."""
    def __init__(self, env):
        cls.__init__(self, env, **fixed)
    return type(name, (cls,), {'__init__': __init__, '__doc__': cls.__doc__})

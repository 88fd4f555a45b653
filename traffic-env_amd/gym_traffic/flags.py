"""Access to the run flags the env reads every tick.

The reference keeps flags in a process-global `args.FLAGS` namespace (reference args.py:1-38) that
the agents and `traffic_test.py` also read and mutate.  When that top-level `args` module is
importable (a user running the reference's drivers against this package) the SAME object is used
and the env's five flags are registered on its parser exactly as the reference's env module does
(traffic_env.py:11-15).  Otherwise a small stand-alone namespace with the same defaults is used.
"""

ENV_DEFAULTS = dict(local_cars_per_sec=0.12, rate=0.5, poisson=True, entry='all', learn_switch=False)


class _Flags(object):
    def __init__(self):
        self.__dict__.update(ENV_DEFAULTS)
        self.mode = 'train'

    def __getattr__(self, name):
        raise AttributeError(name)


def _bind():
    try:
        import args as ref_args  # the reference's flag module, if the user has it on sys.path
    except ImportError:
        return _Flags(), None
    flags = ref_args.FLAGS
    known = getattr(ref_args.PARSER, 'defaults', {})
    for name, default in ENV_DEFAULTS.items():
        if name not in known:
            kw = {} if isinstance(default, str) else {'type': type(default)}
            ref_args.add_argument('--' + name, default, **kw)
    return flags, ref_args


FLAGS, _REF_ARGS = _bind()


def flag(name, default=None):
    """FLAGS.<name>, or `default` when the flag system has no such flag (e.g. `mode` before
    alg_flags was imported - the reference would raise there, traffic_env.py:240)."""
    try:
        return getattr(FLAGS, name)
    except AttributeError:
        return default


def update_flags(**kw):
    if _REF_ARGS is not None:
        _REF_ARGS.update_flags(**kw)
    else:
        FLAGS.__dict__.update(kw)

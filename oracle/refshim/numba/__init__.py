"""Identity stand-in for `numba`, used ONLY by oracle/ref_loader.py to import the
reference package in the build container (numba is not installed; no network).

`jit(...)` returns the decorated function unchanged, so the reference's L1 kernels
run as the pure-NumPy code they are written as.  Type names are inert objects that
accept `[...]` and `(...)` so signature expressions such as
`float32[:](float32, float32[:,:])` evaluate.  TEST INFRASTRUCTURE - never shipped
on the product path.
"""


class _Ty:
    def __getitem__(self, _):
        return self

    def __call__(self, *a, **k):
        return self


void = float64 = float32 = int64 = int32 = uint32 = boolean = _Ty()


def jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not isinstance(args[0], _Ty) and not kwargs:
        return args[0]

    def deco(fn):
        return fn

    return deco

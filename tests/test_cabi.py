"""The C-ABI library loads on a CPU-only box and exports every symbol include/tfx.h declares.
No compute call is made here (there is no GPU); argument validation that happens before any
device work is exercised too."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "tfx.h")
LIB = os.path.join(ROOT, "traffic-env_amd", "lib", "libtfx_hip.so")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tfx_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "traffic-env_amd", "csrc")])
    return C.CDLL(LIB)


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 20 and "tfx_step" in names and "tfx_move_cars" in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert missing == []


def test_abi_version_matches_binding(lib):
    from gym_traffic import _native
    assert lib.tfx_abi_version() == _native.ABI_VERSION
    hdr = int(re.search(r"#define\s+TFX_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert hdr == _native.ABI_VERSION


def test_struct_layouts_match_header():
    """ctypes mirrors of tfx_config / tfx_buffers list the header's fields in the header's order."""
    from gym_traffic import _native
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)

    def fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), src, re.S).group(1)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            out += [re.sub(r"\[.*", "", n.strip().lstrip("*")) for n in re.sub(r"^[a-z0-9_]+\s+", "", decl).split(",")]
        return out
    assert fields("tfx_config") == [f[0] for f in _native.TfxConfig._fields_]
    assert fields("tfx_buffers") == [f[0] for f in _native.TfxBuffers._fields_]


def test_errors_are_codes_not_crashes(lib):
    lib.tfx_last_error.restype = C.c_char_p
    assert lib.tfx_create(None, None) < 0
    assert b"null" in lib.tfx_last_error()
    assert lib.tfx_step(None, 1, None) < 0
    assert lib.tfx_destroy(None) == 0
    from gym_traffic import _native
    cfg = _native.TfxConfig()
    cfg.m, cfg.n, cfg.capacity, cfg.n_envs, cfg.planes = 2, 2, 2, 1, 2     # capacity too small
    h = C.c_void_p()
    assert lib.tfx_create(C.byref(cfg), C.byref(h)) == -1
    assert b"capacity" in lib.tfx_last_error()


def test_product_path_has_no_cpu_fallback():
    """Without a GPU the engine must refuse to run rather than route to any CPU code."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gym_traffic.core import TfxEngine
    from gym_traffic._native import TfxError
    with pytest.raises(TfxError):
        TfxEngine(2, 2, 100.0, 10)
    # and nothing under the product package imports the oracle
    pkg = os.path.join(ROOT, "traffic-env_amd", "gym_traffic")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f

"""The README's quick start, verbatim: the first ```python block of README.md is executed as it stands."""
import os
import re

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_readme_quick_start_runs():
    text = open(os.path.join(ROOT, "README.md")).read()
    block = re.search(r"```python\n(.*?)```", text, re.S).group(1)
    assert "gym.make('traffic-v0')" in block and "TrafficVecEnv" in block
    cwd = os.getcwd()
    os.chdir(ROOT)                          # (the block's sys.path entry is relative to the repo root)
    try:
        scope = {}
        exec(compile(block, "README.md", "exec"), scope)
    finally:
        os.chdir(cwd)
    assert scope["obs"].shape == (81,) and not scope["done"]
    assert tuple(scope["aobs"].shape) == (4096, 2 * 1024 + 256) and tuple(scope["adone"].shape) == (4096,)

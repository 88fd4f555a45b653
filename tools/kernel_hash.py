#!/usr/bin/env python3
"""Hashes of the MACHINE CODE of the kernels in libtfx_hip.so, per kernel family.

    python tools/kernel_hash.py            # {"k_tail": "...", "k_move_tt": "...", ...} as JSON
    python tools/kernel_hash.py k_tail     # one hash

The gfx950 code object is taken out of the shared library (llvm-objdump --offloading), disassembled, and the
instruction text of every instantiation of a kernel template (addresses and encodings stripped) is hashed, sorted by
symbol.  A profile summary carries the hash of the kernel it measured, taken ON THE BOX from the library that ran
(tools/profile_round.sh); bench.py reports `roofline.traffic` only while the library it drives has the same hash for
that kernel.  Comments, host code and other kernels do not unpin a summary; any change to the kernel's code does -
and there is nothing to re-stamp by hand: the hash travels with the counters.
"""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("TFX_LIB", os.path.join(ROOT, "traffic-env_amd", "lib", "libtfx_hip.so"))
LLVM = "/opt/rocm/lib/llvm/bin"


def family(symbol):
    """_ZN3tfx9k_move_ttILb1E... -> k_move_tt"""
    m = re.match(r"_ZN3tfx(\d+)", symbol)
    if not m:
        return None
    n = int(m.group(1))
    name = symbol[len(m.group(0)):len(m.group(0)) + n]
    return name if name.startswith("k_") else None


def kernel_hashes(lib=LIB):
    tmp = tempfile.mkdtemp(prefix="tfxhash")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = [f for f in os.listdir(tmp) if "amdgcn" in f]
        if not cos:
            raise RuntimeError("no gfx950 code object in %s" % lib)
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, cos[0])],
                             check=True, stdout=subprocess.PIPE, universal_newlines=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    per = {}
    cur = None
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            fam = family(m.group(1))
            cur = (fam, m.group(1)) if fam else None
            if cur:
                per.setdefault(fam, {})[m.group(1)] = hashlib.sha256()
            continue
        if cur and line.strip():
            # "\ts_load_dwordx2 s[0:1], s[4:5], 0x0      // 000000001000: ..." -> the instruction text only
            text = line.split("//")[0].strip()
            per[cur[0]][cur[1]].update(text.encode() + b"\n")
    out = {}
    for fam, syms in per.items():
        h = hashlib.sha256()
        for s in sorted(syms):
            h.update(s.encode())
            h.update(syms[s].digest())
        out[fam] = h.hexdigest()[:16]
    return out


if __name__ == "__main__":
    hs = kernel_hashes()
    if len(sys.argv) > 1:
        print(hs[sys.argv[1]])
    else:
        print(json.dumps(hs, indent=1, sort_keys=True))

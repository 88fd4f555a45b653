#!/usr/bin/env python3
"""Instruction-class histogram of a kernel's inner loop from the compiler's own ISA (make -C traffic-env_amd/csrc asm).

  tools/isa_histogram.py [--kernel _ZN3tfx9k_move_ttILb1ELb0ELb0EEEvNS_3DevEii] [--rows 3.34e6]

Takes every basic block the compiler marks as part of a depth-2 loop of the kernel (the walk over the rows of a tile,
unrolled P = 4 times), drops the blocks of the literal-division fallback (they hold three IEEE division expansions; the
fast path holds one per IDM step), and prints vector instructions per ROW by issue class, with the vector-ALU time that
count implies for `--rows` row-iterations per launch:

    cycles = sum(count[class] * issue_cycles[class]),  time = rows * cycles / (1024 SIMDs * 2.4 GHz)

Issue cycles per wave64 instruction on one SIMD with >= 2 wavefronts resident (MI355X_MICROARCH.md, per-instruction
constants): plain fp32 / int VALU 2; transcendental (v_rcp_f32 ...) 4 (twice a plain one, as in the guide's one-wave
row: 8 against 4); binary64 multiply and the f32<->f64 conversions 4 (half rate: 78.6 TF fp64 vector against 157.3 fp32).
"""
import argparse
import collections
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = os.path.join(ROOT, "build", "tfx_hip-hip-amdgcn-amd-amdhsa-gfx950.s")

CLASSES = [
    ("transcendental", 4, re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_")),
    ("binary64 / conversions", 4, re.compile(r"^v_(mul_f64|fma_f64|add_f64|cvt_f64_f32|cvt_f32_f64)")),
    ("division helpers (div_scale/fmas/fixup)", 2, re.compile(r"^v_div_")),
    ("fp32 arithmetic", 2, re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mac|max|min|pk_)")),
    ("compare / select", 2, re.compile(r"^v_(cmp|cndmask|addc|subb)")),
    ("moves", 2, re.compile(r"^v_(mov|readlane|writelane|readfirstlane)")),
    ("integer / address", 2, re.compile(r"^v_")),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="_ZN3tfx9k_move_ttILb1ELb0ELb0EEEvNS_3DevEii")
    ap.add_argument("--rows", type=float, default=4096 * 17 * 47.45, help="row-iterations per launch (cfg2: tiles x mean road length)")
    ap.add_argument("--unroll", type=int, default=4)
    a = ap.parse_args()
    if not os.path.exists(ASM):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "traffic-env_amd", "csrc"), "asm"], stdout=subprocess.DEVNULL)
    text = open(ASM).read()
    start = text.index("\n%s:" % a.kernel)
    body = text[start:text.index(".Lfunc_end", start)]
    blocks, cur = [], None
    for line in body.split("\n"):
        m = re.match(r"^(\.LBB\d+_\d+):|^; %bb\.(\d+)", line)
        if m:
            cur = {"name": m.group(1) or ("%%bb.%s" % m.group(2)), "loop": None, "ins": []}
            blocks.append(cur)
            h = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=2", line)
            if h:
                cur["loop"] = h.group(1)
            continue
        if cur is None:
            continue
        if "This Inner Loop Header: Depth=2" in line:
            cur["loop"] = cur["name"].lstrip(".L")
        h = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=2", line)
        if h and line.lstrip().startswith(";"):
            cur["loop"] = h.group(1)
        t = line.strip().split()
        if t and re.match(r"^[vs]_|^global_|^ds_|^buffer_", t[0]):
            cur["ins"].append(t[0])
    # the walk over the rows is the depth-2 loop with the most instructions (the others: arrivals, reductions)
    size = collections.Counter()
    for b in blocks:
        if b["loop"]:
            size[b["loop"]] += len(b["ins"])
    main_loop = size.most_common(1)[0][0]
    inner = [b for b in blocks if b["loop"] == main_loop]
    fast = [b for b in inner if sum(i.startswith("v_div_fixup") for i in b["ins"]) < 3]
    hist = collections.Counter()
    salu = mem = 0
    for b in fast:
        for i in b["ins"]:
            if i.startswith("s_"):
                salu += 1
            elif i.startswith(("global_", "ds_", "buffer_")):
                mem += 1
            else:
                for name, cyc, rx in CLASSES:
                    if rx.match(i):
                        hist[name] += 1
                        break
    u = float(a.unroll)
    total = cyc_total = 0.0
    print("kernel %s: %d blocks in the inner loop, %d on the fast path; per row (loop unrolled %d times)" %
          (a.kernel, len(inner), len(fast), a.unroll))
    for name, cyc, _ in CLASSES:
        n = hist[name] / u
        total += n
        cyc_total += n * cyc
        print("  %-42s %6.1f  x %d cycles" % (name, n, cyc))
    print("  %-42s %6.1f  (scalar %.1f, memory %.1f)" % ("vector instructions per row", total, salu / u, mem / u))
    t = a.rows * cyc_total / (1024 * 2.4e9)
    print("  issue cycles per row %.0f -> %.3f ms of vector-ALU issue per launch for %.3g rows at 2.4 GHz" %
          (cyc_total, t * 1e3, a.rows))


if __name__ == "__main__":
    main()

"""Registers, LDS, scratch and memory-instruction mix of every kernel in the library's gfx950 code object:
   python3 tools/kernel_regs.py [substring of the demangled name]
A kernel whose per-road pointers were moved onto LDS copies (k_tail, the resident kernels) must show ds_ accesses and no
flat_ ones; a big function that the inliner left as a real call shows up as a stack frame (scratch) and flat accesses -
k_tail<AGENT, HET> lost 26 % of a decision that way until round 4 (tests/test_host_logic.py keeps watch)."""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "traffic-env_amd", "lib", "libtfx_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
KEYS = ("ds_", "flat_", "global_", "scratch_", "s_barrier")


def kernel_table(lib=LIB):
    """[{name, vgpr, sgpr, lds, scratch, insts, ds_, flat_, global_, scratch_, s_barrier}] for every kernel"""
    tmp = tempfile.mkdtemp(prefix="tfxregs")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = os.path.join(tmp, [f for f in os.listdir(tmp) if "amdgcn" in f][0])
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True,
                             stdout=subprocess.PIPE, universal_newlines=True).stdout
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True,
                               stdout=subprocess.PIPE, universal_newlines=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    mix = {}
    for f in re.split(r"\n(?=[0-9a-f]+ <)", dis):
        m = re.match(r"[0-9a-f]+ <([^>]+)>", f)
        if m:
            mix[m.group(1)] = dict({k: len(re.findall(r"\b" + k, f)) for k in KEYS}, insts=f.count("\n"))
    blocks = notes.split("- .agpr_count")[1:]
    names = [re.search(r"\.name:\s+(\S+)", b).group(1) for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, universal_newlines=True).stdout.split("\n")
    out = []
    for b, nm, dn in zip(blocks, names, dem):
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", b).group(1))
        out.append(dict({"name": dn.replace("tfx::", ""), "vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"),
                         "lds": g("group_segment_fixed_size"), "scratch": g("private_segment_fixed_size")}, **mix.get(nm, {})))
    return out


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    for k in kernel_table():
        if pat in k["name"]:
            print("%-72s vgpr %3d sgpr %3d lds %6d scratch %4d  %s" % (k["name"][:72], k["vgpr"], k["sgpr"], k["lds"], k["scratch"],
                  {q: k.get(q) for q in KEYS + ("insts",)}))

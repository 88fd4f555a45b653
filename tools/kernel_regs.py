"""Registers, LDS, scratch and memory-instruction mix of the kernels whose name contains the argument:
   python3 tools/kernel_regs.py k_grid"""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "traffic-env_amd", "lib", "libtfx_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
pat = sys.argv[1] if len(sys.argv) > 1 else "k_"
tmp = tempfile.mkdtemp(prefix="tfxregs")
try:
    so = os.path.join(tmp, "lib.so")
    shutil.copy(LIB, so)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=tmp, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    co = os.path.join(tmp, [f for f in os.listdir(tmp) if "amdgcn" in f][0])
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True,
                         stdout=subprocess.PIPE, universal_newlines=True).stdout
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True,
                           stdout=subprocess.PIPE, universal_newlines=True).stdout
    demangle = lambda n: subprocess.run(["c++filt", n], stdout=subprocess.PIPE,
                                        universal_newlines=True).stdout.strip()
finally:
    shutil.rmtree(tmp, ignore_errors=True)
mix = {}
for f in re.split(r"\n(?=[0-9a-f]+ <)", dis):
    m = re.match(r"[0-9a-f]+ <([^>]+)>", f)
    if m:
        mix[m.group(1)] = {k: len(re.findall(r"\b" + k, f)) for k in ("ds_", "flat_", "global_", "scratch_", "buffer_wbl2", "buffer_inv", "s_barrier")}
        mix[m.group(1)]["insts"] = f.count("\n")
for b in notes.split("- .agpr_count")[1:]:
    nm = re.search(r"\.name:\s+(\S+)", b).group(1)
    dn = demangle(nm)
    if pat not in dn:
        continue
    g = lambda k: re.search(r"\." + k + r":\s+(\d+)", b).group(1)
    print("%-72s vgpr %3s sgpr %3s lds %6s scratch %4s  %s" % (dn.replace("tfx::", "")[:72], g("vgpr_count"), g("sgpr_count"),
          g("group_segment_fixed_size"), g("private_segment_fixed_size"), mix.get(nm, {})))

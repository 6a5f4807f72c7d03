"""Register / scratch / LDS figures of the gfx950 kernels in a built object (code-object metadata).
usage: python tools/kernel_regs.py [object, default nr-slam_amd/build/nrs_engine.hip.o] [name filter ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
args = sys.argv[1:]
obj = args.pop(0) if args and args[0].endswith(".o") else os.path.join(ROOT, "nr-slam_amd/build/nrs_engine.hip.o")
with tempfile.TemporaryDirectory() as td:
    co, fat = os.path.join(td, "k.co"), os.path.join(td, "fat.bin")
    subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
rows = []
for b in notes.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, b).group(1))
    rows.append((name, g("vgpr_count"), int(b.split()[0]), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
dn = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, d in zip(rows, dn):
    d = re.sub(r"^void nrs::", "", d)
    d = re.sub(r"\(.*$", "", d)
    if args and not any(a in d for a in args):
        continue
    print("%-70s vgpr %3d agpr %3d spill v%d s%d scratch %3d lds %d" % (d[:70], r[1], r[2], r[3], r[4], r[5], r[6]))

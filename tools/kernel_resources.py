"""VGPR / SGPR / scratch / LDS of every kernel in liblupin_hip.so, from the code object's metadata notes.
usage: python tools/kernel_resources.py [name pattern]"""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, "lupinpathtracer_amd", "liblupin_hip.so")
pat = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as tmp:
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", so, os.path.join(tmp, "copy.so")])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}", "--unbundle"])
    notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
rows = []
for blk in re.split(r"\n\s*- \.agpr_count", notes)[1:]:
    def g(k):
        m = re.search(r"\." + k + r":\s*(\S+)", blk)
        return m.group(1) if m else "?"
    name = g("name")
    if pat in name:
        rows.append((name, g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    print(f"vgpr {r[1]:>4} sgpr {r[2]:>4} scratch {r[3]:>5} lds {r[4]:>6}  {n[:150]}")

"""Disassembly of one kernel of liblupin_hip.so: python tools/disasm_kernel.py <mangled-name substring> > out.s"""
import os, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
so = os.path.join(ROOT, "lupinpathtracer_amd", "liblupin_hip.so")
with tempfile.TemporaryDirectory() as tmp:
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", so, os.path.join(tmp, "copy.so")])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}", "--unbundle"])
    syms = subprocess.check_output([f"{LLVM}/llvm-readelf", "-sW", co], text=True).split("\n")
    names = [l.split()[-1] for l in syms if " FUNC " in l and sys.argv[1] in l]
    for n in names:
        sys.stdout.write(subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", f"--disassemble-symbols={n}", co], text=True))

"""Static checks on the code object inside liblupin_hip.so (no GPU needed; tests/test_isa_guard.py runs them).

Why: round 2's k_shade<Direct> faulted on hardware once the compiler stopped inlining integrate_vertex -- the out-of-line
callee reached its by-reference arguments (SceneDev / PathRegs / ShadowRays in the caller's scratch frame) through ~380
FLAT loads and stores of generic pointers, the only place in the library where scratch was addressed that way.
Everything is force-inlined since; this keeps it so:
  * no device function call in any kernel (s_swappc / s_call) and no function symbol besides the kernels; s_setpc appears only
    as the tail of a long-branch expansion (s_getpc / s_add / s_addc / s_setpc inside one kernel: the 140 KB shade kernels
    exceed the 16-bit branch range) and is reported, not refused;
  * no kernel needs a dynamic stack;
  * scratch (private segment) per kernel stays below the caps below: 0 for the default pipeline's kernels, a stated size
    for the few that spill a handful of registers.
usage: python tools/isa_guard.py  ->  JSON report on stdout, exit status 1 on a violation
"""
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "lupinpathtracer_amd", "liblupin_hip.so")

# kernels allowed to use scratch at all: demangled-name prefix -> cap in bytes (spilled registers, no stack objects passed by reference)
SCRATCH_CAPS = {
    "void k_shade<1,": 64,       # MIS with the marches inline (LUPIN_LIGHT_STAGE=0; the default, k_shade<1, .., DEFER>, has none): was 448 in round 2
    "void k_shade<3,": 64,       # Direct
    "void k_shade<0, true, true": 32, "void k_shade<0, false, true": 32,   # SIMPLE (matte-only) specialisation at 128 VGPRs
    "void k_shadow<": 64,
    "void k_light_pdf<": 16,     # the opt-in light-pdf stage of the Standard integrator (LUPIN_LIGHT_STAGE=1)
    # the short-stack pass of the tracer is compiled for six waves per SIMD (80 registers): a few kernel-argument pointers are
    # spilled in the prologue and reloaded in the triangle / end-of-traversal phases, none inside the node loop
    "void k_extend_persistent<": 48,
}


def code_object(tmp):
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", SO, os.path.join(tmp, "copy.so")])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}", "--unbundle"])
    return co


def kernels(co):
    notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    out = []
    for blk in re.split(r"\n\s*- \.agpr_count", notes)[1:]:
        def g(k, default="0"):
            m = re.search(r"\." + k + r":\s*(\S+)", blk)
            return m.group(1) if m else default
        out.append({"symbol": g("name", "?"), "vgpr": int(g("vgpr_count")), "sgpr": int(g("sgpr_count")), "scratch": int(g("private_segment_fixed_size")),
                    "dynamic_stack": g("uses_dynamic_stack", "false") == "true"})
    names = subprocess.run(["c++filt"], input="\n".join(k["symbol"] for k in out), capture_output=True, text=True).stdout.split("\n")
    for k, n in zip(out, names):
        k["name"] = n
    return out


def check():
    with tempfile.TemporaryDirectory() as tmp:
        co = code_object(tmp)
        ks = kernels(co)
        dis = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", co], text=True)
        funcs = [l.split()[-1] for l in subprocess.check_output([f"{LLVM}/llvm-readelf", "-sW", co], text=True).split("\n") if " FUNC " in l]
    calls = len(re.findall(r"\bs_swappc_b64\b|\bs_call_b64\b", dis))
    setpc = len(re.findall(r"\bs_setpc_b64\b", dis))
    kernel_symbols = {k["symbol"] for k in ks}
    device_functions = sorted(f for f in funcs if f not in kernel_symbols and not f.endswith(".kd"))
    violations = []
    long_branches = len(re.findall(r"s_getpc_b64[^\n]*\n[^\n]*s_add_u32[^\n]*\n[^\n]*s_addc_u32[^\n]*\n[^\n]*s_setpc_b64", dis))
    if calls:
        violations.append(f"{calls} device calls (s_swappc / s_call) in the code object: a helper is no longer inlined")
    if setpc != long_branches:
        violations.append(f"{setpc} s_setpc but only {long_branches} of them are long-branch expansions")
    if device_functions:
        violations.append(f"out-of-line device functions: {device_functions[:5]}")
    for k in ks:
        if k["dynamic_stack"]:
            violations.append(f"{k['name'][:80]}: uses a dynamic stack")
        cap = max([c for p, c in SCRATCH_CAPS.items() if k["name"].startswith(p)] or [0])
        if k["scratch"] > cap:
            violations.append(f"{k['name'][:100]}: {k['scratch']} B of scratch (cap {cap})")
    return {"kernels": len(ks), "device_calls": calls, "long_branches": long_branches, "device_functions": len(device_functions),
            "kernels_with_scratch": {k["name"][:70]: k["scratch"] for k in ks if k["scratch"]}, "max_vgpr": max(k["vgpr"] for k in ks),
            "so_bytes": os.path.getsize(SO), "violations": violations}


if __name__ == "__main__":
    rep = check()
    print(json.dumps(rep, indent=1))
    sys.exit(1 if rep["violations"] else 0)

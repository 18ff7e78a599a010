"""Fail (exit 1) when a built library contains packed fp32 VALU arithmetic (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32):
    python tools/check_no_packed_fp32.py sign-language-nlp_amd/lib/libslnlp.so
Those instructions compute wrongly in lanes 48-63 while another kernel's waves run MFMA on the same CU (MI355X / ROCm 7.2;
DESIGN.md section 6, tools/probes/PACKED_FP32_REPORT.md).  The Makefile builds without them through an internal target-feature
name; this disassembles every embedded code object so that a toolchain that renames or ignores the flag fails the BUILD, not a
grid search weeks later.  Used by sign-language-nlp_amd/Makefile (post-link) and tests/test_hygiene_cpu.py."""
import os, re, subprocess, sys, tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_registers as kr

def find_objdump():
    """llvm-objdump of the ROCm toolchain in use: next to $HIPCC's clang, under $ROCM_PATH / $HIP_PATH / /opt/rocm, or on PATH."""
    import shutil
    cands = []
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc")
    if hipcc:
        root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
        cands += [os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"), os.path.join(root, "llvm", "bin", "llvm-objdump")]
    for var in ("ROCM_PATH", "HIP_PATH"):
        if os.environ.get(var):
            cands.append(os.path.join(os.environ[var], "lib", "llvm", "bin", "llvm-objdump"))
    cands.append("/opt/rocm/lib/llvm/bin/llvm-objdump")
    for c in cands:
        if os.path.exists(c):
            return c
    return shutil.which("llvm-objdump")


OBJDUMP = find_objdump()


def packed_fp32_hits(lib):
    """(disassembly lines, [matches]) over every AMDGPU code object inside `lib`."""
    lines, hits = 0, []
    for _, elf in kr.code_objects(open(lib, "rb").read()):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            out = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout
        lines += out.count("\n")
        hits += re.findall(r"v_pk_(?:add|mul|fma)_f32", out)      # (v_pk_mov_b32, a move, stays: the builds that pass the probes have it)
    return lines, hits


if __name__ == "__main__":
    lib = sys.argv[1]
    if not OBJDUMP:
        # exit code 2 = "cannot verify" (the Makefile keeps the library and says so), 1 = packed fp32 found (the Makefile deletes it)
        print(f"check_no_packed_fp32: no llvm-objdump (looked beside hipcc, under ROCM_PATH / HIP_PATH, /opt/rocm, on PATH) -- cannot verify {lib}", file=sys.stderr)
        sys.exit(2)
    n, hits = packed_fp32_hits(lib)
    if n < 1000:
        print(f"check_no_packed_fp32: disassembly of {lib} looks empty ({n} lines)", file=sys.stderr)
        sys.exit(1)
    if hits:
        print(f"check_no_packed_fp32: {len(hits)} packed fp32 instructions in {lib} (first: {hits[0]}) -- the toolchain ignored "
              "-target-feature -packed-fp32-ops", file=sys.stderr)
        sys.exit(1)
    print(f"check_no_packed_fp32: {lib}: {n} lines of gfx950 code, no v_pk_{{add,mul,fma}}_f32")

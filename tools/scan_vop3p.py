#!/usr/bin/env python3
"""Scan the device code of a library build for the packed-f32 instruction forms that are UNSAFE next to f16 matrix instructions on
gfx950 (profiles/r4_sp_root_cause.md, tools/probe_pk_opsel.hip):

    v_pk_fma_f32 / v_pk_mul_f32 whose op_sel selects the HIGH register of a source pair for the LOW half of src1 (op_sel[1] = 1),
    or of src2 (op_sel[2] = 1; measured wrong when src2 is the same pair as src1 -- hipcc's `x * {t.x, t.x} + {t.y, t.y}` form --
    and treated as unsafe in general),

return a wrong low half when another wave of the SIMD issues v_mfma_f32_16x16x32_f16 (rarely: _bf16, i32_16x16x64_i8).  With the
split-precision conv path one decoder's f16-MFMA kernels run beside the other decoder's kernels, so NO kernel of the library may
contain these forms.  (v_pk_add_f32 in every form, op_sel on src0 and every op_sel_hi form measured clean.)

    python tools/scan_vop3p.py [path/to/libwslhip.so]        exit status 1 if an unsafe instruction is found
Used by tests/test_abi.py::test_no_packed_f32_forms_unsafe_next_to_f16_mfma."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
PK = re.compile(r"\b(v_pk_fma_f32|v_pk_mul_f32)\b([^;/]*)")
OPSEL = re.compile(r"op_sel:\[([01](?:,[01])*)\]")


def unsafe(line):
    m = PK.search(line)
    if not m:
        return False
    s = OPSEL.search(m.group(2))
    if not s:
        return False
    bits = [int(b) for b in s.group(1).split(",")]
    return any(bits[1:])          # op_sel of src1 (and of src2 for the fma)


def code_objects(lib, work):
    """device code objects (gfx950) embedded in a hipcc-built shared library"""
    tmp = os.path.join(work, os.path.basename(lib))
    shutil.copy(lib, tmp)
    subprocess.run([OBJDUMP, "--offloading", tmp], cwd=work, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return sorted(os.path.join(work, f) for f in os.listdir(work) if "amdgcn" in f)


def scan(lib):
    """-> (number of packed f32 multiply / fma instructions, [(kernel, instruction text)] of the unsafe ones)"""
    total, bad = 0, []
    with tempfile.TemporaryDirectory() as work:
        for co in code_objects(lib, work):
            out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
            kern = "?"
            for line in out.splitlines():
                if line.endswith(">:"):
                    kern = line.split("<")[-1][:-2]
                    continue
                if "v_pk_fma_f32" in line or "v_pk_mul_f32" in line:
                    total += 1
                    if unsafe(line):
                        bad.append((kern, line.split("//")[0].strip()))
    return total, bad


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "wsl4mis_amd", "csrc", "libwslhip.so")
    total, bad = scan(lib)
    per = {}
    for k, _ in bad:
        per[k] = per.get(k, 0) + 1
    print(f"{lib}: {total} packed f32 multiply / fma instructions, {len(bad)} in a form unsafe next to f16 MFMAs, in {len(per)} kernels")
    for k, n in sorted(per.items(), key=lambda kv: -kv[1])[:40]:
        print(f"  {n:5d}  {k}")
    for k, t in bad[:10]:
        print("   e.g.", t)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Disassembly digest of the kernels of a library build (CPU, build container): per kernel whose mangled name contains PATTERN the
instruction mix (MFMA, packed / plain VALU, every ds_* and global_* / buffer_* opcode), scratch use and the register counts of the
kernel descriptor notes -- what a layout or scheduling change did to the code, before any GPU minute is spent.

    python tools/kernel_asm.py PATTERN [lib]          PATTERN e.g. conv_wino2r ; `--dump` as third argument prints the instructions
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from scan_vop3p import OBJDUMP, code_objects  # noqa: E402


def demangle(name):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        return name


def main():
    pat = sys.argv[1]
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else os.path.join(here, "..", "wsl4mis_amd", "csrc", "libwslhip.so")
    dump = "--dump" in sys.argv
    with tempfile.TemporaryDirectory() as work:
        for co in code_objects(lib, work):
            out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
            notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            kern, mix = None, None
            res = {}

            def flush():
                if kern and pat in kern:
                    d = demangle(kern)
                    lines = notes.splitlines()
                    info = ""
                    idx = [i for i, ln in enumerate(lines) if ln.strip() == ".name:           " + kern or ln.strip().endswith(" " + kern) and ".name:" in ln]
                    if idx:
                        i0 = idx[0]
                        lo = i0
                        while lo > 0 and not lines[lo].lstrip().startswith("- .agpr_count"):   # first key of a kernel's entry
                            lo -= 1
                        hi = i0
                        while hi + 1 < len(lines) and not lines[hi + 1].lstrip().startswith("- .agpr_count"):
                            hi += 1
                        seg = "\n".join(lines[lo:hi + 1])
                        for key in (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size"):
                            mm = re.findall(re.escape(key) + r":\s+(\d+)", seg)
                            if mm:
                                info += f" {key[1:]}={mm[0]}"
                    print(f"== {d}\n   {info.strip()}")
                    tot = sum(mix.values())
                    groups = collections.OrderedDict()
                    for op, n in sorted(mix.items(), key=lambda kv: -kv[1]):
                        if op.startswith(("v_mfma", "v_pk_", "ds_", "global_", "buffer_", "s_barrier", "s_waitcnt", "s_nop", "scratch_", "v_accvgpr", "v_mov")):
                            groups[op] = n
                    valu = sum(n for op, n in mix.items() if op.startswith("v_") and not op.startswith("v_mfma"))
                    mf = sum(n for op, n in mix.items() if op.startswith("v_mfma"))
                    print(f"   {tot} instructions: {mf} MFMA, {valu} other VALU, {sum(n for o, n in mix.items() if o.startswith('ds_'))} LDS, "
                          f"{sum(n for o, n in mix.items() if o.startswith(('global_', 'buffer_', 'scratch_')))} vector memory, "
                          f"{sum(n for o, n in mix.items() if o.startswith('s_'))} scalar")
                    print("   " + ", ".join(f"{op} {n}" for op, n in groups.items()))

            for line in out.splitlines():
                if line.endswith(">:"):
                    flush()
                    kern = line.split("<")[-1][:-2]
                    mix = collections.Counter()
                    if dump and pat in kern:
                        print(line)
                    continue
                if kern is None or not line.strip() or line.startswith("Disassembly") or ":" in line.split()[0] and len(line.split()) == 1:
                    continue
                tok = line.split()
                if tok:
                    mix[tok[0]] += 1
                    if dump and pat in kern:
                        print(line.split("//")[0].rstrip())
            flush()


if __name__ == "__main__":
    main()

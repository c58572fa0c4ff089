#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc passes (any of SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_LDS_BANK_CONFLICT,
SQ_LDS_IDX_ACTIVE, SQ_INSTS_VALU, SQ_INSTS_MFMA, SQ_ACTIVE_INST_VALU, GRBM_GUI_ACTIVE) per kernel family and prints ratios.
   python tools/pmc_mfma.py <dir> [<dir> ...] > profiles/<tag>_pmc_sq.md"""
import collections
import csv
import glob
import sys


def family(k):
    for f in ("conv_sp_kernel", "wgrad_sp_kernel", "conv_wino2r_kernel", "conv_wino2_kernel", "wgrad_wino_kernel", "conv_wino_kernel", "conv_mfma2l_kernel", "wgrad_mfma2s_kernel", "wgrad_mfma2l_kernel", "conv_cls_kernel", "conv_nk16_kernel", "wgrad_small_kernel",
              "conv_mfma2_kernel", "bnact_bwd_apply", "bnact_bwd_reduce", "gatedcrf_fwd", "feat_grad_combine", "bilinear_up2"):
        if f in k:
            return f
    return None


# counters are summed per PASS (directory): SQ_BUSY_CYCLES is collected in more than one pass, and a ratio must use the
# busy cycles of the pass its numerator comes from
acc = collections.defaultdict(lambda: collections.defaultdict(float))
per_pass = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            fam = family(r["Kernel_Name"])
            if fam:
                per_pass[d][fam][r["Counter_Name"]] += float(r["Counter_Value"])
for d, fams in per_pass.items():
    for fam, c in fams.items():
        for k, v in c.items():
            if k == "SQ_BUSY_CYCLES" and "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
                acc[fam]["SQ_BUSY_CYCLES@" + ("valu" if "SQ_ACTIVE_INST_VALU" in c else "other")] += v
            else:
                acc[fam][k] += v
print("| kernel family | MFMA pipe busy / SQ busy | vector-instruction active (SQ_ACTIVE_INST_VALU x 4 per SIMD) / SQ busy | LDS bank-conflict cycles / LDS active | VALU instr (incl. MFMA) per MFMA instr |")
print("|---|---|---|---|---|")
for fam, c in sorted(acc.items()):
    def ratio(a, b):
        return f"{c[a] / c[b]:.3f}" if c.get(a) is not None and c.get(b) else "–"
    # SQ_BUSY_CYCLES is reported per shader engine (32 on MI355X), SQ_VALU_MFMA_BUSY_CYCLES per SIMD (1024): the busy
    # fraction of one SIMD's matrix pipe while its shader engine is busy = MFMA / 1024 / (BUSY / 32)
    mf = f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['SQ_BUSY_CYCLES'] * 32):.3f}" if c.get("SQ_BUSY_CYCLES") and c.get("SQ_VALU_MFMA_BUSY_CYCLES") else "–"
    # SQ_ACTIVE_INST_VALU: cycles (in quad-cycle units) waves spend executing vector-ALU instructions, summed over SIMDs
    va = f"{c['SQ_ACTIVE_INST_VALU'] * 4 / (c['SQ_BUSY_CYCLES@valu'] * 32):.3f}" if c.get("SQ_BUSY_CYCLES@valu") and c.get("SQ_ACTIVE_INST_VALU") else "–"
    print(f"| `{fam}` | {mf} | {va} | {ratio('SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE')} | {ratio('SQ_INSTS_VALU', 'SQ_INSTS_MFMA')} |")


#!/usr/bin/env bash
# SQ counter passes of the f32 step on the final tree (the record script collects them for the split step): profiles/r4_pmc_sq_f32.md
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd /tmp; export TMPDIR=/tmp
S="python $R/bench.py --no-cpu-baseline --no-split-record --steps 2 --warmup 1 --serial-decoders --no-prof"
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d "$R/$O/pmc_sq1" -- $S > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$R/$O/pmc_sq2" -- $S > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d "$R/$O/pmc_sq3" -- $S > /dev/null 2>&1
cd "$R"; python tools/pmc_mfma.py "$O/pmc_sq1" "$O/pmc_sq2" "$O/pmc_sq3" > "$O/pmc_sq_f32.md" 2>/dev/null; rm -rf "$O/pmc_sq1" "$O/pmc_sq2" "$O/pmc_sq3"; cat "$O/pmc_sq_f32.md"

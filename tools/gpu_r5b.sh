#!/usr/bin/env bash
# round 5, bundle b: SQ counter passes of ONE layer shape (SHAPE="Ci,Co,S", default 64,64,64) through the three Winograd kernels
# (BN-source conv, plain-source conv, weight gradient), experiments build; LIBSEL=old for round 4's build
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd /tmp; export TMPDIR=/tmp
export SWEEP_SHAPES="${SHAPE:-64,64,64}" SWEEP_REPS=5
[ "${LIBSEL:-new}" = old ] && export WSL_EXP_LIB=old
S="python $R/tools/sweep_layers.py"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_FLAT"
P4="SQ_WAVE_CYCLES SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH SQ_WAVES"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $P --output-format csv -d "$R/$O/p$i" -- $S > "$R/$O/p$i.log" 2>&1
done
cd "$R"; python tools/pmc_sq_table.py "$O/p1" "$O/p2" "$O/p3" "$O/p4" > "$O/sq_table_${LIBSEL:-new}_${SHAPE:-64,64,64}.md"; rm -rf "$O/p1" "$O/p2" "$O/p3" "$O/p4"
cat "$O"/sq_table_*.md | grep -v "pack\|reduce" | head -80

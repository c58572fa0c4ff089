#!/usr/bin/env bash
# product build (in-place MFMA + fences), repeated runs of the configurations that failed
for a in "8 128 128 16 32 bn" "8 128 128 32 32 bn" "32 64 64 64 64 bn" "8 128 128 64 32 bn" "48 32 32 128 128 bn" "16 256 256 16 16 bn" "16 128 128 64 32"; do
  bad=0; for i in 1 2 3 4 5 6; do r=$(python tools/debug_sp_case.py $a 2>&1 | grep -E "max err" | sed 's/.*bad elements \([0-9]*\) of.*/\1/'); [ "$r" != "0" ] && bad=$((bad+1)); done; echo "product  [$a]: $bad of 6 runs with bad elements"
done

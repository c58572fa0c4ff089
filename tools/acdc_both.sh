#!/usr/bin/env bash
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$R"
bash tools/acdc_convergence.sh "$O/trial" 300
ok=1; for l in pce pce_tv pce_gatedcrf; do [ -s "$O/trial/curve_$l.json" ] || ok=0; done
if [ $ok = 1 ]; then echo "trial ok"; bash tools/acdc_convergence.sh "$O/full" 60000; else echo "TRIAL FAILED"; tail -20 "$O"/trial/train_*.log; fi

#!/usr/bin/env bash
# wgrad_sp_kernel with the lower-conflict image layout (-DWSL_SP_WG_LAYOUT2, tools/exp/libwslhip_exp_l2.so) against the compact images:
# the split step alternating on one box (value + last loss = same numerics), then SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of both
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
B="tools/bench_exp.py --conv-precision split_f16x3 --no-cpu-baseline --no-split-record --no-pmc-refresh"
for rep in 1 2 3; do for t in "" l2; do
  WSL_EXP_LIB=$t python $B --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('rep $rep lib [$t]', d['value'], d['repeats']['values'], 'loss', d['last_losses']['loss'], 'wgrad_sp', [v['avg_launch_us'] for n,v in k.items() if n.startswith('wgrad_sp')])"
done; done | tee "$O/ab.log"
cd /tmp; export TMPDIR=/tmp
for t in "" l2; do
  WSL_EXP_LIB=$t timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$R/$O/pmc_$t" -- python $R/$B --steps 2 --warmup 1 --serial-decoders --no-prof --repeats 1 > /dev/null 2>&1
done
cd "$R"
python - "$O" <<'PY'
import csv, glob, sys, collections
for t in ("", "l2"):
    acc = collections.defaultdict(lambda: [0.0, 0.0])
    for f in glob.glob(f"{sys.argv[1]}/pmc_{t}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "_sp_kernel" in n:
                k = n.split("<")[0].replace("void ", "")
                acc[k][0 if r["Counter_Name"] == "SQ_LDS_BANK_CONFLICT" else 1] += float(r["Counter_Value"])
    print(f"lib [{t}]", {k: round(v[0] / max(v[1], 1.0), 3) for k, v in acc.items()})
PY
rm -rf "$O"/pmc_*

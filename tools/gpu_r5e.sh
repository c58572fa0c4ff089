#!/usr/bin/env bash
# round 5, bundle e: tile walk order of wgrad_wino_kernel -- contiguous runs (rounds 2-4) against interleaved, XCD-grouped (round 5):
# per-layer sweep of both, L2 counters of the full-resolution layers with the new order, then the f32 step of both builds alternating.
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
for il in 0 1; do echo "== WSL_WGWINO_INTERLEAVE=$il"; WSL_WGWINO_INTERLEAVE=$il python tools/sweep_layers.py 2>&1 | grep "^|"; done > "$O/sweep_interleave.log"
(echo "== WSL_WGWINO_INTERLEAVE=1 WSL_WGRAD_WGS=512"; WSL_WGRAD_WGS=512 python tools/sweep_layers.py 2>&1 | grep "^|") >> "$O/sweep_interleave.log"
cd /tmp; export TMPDIR=/tmp
i=0
for P in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  SWEEP_SHAPES="16,16,256;32,16,256" SWEEP_REPS=5 timeout 200 rocprofv3 --pmc $P --output-format csv -d "$R/$O/q$i" -- python $R/tools/sweep_layers.py > "$R/$O/q$i.log" 2>&1
  SWEEP_SHAPES="32,32,128;64,64,64" SWEEP_REPS=5 timeout 200 rocprofv3 --pmc $P --output-format csv -d "$R/$O/r$i" -- python $R/tools/sweep_layers.py > "$R/$O/r$i.log" 2>&1
done
cd "$R"
python - "$O" <<'PY' > "$O/traffic_interleaved.md"
import collections, csv, glob, re, sys
o = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(o + "/[qr]*/**/*counter_collection.csv", recursive=True):
    tag = "level 0" if "/q" in f else "32ch@128 + 64ch@64"
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wgrad_wino" not in k: continue
        key = (tag, re.sub(r"\(.*", "", k)[:60], r["Grid_Size"])
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])].add(r["Dispatch_Id"])
cs = sorted({c for v in acc.values() for c in v})
print("| layers | kernel | grid | " + " | ".join(cs) + " |"); print("|---|---|---|" + "---|" * len(cs))
for key, v in sorted(acc.items()):
    print(f"| {key[0]} | {key[1]} | {key[2]} | " + " | ".join(f"{v[c] / max(len(n[(key, c)]), 1):.4g}" for c in cs) + " |")
PY
rm -rf "$O"/q[0-9] "$O"/r[0-9]
cat "$O/sweep_interleave.log" "$O/traffic_interleaved.md"
VARIANTS="product prev" PREC=f32 REPS=3 bash tools/gpu_step_ab.sh "$O" 2>&1 | tail -12

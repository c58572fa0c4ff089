#!/usr/bin/env bash
# round 5, bundle d: the full-resolution layers (16 -> 16 and 32 -> 16 @ 256 x 256, N = 64) under (1) the phase ablations of the
# raw-source conv and (2) per-layer HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) + L2 hit / miss of all three Winograd kernels.
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(for a in 0 1 4 2 6 7; do echo "== WSL_WINO2R_ABLATE=$a"; for c in "64 16 16 256 256" "64 32 16 256 256"; do WSL_WINO2R_ABLATE=$a MB_WINO=1 MB_RAW=1 python tools/microbench_conv.py $c 2>&1 | grep "us"; done; done) > "$O/abl_level0.log" 2>&1
cd /tmp; export TMPDIR=/tmp
export SWEEP_SHAPES="${SHAPES:-16,16,256;32,16,256}" SWEEP_REPS=5
S="python $R/tools/sweep_layers.py"
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $P --output-format csv -d "$R/$O/q$i" -- $S > "$R/$O/q$i.log" 2>&1
done
cd "$R"
python - "$O" <<'PY' > "$O/traffic_level0.md"
import collections, csv, glob, re, sys
o = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(o + "/q*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wino" not in k or "pack" in k: continue
        key = (re.sub(r"\(.*", "", k)[:60], r["Grid_Size"])
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])].add(r["Dispatch_Id"])
cs = sorted({c for v in acc.values() for c in v})
print("| kernel | grid | " + " | ".join(cs) + " |"); print("|---|---|" + "---|" * len(cs))
for key, v in sorted(acc.items()):
    print(f"| {key[0]} | {key[1]} | " + " | ".join(f"{v[c] / max(len(n[(key, c)]), 1):.4g}" for c in cs) + " |")
PY
rm -rf "$O"/q[0-9]
cat "$O/abl_level0.log" "$O/traffic_level0.md"; tail -5 "$O/q1.log"

#!/usr/bin/env bash
# HBM traffic of EVERY kernel of the f32 step (FETCH_SIZE / WRITE_SIZE, separate passes, decoders serialised), per kernel name
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-split-record --steps 2 --warmup 1 --serial-decoders --no-prof"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --output-format csv -d "$R/$O/pmc_$d" -- $B > /dev/null 2>&1
done
cd "$R"
python - "$O" <<'PY' > "$O/traffic_all_kernels.md"
import collections, csv, glob, re, sys
o = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(o + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void ", "", r["Kernel_Name"]); k = re.sub(r"\(.*", "", k); k = k.replace("wsl::", "")[:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
print("| kernel | launches (3 steps) | fetch MB / launch (FETCH_SIZE x 2) | write MB / launch | total GB over 3 steps | L2 hit share |"); print("|---|---|---|---|---|---|")
rows = []
for k, v in acc.items():
    nl = max(len(n[(k, "FETCH_SIZE")]), 1)
    f, w = 2 * 1024 * v["FETCH_SIZE"], 1024 * v["WRITE_SIZE"]
    hit = v["TCC_HIT_sum"] / max(v["TCC_HIT_sum"] + v["TCC_MISS_sum"], 1)
    rows.append((f + w, k, nl, f / nl / 1e6, w / nl / 1e6, hit))
for t, k, nl, f, w, hit in sorted(rows, reverse=True)[:45]:
    print(f"| {k} | {nl} | {f:.1f} | {w:.1f} | {t / 1e9:.2f} | {hit:.2f} |")
PY
rm -rf "$O"/pmc_*
cat "$O/traffic_all_kernels.md"

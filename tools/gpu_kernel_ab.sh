#!/usr/bin/env bash
# per-KERNEL A / B of two library builds on one box: rocprofv3 kernel summaries of the serialised step with each build, the rows whose name
# matches PATTERN side by side (average microseconds per launch)
#   VARIANTS="product r6base" PATTERN="conv_cls|conv_nk16|wgrad_small" bash tools/gpu_kernel_ab.sh <out dir>
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd /tmp; export TMPDIR=/tmp
for v in ${VARIANTS:-product r6base}; do
  lib=""; [ "$v" != product ] && lib="--lib $R/tools/exp/libwslhip_$v.so"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_$v" -- python $R/bench.py --no-cpu-baseline --no-split-record --no-pmc-refresh --no-prof --repeats 1 --steps 8 --warmup 3 --serial-decoders $lib > "$R/$O/bench_$v.log" 2>/dev/null
  rm -f "$R/$O"/prof_$v/*/*kernel_trace.csv
done
cd "$R"
python - "$O" "${PATTERN:-conv_cls|conv_nk16|wgrad_small}" ${VARIANTS:-product r6base} <<'PY'
import csv, glob, re, sys
o, pat, vs = sys.argv[1], re.compile(sys.argv[2]), sys.argv[3:]
tab = {}
for v in vs:
    for f in glob.glob(f"{o}/prof_{v}/*/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if pat.search(r["Name"]):
                tab.setdefault(r["Name"].replace("void ", "").replace("wsl::", "")[:60], {})[v] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
print("| kernel | " + " | ".join(f"{v} us (launches)" for v in vs) + " |")
print("|---|" + "---|" * len(vs))
for k, d in sorted(tab.items()):
    print(f"| `{k}` | " + " | ".join(f"{d[v][0]:.1f} ({d[v][1]})" if v in d else "-" for v in vs) + " |")
PY

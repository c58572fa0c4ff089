#!/usr/bin/env bash
# in-step A / B on ONE box: the step (PREC=f32 | split_f16x3) with the product library against other builds (VARIANTS="product r4": tools/exp/libwslhip_<tag>.so),
# alternating, REPS rounds -- how a kernel change is priced when boxes differ by 1-3 %
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
for rep in $(seq 1 ${REPS:-3}); do
  for v in ${VARIANTS:-product prev}; do
    lib=""; [ "$v" != product ] && lib="--lib tools/exp/libwslhip_$v.so"
    python bench.py --conv-precision ${PREC:-split_f16x3} --steps 40 --warmup 10 --no-split-record --no-cpu-baseline --no-pmc-refresh $lib 2>/dev/null | tail -1 > "$O/bench_${v}_$rep.json"
    python - "$O/bench_${v}_$rep.json" $v $rep <<'PY' | tee -a "$O/ab.log"
import json, sys
d = json.loads(open(sys.argv[1]).read()); k = d["roofline"]["kernels"]
g = lambda n: next((v for kk, v in k.items() if kk.startswith(n)), {})
print(sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], "conv_wino2", g("conv_wino2").get("avg_launch_us"), "conv_sp", g("conv_sp").get("avg_launch_us"), "wgrad_sp", g("wgrad_sp").get("avg_launch_us"),
      "wgrad_wino", g("wgrad_wino").get("avg_launch_us"), "conv_mfma2l", g("conv_mfma2l").get("avg_launch_us"))
PY
  done
done

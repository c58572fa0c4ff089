#!/usr/bin/env bash
# Builds libwslhip.so (the product: hand-written HIP for gfx950) in-tree next to the sources.
#   ./build.sh          product library only (no environment knobs, no ablation / probe / experiment code)
#   ./build.sh emul     ALSO the test-only host emulation build (tests/emul/libwslhip_emul.so; -DWSL_EXPERIMENTS so the
#                       experiment variants stay logic-checked on the CPU)
#   ./build.sh exp      ALSO tools/exp/libwslhip_exp.so: the same sources with -DWSL_EXPERIMENTS (env tuning knobs, ablation
#                       switches, measured-slower kernel families, machine probes) for the tuning tools -- never loaded by
#                       wsl4mis_amd/, bench.py or the tests
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
root="$(cd "$here/../.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
srcs=(wsl_api wsl_conv wsl_conv2 wsl_conv4 wsl_conv5 wsl_convsp wsl_bn wsl_convt wsl_loss wsl_optim wsl_net wsl_data)
mkdir -p "$here/build"
objs=()
pids=()
for s in "${srcs[@]}"; do
  [ -f "$here/$s.hip" ] || continue
  o="$here/build/$s.o"
  objs+=("$o")
  if [ ! -f "$o" ] || [ "$here/$s.hip" -nt "$o" ] || [ "$here/wsl_rt.h" -nt "$o" ] || [ "$here/wsl_debug.h" -nt "$o" ] || [ "$root/include/wsl_hip.h" -nt "$o" ]; then
    extra=""
    # the Winograd kernels are bound by their vector-instruction count: the SLP vectoriser packs the output transforms into
    # v_pk_add_f32 and then pays more v_mov_b32 to un-interleave the results than it saved (-8 % vector instructions without)
    [ "$s" = "wsl_conv5" ] && [ "${WSL_NO_SLP:-1}" = "1" ] && extra="-fno-slp-vectorize"
    # (wsl_convsp: measured neutral there -- it removes 24 v_mov_b32 per staged task and costs the 16-wide data-gradient kernel a spill)
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra -c "$here/$s.hip" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$here/libwslhip.so"
echo "built $here/libwslhip.so"

if [ "${1:-}" = "exp" ] || [ "${2:-}" = "exp" ]; then
  xd="$root/tools/exp"
  mkdir -p "$xd/build"
  xobjs=()
  pids=()
  for s in "${srcs[@]}"; do
    [ -f "$here/$s.hip" ] || continue
    o="$xd/build/$s.o"
    xobjs+=("$o")
    if [ ! -f "$o" ] || [ "$here/$s.hip" -nt "$o" ] || [ "$here/wsl_rt.h" -nt "$o" ] || [ "$here/wsl_debug.h" -nt "$o" ] || [ "$root/include/wsl_hip.h" -nt "$o" ]; then
      extra=""
      [ "$s" = "wsl_conv5" ] && [ "${WSL_NO_SLP:-1}" = "1" ] && extra="-fno-slp-vectorize"
      "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DWSL_EXPERIMENTS $extra -c "$here/$s.hip" -o "$o" &
      pids+=($!)
    fi
  done
  for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC "${xobjs[@]}" -o "$xd/libwslhip_exp.so"
  echo "built $xd/libwslhip_exp.so"
fi

if [ "${1:-}" = "emul" ] || [ "${2:-}" = "emul" ]; then
  CXX="${EMUL_CXX:-/opt/rocm/lib/llvm/bin/clang++}"
  em="$root/tests/emul"
  mkdir -p "$em/build"
  eobjs=()
  pids=()
  for s in "${srcs[@]}"; do
    [ -f "$here/$s.hip" ] || continue
    o="$em/build/$s.o"
    eobjs+=("$o")
    if [ ! -f "$o" ] || [ "$here/$s.hip" -nt "$o" ] || [ "$here/wsl_rt.h" -nt "$o" ] || [ "$em/hip_emul.h" -nt "$o" ] || [ "$root/include/wsl_hip.h" -nt "$o" ]; then
      "$CXX" -x c++ -std=c++17 -O2 -g -fPIC -ffp-contract=off -DWSL_HOST_EMUL -DWSL_EXPERIMENTS -I"$em" -Wall -Wno-unused-function \
        -Wno-unknown-pragmas -Wno-pass-failed -c "$here/$s.hip" -o "$o" &
      pids+=($!)
    fi
  done
  "$CXX" -std=c++17 -O2 -g -fPIC -c "$em/hip_emul.cpp" -o "$em/build/hip_emul.o" &
  pids+=($!)
  for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
  "$CXX" -shared -fPIC -pthread "${eobjs[@]}" "$em/build/hip_emul.o" -o "$em/libwslhip_emul.so"
  echo "built $em/libwslhip_emul.so"
fi

#!/usr/bin/env bash
# Builds libwslhip.so (the product: hand-written HIP for gfx950) in-tree next to the sources.
#   ./build.sh          product library only (no environment knobs, no routing hooks, no ablation / probe / experiment code)
#   ./build.sh emul     ALSO the test-only host emulation build (tests/emul/libwslhip_emul.so; -DWSL_EXPERIMENTS so the
#                       experiment variants stay logic-checked on the CPU)
#   ./build.sh exp      ALSO tools/exp/libwslhip_exp.so: the same sources with -DWSL_EXPERIMENTS (env tuning knobs, routing overrides,
#                       ablation switches, measured-slower kernel families, machine probes) for the tuning tools and for the GPU tests
#                       that force a route (tests/backends.py::HipExpBackend) -- never loaded by wsl4mis_amd/ or bench.py
#
# What ties a binary to the tree:
#   * SRC_SHA = SHA-256 over the bytes of csrc/*.hip, csrc/*.h (sorted by name, C locale) and include/wsl_hip.h, in that order --
#     wsl4mis_amd/_lib.py::source_sha256() recomputes exactly this.  It is compiled into wsl_api.o of every variant
#     (-DWSL_SRC_SHA256=...), wsl_build_info() returns it, bench.py refuses to run a library whose hash differs from the tree's
#     and tests/test_abi.py asserts it.
#   * an object is rebuilt when the HASH of (its source, the shared headers, its flags) differs from the one recorded next to it
#     (build/<name>.o.key) -- not when a modification time says so: a checkout or a copy cannot leave a stale object behind.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
root="$(cd "$here/../.." && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
srcs=(wsl_api wsl_conv wsl_conv2 wsl_conv4 wsl_conv5 wsl_convsp wsl_bn wsl_convt wsl_loss wsl_optim wsl_net wsl_data)
export LC_ALL=C
SRC_SHA="$(cat $(ls "$here"/*.hip "$here"/*.h | sort) "$root/include/wsl_hip.h" | sha256sum | cut -d' ' -f1)"
HDR_SHA="$(cat "$here/wsl_rt.h" "$here/wsl_debug.h" "$root/include/wsl_hip.h" | sha256sum | cut -d' ' -f1)"

# build_variant <object dir> <output .so> <compiler> <link flags> <extra header for the key> <compile flags...>
build_variant() {
  local odir="$1" out="$2" cxx="$3" link="$4" xhdr="$5"; shift 5
  local flags=("$@") objs=() pids=() s o key extra xsha
  mkdir -p "$odir"
  for s in "${srcs[@]}"; do
    [ -f "$here/$s.hip" ] || continue
    o="$odir/$s.o"; objs+=("$o"); extra=()
    # the Winograd kernels are bound by their vector-instruction count: the SLP vectoriser packs the output transforms into
    # v_pk_add_f32 and then pays more v_mov_b32 to un-interleave the results than it saved (-8 % vector instructions without)
    # (wsl_convsp: measured neutral there -- it removes 24 v_mov_b32 per staged task and costs the 16-wide data-gradient kernel a spill)
    [ "$s" = "wsl_conv5" ] && [ "${WSL_NO_SLP:-1}" = "1" ] && [ "$cxx" = "$HIPCC" ] && extra=(-fno-slp-vectorize)
    [ "$s" = "wsl_api" ] && extra+=("-DWSL_SRC_SHA256=\"$SRC_SHA\"")
    xsha=""; if [ -n "$xhdr" ]; then xsha="$(sha256sum < "$xhdr" | cut -d' ' -f1)"; fi
    key="$(sha256sum < "$here/$s.hip" | cut -d' ' -f1) $HDR_SHA $xsha $cxx ${flags[*]} ${extra[*]:-}"
    if [ ! -f "$o" ] || [ ! -f "$o.key" ] || [ "$(cat "$o.key")" != "$key" ]; then
      rm -f "$o.key"
      ( "$cxx" "${flags[@]}" "${extra[@]}" -c "$here/$s.hip" -o "$o" && printf '%s' "$key" > "$o.key" ) &
      pids+=($!)
    fi
  done
  local rc=0 p
  for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || rc=1; }; done
  [ $rc = 0 ] || { echo "build of $out failed" >&2; exit 1; }
  BUILT_OBJS=("${objs[@]}")
}

build_variant "$here/build" "$here/libwslhip.so" "$HIPCC" "" "" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${BUILT_OBJS[@]}" -o "$here/libwslhip.so"
echo "built $here/libwslhip.so (sources sha256:$SRC_SHA)"

if [ "${1:-}" = "exp" ] || [ "${2:-}" = "exp" ]; then
  xd="$root/tools/exp"
  build_variant "$xd/build" "$xd/libwslhip_exp.so" "$HIPCC" "" "" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DWSL_EXPERIMENTS
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC "${BUILT_OBJS[@]}" -o "$xd/libwslhip_exp.so"
  echo "built $xd/libwslhip_exp.so"
fi

# ./build.sh expvar   with WSL_EXP_TAG=<tag> WSL_EXP_FLAGS="-DWSL_X=1 ...": a tagged experiments build with extra compile flags
#                     (tools/exp/libwslhip_exp_<tag>.so; the tuning tools pick it with WSL_EXP_LIB=<tag>) -- same-box A/B of source variants
if [ "${1:-}" = "expvar" ]; then
  xd="$root/tools/exp"; tag="${WSL_EXP_TAG:?WSL_EXP_TAG}"
  # shellcheck disable=SC2086
  build_variant "$xd/build_$tag" "$xd/libwslhip_exp_$tag.so" "$HIPCC" "" "" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DWSL_EXPERIMENTS ${WSL_EXP_FLAGS:-}
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC "${BUILT_OBJS[@]}" -o "$xd/libwslhip_exp_$tag.so"
  echo "built $xd/libwslhip_exp_$tag.so"
fi

if [ "${1:-}" = "emul" ] || [ "${2:-}" = "emul" ]; then
  CXX="${EMUL_CXX:-/opt/rocm/lib/llvm/bin/clang++}"
  em="$root/tests/emul"
  mkdir -p "$em/build"
  "$CXX" -std=c++17 -O2 -g -fPIC -c "$em/hip_emul.cpp" -o "$em/build/hip_emul.o" &
  epid=$!
  build_variant "$em/build" "$em/libwslhip_emul.so" "$CXX" "" "$em/hip_emul.h" -x c++ -std=c++17 -O2 -g -fPIC -ffp-contract=off -DWSL_HOST_EMUL \
    -DWSL_EXPERIMENTS -I"$em" -Wall -Wno-unused-function -Wno-unknown-pragmas -Wno-pass-failed
  wait $epid
  "$CXX" -shared -fPIC -pthread "${BUILT_OBJS[@]}" "$em/build/hip_emul.o" -o "$em/libwslhip_emul.so"
  echo "built $em/libwslhip_emul.so"
fi

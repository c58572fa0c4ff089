#!/usr/bin/env bash
# A / B builds of the split-precision kernels for the open question of profiles/r3_sp_hunt.md (cause 2): which property of the
# compiler's f16 MFMA chains is the wrong one, and which ingredients of the inline-assembly form are needed.  Builds
# tools/exp/libwslhip_sp_<name>.so = the product's objects with wsl_convsp.hip recompiled under the given switches (wsl_rt.h);
# measure each with   WSL_LIB=tools/exp/libwslhip_sp_<name>.so python tools/ab_split_fullsize.py 2   on the GPU box
# (right: 2.23e-3 on every repetition; wrong: 4e-3 .. 1e-2 and different from run to run).
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$root/wsl4mis_amd/csrc"
"$src/build.sh" > /dev/null
mkdir -p "$root/tools/exp/build"
objs=$(ls "$src"/build/*.o | grep -v wsl_convsp.o)
build() {   # name, flags...
  local name="$1"; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c "$src/wsl_convsp.hip" -o "$root/tools/exp/build/convsp_$name.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs "$root/tools/exp/build/convsp_$name.o" -o "$root/tools/exp/libwslhip_sp_$name.so"
  echo "built tools/exp/libwslhip_sp_$name.so"
}
build compiler_chains   -DWSL_SP_AB_FORM=1
build renamed_no_overlap -DWSL_SP_AB_FORM=2
build no_fence          -DWSL_SP_AB_NO_FENCE
build no_release        -DWSL_SP_AB_NO_RELEASE
build short_drain       -DWSL_SP_AB_SHORT_DRAIN
build bare_inplace      -DWSL_SP_AB_NO_FENCE -DWSL_SP_AB_NO_RELEASE -DWSL_SP_AB_SHORT_DRAIN
# speed (tools/sweep_layers_sp.py with WSL_LIB=...), not a correctness question: weight-gradient images on the pitch / plane offset of
# profiles/r3_wgrad_sp_lds_conflicts.md
build wg_layout2         -DWSL_SP_WG_LAYOUT2

#!/usr/bin/env bash
# A / B builds for the spill question: the kernels that spill under a 3-waves-per-SIMD register cap, rebuilt with
# a cap of 2 (no spills, one resident workgroup per CU fewer).  Measure with tools/microbench_nk16.py / tools/sweep_layers_sp.py and
# WSL_LIB=tools/exp/libwslhip_minw2.so
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$root/wsl4mis_amd/csrc"
"$src/build.sh" > /dev/null
mkdir -p "$root/tools/exp/build"
objs=$(ls "$src"/build/*.o | grep -v "wsl_convsp.o\|wsl_conv4.o")
for s in wsl_convsp wsl_conv4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DWSL_NK16_MINW=2 -DWSL_SP_MINW16=2 -c "$src/$s.hip" -o "$root/tools/exp/build/${s}_minw2.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs "$root/tools/exp/build/wsl_convsp_minw2.o" "$root/tools/exp/build/wsl_conv4_minw2.o" -o "$root/tools/exp/libwslhip_minw2.so"
echo "built tools/exp/libwslhip_minw2.so"

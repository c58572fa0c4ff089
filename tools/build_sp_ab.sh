#!/usr/bin/env bash
# A / B build of the split-precision conv kernels: the product's objects with wsl_convsp.hip recompiled under extra -D switches.
#   tools/build_sp_ab.sh <name> [-DWSL_SP_DB=0 ...]   ->  tools/exp/libwslhip_<name>.so   (select with WSL_LIB= in the tuning tools)
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$root/wsl4mis_amd/csrc"
name="$1"; shift
[ -f "$src/libwslhip.so" ] || "$src/build.sh" > /dev/null
mkdir -p "$root/tools/exp/build"
objs=$(ls "$src"/build/*.o | grep -v "wsl_convsp.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c "$src/wsl_convsp.hip" -o "$root/tools/exp/build/wsl_convsp_$name.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs "$root/tools/exp/build/wsl_convsp_$name.o" -o "$root/tools/exp/libwslhip_$name.so"
echo "built tools/exp/libwslhip_$name.so"

#!/usr/bin/env bash
# which ingredient of the proven split kernels the full network needs: D = compiler MFMA chains + full waits around the commits,
# E = inline-assembly in-place MFMA chains + fences, counted waits around the commits
mkdir -p gpurun_out/r3v
for lib in tools/exp/libwslhip_D.so tools/exp/libwslhip_E.so; do WSL_LIB=$lib timeout 30 python tools/ab_split_fullsize.py 2 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r3v/ab2.log

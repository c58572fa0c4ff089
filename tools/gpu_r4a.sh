#!/usr/bin/env bash
# Next round's first GPU call (~25 s of box time): which property of the compiler's f16 MFMA chains is wrong, which ingredients of the
# inline-assembly form are needed (profiles/r3_sp_hunt.md, cause 2).  Build the variants first, in the build container:
#     tools/build_sp_variants.sh
# then   gpurun --timeout 200 -- 'bash tools/gpu_r4a.sh'
mkdir -p gpurun_out/r4a
for v in "" compiler_chains renamed_no_overlap no_fence no_release short_drain bare_inplace; do
  lib=""; [ -n "$v" ] && lib=tools/exp/libwslhip_sp_$v.so
  WSL_LIB=$lib timeout 40 python tools/ab_split_fullsize.py 2 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r4a/ab.log
# speed A/B of the weight-gradient image layout (profiles/r3_wgrad_sp_lds_conflicts.md): sums of the per-layer sweep, product first
for lib in "" tools/exp/libwslhip_sp_wg_layout2.so; do echo "== ${lib:-product}"; WSL_LIB=$lib timeout 200 python tools/sweep_layers_sp.py --dec --only-sp 2>&1 | grep -v amdgpu.ids | tail -15; done | tee gpurun_out/r4a/sweep_wg_layout.log

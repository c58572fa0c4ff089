# after the WSL_DETACH32 fix (profiles/r4_sp_root_cause.md): the f32 victims beside synthetic f16-MFMA aggressors, the compiler-chain
# build of the split kernels at network level, and the instruction-form table
mkdir -p gpurun_out/r4d
( timeout 250 tools/exp/probe_pk_opsel 4000
  for v in c1:64,32,32,128,64 c3:64,32,32,128,64 wino:64,32,32,128,128; do timeout 200 python tools/pair_race2.py $v 30 inplace,sparse,renamed_nops,inplace_lds 512; done
  for lib in tools/exp/libwslhip_sp_compiler_chains.so tools/exp/libwslhip_sp_bare_inplace.so ""; do
    WSL_LIB=$lib timeout 100 python tools/pair_race.py c1:64,32,32,128,64 sp:64,32,32,256,128 50
    WSL_LIB=$lib timeout 100 python tools/pair_race.py wino:64,32,32,128,128 sp:64,32,32,256,128 50
    WSL_LIB=$lib timeout 100 python tools/ab_split_fullsize.py 3
    WSL_LIB=$lib timeout 150 python tools/diff_runs_split.py 4
  done ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4d/verify.log

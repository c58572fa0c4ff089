/* PRIVATE header of libwslhip.so: read-only queries for the test-suite and the tuning tools.  Not part of the drop-in boundary
 * (include/wsl_hip.h); nothing in wsl4mis_amd/ (the product's host side) calls these.
 *
 * In every build: three QUERIES (no state: they read a workspace / a plan and return).  The product library has no routing switch,
 * no test hook that changes which kernel a launch takes and no ablation template arm: those exist only with -DWSL_EXPERIMENTS
 * (second half of this header). */
#pragma once
#ifdef __cplusplus
extern "C" {
#endif

/* The discrete decisions of the training forward held in a network workspace (wsl_net_forward(training = 1) ran on `ws`), as the
 * kernels take them -- the full-size gradient-parity test replays them in the oracle, so that two fp32 implementations are compared
 * on the SAME piecewise-linear function instead of on whichever side of a LeakyReLU kink round-off put them.
 *   which 0: index = BatchNorm layer in state_dict order (0 .. 25 for unet_cct): out uint8 [N, C, H, W], 1 where the LeakyReLU
 *            argument  fma(y, scale, shift)  is > 0 (the loader's expression, bit for bit), else 0;
 *   which 1: index = encoder level l = 1 .. 4: out uint8 [N, C, H/2, W/2] of level l - 1's feature, the position 0 .. 3 (row-major
 *            in the 2 x 2 window) the max-pool takes (first maximum, strict >: the forward kernel's scan). */
struct WslNetDesc;
int wsl_debug_net_decisions(const struct WslNetDesc* d, const void* ws, size_t ws_bytes, int which, int index, unsigned char* out,
                            void* stream);

/* Named regions of a network workspace in allocation order (raw conv outputs, decoder tensors, scratch sets, weight images):
 * index 0, 1, ... until the return value is 1.  Offsets / sizes in floats.  For tools that compare the workspaces of two runs
 * (tools/diff_runs_split.py). */
int wsl_debug_net_ws_region(const struct WslNetDesc* d, int index, char* name, size_t name_len, size_t* off_floats, size_t* n_floats);

/* What a split-precision conv launch of this layer shape would use: output tile, block width, LDS bytes per workgroup and the number of
 * workgroups that stay resident per CU (LDS in units of 1280 bytes, register cap).  Returns 1 when the shape has no split kernel.  For the
 * test that pins the shapes sitting exactly at an LDS allocation-unit edge (tests/test_ops_convsp.py). */
int wsl_debug_sp_conv_residency(int N, int H, int W, int Ci, int Co, int want_bn_epilogue, int* tile_h, int* tile_w, int* co_t,
                                size_t* lds_bytes, int* per_cu);

/* Only in the EXPERIMENTS build (-DWSL_EXPERIMENTS: `build.sh exp` -> tools/exp/libwslhip_exp.so, loaded by the tuning tools and by the
 * GPU tests that force a route (tests/backends.py::HipExpBackend), and the test-only host emulator): routing overrides (process-wide
 * atomics: set them from the thread that launches, before the ws_bytes query of the same call), ablation switches (env WSL_CONV_ABLATE /
 * WSL_WGRAD_ABLATE and the ABL template arms: they skip work, results are WRONG by design), ~20 env tuning knobs (WSL_TUNE in
 * wsl_rt.h) and four machine probes. */
/* Force the conv tile shape (rows, columns, output-channel block) wherever it divides the layer; th <= 0 restores the
 * built-in per-layer table. */
int wsl_debug_conv_plan(int th, int tw, int co_t);
/* Winograd F(2x2,3x3) path: 0 off (wsl_conv2d_wino_ok() returns 0), 1 only layers with Co % 32 == 0, 2 (default) also
 * layers with Co % 16 == 0; -1 restores the default. */
int wsl_debug_conv_wino(int on);
/* Number of persistent workgroups a weight-gradient launch (f32 and split-precision plans) and the classifier's forward kernel aim at
 * (defaults: 768 = 3 per CU, 2 per CU); n <= 0 restores them.  Tests force a few so that a small layer's workgroups walk several tiles
 * each (the interleaved, XCD-grouped tile order of wgrad_wino_kernel and its siblings).  Set it before wsl_conv2d_wgrad_ws_bytes() /
 * wsl_sp_conv2d_wgrad_ws_bytes(): the workspaces are sized for the plan. */
int wsl_debug_wgrad_workgroups(int n);

int wsl_debug_mfma4_probe(const float* a, const float* b, float* d, void* stream);          /* tools/probe_mfma4.py */
int wsl_debug_mfma_stream(int shape, int blocks, int iters, float* out, void* stream);      /* tools/mfma_ceiling.py */
int wsl_debug_pk_probe(const float* in, float* out, void* stream);                            /* tools/probe_pk.py */
int wsl_debug_lds_dma_probe(const float* g, float* out, void* stream);                      /* tools/probe_lds_dma.py */
#ifdef __cplusplus
}
#endif

/* Validation hooks and diagnostics of libmural_hip -- NOT part of the product library.
 *
 * `libmural_hip.so` exports none of these (nm -D libmural_hip.so | grep -c mural_debug = 0).  They live in csrc/debug_hooks.hip, which
 * is linked into the second flavour `libmural_hip_debug.so` only (same objects + that file; MURAL_HIP_FLAVOR=debug selects it in
 * mural_amd/_lib.py, tests/conftest.py does).  The debug flavour is also the only one that honours the development switches of
 * csrc/common.h (dev_env): the product library reads MURAL_HOST_THREADS and TMPDIR and nothing else from the environment.          */
#ifndef MURAL_HIP_DEBUG_H
#define MURAL_HIP_DEBUG_H
#include "mural_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* validation: (offset, bytes) pairs of the regions of the calling thread's latest workspace carve (mural_snv_forward_* /
 * mural_snv_forward_packed_reuse and their *_workspace_bytes queries); returns the number of pairs.  With
 * MURAL_DEBUG_WS_GUARD=<bytes> in the environment every region is followed by that many unused bytes, so that a test can poison a
 * workspace, run a call and check that nothing was written outside the regions. */
int mural_debug_last_ws_layout(size_t* out_pairs, int32_t max_pairs);

/* Validation hooks of the parity tests: the channel-last conv kernels of the composed training step on their own.  Tensors are
 * [B][L][32]; acc blocks are double[MURAL_BN_SLOTS][2][32] (the forward reads the batch sums of act(x) from `acc` and finalises the
 * BatchNorm itself; acc_out / stat_out zeroed by the caller); part: 1024 * (32*32*3 + 32) floats of partial rows, *nrow of them written. */
int mural_debug_cl_conv32_fwd(const float* x, int64_t B, int32_t L, int32_t pre_relu, const double* acc, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, float* state, const float* W,
                              const float* bias, int32_t post_relu, const float* res1, const float* res2, double* acc_out,
                              int32_t out_relu, float* y, void* stream);
int mural_debug_cl_conv32_bwd(const float* dy, const float* x, const float* W, int64_t B, int32_t L, const float* state,
                              int32_t pre_relu, float* dz, double* stat_out, float* part, int32_t* nrow, void* stream);
int mural_debug_cl_bn_stats(const float* x, int64_t rows, int32_t relu, double* acc, void* stream);
/* the same two layers on the wave-private kernels (csrc/conv32_wave.hip); wfrag_scratch: the 6144 floats mural_debug_cw_wfrag wrote
 * for W (forward | input-gradient filter fragments, the per-step relayout of the composed step) or NULL: the conv gathers its
 * fragments from a copy of W in LDS */
int mural_debug_cw_wfrag(const float* W, float* out6144, void* stream);
/* diagnostic: per-workgroup wall-clock stamps (100 MHz) of the wave-private forward kernel's phases, uint64 [workgroups][4]; NULL: off */
int mural_debug_cw_set_stamps(void* dev_ptr);
/* the same for the training-mode first-layer kernels (csrc/snv_stage1.hip: first_train_kernel), uint64 [workgroups][8]: entry, tables
 * ready, wave 0's window in LDS, its window indices built, its row done, every wave done, exit; NULL: off */
int mural_debug_first_set_stamps(void* dev_ptr);
/* the same for the three backward launches of the fused local branch (csrc/snv_local_train.h), uint64 [3][256][8]; NULL: off */
int mural_debug_lt_set_stamps(void* dev_ptr);
int mural_debug_cw_conv32_fwd(const float* x, int64_t B, int32_t L, int32_t pre_relu, const double* acc, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, float* state, const float* W,
                              const float* bias, int32_t post_relu, const float* res1, const float* res2, double* acc_out,
                              int32_t out_relu, float* y, float* wfrag_scratch, void* stream);
int mural_debug_cw_conv32_bwd(const float* dy, const float* x, const float* W, int64_t B, int32_t L, const float* state,
                              const float* gamma, int32_t pre_relu, float* dz, double* stat_out, float* part, int32_t* nrow,
                              float* wfrag_scratch, void* stream);

/* Validation hook: the generic Conv1d of the INDEL path with every geometry knob (stride, nearest-neighbour upsampling `up`,
 * activation 0 none / 1 ReLU / 2 SiLU / 3 Softplus, two residuals), weights wt laid out [Cin][K][Cout]; engine 0 = vector-ALU
 * kernel, 1 = MFMA implicit-GEMM kernel, 2 = the router's choice, 3 = MFMA kernel on the polyphase form of an upsampled conv,
 * 4 = the barrier-free long-row MFMA kernel (3 / 5 / 7 taps, <= 32 channels, any stride), 5 = its polyphase form for upsampled
 * convs (24 / 32 input channels); engine | 0x100 first fills every CU's LDS
 * with NaN. */
int mural_debug_conv1d(const float* in, const float* wt, const float* bias, float* out, int64_t B, int32_t Cin, int32_t Lin,
                       int32_t Cout, int32_t Lout, int32_t K, int32_t stride, int32_t up, int32_t act, const float* res1,
                       const float* res2, int32_t engine, void* stream);

/* Validation hook: fills the whole LDS of every CU with NaN (a 160 KB-per-workgroup kernel on `stream`).  Called in front of a product
 * call whose result is then checked: a kernel that depends on LDS it has not written fails the comparison. */
int mural_debug_poison_lds(void* stream);

/* Validation hook: one fused ConvBlock launch of the INDEL forward (model_indel.py:6-19: x + W1 . SiLU(W5 * x + b5) + b1, BatchNorms
 * folded; w5 [C][5][2C], w1 [2C][C]) with its optional front (f_in [B][Cf][L / f_up], k = 7 conv Cf -> C with weights f_w [Cf][7][C]
 * on the input upsampled f_up times: the block input x is then produced inside the launch; f_pw, optional with f_up == 4: the same
 * weights in polyphase form [4 phases][Cf][3 source columns][C], taps that share a source column summed), skip tensor res2 and tail (max over the
 * workgroup's positions of Softplus(Wb . ReLU(Wa . out + ba) + bb), weights [Cin][Cout]; tail_max [B][tiles][C], tiles =
 * ceil(L / 256) without a front, ceil(L / 252) with one, ceil(L / 248) for the split form with f_pw and Cf = 16; entries the launch
 * does not write keep the caller's values).  form: 0 = the 8-channel block entirely on the vector ALU, 1 = its split
 * form (convs on the matrix cores, front / SiLU / adds on the vector ALU), -1 = the library's choice; form | 0x100 (form in the low byte, 0xff = the library's choice) first
 * fills every CU's LDS with NaN: a launch that depends on LDS it has not written then fails the comparison.
 * C = 32 with L <= 80, C = 40 with L = 16 and C = 48 with L = 8 are the one-launch blocks of the three deepest INDEL levels
 * (csrc/convblock_deep.hip; no tail); with C = 32, f_up = -s asks for a STRIDED k = 7 front instead (stride s, f_in [B][Cf][L * s]). */
int mural_debug_convblock(const float* x, const float* w5, const float* b5, const float* w1, const float* b1, const float* res2,
                          float* out, int64_t B, int32_t C, int32_t L, const float* f_in, const float* f_w, const float* f_b,
                          int32_t Cf, int32_t f_up, const float* f_pw, const float* ta_w, const float* ta_b, const float* tb_w,
                          const float* tb_b, float* tail_max, int32_t form, void* stream);

/* Diagnostic: the MFMA conv's workgroups of the following launches record 5 s_memrealtime values each (start, tile staged, MFMAs done,
 * stores issued, stores landed) into `stamps` (device memory, 5 x workgroups entries); NULL switches it off. */
int mural_debug_conv1d_set_stamps(unsigned long long* stamps);
/* The same for the level-0 MFMA ConvBlock kernel: 8 accumulators per workgroup (phase time sums over its tiles, word 7 = tiles). */
int mural_debug_cb8_set_stamps(unsigned long long* stamps);

/* Diagnostic only: while a device buffer of 2048*32 uint64 is set, the packed-path tower kernel adds wave 0's
 * cycles per phase into it (see tools/phase_stamps.py); pass NULL to switch off.                          */
int mural_debug_set_stamps(void* dev_ptr);

/* the development switches the debug flavour honours (csrc/common.h: dev_env), one "NAME\tdescription" line each, into buf (NUL-
 * terminated, truncated to cap bytes); returns the number of switches                                                            */
int mural_debug_list_switches(char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif

// Validation hooks and diagnostics (include/mural_hip_debug.h).  This file is linked into libmural_hip_debug.so ONLY: the product
// library exports no mural_debug_* symbol and reads no development switch.  Every hook is a thin wrapper around an internal entry of
// the library (declared here or in the internal headers) or a setter of a diagnostic pointer; loading this object also switches the
// development switches of common.h on (dev_env).
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mural_hip_debug.h"
#include "common.h"
#include "conv1d.h"
#include "conv32_cl.h"
#include "snv.h"

namespace mural {
// conv32_cl.hip / conv32_wave.hip
int cl_conv32_fwd(const float* x, int64_t B, int L, int pre_relu, const double* acc, const float* gamma, const float* beta, float eps,
                  float momentum, float* running_mean, float* running_var, float* state, const float* W, const float* bias, int post_relu,
                  const float* res1, const float* res2, double* acc_out, int out_relu, float* y, hipStream_t stream);
int cl_conv32_bwd(const float* dy, const float* x, const float* W, int64_t B, int L, const float* state, int pre_relu, float* dz,
                  double* stat_out, float* part, int* nrow, hipStream_t stream);
int cl_bn_stats(const float* x, int64_t rows, int relu, double* acc, hipStream_t stream);
int cw_conv32_fwd(const float* x, int64_t B, int L, int pre_relu, const double* acc, const float* gamma, const float* beta, float eps,
                  float momentum, float* running_mean, float* running_var, float* state, const float* W, const float* wfrag, const float* bias,
                  int post_relu, const float* res1, const float* res2, double* acc_out, int out_relu, float* y, hipStream_t stream);
int cw_conv32_bwd(const float* dy, const float* x, const float* W, const float* wfrag, int64_t B, int L, const float* state, const float* gamma,
                  int pre_relu, float* dz, double* stat_out, float* part, int* nrow, hipStream_t stream);
int cw_wfrag_build(const float* const* W, int n, float* out, hipStream_t stream);
extern unsigned long long* g_cw_stamps;
// encode.hip
extern thread_local std::vector<size_t> g_ws_layout;
// snv_model.hip, train_ops.hip (snv_local_train.h), conv1d_mfma.hip, conv1d.hip
extern unsigned long long* g_tower_stamps;
namespace ltrain { extern unsigned long long* g_lt_stamps; }
void conv1d_mfma_set_stamps(unsigned long long* p);
extern unsigned long long* g_cb8_stamps;
extern int g_convblock8_form;
}  // namespace mural

using namespace mural;

#define STREAM ((hipStream_t)stream)

namespace {
struct EnableDevSwitches {
  EnableDevSwitches() { mural::g_dev_switches = true; }
} g_enable_dev_switches;
}  // namespace

extern "C" int mural_debug_list_switches(char* buf, size_t cap) {
  std::string text;
  int n = 0;
  for (const mural::DevSwitch* s = mural::dev_switch_table(); s->name; ++s, ++n) text += std::string(s->name) + "\t" + s->what + "\n";
  if (buf && cap) {
    std::strncpy(buf, text.c_str(), cap - 1);
    buf[cap - 1] = '\0';
  }
  return n;
}

// (validation only) every CU's whole LDS filled with NaN: what a kernel reads from LDS without having written it shows up in its results
__global__ __launch_bounds__(256) void lds_poison_kernel(float* sink) {
  extern __shared__ float lds_all[];
  for (int i = threadIdx.x; i < 160 * 256; i += 256) lds_all[i] = __builtin_nanf("");
  __syncthreads();
  if (sink && lds_all[(threadIdx.x * 97) % (160 * 256)] == 1.f) sink[0] = 1.f;
}
static int poison_lds(hipStream_t stream) {
  MURAL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL(lds_poison_kernel, dim3(2048), dim3(256), 160 * 1024, stream, (float*)nullptr);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

// diagnostic (tools/phase_stamps_conv1d.py): the MFMA conv's workgroups record 5 s_memrealtime values each (start, tile staged, MFMAs
// done, stores issued, stores landed) into `stamps` (device, 5 x workgroups of the next launches; NULL switches it off)
extern "C" int mural_debug_conv1d_set_stamps(unsigned long long* stamps) {
  mural::conv1d_mfma_set_stamps(stamps);
  return MURAL_OK;
}

// the same for the level-0 ConvBlock kernels (indel_level0.hip, conv1d.hip): 8 accumulators per workgroup -- the time between the phase
// boundaries of a tile summed over its tiles (100 MHz units), word 7 = tiles walked
extern "C" int mural_debug_cb8_set_stamps(unsigned long long* stamps) {
  mural::g_cb8_stamps = stamps;
  return MURAL_OK;
}

// validation hook: fill every CU's LDS with NaN on `stream` (tests call it in front of a product call whose result they check: a
// kernel whose result depends on LDS it has not written then fails its parity comparison instead of passing by the luck of the leftovers)
extern "C" int mural_debug_poison_lds(void* stream) { return poison_lds(STREAM); }

// validation hook (tests/test_gpu_indel.py): the generic conv with every geometry knob of Conv1dArgs, on the vector-ALU kernel
// (engine 0), the MFMA implicit-GEMM kernel (engine 1), the router's choice (engine 2), the polyphase form (3) or the barrier-free
// long-row kernel (engine 4, conv1d_direct.hip)
extern "C" int mural_debug_conv1d(const float* in, const float* wt, const float* bias, float* out, int64_t B, int32_t Cin, int32_t Lin,
                                  int32_t Cout, int32_t Lout, int32_t K, int32_t stride, int32_t up, int32_t act, const float* res1,
                                  const float* res2, int32_t engine, void* stream) {
  Conv1dArgs a;
  std::memset(&a, 0, sizeof(a));
  a.in = in; a.wt = wt; a.bias = bias; a.out = out;
  a.B = (int)B; a.Cin = Cin; a.Lin = Lin; a.Cout = Cout; a.Lout = Lout;
  a.K = K; a.stride = stride; a.pad = (K - 1) / 2; a.up = up;
  a.act = act; a.res1 = res1; a.res2 = res2;
  if (engine & 0x100) {      // every CU's LDS filled with NaN first
    engine &= 0xff;
    if (int rc = poison_lds(STREAM)) return rc;
  }
  if (engine == 1) {
    MURAL_REQUIRE(conv1d_mfma_supported(a), "conv1d (MFMA): unsupported geometry");
    return launch_conv1d_mfma(a, STREAM);
  }
  if (engine == 4) {
    MURAL_REQUIRE(conv1d_direct_supported(a), "conv1d (direct MFMA): unsupported geometry");
    return launch_conv1d_direct(a, STREAM);
  }
  if (engine == 3 || engine == 5) {     // polyphase form of the upsampled conv: weights expanded on the host once per weight tensor (kept for repeats)
    MURAL_REQUIRE(up > 1 && stride == 1, "polyphase conv: needs up > 1, stride 1");
    static const float* last_wt = nullptr;
    static float* dw = nullptr;
    static int kj = 0, padj = 0;
    if (last_wt != wt) {
      std::vector<float> hw((size_t)Cin * K * Cout), pw;
      MURAL_HIP_CHECK(hipMemcpy(hw.data(), wt, hw.size() * 4, hipMemcpyDeviceToHost));
      conv1d_phase_weights(hw.data(), Cin, K, Cout, up, &pw, &kj, &padj);
      if (dw) (void)hipFree(dw);
      MURAL_HIP_CHECK(hipMalloc(&dw, pw.size() * 4));
      MURAL_HIP_CHECK(hipMemcpy(dw, pw.data(), pw.size() * 4, hipMemcpyHostToDevice));
      last_wt = wt;
    }
    a.wt = dw; a.K = kj; a.pad = padj; a.up = 1; a.phases = up;
    if (engine == 5) {
      MURAL_REQUIRE(conv1d_direct_poly_supported(a), "conv1d (direct MFMA, polyphase): unsupported geometry");
      return launch_conv1d_direct_poly(a, STREAM);
    }
    MURAL_REQUIRE(conv1d_mfma_supported(a), "conv1d (MFMA, polyphase): unsupported geometry");
    return launch_conv1d_mfma(a, STREAM);
  }
  return engine == 0 ? launch_conv1d_valu(a, STREAM) : launch_conv1d(a, STREAM);
}

extern "C" int mural_debug_convblock(const float* x, const float* w5, const float* b5, const float* w1, const float* b1,
                                     const float* res2, float* out, int64_t B, int32_t Cch, int32_t L, const float* f_in,
                                     const float* f_w, const float* f_b, int32_t Cf, int32_t f_up, const float* f_pw, const float* ta_w,
                                     const float* ta_b, const float* tb_w, const float* tb_b, float* tail_max, int32_t form, void* stream) {
  ConvBlockArgs a;
  std::memset(&a, 0, sizeof(a));
  a.x = x; a.w5 = w5; a.b5 = b5; a.w1 = w1; a.b1 = b1; a.res2 = res2; a.out = out;
  a.B = (int)B; a.C = Cch; a.L = L;
  if (f_in && f_up < 0) {      // a STRIDED k = 7 front (stride -f_up, source rows of L * stride columns): convblock_deep.hip
    a.f_in = f_in; a.f_w = f_w; a.f_b = f_b; a.Cf = Cf; a.f_up = 1; a.f_stride = -f_up; a.Lf = L * a.f_stride;
  } else if (f_in) {
    MURAL_REQUIRE(f_up >= 1 && L % f_up == 0, "convblock: the front's upsampling factor must divide the row length");
    a.f_in = f_in; a.f_w = f_w; a.f_b = f_b; a.Cf = Cf; a.f_up = f_up; a.Lf = L / f_up;
    a.f_pw = f_pw;      // optional (f_up == 4): the front's polyphase weights [4][Cf][3][C]
  }
  a.ta_w = ta_w; a.ta_b = ta_b; a.tb_w = tb_w; a.tb_b = tb_b; a.tail_max = tail_max;
  if (form >= 0 && (form & 0x100)) {      // poison LDS first; the low byte is the form (0xff: the library's choice)
    form = (form & 0xff) == 0xff ? -1 : (form & 0xff);
    if (int rc = poison_lds(STREAM)) return rc;
  }
  mural::g_convblock8_form = form;
  const int rc = launch_convblock(a, STREAM);
  mural::g_convblock8_form = -1;
  return rc;
}

// ---- validation hooks (tests/test_gpu_train.py, tools/gpu_debug_conv32_cl.py): the channel-last conv kernels on their own -----
extern "C" int mural_debug_cl_conv32_fwd(const float* x, int64_t B, int32_t L, int32_t pre_relu, const double* acc, const float* gamma,
                                         const float* beta, float* running_mean, float* running_var, float* state, const float* W,
                                         const float* bias, int32_t post_relu, const float* res1, const float* res2, double* acc_out,
                                         int32_t out_relu, float* y, void* stream) {
  return mural::cl_conv32_fwd(x, B, L, pre_relu, acc, gamma, beta, 1e-5f, 0.1f, running_mean, running_var, state, W, bias, post_relu, res1, res2,
                              acc_out, out_relu, y, (hipStream_t)stream);
}

extern "C" int mural_debug_cl_conv32_bwd(const float* dy, const float* x, const float* W, int64_t B, int32_t L, const float* state,
                                         int32_t pre_relu, float* dz, double* stat_out, float* part, int32_t* nrow, void* stream) {
  int n = 0;
  const int rc = mural::cl_conv32_bwd(dy, x, W, B, L, state, pre_relu, dz, stat_out, part, &n, (hipStream_t)stream);
  if (nrow) *nrow = n;
  return rc;
}

extern "C" int mural_debug_cl_bn_stats(const float* x, int64_t rows, int32_t relu, double* acc, void* stream) {
  return mural::cl_bn_stats(x, rows, relu, acc, (hipStream_t)stream);
}

// ---- validation hooks (tests/test_gpu_train.py, tools/gpu_debug_conv32_cl.py): the wave-private conv kernels on their own -----
extern "C" int mural_debug_cw_conv32_fwd(const float* x, int64_t B, int32_t L, int32_t pre_relu, const double* acc, const float* gamma,
                                         const float* beta, float* running_mean, float* running_var, float* state, const float* W,
                                         const float* bias, int32_t post_relu, const float* res1, const float* res2, double* acc_out,
                                         int32_t out_relu, float* y, float* wfrag_scratch, void* stream) {
  // wfrag_scratch != NULL: the 6144 floats mural_debug_cw_wfrag wrote for W (the path of the composed step), NULL: the conv gathers
  // the fragments from a copy of W in LDS
  return mural::cw_conv32_fwd(x, B, L, pre_relu, acc, gamma, beta, 1e-5f, 0.1f, running_mean, running_var, state, W, wfrag_scratch, bias, post_relu,
                              res1, res2, acc_out, out_relu, y, (hipStream_t)stream);
}

extern "C" int mural_debug_cw_conv32_bwd(const float* dy, const float* x, const float* W, int64_t B, int32_t L, const float* state,
                                         const float* gamma, int32_t pre_relu, float* dz, double* stat_out, float* part, int32_t* nrow,
                                         float* wfrag_scratch, void* stream) {
  int n = 0;
  const int rc = mural::cw_conv32_bwd(dy, x, W, wfrag_scratch ? wfrag_scratch + 3072 : nullptr, B, L, state, gamma, pre_relu, dz, stat_out, part, &n,
                                      (hipStream_t)stream);
  if (nrow) *nrow = n;
  return rc;
}

extern "C" int mural_debug_cw_wfrag(const float* W, float* out6144, void* stream) {
  return mural::cw_wfrag_build(&W, 1, out6144, (hipStream_t)stream);
}

// diagnostic: per-workgroup wall-clock stamps of the forward kernel's phases (tools/phase_stamps_cw.py); NULL switches them off
extern "C" int mural_debug_cw_set_stamps(void* dev_ptr) {
  mural::g_cw_stamps = static_cast<unsigned long long*>(dev_ptr);
  return MURAL_OK;
}

// (offset, bytes) pairs of the regions of the calling thread's latest workspace carve (forward / reuse entry points and their
// *_workspace_bytes queries); returns the number of pairs written (at most max_pairs)
extern "C" int mural_debug_last_ws_layout(size_t* out, int32_t max_pairs) {
  const int n = (int)(mural::g_ws_layout.size() / 2);
  const int m = n < max_pairs ? n : max_pairs;
  for (int i = 0; i < 2 * m; ++i) out[i] = mural::g_ws_layout[i];
  return m;
}

extern "C" int mural_debug_set_stamps(void* dev_ptr) {
  mural::g_tower_stamps = static_cast<unsigned long long*>(dev_ptr);
  return MURAL_OK;
}

// diagnostic: per-workgroup wall-clock stamps of the training-mode first-layer kernels' phases (tools/phase_stamps_first.py); NULL: off
extern "C" int mural_debug_first_set_stamps(void* dev_ptr) {
  mural::g_first_stamps = static_cast<unsigned long long*>(dev_ptr);
  return MURAL_OK;
}

// diagnostic: wall-clock stamps of the fused local branch's three backward launches (uint64 [3][256][8]); NULL: off
extern "C" int mural_debug_lt_set_stamps(void* dev_ptr) {
  mural::ltrain::g_lt_stamps = static_cast<unsigned long long*>(dev_ptr);
  return MURAL_OK;
}


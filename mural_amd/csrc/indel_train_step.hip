// The INDEL training step as one C call per direction (mural_indel_forward / mural_indel_backward).
//
// Reference: UNet_Small.forward under model.train() (MuRaL/model/model_indel.py:151-176; ConvBlock :6-19) inside the step of
// MuRaL/training.py:424-436.  mural_indel_train_forward / _backward compose the per-unit ops of indel_train.hip / train_ops.hip
// (mural_op_convg_bn_fwd / _bwd: Conv1d -> batch-statistics BatchNorm -> activation -> residual adds as one call) over a
// caller-provided workspace, in the order the reference's modules run; what stock autograd does between the modules -- summing the
// gradients of tensors with two consumers, the flips of the strand-symmetrising layer -- is spelled out here.  Parameters come as
// DEVICE pointers in the state_dict naming (MuralIndelParams), gradients land in a second MuralIndelParams whose pointers the
// caller lays out (one flat buffer in the Python mirror, model/indel_train_step.py).  A host in any language needs two calls per step.
#include <algorithm>
#include <cstring>
#include <vector>

#include "common.h"

using namespace mural;

namespace mural {
void wgrad_defer_begin();                 // indel_train.hip: collect the weight-gradient partial rows of the layers that follow ...
int wgrad_defer_flush(hipStream_t st);    // ... and reduce them all in one launch
int convg_bn_bwd_add(const float* dz, const float* x, const float* W, const float* y0, const float* state, const float* gamma, int64_t B,
                     int32_t Cin, int32_t Lin, int32_t Cout, int32_t K, int32_t stride, int32_t pad, int32_t up, int32_t act, double* acc,
                     float* dy0, float* dx, const float* dx_add, float* dW, float* db, float* dgamma, float* dbeta, float* part,
                     size_t part_floats, const float* wt_dgrad, void* stream);      // indel_train.hip
}

namespace {

constexpr int IL = 6;      // U-Net levels
constexpr float kEps = 1e-5f;

// y = a + b (y may be a or b)
__global__ void add2_kernel(const float* a, const float* b, float* y, int64_t n4) {
  using f4 = __attribute__((ext_vector_type(4))) float;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f4 u = reinterpret_cast<const f4*>(a)[i], v = reinterpret_cast<const f4*>(b)[i];
    reinterpret_cast<f4*>(y)[i] = u + v;
  }
}
__global__ void add2_tail_kernel(const float* a, const float* b, float* y, int64_t lo, int64_t n) {
  const int64_t i = lo + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) y[i] = a[i] + b[i];
}
// y[b][c][l] = [add[b][c][l] +] x[b][C-1-c][L-1-l] (flip_c) or x[b][c][L-1-l]
__global__ void flip_kernel(const float* __restrict__ x, const float* __restrict__ add, float* __restrict__ y, int64_t rows, int C, int L,
                            int flip_c) {
  const int64_t total = rows * L;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / L;
    const int l = (int)(i - row * L);
    const int64_t b = row / C;
    const int c = (int)(row - b * C);
    const int64_t src_row = flip_c ? b * C + (C - 1 - c) : row;
    const float v = x[src_row * L + (L - 1 - l)];
    y[i] = add ? add[i] + v : v;
  }
}

// the relayout job table travels as a kernel argument (40 jobs = 1.9 KB): no host-to-device copy, so the step stays capturable
// into a HIP graph
constexpr int kMaxJobs = 40;
struct JobPack { MuralRelayoutJob j[kMaxJobs]; };
__global__ void fill_jobs_kernel(const JobPack pack, int n, MuralRelayoutJob* dst) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = pack.j[threadIdx.x];
}

int add2(const float* a, const float* b, float* y, int64_t n, hipStream_t st) {
  if (n <= 0) return MURAL_OK;
  const bool al = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(y)) & 15u) == 0;
  const int64_t n4 = al ? n / 4 : 0;
  if (n4 > 0) {
    const int64_t g = (n4 + 255) / 256;
    hipLaunchKernelGGL(add2_kernel, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, st, a, b, y, n4);
  }
  if (n4 * 4 < n) hipLaunchKernelGGL(add2_tail_kernel, dim3((unsigned)((n - n4 * 4 + 255) / 256)), dim3(256), 0, st, a, b, y, n4 * 4, n);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}
int flip(const float* x, const float* add, float* y, int64_t B, int C, int L, int flip_c, hipStream_t st) {
  const int64_t total = B * C * L;
  if (total == 0) return MURAL_OK;
  const int64_t g = (total + 255) / 256;
  hipLaunchKernelGGL(flip_kernel, dim3((unsigned)(g > 16384 ? 16384 : g)), dim3(256), 0, st, x, add, y, B * C, C, L, flip_c);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

// one Conv1d -> BatchNorm1d unit of the model
struct Unit {
  const float *W, *bias;                // conv (device); bias may be NULL (ConvBlock convs)
  const float *gamma, *beta;
  float *rmean, *rvar;
  float *gW, *gbias, *ggamma, *gbeta;   // gradient destinations (backward)
  int Cin, Lin, Cout, Lout, K, stride, pad, up, act;
  const float* x;                       // input (forward-saved)
  const float *res1, *res2;
  float *y0, *z, *state;
  double *acc_f, *acc_b;
  float *wt_fwd, *wt_dgrad;             // this step's kernel layouts of W
  float* part;                          // this unit's weight-gradient partial rows (reduced at the end of the backward)
  size_t part_floats;
  int job;                              // index into the relayout job table
};

struct Plan {
  int B, C[IL], L[IL], Lx, n_class;
  bool rev;
  std::vector<Unit> u;                  // forward order
  // workspace regions
  MuralRelayoutJob* jobs_dev = nullptr;
  float* wt = nullptr;
  int64_t wt_total = 0;
  double *acc_f = nullptr, *acc_b = nullptr;
  size_t acc_f_bytes = 0, acc_b_bytes = 0;
  float *xf = nullptr, *s2f = nullptr;  // flipped input; h0 = first term + flipped second term
  float *o2 = nullptr, *sp = nullptr, *feat = nullptr, *fc_state = nullptr, *fbn = nullptr, *fdrop = nullptr, *lin = nullptr;
  int32_t* arg = nullptr;
  double *acc_fc_f = nullptr, *acc_fc_b = nullptr;
  float* wt_out2 = nullptr;
  // backward temporaries: per level three C x L buffers and one 2C x L buffer; one scratch of the largest conv output; partial rows
  float *g[IL][3], *gh[IL], *dy0 = nullptr, *part = nullptr;
  size_t part_floats = 0;
  float *tmpW = nullptr;                // second strand-symmetry term's parameter gradients before they are added
  size_t bytes = 0;
};

struct Arena {
  char* base;
  size_t off = 0, guard;
  explicit Arena(void* b) : base(static_cast<char*>(b)), guard(ws_guard_bytes()) { ws_layout_reset(); }
  void* take(size_t bytes) {
    const size_t o = off;
    if (bytes) ws_layout_add(o, bytes);
    off = (off + bytes + (bytes ? guard : 0) + 255) & ~size_t(255);
    return base ? base + o : nullptr;
  }
  float* f(size_t n) { return static_cast<float*>(take(n * 4)); }
  double* d(size_t n) { return static_cast<double*>(take(n * 8)); }
};

int conv_out(int Lin, int K, int stride, int pad, int up) { return mural_op_convg_out_length(Lin, K, stride, pad, up); }

// lays out the units and the workspace; base == nullptr: sizes only.  params / grads may be NULL for a sizing call.
int make_plan(const MuralIndelShape& sh, const MuralIndelParams* p, const MuralIndelParams* gr, const float* x, int64_t B64, void* base, Plan* out) {
  Plan& P = *out;
  MURAL_REQUIRE(B64 >= 1 && B64 <= (1 << 20), "batch out of range");
  MURAL_REQUIRE(sh.channels >= 4 && sh.channels % 4 == 0 && sh.ksize >= 1 && (sh.ksize & 1) && sh.length >= 1 && sh.n_class >= 1,
                "bad INDEL shape");
  P.B = (int)B64;
  P.Lx = sh.length;
  P.n_class = sh.n_class;
  P.rev = sh.use_reverse != 0;
  const int K = sh.ksize, pad = (K - 1) / 2;
  int L = sh.length;
  for (int i = 0; i < IL; ++i) {
    MURAL_REQUIRE(sh.down[i] >= 1, "down_list entries must be >= 1");
    P.C[i] = sh.channels * (i + 1);
    L = conv_out(L, K, sh.down[i], pad, 1);
    MURAL_REQUIRE(L >= 1, "input length %d is too short for down_list", sh.length);
    P.L[i] = L;
  }
  for (int i = IL - 1; i >= 1; --i)
    MURAL_REQUIRE(P.L[i] * sh.down[i] == P.L[i - 1], "input length %d is not compatible with down_list (level %d)", sh.length, i);
  Arena A(base);
  const size_t Bz = (size_t)P.B;
  auto mk = [&](const MuralAffine* cv, const float* w_only, const MuralBN* bn, const MuralAffine* gcv, const float* gw_only, const MuralBN* gbn,
                int Cin, int Lin, int Cout, int k, int stride, int up, int act) {
    Unit u;
    std::memset(&u, 0, sizeof(u));
    u.W = cv ? cv->weight : w_only;
    u.bias = cv ? cv->bias : nullptr;
    u.gamma = bn ? bn->weight : nullptr;
    u.beta = bn ? bn->bias : nullptr;
    u.rmean = bn ? const_cast<float*>(bn->running_mean) : nullptr;
    u.rvar = bn ? const_cast<float*>(bn->running_var) : nullptr;
    u.gW = const_cast<float*>(gcv ? gcv->weight : gw_only);
    u.gbias = const_cast<float*>(gcv ? gcv->bias : nullptr);
    u.ggamma = const_cast<float*>(gbn ? gbn->weight : nullptr);
    u.gbeta = const_cast<float*>(gbn ? gbn->bias : nullptr);
    u.Cin = Cin; u.Lin = Lin; u.Cout = Cout; u.K = k; u.stride = stride; u.pad = (k - 1) / 2; u.up = up; u.act = act;
    u.Lout = conv_out(Lin, k, stride, u.pad, up);
    u.y0 = A.f(Bz * Cout * u.Lout);
    u.z = A.f(Bz * Cout * u.Lout);
    u.state = A.f((size_t)4 * Cout);
    u.job = -1;
    P.u.push_back(u);
    return (int)P.u.size() - 1;
  };
  const MuralIndelParams zero{};
  const MuralIndelParams& pp = p ? *p : zero;
  const MuralIndelParams& gg = gr ? *gr : zero;
  // ---- units in forward order (indices are used by the drivers below)
  if (P.rev) {
    mk(&pp.sym.conv, nullptr, &pp.sym.bn, &gg.sym.conv, nullptr, &gg.sym.bn, 4, P.Lx, 4, K, 1, 1, 0);      // on the input
    mk(&pp.sym.conv, nullptr, &pp.sym.bn, &gg.sym.conv, nullptr, &gg.sym.bn, 4, P.Lx, 4, K, 1, 1, 0);      // on the flipped input
  }
  for (int i = 0; i < IL; ++i) {
    const int cin = i == 0 ? 4 : P.C[i - 1], lin = i == 0 ? P.Lx : P.L[i - 1], c = P.C[i];
    mk(&pp.up_l[i].conv, nullptr, &pp.up_l[i].bn, &gg.up_l[i].conv, nullptr, &gg.up_l[i].bn, cin, lin, c, K, sh.down[i], 1, 0);
    mk(nullptr, pp.up_b[i].conv5_w, &pp.up_b[i].bn1, nullptr, gg.up_b[i].conv5_w, &gg.up_b[i].bn1, c, P.L[i], 2 * c, 5, 1, 1, 2);
    mk(nullptr, pp.up_b[i].conv1_w, &pp.up_b[i].bn2, nullptr, gg.up_b[i].conv1_w, &gg.up_b[i].bn2, 2 * c, P.L[i], c, 1, 1, 1, 0);
  }
  for (int j = 0; j < IL - 1; ++j) {
    const int lvl = IL - 2 - j, cin = P.C[lvl + 1], c = P.C[lvl];
    mk(&pp.down_l[j].conv, nullptr, &pp.down_l[j].bn, &gg.down_l[j].conv, nullptr, &gg.down_l[j].bn, cin, P.L[lvl + 1], c, K, 1, sh.down[lvl + 1], 0);
    mk(nullptr, pp.down_b[j].conv5_w, &pp.down_b[j].bn1, nullptr, gg.down_b[j].conv5_w, &gg.down_b[j].bn1, c, P.L[lvl], 2 * c, 5, 1, 1, 2);
    mk(nullptr, pp.down_b[j].conv1_w, &pp.down_b[j].bn2, nullptr, gg.down_b[j].conv1_w, &gg.down_b[j].bn2, 2 * c, P.L[lvl], c, 1, 1, 1, 0);
  }
  mk(&pp.out1, nullptr, &pp.out_bn, &gg.out1, nullptr, &gg.out_bn, P.C[0], P.L[0], P.C[0], 1, 1, 1, 1);
  // ---- relayout jobs: one per distinct conv weight (the strand-symmetry conv serves two units), + out_conv's second conv
  int64_t total = 0;
  int njobs = 0;
  for (size_t i = 0; i < P.u.size(); ++i) {
    if (P.rev && i == 1) { P.u[1].job = P.u[0].job; continue; }
    P.u[i].job = njobs++;
    total += (int64_t)P.u[i].Cout * P.u[i].Cin * P.u[i].K;
  }
  const int64_t n_out2 = (int64_t)P.C[0] * P.C[0];
  P.wt_total = total;
  P.jobs_dev = static_cast<MuralRelayoutJob*>(A.take((size_t)kMaxJobs * sizeof(MuralRelayoutJob)));
  P.wt = A.f((size_t)2 * total);
  P.wt_out2 = A.f((size_t)n_out2);
  {
    int64_t off_f = 0, off_d = total;
    for (size_t i = 0; i < P.u.size(); ++i) {
      Unit& u = P.u[i];
      if (P.rev && i == 1) { u.wt_fwd = P.u[0].wt_fwd; u.wt_dgrad = P.u[0].wt_dgrad; continue; }
      const int64_t n = (int64_t)u.Cout * u.Cin * u.K;
      u.wt_fwd = P.wt ? P.wt + off_f : nullptr;
      u.wt_dgrad = (u.stride == 1 && P.wt) ? P.wt + off_d : nullptr;
      off_f += n;
      if (u.stride == 1) off_d += n;
    }
  }
  // ---- accumulator blocks: forward and backward, one memset each
  size_t accn = 0;
  for (Unit& u : P.u) accn += (size_t)MURAL_BN_SLOTS * 2 * u.Cout;
  accn += (size_t)MURAL_BN_SLOTS * 2 * P.C[0];
  P.acc_f = A.d(accn);
  P.acc_b = A.d(accn);
  P.acc_f_bytes = P.acc_b_bytes = accn * 8;
  {
    size_t o = 0;
    for (Unit& u : P.u) {
      u.acc_f = P.acc_f ? P.acc_f + o : nullptr;
      u.acc_b = P.acc_b ? P.acc_b + o : nullptr;
      o += (size_t)MURAL_BN_SLOTS * 2 * u.Cout;
    }
    P.acc_fc_f = P.acc_f ? P.acc_f + o : nullptr;
    P.acc_fc_b = P.acc_b ? P.acc_b + o : nullptr;
  }
  // ---- the rest of the forward state
  const size_t in_floats = Bz * 4 * P.Lx, l0 = Bz * P.C[0] * P.L[0];
  if (P.rev) {
    P.xf = A.f(in_floats);
    P.s2f = A.f(in_floats);
  }
  P.o2 = A.f(l0);
  P.sp = A.f(l0);
  P.arg = static_cast<int32_t*>(A.take(Bz * P.C[0] * 4));
  P.feat = A.f(Bz * P.C[0]);
  P.fc_state = A.f((size_t)4 * P.C[0]);
  P.fbn = A.f(Bz * P.C[0]);
  P.fdrop = A.f(Bz * P.C[0]);
  P.lin = A.f(Bz * P.n_class);
  // ---- backward temporaries
  size_t dy0max = l0, partmax = 0;
  for (int i = 0; i < IL; ++i) {
    const size_t n = Bz * P.C[i] * P.L[i];
    for (int k = 0; k < 3; ++k) P.g[i][k] = A.f(n);
    P.gh[i] = A.f(2 * n);
    dy0max = std::max(dy0max, 2 * n);
  }
  for (Unit& u : P.u) {
    u.part_floats = mural_op_convg_bwd_scratch(u.Cin, u.Cout, u.K);
    u.part = A.f(u.part_floats);
  }
  partmax = mural_op_convg_bwd_scratch(P.C[0], P.C[0], 1);
  P.dy0 = A.f(dy0max);
  P.part = A.f(partmax);
  P.part_floats = partmax;
  P.tmpW = A.f((size_t)4 * 4 * K + 4 + 4 + 4 + 64);
  P.bytes = A.off;
  (void)x;
  return MURAL_OK;
}

int unit_fwd(const Plan& P, Unit& u, const float* x, const float* res1, const float* res2, float momentum, hipStream_t st) {
  u.x = x; u.res1 = res1; u.res2 = res2;
  return mural_op_convg_bn_fwd(x, nullptr, u.bias, u.wt_fwd, u.y0, P.B, u.Cin, u.Lin, u.Cout, u.K, u.stride, u.pad, u.up, u.gamma, u.beta, kEps,
                               momentum, u.rmean, u.rvar, u.acc_f, u.state, u.act, res1, res2, u.z, st);
}

// dW / db / dgamma / dbeta of the unit go to its gradient slots (or to alternative destinations), dx optional; dx_add (optional, may be
// dx itself): a second gradient of the unit's input, added while dx is written
int unit_bwd(const Plan& P, const Unit& u, const float* x, const float* dz, float* dx, const float* dx_add, float* gW, float* gb, float* gga,
             float* gbe, hipStream_t st) {
  return convg_bn_bwd_add(dz, x, u.W, u.y0, u.state, u.gamma, P.B, u.Cin, u.Lin, u.Cout, u.K, u.stride, u.pad, u.up, u.act, u.acc_b, P.dy0, dx,
                          dx_add, gW, u.bias ? gb : nullptr, gga, gbe, u.part, u.part_floats, u.up == 1 ? u.wt_dgrad : nullptr, st);
}

// four short vectors += their second strand-symmetry term, one launch
struct Add4 { float* y[4]; const float* t[4]; int n[4]; };
__global__ void add4_kernel(const Add4 j) {
  const int k = blockIdx.x;
  for (int i = threadIdx.x; i < j.n[k]; i += blockDim.x) j.y[k][i] += j.t[k][i];
}

int relayout_all(const Plan& P, const MuralIndelParams& p, hipStream_t st) {
  JobPack pack;
  std::memset(&pack, 0, sizeof(pack));
  int n = 0;
  int64_t start = 0;
  for (size_t i = 0; i < P.u.size(); ++i) {
    if (P.rev && i == 1) continue;
    const Unit& u = P.u[i];
    MURAL_REQUIRE(n < kMaxJobs, "internal: relayout job table too small");
    MuralRelayoutJob& j = pack.j[n++];
    j.W = u.W; j.wt_fwd = u.wt_fwd; j.wt_dgrad = u.wt_dgrad; j.Cout = u.Cout; j.Cin = u.Cin; j.K = u.K; j.start = start;
    start += (int64_t)u.Cout * u.Cin * u.K;
  }
  hipLaunchKernelGGL(fill_jobs_kernel, dim3(1), dim3(64), 0, st, pack, n, P.jobs_dev);
  MURAL_HIP_CHECK(hipGetLastError());
  (void)p;
  return mural_op_relayout_multi(P.jobs_dev, n, start, st);
}

int check_params(const MuralIndelShape& sh, const MuralIndelParams& p, bool grads) {
  auto bn = [&](const MuralBN& b) { return b.weight && b.bias && (grads || (b.running_mean && b.running_var)); };
  bool ok = true;
  if (sh.use_reverse) ok = ok && p.sym.conv.weight && p.sym.conv.bias && bn(p.sym.bn);
  for (int i = 0; i < IL; ++i)
    ok = ok && p.up_l[i].conv.weight && p.up_l[i].conv.bias && bn(p.up_l[i].bn) && p.up_b[i].conv5_w && bn(p.up_b[i].bn1) && p.up_b[i].conv1_w &&
         bn(p.up_b[i].bn2);
  for (int j = 0; j < IL - 1; ++j)
    ok = ok && p.down_l[j].conv.weight && p.down_l[j].conv.bias && bn(p.down_l[j].bn) && p.down_b[j].conv5_w && bn(p.down_b[j].bn1) &&
         p.down_b[j].conv1_w && bn(p.down_b[j].bn2);
  ok = ok && p.out1.weight && p.out1.bias && bn(p.out_bn) && p.out2.weight && p.out2.bias && bn(p.fc_bn) && p.fc.weight && p.fc.bias;
  MURAL_REQUIRE(ok, grads ? "INDEL gradient pointer is NULL" : "INDEL parameter pointer is NULL");
  return MURAL_OK;
}

}  // namespace

extern "C" size_t mural_indel_train_workspace_bytes(const MuralIndelShape* shape, int64_t B) {
  if (!shape || B < 1) return 0;
  Plan P;
  if (make_plan(*shape, nullptr, nullptr, nullptr, B, nullptr, &P)) return 0;
  return P.bytes;
}

// out [B][n_class] = UNet_Small.forward(x) in training mode (positive Softplus scores); everything the backward needs stays in the
// workspace.  x: dev float [B][4][length].  dropout_p / seed / seed_dev: out_fc's Dropout (counter-based generator, mural_op_dropout).
extern "C" int mural_indel_train_forward(const MuralIndelShape* shape, const MuralIndelParams* params, const float* x, int64_t B,
                                         float dropout_p, uint64_t seed, const uint64_t* seed_dev, float momentum, float* out, void* workspace,
                                         size_t workspace_bytes, void* stream_) {
  MURAL_REQUIRE(shape && params && x && out && workspace, "NULL argument");
  if (int rc = check_params(*shape, *params, false)) return rc;
  Plan P;
  if (int rc = make_plan(*shape, params, nullptr, x, B, workspace, &P)) return rc;
  if (workspace_bytes < P.bytes) {
    set_error("workspace too small: need %zu bytes, got %zu", P.bytes, workspace_bytes);
    return MURAL_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream_;
  const MuralIndelParams& p = *params;
  MURAL_HIP_CHECK(hipMemsetAsync(P.acc_f, 0, P.acc_f_bytes, st));
  if (int rc = relayout_all(P, p, st)) return rc;
  size_t ui = 0;
  const float* h = x;
  if (P.rev) {
    // conv(x) + flip_L(conv(flip_CL(x))) (model_indel.py:154-155): the same Conv1d + BatchNorm module applied twice -- first to x,
    // then to the flipped x -- so its batch statistics are taken and its running statistics updated twice, in that order
    if (int rc = unit_fwd(P, P.u[0], x, nullptr, nullptr, momentum, st)) return rc;
    if (int rc = flip(x, nullptr, P.xf, P.B, 4, P.Lx, 1, st)) return rc;
    if (int rc = unit_fwd(P, P.u[1], P.xf, nullptr, nullptr, momentum, st)) return rc;
    if (int rc = flip(P.u[1].z, P.u[0].z, P.s2f, P.B, 4, P.Lx, 0, st)) return rc;
    h = P.s2f;
    ui = 2;
  }
  const float* enc[IL];
  for (int i = 0; i < IL; ++i) {
    Unit &a = P.u[ui], &b5 = P.u[ui + 1], &b1 = P.u[ui + 2];
    if (int rc = unit_fwd(P, a, h, nullptr, nullptr, momentum, st)) return rc;
    if (int rc = unit_fwd(P, b5, a.z, nullptr, nullptr, momentum, st)) return rc;
    if (int rc = unit_fwd(P, b1, b5.z, a.z, nullptr, momentum, st)) return rc;      // x + BN(1x1(SiLU(BN(k5(x)))))
    h = enc[i] = b1.z;
    ui += 3;
  }
  for (int j = 0; j < IL - 1; ++j) {
    const int lvl = IL - 2 - j;
    Unit &a = P.u[ui], &b5 = P.u[ui + 1], &b1 = P.u[ui + 2];
    if (int rc = unit_fwd(P, a, h, nullptr, nullptr, momentum, st)) return rc;
    if (int rc = unit_fwd(P, b5, a.z, nullptr, nullptr, momentum, st)) return rc;
    if (int rc = unit_fwd(P, b1, b5.z, a.z, enc[lvl], momentum, st)) return rc;     // ConvBlock(x) + encoder skip (:168-170)
    h = b1.z;
    ui += 3;
  }
  Unit& o1 = P.u[ui];
  if (int rc = unit_fwd(P, o1, h, nullptr, nullptr, momentum, st)) return rc;
  const int C0 = P.C[0], L0 = P.L[0];
  if (int rc = mural_op_convg_fwd(o1.z, p.out2.weight, p.out2.bias, P.wt_out2, P.o2, P.B, C0, L0, C0, 1, 1, 0, 1, st)) return rc;
  if (int rc = mural_op_act_fwd(P.o2, (int64_t)P.B * C0 * L0, 3, P.sp, st)) return rc;
  if (int rc = mural_op_maxpool_fwd(P.sp, (int64_t)P.B * C0, L0, L0, L0, 0, P.feat, P.arg, st)) return rc;
  // out_fc: BatchNorm1d (batch statistics over B) -> Dropout -> Linear -> Softplus
  if (int rc = mural_op_bn_stats(P.feat, P.B, C0, 1, 0, P.acc_fc_f, st)) return rc;
  if (int rc = mural_op_bn_finalize(P.acc_fc_f, (double)P.B, C0, p.fc_bn.weight, p.fc_bn.bias, kEps, momentum, const_cast<float*>(p.fc_bn.running_mean),
                                    const_cast<float*>(p.fc_bn.running_var), P.fc_state, P.fc_state + C0, P.fc_state + 2 * C0, P.fc_state + 3 * C0, st))
    return rc;
  if (int rc = mural_op_bn_apply(P.feat, P.B, C0, 1, 0, P.fc_state, P.fc_state + C0, P.fbn, st)) return rc;
  const float* fin = P.fbn;
  if (dropout_p > 0.f) {
    if (int rc = mural_op_dropout(P.fbn, (int64_t)P.B * C0, dropout_p, seed, seed_dev, P.fdrop, st)) return rc;
    fin = P.fdrop;
  }
  if (int rc = mural_op_linear_fwd(fin, p.fc.weight, p.fc.bias, P.B, C0, P.n_class, P.lin, st)) return rc;
  return mural_op_act_fwd(P.lin, (int64_t)P.B * P.n_class, 3, out, st);
}

// gradients of every parameter from dout = dL/d(out); `grads` mirrors `params` (running statistics unused).  Must follow the
// forward of the same batch on the same workspace; x is the forward's input.
extern "C" int mural_indel_train_backward(const MuralIndelShape* shape, const MuralIndelParams* params, const MuralIndelParams* grads,
                                          const float* x, const float* dout, int64_t B, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                                          void* workspace, size_t workspace_bytes, void* stream_) {
  MURAL_REQUIRE(shape && params && grads && x && dout && workspace, "NULL argument");
  if (int rc = check_params(*shape, *params, false)) return rc;
  if (int rc = check_params(*shape, *grads, true)) return rc;
  Plan P;
  if (int rc = make_plan(*shape, params, grads, x, B, workspace, &P)) return rc;
  MURAL_REQUIRE(workspace_bytes >= P.bytes, "workspace too small");
  hipStream_t st = (hipStream_t)stream_;
  const MuralIndelParams &p = *params, &g = *grads;
  MURAL_HIP_CHECK(hipMemsetAsync(P.acc_b, 0, P.acc_b_bytes, st));
  const int C0 = P.C[0], L0 = P.L[0];
  const int64_t n0 = (int64_t)P.B * C0 * L0;
  // every conv layer leaves its weight-gradient partial rows in its own region; ONE launch reduces them all (before the two
  // strand-symmetry terms are summed).  The guard object ends the collection on every exit path.
  struct DeferGuard {
    hipStream_t st;
    bool flushed = false;
    ~DeferGuard() { if (!flushed) (void)wgrad_defer_flush(st); }
  } defer{st};
  wgrad_defer_begin();
  // ---- the forward's tensor links (make_plan lays the units out; the inputs / residuals are re-derived, not stored)
  size_t ui = P.rev ? 2 : 0;
  const float* h = P.rev ? P.s2f : x;
  const float* enc[IL];
  std::vector<const float*> unit_in(P.u.size(), nullptr);
  if (P.rev) { unit_in[0] = x; unit_in[1] = P.xf; }
  for (int i = 0; i < IL; ++i) {
    unit_in[ui] = h; unit_in[ui + 1] = P.u[ui].z; unit_in[ui + 2] = P.u[ui + 1].z;
    h = enc[i] = P.u[ui + 2].z;
    ui += 3;
  }
  const size_t dec0 = ui;
  for (int j = 0; j < IL - 1; ++j) {
    unit_in[ui] = h; unit_in[ui + 1] = P.u[ui].z; unit_in[ui + 2] = P.u[ui + 1].z;
    h = P.u[ui + 2].z;
    ui += 3;
  }
  const size_t iout = ui;
  unit_in[iout] = h;
  // ---- head: Softplus <- Linear <- Dropout <- BatchNorm1d <- max over positions <- Softplus <- Conv1d(1x1)
  float *t0 = P.g[0][0], *t1 = P.g[0][1], *t2 = P.g[0][2];
  float* dlin = P.gh[IL - 1];       // small scratch: (B, n_class) and three (B, C0) vectors (free until the encoder's deepest level)
  float* dfe = dlin + (size_t)P.B * P.n_class;
  float* dfe2 = dfe + (size_t)P.B * C0;
  float* dfe3 = dfe2 + (size_t)P.B * C0;
  if (int rc = mural_op_act_bwd(dout, P.lin, (int64_t)P.B * P.n_class, 3, dlin, st)) return rc;
  const float* fin = dropout_p > 0.f ? P.fdrop : P.fbn;
  if (int rc = mural_op_linear_bwd(dlin, fin, p.fc.weight, P.B, C0, P.n_class, dfe, const_cast<float*>(g.fc.weight), const_cast<float*>(g.fc.bias), st))
    return rc;
  const float* d = dfe;
  if (dropout_p > 0.f) {
    if (int rc = mural_op_dropout(dfe, (int64_t)P.B * C0, dropout_p, seed, seed_dev, dfe2, st)) return rc;
    d = dfe2;
  }
  if (int rc = mural_op_bn_backward(d, P.feat, P.B, C0, 1, 0, P.fc_state + 2 * C0, P.fc_state + 3 * C0, p.fc_bn.weight, P.acc_fc_b, 0, nullptr, nullptr,
                                    dfe3, const_cast<float*>(g.fc_bn.weight), const_cast<float*>(g.fc_bn.bias), st))
    return rc;
  if (int rc = mural_op_maxpool_bwd(dfe3, P.arg, (int64_t)P.B * C0, L0, 1, L0, L0, 0, t0, st)) return rc;          // t0 = d sp
  if (int rc = mural_op_act_bwd(t0, P.o2, n0, 3, t1, st)) return rc;                                                // t1 = d o2
  Unit& o1 = P.u[iout];
  if (int rc = mural_op_convg_bwd(t1, o1.z, p.out2.weight, P.B, C0, L0, C0, 1, 1, 0, 1, t0, const_cast<float*>(g.out2.weight),
                                  const_cast<float*>(g.out2.bias), P.part, P.part_floats, st)) return rc;          // t0 = d o1
  if (int rc = unit_bwd(P, o1, unit_in[iout], t0, t2, nullptr, o1.gW, o1.gbias, o1.ggamma, o1.gbeta, st)) return rc;         // t2 = d (decoder out)
  // ---- decoder, last level first.  G = gradient of the level's output d_lvl = ConvBlock(u) + enc: it is the 1x1 unit's dz AND the
  // gradient of both residuals; d u = G + dx(k5 unit); the upsampling conv's dx is the next (coarser) level's G.
  float* G = t2;                      // lives in P.g[lvl][2] of the current level
  float* d_enc[IL];                   // gradient arriving at enc[lvl] from the decoder (nullptr: none)
  for (int i = 0; i < IL; ++i) d_enc[i] = nullptr;
  for (int j = IL - 2; j >= 0; --j) {
    const int lvl = IL - 2 - j;
    const size_t a = dec0 + 3 * (size_t)j;
    const Unit &ua = P.u[a], &u5 = P.u[a + 1], &u1 = P.u[a + 2];
    float* s1 = P.g[lvl][1];
    if (int rc = unit_bwd(P, u1, unit_in[a + 2], G, P.gh[lvl], nullptr, u1.gW, nullptr, u1.ggamma, u1.gbeta, st)) return rc;    // gh = d hidden
    if (int rc = unit_bwd(P, u5, unit_in[a + 1], P.gh[lvl], s1, G, u5.gW, nullptr, u5.ggamma, u5.gbeta, st)) return rc;        // s1 = d u = G + dx of the k5 unit
    d_enc[lvl] = G;                                                                                                   // skip gradient: G itself
    float* Gnext = P.g[lvl + 1][2];
    if (int rc = unit_bwd(P, ua, unit_in[a], s1, Gnext, nullptr, ua.gW, ua.gbias, ua.ggamma, ua.gbeta, st)) return rc;
    G = Gnext;
  }
  // ---- encoder, deepest level first.  d e_5 = G (from the decoder's first unit); d e_i (i < 5) = dx of level i + 1's strided conv
  // + the decoder's skip gradient.
  float* de = G;                      // gradient of enc[IL - 1], in P.g[IL - 1][2]
  for (int i = IL - 1; i >= 0; --i) {
    const size_t a = (P.rev ? 2 : 0) + 3 * (size_t)i;
    const Unit &ua = P.u[a], &u5 = P.u[a + 1], &u1 = P.u[a + 2];
    float* s1 = P.g[i][1];
    if (int rc = unit_bwd(P, u1, unit_in[a + 2], de, P.gh[i], nullptr, u1.gW, nullptr, u1.ggamma, u1.gbeta, st)) return rc;
    if (int rc = unit_bwd(P, u5, unit_in[a + 1], P.gh[i], s1, de, u5.gW, nullptr, u5.ggamma, u5.gbeta, st)) return rc;      // s1 = d a_i = de + dx of the k5 unit
    if (i > 0) {
      // dx of the strided conv + the decoder's skip gradient, which sits in P.g[i-1][2]: summed in place while dx is written
      MURAL_REQUIRE(d_enc[i - 1] == P.g[i - 1][2], "internal: skip gradient is not where the encoder expects it");
      if (int rc = unit_bwd(P, ua, unit_in[a], s1, P.g[i - 1][2], d_enc[i - 1], ua.gW, ua.gbias, ua.ggamma, ua.gbeta, st)) return rc;
      de = P.g[i - 1][2];
    } else if (P.rev) {
      float* dh0 = P.gh[0];           // (B, 4, L): fits the 2C x L buffer
      if (int rc = unit_bwd(P, ua, unit_in[a], s1, dh0, nullptr, ua.gW, ua.gbias, ua.ggamma, ua.gbeta, st)) return rc;
      // h0 = unit0(x) + flip_L(unit1(flip_CL(x))): dz of unit 0 is d h0, dz of unit 1 is flip_L(d h0); the input needs no gradient.
      // Both are the same module: the second call's parameter gradients go to a scratch and are added to the first's.
      const Unit &s_a = P.u[0], &s_b = P.u[1];
      const int nW = 4 * 4 * s_a.K;
      if (int rc = unit_bwd(P, s_a, x, dh0, nullptr, nullptr, s_a.gW, s_a.gbias, s_a.ggamma, s_a.gbeta, st)) return rc;
      float* dflip = P.g[0][0];
      if (int rc = flip(dh0, nullptr, dflip, P.B, 4, P.Lx, 0, st)) return rc;
      float *tW = P.tmpW, *tb = P.tmpW + ((nW + 3) & ~3), *tg = tb + 4, *tbe = tg + 4;
      if (int rc = unit_bwd(P, s_b, P.xf, dflip, nullptr, nullptr, tW, tb, tg, tbe, st)) return rc;
      defer.flushed = true;
      if (int rc = wgrad_defer_flush(st)) return rc;
      const Add4 j4{{s_a.gW, s_a.gbias, s_a.ggamma, s_a.gbeta}, {tW, tb, tg, tbe}, {nW, 4, 4, 4}};
      hipLaunchKernelGGL(add4_kernel, dim3(4), dim3(128), 0, st, j4);
      MURAL_HIP_CHECK(hipGetLastError());
    } else {
      if (int rc = unit_bwd(P, ua, unit_in[a], s1, nullptr, nullptr, ua.gW, ua.gbias, ua.ggamma, ua.gbeta, st)) return rc;
    }
  }
  if (!defer.flushed) {
    defer.flushed = true;
    return wgrad_defer_flush(st);
  }
  return MURAL_OK;
}

// One C call per direction for the SNV training step (SURVEY.md section 8b "mural_snv_backward"; reference: the forward of
// MuRaL/model/model_snv.py:439-525 under model.train() and loss.backward() of MuRaL/training.py:424-427).
//
// mural_snv_train_forward  : training-mode forward of Network0 / Network1 / Network2 (batch-statistics BatchNorm with running
//                            statistics updated in place, dropout) -> (B, n_class) outputs; every tensor the backward needs stays in
//                            the caller's workspace.
// mural_snv_train_backward : gradient of every parameter tensor (written to the caller's buffers, laid out like the parameters)
//                            from d(loss)/d(output) and that workspace.
// The host composes the training kernels of train_ops.hip / conv32_mfma.hip / snv_stage1.hip here in C++, so a non-Python host needs
// nothing but these two calls, an optimiser and a loss.  Parameters are DEVICE pointers in the reference's state_dict naming
// (the MuralSnvParams structs of the eval path, here with device addresses).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <utility>

#include "snv.h"
#include "conv32_jobs.h"

using namespace mural;

namespace mural {   // conv32_cl.hip: every activation of a tower is channel-last [B][L][32] inside the step
int cl_conv32_supported(int L);
int cl_conv32_fwd(const float* x, int64_t B, int L, int pre_relu, const double* acc, const float* gamma, const float* beta, float eps,
                  float momentum, float* running_mean, float* running_var, float* state, const float* W, const float* bias, int post_relu,
                  const float* res1, const float* res2, double* acc_out, int out_relu, float* y, hipStream_t stream);
size_t cl_conv32_part_floats();
int cl_conv32_bwd(const float* dy, const float* x, const float* W, int64_t B, int L, const float* state, int pre_relu, float* dz,
                  double* stat_out, float* part, int* nrow, hipStream_t stream);
// conv32_wave.hip: the same two layers with wave-private units (raw output, batch sums of relu(y))
int cw_conv32_supported(int L);
int cw_conv32_fwd(const float* x, int64_t B, int L, int pre_relu, const double* acc, const float* gamma, const float* beta, float eps,
                  float momentum, float* running_mean, float* running_var, float* state, const float* W, const float* wfrag, const float* bias,
                  int post_relu, const float* res1, const float* res2, double* acc_out, int out_relu, float* y, hipStream_t stream);
int cw_conv32_bwd(const float* dy, const float* x, const float* W, const float* wfrag, int64_t B, int L, const float* state, const float* gamma,
                  int pre_relu, float* dz, double* stat_out, float* part, int* nrow, hipStream_t stream);
size_t cw_wfrag_floats();
int cw_wfrag_build(const float* const* W, int n, float* out, hipStream_t stream);
int cl_bn_stats(const float* x, int64_t rows, int relu, double* acc, hipStream_t stream);
int cl_bn_bwd_apply(const float* dz, const float* x, int64_t rows, int relu, const float* state, const float* gamma, const double* acc,
                    const float* add1, const float* add2, float* dx, float* dgamma, float* dbeta, hipStream_t stream);
int cl_maxpool_fwd(const float* x, int64_t B, int L, int k, int s, int p, float* y, int32_t* arg, double* acc, hipStream_t stream);
int cl_maxpool_bwd(const float* dy, const int32_t* arg, int64_t B, int L, int Lout, int k, int s, int p, float* dx, hipStream_t stream);
int cl_maxpool_bwd_fold(const BnApplyJob& f, const int32_t* arg, int64_t B, int L, int Lout, int k, int s, int p, float* dx, hipStream_t stream);
int cl_gmax_fwd_jobs(const GmaxFwdJob* jobs, int n, hipStream_t stream);
int cw_conv32_bwd_jobs(ConvBwdJob* jobs, int n, hipStream_t stream);
int cl_gmax_relu_bwd(const float* dfeat, const int32_t* arg, const float* c3, int64_t B, int L, float* dx, hipStream_t stream);
// train_ops.hip / snv_stage1.hip: first layer with channel-last output
int train_first_fwd_cl(const uint8_t* sym, int64_t B, int Lwin, int col0, int L1, int pk, int ps, int pp, const float* gamma, const float* beta,
                       const float* W, const float* bias, float eps, float momentum, float* running_mean, float* running_var,
                       unsigned long long* counts, float* tab, float* y, void* arg, double* stat, hipStream_t stream);
int train_first_prepare2(const uint8_t* sym, int64_t B, int Lwin, const int* col0, const int* L1, const float* const* gamma,
                         const float* const* beta, const float* const* W, const float* const* bias, float* const* running_mean,
                         float* const* running_var, unsigned long long* const* counts, float* const* tab, float eps, float momentum,
                         hipStream_t stream);
int train_first_fwd_cl_prepared(const uint8_t* sym, int64_t B, int Lwin, int col0, int L1, int pk, int ps, int pp, const float* tab, float* y,
                                void* arg, double* stat, hipStream_t stream);
int train_first_bwd_cl(const float* dy, const void* arg, const uint8_t* sym, int64_t B, int Lwin, int col0, int L1, int pk, int ps, int pp,
                       const float* tab, const float* W, float* scratch, float* dW, float* dbias, float* dgamma, float* dbeta,
                       const FirstFold* fold, hipStream_t stream);
// conv32_mfma.hip
int train_reduce_parts(const float* const* part, const int* nrow, float* const* dW, float* const* db, int njobs, hipStream_t stream);
// train_ops.hip
int train_bn2d_apply_dropout(const float* x, int64_t B, int C, int relu, const double* acc, const float* gamma, const float* beta, float eps,
                             float momentum, float* running_mean, float* running_var, float* state, float p, uint64_t seed,
                             const uint64_t* seed_dev, float* y_bn, float* y, hipStream_t stream);
// snv_head_train.h (train_ops.hip): a tower's head in two launches per direction
bool head_train_fused_ok(int nc);
int head_train_fwd(const float* c3, int64_t B, int L, float* feat, int32_t* arg, double* acc, const float* gamma, const float* beta, float eps,
                   float momentum, float* running_mean, float* running_var, float* state, float p, uint64_t seed, const uint64_t* seed_dev,
                   float* fd, const float* W, const float* bias, int nc, float* logits, hipStream_t stream);
int head_train_bwd(const float* dlogits, const float* W, int nc, int64_t B, int L, const float* feat, const float* state, const float* gamma,
                   float p, uint64_t seed, const uint64_t* seed_dev, float* dd, double* acc, const int32_t* arg, const float* c3, float* dx,
                   float* dgamma, float* dbeta, hipStream_t stream);
int head_train_wgrad(const float* dlogits, const float* fd, int64_t B, int nc, float* dW, float* db, hipStream_t stream);
// snv_local_train.h (train_ops.hip): the local branch in three launches per direction
bool local_train_fused_ok(int in1, int h1, int h2, int nc, int emb_rows, int64_t B);
int local_train_fwd(const int64_t* cat, const float* E, int cols, int emb_rows, int64_t B, const int* dims, const float* const* W,
                    const float* const* bias, const float* const* gamma, const float* const* beta, float* const* running_mean,
                    float* const* running_var, float* const* state, double* const* acc_f, const float* drop, const uint64_t* seeds,
                    const uint64_t* seed_dev, float eps, float momentum, float* const* xt, float* const* lin, float* logits, hipStream_t stream);
int local_train_bwd(const int64_t* cat, int cols, int emb_rows, int64_t B, const int* dims, const float* dlogits, const float* const* W,
                    const float* const* gamma, const float* const* state, double* const* acc_b, const float* drop, const uint64_t* seeds,
                    const uint64_t* seed_dev, const float* const* xt, const float* const* lin, float* const* dd, float* const* g,
                    float* const* dW, float* const* db, float* const* dgamma, float* const* dbeta, float* dE, hipStream_t stream);
}

namespace {

constexpr int TR_C = 32;
constexpr size_t ACC_DOUBLES_32 = (size_t)MURAL_BN_SLOTS * 2 * TR_C;

struct Arena {            // bump allocator over the caller's workspace; base == nullptr: dry run that only measures
  char* base;
  size_t off = 0;
  size_t guard;           // validation only (MURAL_DEBUG_WS_GUARD, common.h): unused bytes behind every region
  explicit Arena(void* b) : base(static_cast<char*>(b)), guard(ws_guard_bytes()) { ws_layout_reset(); }
  void* take(size_t bytes) {
    const size_t o = off;
    if (bytes) ws_layout_add(o, bytes);
    off = (off + bytes + (bytes ? guard : 0) + 255) & ~size_t(255);
    return base ? base + o : nullptr;
  }
  float* f(size_t n) { return static_cast<float*>(take(n * 4)); }
  double* d(size_t n) { return static_cast<double*>(take(n * 8)); }
  int32_t* i(size_t n) { return static_cast<int32_t*>(take(n * 4)); }
  uint8_t* b(size_t n) { return static_cast<uint8_t*>(take(n)); }
};

struct TowerGeo { int L1, col0, L[3], pk[3], ps[3], pp[3]; };

struct StageBufs {        // a run of two ResBlocks on [B][32][L]
  float* t[4];            // h, x1, h2, xo
  float* state[4];        // [4][32] scale | shift | mean | invstd per layer
  double* acc_f[4];       // forward: sums feeding BN of layer i (acc_f[0] may be produced by the previous op)
  double* acc_b[4];       // backward: BatchNorm-backward sums of layer i
};

struct TowerBufs {
  unsigned long long* counts;
  float* tab;
  uint8_t* arg1;
  float* x0;
  StageBufs s2;
  float* p2; int32_t* arg2;
  float* x0b; float* state_c2; double* acc_c2_f; double* acc_c2_b;
  StageBufs s3;
  float* p3; int32_t* arg3;
  float* c3; float* state_c3; double* acc_c3_f; double* acc_c3_b;
  float* feat; int32_t* argg;
  float* fc_state; double* acc_fc_f; double* acc_fc_b;
  float* fb; float* fd; float* logits;
};

struct LocalBufs {
  float* emb; float* emb_do;
  float* lin[2]; float* bn_state[2]; double* acc_f[2]; double* acc_b[2];
  float* bn_out[2]; float* dout[2];
  float* logits;
};

struct Plan {
  int B, nc, Lwin;
  TowerGeo geo[2];          // 0 = mid ("" suffix), 1 = large ("_2")
  uint8_t* sym;
  TowerBufs tw[2];
  LocalBufs loc;
  // accumulator region (zeroed once per direction)
  double* acc_begin; size_t acc_bytes;
  // backward temporaries
  float* g[4];              // gradient buffers of the largest activation shape (local branch, large tower)
  float* g_mid[4];          // the mid tower's own set: the two towers run on two streams
  float* g_loc[3];          // the local branch's (it shares the mid tower's stream, but not its buffers: no ordering to maintain)
  float* part[24]; size_t part_floats;   // one partial-row region per conv32 layer (reduced by ONE launch at the end of the backward)
  float* dlogit[3];         // gradients of the local / mid / large logits
  float* first_scratch[2];  // per tower
  float* wfrag;             // filter fragments of the 20 conv32 layers (forward | input gradient), written once per step by the forward
  size_t total;
};

int pool_len(int L, int k, int s, int p) { return (L + 2 * p - k) / s + 1; }

void stage_bufs(Arena& A, StageBufs& s, size_t n) {
  for (int i = 0; i < 4; ++i) {
    s.t[i] = A.f(n);
    s.state[i] = A.f(4 * TR_C);
  }
}

// lay the workspace out; all accumulator blocks are carved from one contiguous region
int make_plan(const MuralSnvShape& sh, int64_t B, void* ws, Plan* P) {
  static const int pools[2][3][3] = {{{3, 3, 1}, {3, 3, 1}, {3, 3, 1}}, {{15, 15, 7}, {7, 7, 3}, {3, 3, 1}}};
  std::memset(P, 0, sizeof(*P));
  P->B = (int)B;
  P->nc = sh.n_class;
  P->Lwin = sh.distal_len;
  Arena A(ws);
  const bool towers = sh.model_no != 0, local = sh.model_no != 1;
  size_t max_act = 0;
  if (towers) {
    P->sym = A.b((size_t)B * sh.distal_len);
    for (int t = 0; t < 2; ++t) {
      TowerGeo& g = P->geo[t];
      g.L1 = t == 0 ? 2 * SNV_MID_HALF + 1 : sh.distal_len;
      g.col0 = t == 0 ? sh.distal_len / 2 - SNV_MID_HALF : 0;
      int L = g.L1;
      for (int i = 0; i < 3; ++i) {
        g.pk[i] = pools[t][i][0]; g.ps[i] = pools[t][i][1]; g.pp[i] = pools[t][i][2];
        L = pool_len(L, g.pk[i], g.ps[i], g.pp[i]);
        MURAL_REQUIRE(L >= 1, "distal window too short for the pooling pyramid");
        g.L[i] = L;
      }
      MURAL_REQUIRE(cl_conv32_supported(g.L[0]), "training: pooled rows of %d columns do not fit the MFMA conv tile", g.L[0]);
      TowerBufs& b = P->tw[t];
      int64_t tabf, argb, scr;
      mural_op_first_plan(TR_C, g.pk[0], &tabf, &argb, &scr);
      b.tab = A.f((size_t)tabf);
      const size_t n2 = (size_t)B * TR_C * g.L[0], n3 = (size_t)B * TR_C * g.L[1], n4 = (size_t)B * TR_C * g.L[2];
      max_act = std::max(max_act, n2);
      b.arg1 = A.b(n2 * (size_t)argb);
      b.x0 = A.f(n2);
      stage_bufs(A, b.s2, n2);
      b.p2 = A.f(n3); b.arg2 = A.i(n3);
      b.x0b = A.f(n3); b.state_c2 = A.f(4 * TR_C);
      stage_bufs(A, b.s3, n3);
      b.p3 = A.f(n4); b.arg3 = A.i(n4);
      b.c3 = A.f(n4); b.state_c3 = A.f(4 * TR_C);
      b.feat = A.f((size_t)B * TR_C); b.argg = A.i((size_t)B * TR_C);
      b.fc_state = A.f(4 * TR_C);
      b.fb = A.f((size_t)B * TR_C); b.fd = A.f((size_t)B * TR_C);
      b.logits = A.f((size_t)B * sh.n_class);
    }
  }
  if (local) {
    LocalBufs& l = P->loc;
    const int in1 = 5 * sh.local_cols, h[2] = {sh.hidden1, sh.hidden2};
    l.emb = A.f((size_t)B * in1); l.emb_do = A.f((size_t)B * in1);
    for (int i = 0; i < 2; ++i) {
      l.lin[i] = A.f((size_t)B * h[i]);
      l.bn_state[i] = A.f(4 * (size_t)h[i]);
      l.bn_out[i] = A.f((size_t)B * h[i]);
      l.dout[i] = A.f((size_t)B * h[i]);
    }
    l.logits = A.f((size_t)B * sh.n_class);
    max_act = std::max(max_act, (size_t)B * std::max(in1, std::max(h[0], h[1])));
  }
  // ---- accumulators
  const size_t acc0 = A.off;
  char* acc_base = static_cast<char*>(A.take(0));
  const size_t guard_outside = A.guard;
  A.guard = 0;            // the accumulator blocks are one range, cleared by one memset per step
  if (towers) {
    for (int t = 0; t < 2; ++t) {
      TowerBufs& b = P->tw[t];
      for (StageBufs* s : {&b.s2, &b.s3})
        for (int i = 0; i < 4; ++i) { s->acc_f[i] = A.d(ACC_DOUBLES_32); s->acc_b[i] = A.d(ACC_DOUBLES_32); }
      b.acc_c2_f = A.d(ACC_DOUBLES_32); b.acc_c2_b = A.d(ACC_DOUBLES_32);
      b.acc_c3_f = A.d(ACC_DOUBLES_32); b.acc_c3_b = A.d(ACC_DOUBLES_32);
      b.acc_fc_f = A.d(ACC_DOUBLES_32); b.acc_fc_b = A.d(ACC_DOUBLES_32);
      b.counts = static_cast<unsigned long long*>(A.take(16 * 8));     // symbol histogram of the first layer (forward only)
    }
  }
  if (local) {
    const int h[2] = {sh.hidden1, sh.hidden2};
    for (int i = 0; i < 2; ++i) {
      P->loc.acc_f[i] = A.d((size_t)MURAL_BN_SLOTS * 2 * h[i]);
      P->loc.acc_b[i] = A.d((size_t)MURAL_BN_SLOTS * 2 * h[i]);
    }
  }
  P->acc_begin = reinterpret_cast<double*>(acc_base);
  P->acc_bytes = A.off - acc0;
  A.guard = guard_outside;
  A.off += A.guard;       // (behind the accumulator range as a whole)
  // ---- backward temporaries
  for (int i = 0; i < 4; ++i) P->g[i] = A.f(max_act);
  if (towers)
    for (int i = 0; i < 4; ++i) P->g_mid[i] = A.f((size_t)B * TR_C * P->geo[0].L[0]);
  if (local)
    for (int i = 0; i < 3; ++i) P->g_loc[i] = A.f((size_t)B * std::max(5 * sh.local_cols, std::max(sh.hidden1, sh.hidden2)));
  P->part_floats = cl_conv32_part_floats();
  if (towers)
    for (int i = 0; i < 20; ++i) P->part[i] = A.f(P->part_floats);
  for (int i = 0; i < 3; ++i) P->dlogit[i] = A.f((size_t)B * sh.n_class);
  if (towers) {
    int64_t tabf, argb, scr = 0;
    for (int t = 0; t < 2; ++t) {
      mural_op_first_plan(TR_C, P->geo[t].pk[0], &tabf, &argb, &scr);
      P->first_scratch[t] = A.f((size_t)scr);
    }
  }
  if (towers) P->wfrag = A.f(20 * cw_wfrag_floats());
  P->total = A.off;
  return MURAL_OK;
}

const float EPS = 1e-5f;

// which conv kernels the step runs: the wave-private ones (conv32_wave.hip) where they apply, unless MURAL_TRAIN_CONV_CL=1 asks for the
// workgroup-tile kernels everywhere (A/B runs, parity tests of both)
bool use_wave_conv(int L, int post_relu, bool stats, int out_relu) {
  const char* e = dev_env("MURAL_TRAIN_CONV_CL");
  if (e && atoi(e) != 0) return false;
  return cw_conv32_supported(L) && !post_relu && (!stats || out_relu);
}

// The two towers share nothing but the symbols, so the mid tower runs on a side stream next to the large one: its short rows
// (67 / 23 / 8 columns) leave most of the chip idle, and the fixed cost of its ~60 launches per direction hides behind the large
// tower's kernels (SideStream, common.h).

struct Ctx {
  const MuralSnvShape* sh;
  const MuralSnvParams* p;
  const MuralSnvParams* gr;   // gradient destinations (backward)
  Plan* P;
  float momentum;
  void* stream;
  const float* conv_w[20];    // the conv32 weights in the order of Plan::wfrag
  // weight-gradient partial rows waiting for the final reduction
  int njobs = 0;
  const float* job_part[24];
  int job_nrow[24];
  float* job_dW[24];
  float* job_db[24];
  bool first_prepared = false;   // forward: the first layers' histograms / tables are in place (train_first_prepare2)
};

// the conv32 weights of the step in a fixed order (per tower: RBs1 convs, conv2, RBs2 convs, conv3) and their fragments in the workspace
void conv_weight_table(Ctx& c) {
  int n = 0;
  for (const MuralTower* T : {&c.p->mid, &c.p->large}) {
    for (const MuralResBlock* rb : {T->rbs1, T->rbs2}) {
      for (int i = 0; i < 2; ++i) {
        c.conv_w[n++] = rb[i].conv1.weight;
        c.conv_w[n++] = rb[i].conv2.weight;
      }
      c.conv_w[n++] = rb == T->rbs1 ? T->conv_mid.weight : T->conv_out.weight;
    }
  }
}
const float* wfrag_of(const Ctx& c, const float* W, int dgrad) {
  for (int i = 0; i < 20; ++i)
    if (c.conv_w[i] == W) return c.P->wfrag + ((size_t)2 * i + dgrad) * (cw_wfrag_floats() / 2);
  return nullptr;
}

// ---- forward of one BN -> conv32 layer
int bnconv_f(Ctx& c, const float* x, int L, int pre_relu, double* acc, bool have_acc, const MuralBN& bn, const MuralAffine& cv,
             float* state, int post_relu, const float* r1, const float* r2, double* acc_out, int out_relu, float* y) {
  if (!have_acc)
    if (int rc = cl_bn_stats(x, (int64_t)c.P->B * L, pre_relu, acc, (hipStream_t)c.stream)) return rc;
  if (use_wave_conv(L, post_relu, acc_out != nullptr, out_relu))
    return cw_conv32_fwd(x, c.P->B, L, pre_relu, acc, bn.weight, bn.bias, EPS, c.momentum, const_cast<float*>(bn.running_mean),
                         const_cast<float*>(bn.running_var), state, cv.weight, wfrag_of(c, cv.weight, 0), cv.bias, post_relu, r1, r2, acc_out,
                         out_relu, y, (hipStream_t)c.stream);
  return cl_conv32_fwd(x, c.P->B, L, pre_relu, acc, bn.weight, bn.bias, EPS, c.momentum, const_cast<float*>(bn.running_mean),
                       const_cast<float*>(bn.running_var), state, cv.weight, cv.bias, post_relu, r1, r2, acc_out, out_relu, y,
                       (hipStream_t)c.stream);
}

int stage_f(Ctx& c, const MuralResBlock* rb, const float* x_in, int L, double* acc_in, bool have_in, StageBufs& s) {
  // every conv takes the batch sums of relu(its output) in its epilogue for the BatchNorm of the next layer
  if (int rc = bnconv_f(c, x_in, L, 1, have_in ? acc_in : s.acc_f[0], have_in, rb[0].bn1, rb[0].conv1, s.state[0], 0, nullptr, nullptr,
                        s.acc_f[1], 1, s.t[0])) return rc;
  if (int rc = bnconv_f(c, s.t[0], L, 1, s.acc_f[1], true, rb[0].bn2, rb[0].conv2, s.state[1], 0, x_in, nullptr, s.acc_f[2], 1, s.t[1]))
    return rc;
  if (int rc = bnconv_f(c, s.t[1], L, 1, s.acc_f[2], true, rb[1].bn1, rb[1].conv1, s.state[2], 0, nullptr, nullptr, s.acc_f[3], 1, s.t[2]))
    return rc;
  // the second block's own residual (x1) plus the outer skip (x_in), model_snv.py:477-479
  return bnconv_f(c, s.t[2], L, 1, s.acc_f[3], true, rb[1].bn2, rb[1].conv2, s.state[3], 0, s.t[1], x_in, nullptr, 0, s.t[3]);
}

int dropout_f(Ctx& c, const float* x, int64_t n, float p, uint64_t seed, const uint64_t* seed_dev, float* y, const float** out) {
  if (p <= 0.f) { *out = x; return MURAL_OK; }
  *out = y;
  return mural_op_dropout(x, n, p, seed, seed_dev, y, c.stream);
}

int tower_f(Ctx& c, int t, const MuralTower& T, float drop_p, uint64_t seed, const uint64_t* seed_dev) {
  Plan& P = *c.P;
  const TowerGeo& g = P.geo[t];
  TowerBufs& b = P.tw[t];
  const int B = P.B;
  hipStream_t st = (hipStream_t)c.stream;
  if (c.first_prepared) {      // histograms and tables of both towers were made in front of the fork (train_first_prepare2)
    if (int rc = train_first_fwd_cl_prepared(P.sym, B, P.Lwin, g.col0, g.L1, g.pk[0], g.ps[0], g.pp[0], b.tab, b.x0, b.arg1, b.s2.acc_f[0], st))
      return rc;
  } else if (int rc = train_first_fwd_cl(P.sym, B, P.Lwin, g.col0, g.L1, g.pk[0], g.ps[0], g.pp[0], T.bn_in.weight, T.bn_in.bias,
                                         T.conv_in.weight, T.conv_in.bias, EPS, c.momentum, const_cast<float*>(T.bn_in.running_mean),
                                         const_cast<float*>(T.bn_in.running_var), b.counts, b.tab, b.x0, b.arg1, b.s2.acc_f[0], st)) {
    return rc;
  }
  if (int rc = stage_f(c, T.rbs1, b.x0, g.L[0], b.s2.acc_f[0], true, b.s2)) return rc;
  if (int rc = cl_maxpool_fwd(b.s2.t[3], B, g.L[0], g.pk[1], g.ps[1], g.pp[1], b.p2, b.arg2, b.acc_c2_f, st)) return rc;
  if (int rc = bnconv_f(c, b.p2, g.L[1], 0, b.acc_c2_f, true, T.bn_mid, T.conv_mid, b.state_c2, 0, nullptr, nullptr, b.s3.acc_f[0], 1,
                        b.x0b)) return rc;
  if (int rc = stage_f(c, T.rbs2, b.x0b, g.L[1], b.s3.acc_f[0], true, b.s3)) return rc;
  if (int rc = cl_maxpool_fwd(b.s3.t[3], B, g.L[1], g.pk[2], g.ps[2], g.pp[2], b.p3, b.arg3, b.acc_c3_f, st)) return rc;
  // conv3 writes its RAW output: its ReLU (model_snv.py:386-387) is taken by the global max behind it (max_l relu(v) = relu(max_l v),
  // and the backward masks on v > 0 either way), so the layer runs on the same kernels as the others
  if (int rc = bnconv_f(c, b.p3, g.L[2], 0, b.acc_c3_f, true, T.bn_out, T.conv_out, b.state_c3, 0, nullptr, nullptr, nullptr, 0, b.c3))
    return rc;
  if (head_train_fused_ok(P.nc))      // global max + statistics, then BatchNorm + dropout + Linear: two launches (snv_head_train.h)
    return head_train_fwd(b.c3, B, g.L[2], b.feat, b.argg, b.acc_fc_f, T.fc_bn.weight, T.fc_bn.bias, EPS, c.momentum,
                          const_cast<float*>(T.fc_bn.running_mean), const_cast<float*>(T.fc_bn.running_var), b.fc_state, drop_p, seed, seed_dev,
                          b.fd, T.fc.weight, T.fc.bias, P.nc, b.logits, st);
  const GmaxFwdJob gj{b.c3, B, g.L[2], 1, b.feat, b.argg};
  if (int rc = cl_gmax_fwd_jobs(&gj, 1, st)) return rc;
  // distal_fc: BatchNorm1d -> Dropout -> Linear on (B, 32)
  if (int rc = mural_op_bn_stats(b.feat, B, TR_C, 1, 0, b.acc_fc_f, c.stream)) return rc;
  if (int rc = train_bn2d_apply_dropout(b.feat, B, TR_C, 0, b.acc_fc_f, T.fc_bn.weight, T.fc_bn.bias, EPS, c.momentum,
                                        const_cast<float*>(T.fc_bn.running_mean), const_cast<float*>(T.fc_bn.running_var), b.fc_state, drop_p,
                                        seed, seed_dev, nullptr, b.fd, (hipStream_t)c.stream)) return rc;
  const float* fin = b.fd;
  return mural_op_linear_fwd(fin, T.fc.weight, T.fc.bias, B, TR_C, P.nc, b.logits, c.stream);
}

int local_f(Ctx& c, const int64_t* cat, const float* drop, const uint64_t* seeds, const uint64_t* seed_dev) {
  Plan& P = *c.P;
  const MuralSnvShape& sh = *c.sh;
  const MuralLocal& L = c.p->local;
  LocalBufs& l = P.loc;
  const int B = P.B, in1 = 5 * sh.local_cols, h[2] = {sh.hidden1, sh.hidden2};
  if (local_train_fused_ok(in1, h[0], h[1], P.nc, sh.emb_rows, B)) {
    const int dims[4] = {in1, h[0], h[1], P.nc};
    const float* W[3] = {L.lin[0].weight, L.lin[1].weight, L.out.weight};
    const float* bias[3] = {L.lin[0].bias, L.lin[1].bias, L.out.bias};
    const float* gamma[2] = {L.bn[0].weight, L.bn[1].weight};
    const float* beta[2] = {L.bn[0].bias, L.bn[1].bias};
    float* rmean[2] = {const_cast<float*>(L.bn[0].running_mean), const_cast<float*>(L.bn[1].running_mean)};
    float* rvar[2] = {const_cast<float*>(L.bn[0].running_var), const_cast<float*>(L.bn[1].running_var)};
    float* xt[3] = {l.emb_do, l.dout[0], l.dout[1]};
    return local_train_fwd(cat, L.emb, sh.local_cols, sh.emb_rows, B, dims, W, bias, gamma, beta, rmean, rvar, l.bn_state, l.acc_f, drop, seeds,
                           seed_dev, EPS, c.momentum, xt, l.lin, l.logits, (hipStream_t)c.stream);
  }
  if (int rc = mural_op_embedding_fwd(cat, L.emb, B, sh.local_cols, sh.emb_rows, l.emb, c.stream)) return rc;
  const float* x;
  if (int rc = dropout_f(c, l.emb, (int64_t)B * in1, drop[0], seeds[0], seed_dev, l.emb_do, &x)) return rc;
  int in = in1;
  for (int i = 0; i < 2; ++i) {   // Linear -> ReLU -> BN -> Dropout (model_snv.py:466-467)
    if (int rc = mural_op_linear_fwd(x, L.lin[i].weight, L.lin[i].bias, B, in, h[i], l.lin[i], c.stream)) return rc;
    if (int rc = mural_op_bn_stats(l.lin[i], B, h[i], 1, 1, l.acc_f[i], c.stream)) return rc;
    if (int rc = train_bn2d_apply_dropout(l.lin[i], B, h[i], 1, l.acc_f[i], L.bn[i].weight, L.bn[i].bias, EPS, c.momentum,
                                          const_cast<float*>(L.bn[i].running_mean), const_cast<float*>(L.bn[i].running_var), l.bn_state[i],
                                          drop[1 + i], seeds[1 + i], seed_dev, nullptr, l.dout[i], (hipStream_t)c.stream)) return rc;
    x = l.dout[i];
    in = h[i];
  }
  return mural_op_linear_fwd(x, L.out.weight, L.out.bias, B, in, P.nc, l.logits, c.stream);
}

// ---- backward of one BN -> conv32 layer: dx (+ add1 + add2), parameter gradients to their destinations
// defer != nullptr (wave kernels, no residual gradients): the BatchNorm-backward apply is described in *defer instead of launched -- the
// caller's next kernel (the pool's backward) makes dx per element from (dz, x, sums); defer->dz stays nullptr where the apply did run
int bnconv_b(Ctx& c, const float* dy, const float* x, int L, int pre_relu, const float* state, const MuralBN& bn, const MuralAffine& cv,
             double* acc, const float* add1, const float* add2, const MuralBN& gbn, const MuralAffine& gcv, float* dz, float* dx,
             BnApplyJob* defer = nullptr) {
  if (defer) defer->dz = nullptr;
  const int j = c.njobs++;
  MURAL_REQUIRE(j < 20, "internal: more conv layers than partial-row regions");
  if (use_wave_conv(L, 0, false, 0)) {
    if (int rc = cw_conv32_bwd(dy, x, cv.weight, wfrag_of(c, cv.weight, 1), c.P->B, L, state, bn.weight, pre_relu, dz, acc, c.P->part[j],
                               &c.job_nrow[j], (hipStream_t)c.stream)) return rc;
  } else if (int rc = cl_conv32_bwd(dy, x, cv.weight, c.P->B, L, state, pre_relu, dz, acc, c.P->part[j], &c.job_nrow[j], (hipStream_t)c.stream)) {
    return rc;
  }
  c.job_part[j] = c.P->part[j];
  c.job_dW[j] = const_cast<float*>(gcv.weight);
  c.job_db[j] = const_cast<float*>(gcv.bias);
  if (defer && !add1 && !add2 && !dev_env("MURAL_TRAIN_NO_POOL_FOLD")) {
    *defer = BnApplyJob{dz, x, (int64_t)c.P->B * L, pre_relu, state, bn.weight, acc, nullptr, nullptr, nullptr, const_cast<float*>(gbn.weight),
                        const_cast<float*>(gbn.bias)};
    return MURAL_OK;
  }
  return cl_bn_bwd_apply(dz, x, (int64_t)c.P->B * L, pre_relu, state, bn.weight, acc, add1, add2, dx, const_cast<float*>(gbn.weight),
                         const_cast<float*>(gbn.bias), (hipStream_t)c.stream);
}

// d_out: gradient arriving at the stage output (kept intact); d_in: receives the gradient of the stage input; tmp: 3 buffers
// conv backward of one layer on the wave kernels with the BatchNorm-backward apply of the layer behind it folded into its staging
int conv_b_fold(Ctx& c, const float* dy, const float* x, int L, const float* state, const MuralBN& bn, const MuralAffine& cv, double* acc,
                const MuralAffine& gcv, float* dz, const ConvBwdFold& fold, int pre_relu = 1) {
  const int j = c.njobs++;
  MURAL_REQUIRE(j < 20, "internal: more conv layers than partial-row regions");
  ConvBwdJob job{dy, x, cv.weight, wfrag_of(c, cv.weight, 1), c.P->B, L, state, bn.weight, pre_relu, dz, acc, c.P->part[j], 0, fold};
  if (int rc = cw_conv32_bwd_jobs(&job, 1, (hipStream_t)c.stream)) return rc;
  c.job_part[j] = c.P->part[j];
  c.job_nrow[j] = job.nrow;
  c.job_dW[j] = const_cast<float*>(gcv.weight);
  c.job_db[j] = const_cast<float*>(gcv.bias);
  return MURAL_OK;
}

// defer != nullptr: the caller's next kernel makes the stage's input gradient itself (the first layer's backward, FirstFold) -- on the
// folded path the stage's last BatchNorm-backward apply is then described in *defer instead of launched (defer->dz stays nullptr where
// the apply did run and d_in holds the gradient)
int stage_b(Ctx& c, const MuralResBlock* rb, const MuralResBlock* grb, const float* x_in, int L, StageBufs& s, const float* d_out,
            float* d_in, float* const* tmp, FirstFold* defer = nullptr) {
  float *dz = tmp[0], *ga = tmp[1], *gb = tmp[2];
  if (defer) defer->dz = nullptr;
  if (use_wave_conv(L, 0, false, 0) && !dev_env("MURAL_TRAIN_NO_FOLD")) {
    // Three of the four BatchNorm-backward applies of the stage never run as passes of their own: the conv backward of the layer in
    // front makes its dy from (dz, saved input, sums) of the layer behind while it stages it (conv32_wave.hip, FOLD) -- a read of two
    // tensors instead of one there against a pass of two reads and a write here.  Only d x1, which two consumers need, is also
    // written out (by the launch that makes it); the stage's input gradient keeps its own pass (it adds two residual gradients and
    // feeds another kind of kernel).
    float *A = dz, *Bz = ga;
    auto gbn = [](const MuralBN& g) { return std::pair<float*, float*>(const_cast<float*>(g.weight), const_cast<float*>(g.bias)); };
    // layer 3: y = conv(BN(relu(h2))) + x1 + x_in; dy = d_out as it is
    if (int rc = conv_b_fold(c, d_out, s.t[2], L, s.state[3], rb[1].bn2, rb[1].conv2, s.acc_b[3], grb[1].conv2, A, ConvBwdFold{})) return rc;
    // layer 2: dy = d h2 = apply of layer 3
    ConvBwdFold f2{A, s.t[2], s.state[3], rb[1].bn2.weight, s.acc_b[3], 1, gbn(grb[1].bn2).first, gbn(grb[1].bn2).second, nullptr, nullptr};
    if (int rc = conv_b_fold(c, nullptr, s.t[1], L, s.state[2], rb[1].bn1, rb[1].conv1, s.acc_b[2], grb[1].conv1, Bz, f2)) return rc;
    // layer 1: dy = d x1 (total) = apply of layer 2 + d_out, also needed by the stage's last apply -- which adds d_out once more (the
    // outer skip): the launch writes gb = dy + d_out, ONE tensor for that apply to read instead of two
    ConvBwdFold f1{Bz, s.t[1], s.state[2], rb[1].bn1.weight, s.acc_b[2], 1, gbn(grb[1].bn1).first, gbn(grb[1].bn1).second, d_out, gb};
    f1.dy_out_plus_add1 = 1;
    if (int rc = conv_b_fold(c, nullptr, s.t[0], L, s.state[1], rb[0].bn2, rb[0].conv2, s.acc_b[1], grb[0].conv2, A, f1)) return rc;
    // layer 0: dy = d h = apply of layer 1
    ConvBwdFold f0{A, s.t[0], s.state[1], rb[0].bn2.weight, s.acc_b[1], 1, gbn(grb[0].bn2).first, gbn(grb[0].bn2).second, nullptr, nullptr};
    if (int rc = conv_b_fold(c, nullptr, x_in, L, s.state[0], rb[0].bn1, rb[0].conv1, s.acc_b[0], grb[0].conv1, Bz, f0)) return rc;
    if (defer && !dev_env("MURAL_TRAIN_NO_FIRST_FOLD")) {
      *defer = FirstFold{Bz, x_in, gb, nullptr, s.state[0], rb[0].bn1.weight, s.acc_b[0], (double)c.P->B * L, const_cast<float*>(grb[0].bn1.weight),
                         const_cast<float*>(grb[0].bn1.bias)};
      return MURAL_OK;
    }
    return cl_bn_bwd_apply(Bz, x_in, (int64_t)c.P->B * L, 1, s.state[0], rb[0].bn1.weight, s.acc_b[0], gb, nullptr, d_in,
                           const_cast<float*>(grb[0].bn1.weight), const_cast<float*>(grb[0].bn1.bias), (hipStream_t)c.stream);
  }
  // layer 3: y = conv(BN(relu(h2))) + x1 + x_in
  if (int rc = bnconv_b(c, d_out, s.t[2], L, 1, s.state[3], rb[1].bn2, rb[1].conv2, s.acc_b[3], nullptr, nullptr, grb[1].bn2, grb[1].conv2,
                        dz, ga)) return rc;                                       // ga = d h2
  // layer 2: h2 = conv(BN(relu(x1))); x1 also feeds layer 3's residual
  if (int rc = bnconv_b(c, ga, s.t[1], L, 1, s.state[2], rb[1].bn1, rb[1].conv1, s.acc_b[2], d_out, nullptr, grb[1].bn1, grb[1].conv1, dz,
                        gb)) return rc;                                           // gb = d x1 (total)
  // layer 1: x1 = conv(BN(relu(h))) + x_in
  if (int rc = bnconv_b(c, gb, s.t[0], L, 1, s.state[1], rb[0].bn2, rb[0].conv2, s.acc_b[1], nullptr, nullptr, grb[0].bn2, grb[0].conv2, dz,
                        ga)) return rc;                                           // ga = d h
  // layer 0: h = conv(BN(relu(x_in))); x_in also feeds x1 (residual) and the stage output (outer skip)
  return bnconv_b(c, ga, x_in, L, 1, s.state[0], rb[0].bn1, rb[0].conv1, s.acc_b[0], gb, d_out, grb[0].bn1, grb[0].conv1, dz, d_in);
}

int dropout_b(Ctx& c, const float* dy, int64_t n, float p, uint64_t seed, const uint64_t* seed_dev, float* dx, const float** out) {
  if (p <= 0.f) { *out = dy; return MURAL_OK; }
  *out = dx;
  return mural_op_dropout(dy, n, p, seed, seed_dev, dx, c.stream);
}

int tower_b(Ctx& c, int t, const MuralTower& T, const MuralTower& G, const float* dlogits, float drop_p, uint64_t seed,
            const uint64_t* seed_dev) {
  Plan& P = *c.P;
  const TowerGeo& g = P.geo[t];
  TowerBufs& b = P.tw[t];
  const int B = P.B;
  float* const* gs = t == 0 ? P.g_mid : P.g;
  float *g0 = gs[0], *g1 = gs[1], *g2 = gs[2], *g3 = gs[3];
  const float* fin = b.fd;
  hipStream_t st = (hipStream_t)c.stream;
  const bool head_fused = head_train_fused_ok(P.nc);
  if (head_fused) {      // (the Linear's weight gradient, which nothing in the step consumes, is launched behind the tower's chain)
    if (int rc = head_train_bwd(dlogits, T.fc.weight, P.nc, B, g.L[2], b.feat, b.fc_state, T.fc_bn.weight, drop_p, seed, seed_dev, g0, b.acc_fc_b,
                                b.argg, b.c3, g1, const_cast<float*>(G.fc_bn.weight), const_cast<float*>(G.fc_bn.bias), st)) return rc;
  } else {
  // Linear -> Dropout -> BatchNorm1d on (B, 32)
  if (int rc = mural_op_linear_bwd(dlogits, fin, T.fc.weight, B, TR_C, P.nc, g0, const_cast<float*>(G.fc.weight),
                                   const_cast<float*>(G.fc.bias), c.stream)) return rc;
  const float* d;
  if (int rc = dropout_b(c, g0, (int64_t)B * TR_C, drop_p, seed, seed_dev, g1, &d)) return rc;
  if (int rc = mural_op_bn_backward(d, b.feat, B, TR_C, 1, 0, b.fc_state + 2 * TR_C, b.fc_state + 3 * TR_C, T.fc_bn.weight, b.acc_fc_b, 0,
                                    nullptr, nullptr, g2, const_cast<float*>(G.fc_bn.weight), const_cast<float*>(G.fc_bn.bias), c.stream))
    return rc;                                                                      // g2 = d feat
  // global max, ReLU of conv3
  if (int rc = cl_gmax_relu_bwd(g2, b.argg, b.c3, B, g.L[2], g1, st)) return rc;
  }
  // the BatchNorm-backward applies in front of the two pools never run as passes of their own on the wave kernels: the pool's backward
  // makes its gradient per element from (dz, pooled input, sums) (PoolBwdJob: fold)
  const bool wave3 = use_wave_conv(g.L[2], 0, false, 0), wave2 = use_wave_conv(g.L[1], 0, false, 0);
  BnApplyJob ap3{}, ap2{};
  if (int rc = bnconv_b(c, g1, b.p3, g.L[2], 0, b.state_c3, T.bn_out, T.conv_out, b.acc_c3_b, nullptr, nullptr, G.bn_out, G.conv_out, g0,
                        g2, wave3 ? &ap3 : nullptr)) return rc;                    // g2 = d p3 (unless deferred)
  if (ap3.dz) {
    if (int rc = cl_maxpool_bwd_fold(ap3, b.arg3, B, g.L[1], g.L[2], g.pk[2], g.ps[2], g.pp[2], g3, st)) return rc;
  } else if (int rc = cl_maxpool_bwd(g2, b.arg3, B, g.L[1], g.L[2], g.pk[2], g.ps[2], g.pp[2], g3, st)) {
    return rc;
  }
  float* tmp[3] = {g0, g1, g2};
  // second ResBlock stage: the gradient of its output sits in g3, g0..g2 are the stage's temporaries; the gradient of its input
  // lands in the forward's copy of the stage output, which no backward reads (the pooling behind it kept its arg-max) -- or, on the
  // wave kernels, is never materialised: conv_mid's backward makes it while staging (FOLD with the stage's two residual gradients)
  float* d_in3 = b.s3.t[3];
  FirstFold f3{};
  if (int rc = stage_b(c, T.rbs2, G.rbs2, b.x0b, g.L[1], b.s3, g3, d_in3, tmp, wave2 && !dev_env("MURAL_TRAIN_NO_MID_FOLD") ? &f3 : nullptr)) return rc;
  if (f3.dz) {
    // (stage_b's temporaries: dz = tmp[0] = g0, ga = tmp[1] = g1 (= f3.dz), gb = tmp[2] = g2 (= f3.add1); g3 = f3.add2: the conv's own
    // input gradient goes to g0, free again)
    const ConvBwdFold fold{f3.dz, f3.x, f3.state, f3.gamma, f3.acc, 1, f3.dgamma, f3.dbeta, f3.add1, nullptr, f3.add2};
    if (int rc = conv_b_fold(c, nullptr, b.p2, g.L[1], b.state_c2, T.bn_mid, T.conv_mid, b.acc_c2_b, G.conv_mid, g0, fold, 0)) return rc;
    ap2 = BnApplyJob{g0, b.p2, (int64_t)B * g.L[1], 0, b.state_c2, T.bn_mid.weight, b.acc_c2_b, nullptr, nullptr, nullptr,
                     const_cast<float*>(G.bn_mid.weight), const_cast<float*>(G.bn_mid.bias)};
    if (dev_env("MURAL_TRAIN_NO_POOL_FOLD")) {
      if (int rc = cl_bn_bwd_apply(g0, b.p2, (int64_t)B * g.L[1], 0, b.state_c2, T.bn_mid.weight, b.acc_c2_b, nullptr, nullptr, g1,
                                   const_cast<float*>(G.bn_mid.weight), const_cast<float*>(G.bn_mid.bias), st)) return rc;
      ap2.dz = nullptr;
    }
  } else if (int rc = bnconv_b(c, d_in3, b.p2, g.L[1], 0, b.state_c2, T.bn_mid, T.conv_mid, b.acc_c2_b, nullptr, nullptr, G.bn_mid, G.conv_mid,
                               g0, g1, wave2 ? &ap2 : nullptr)) {
    return rc;                                                                     // g1 = d p2 (unless deferred)
  }
  if (ap2.dz) {
    if (int rc = cl_maxpool_bwd_fold(ap2, b.arg2, B, g.L[0], g.L[1], g.pk[1], g.ps[1], g.pp[1], g3, st)) return rc;
  } else if (int rc = cl_maxpool_bwd(g1, b.arg2, B, g.L[0], g.L[1], g.pk[1], g.ps[1], g.pp[1], g3, st)) {
    return rc;
  }
  float* d_in2 = b.s2.t[3];
  // the first ResBlock stage's last BatchNorm-backward apply (four reads and a write of the stage tensor) never runs as a pass of its own:
  // the first layer's backward makes its pooled gradient from the apply's operands element by element (FirstFold, snv_stage1.hip)
  FirstFold fold{};
  if (int rc = stage_b(c, T.rbs1, G.rbs1, b.x0, g.L[0], b.s2, g3, d_in2, tmp, &fold)) return rc;
  if (head_fused)
    if (int rc = head_train_wgrad(dlogits, fin, B, P.nc, const_cast<float*>(G.fc.weight), const_cast<float*>(G.fc.bias), st)) return rc;
  return train_first_bwd_cl(fold.dz ? nullptr : d_in2, b.arg1, P.sym, B, P.Lwin, g.col0, g.L1, g.pk[0], g.ps[0], g.pp[0], b.tab, T.conv_in.weight,
                            P.first_scratch[t], const_cast<float*>(G.conv_in.weight), const_cast<float*>(G.conv_in.bias),
                            const_cast<float*>(G.bn_in.weight), const_cast<float*>(G.bn_in.bias), fold.dz ? &fold : nullptr, st);
}

int local_b(Ctx& c, const int64_t* cat, const float* dlogits, const float* drop, const uint64_t* seeds, const uint64_t* seed_dev) {
  Plan& P = *c.P;
  const MuralSnvShape& sh = *c.sh;
  const MuralLocal& L = c.p->local;
  const MuralLocal& G = c.gr->local;
  LocalBufs& l = P.loc;
  const int B = P.B, in1 = 5 * sh.local_cols, h[2] = {sh.hidden1, sh.hidden2};
  float *g0 = P.g_loc[0], *g1 = P.g_loc[1], *g2 = P.g_loc[2];
  if (local_train_fused_ok(in1, h[0], h[1], P.nc, sh.emb_rows, B)) {
    const int dims[4] = {in1, h[0], h[1], P.nc};
    const float* W[3] = {L.lin[0].weight, L.lin[1].weight, L.out.weight};
    const float* gamma[2] = {L.bn[0].weight, L.bn[1].weight};
    const float* state[2] = {l.bn_state[0], l.bn_state[1]};
    const float* xt[3] = {l.emb_do, l.dout[0], l.dout[1]};
    const float* lin[2] = {l.lin[0], l.lin[1]};
    float* dd[2] = {g2, g0};                    // gradients of the BatchNorm outputs behind the dropout masks: [B][h1], [B][h2]
    float* gl[2] = {l.bn_out[0], g1};           // gradients of the Linear outputs (for the weight gradients): [B][h1], [B][h2]
    float* dW[3] = {const_cast<float*>(G.lin[0].weight), const_cast<float*>(G.lin[1].weight), const_cast<float*>(G.out.weight)};
    float* db[3] = {const_cast<float*>(G.lin[0].bias), const_cast<float*>(G.lin[1].bias), const_cast<float*>(G.out.bias)};
    float* dgamma[2] = {const_cast<float*>(G.bn[0].weight), const_cast<float*>(G.bn[1].weight)};
    float* dbeta[2] = {const_cast<float*>(G.bn[0].bias), const_cast<float*>(G.bn[1].bias)};
    return local_train_bwd(cat, sh.local_cols, sh.emb_rows, B, dims, dlogits, W, gamma, state, l.acc_b, drop, seeds, seed_dev, xt, lin, dd, gl,
                           dW, db, dgamma, dbeta, const_cast<float*>(G.emb), (hipStream_t)c.stream);
  }
  const float* x_last = l.dout[1];
  if (int rc = mural_op_linear_bwd(dlogits, x_last, L.out.weight, B, h[1], P.nc, g0, const_cast<float*>(G.out.weight),
                                   const_cast<float*>(G.out.bias), c.stream)) return rc;
  const float* d = g0;
  for (int i = 1; i >= 0; --i) {
    const float* dd;
    if (int rc = dropout_b(c, d, (int64_t)B * h[i], drop[1 + i], seeds[1 + i], seed_dev, g1, &dd)) return rc;
    float* st = l.bn_state[i];
    if (int rc = mural_op_bn_backward(dd, l.lin[i], B, h[i], 1, 1, st + 2 * h[i], st + 3 * h[i], L.bn[i].weight, l.acc_b[i], 0, nullptr,
                                      nullptr, g2, const_cast<float*>(G.bn[i].weight), const_cast<float*>(G.bn[i].bias), c.stream)) return rc;
    const int in = i == 0 ? in1 : h[0];
    const float* xin = i == 0 ? (drop[0] > 0.f ? l.emb_do : l.emb) : l.dout[0];
    if (int rc = mural_op_linear_bwd(g2, xin, L.lin[i].weight, B, in, h[i], g0, const_cast<float*>(G.lin[i].weight),
                                     const_cast<float*>(G.lin[i].bias), c.stream)) return rc;
    d = g0;
  }
  const float* de;
  if (int rc = dropout_b(c, d, (int64_t)B * in1, drop[0], seeds[0], seed_dev, g1, &de)) return rc;
  MURAL_HIP_CHECK(hipMemsetAsync(const_cast<float*>(G.emb), 0, (size_t)sh.emb_rows * 5 * 4, (hipStream_t)c.stream));
  return mural_op_embedding_bwd(cat, de, B, sh.local_cols, sh.emb_rows, const_cast<float*>(G.emb), c.stream);
}

int check_shape(const MuralSnvShape* sh) {
  MURAL_REQUIRE(sh, "shape is NULL");
  MURAL_REQUIRE(sh->model_no >= 0 && sh->model_no <= 2, "model_no for snv must be one of [0, 1, 2], got %d", sh->model_no);
  MURAL_REQUIRE(sh->n_class >= 1 && sh->n_class <= SNV_MAXCLASS, "n_class must be in [1,%d], got %d", SNV_MAXCLASS, sh->n_class);
  if (sh->model_no != 0) {
    MURAL_REQUIRE(sh->channels == SNV_C && sh->ksize == SNV_K,
                  "the fused training step is built for CNN_out_channels=32, CNN_kernel_size=3 (got %d, %d)", sh->channels, sh->ksize);
    MURAL_REQUIRE(sh->distal_len > 200, "Error: distal seq len must be >200");
  }
  if (sh->model_no != 1) MURAL_REQUIRE(sh->local_cols >= 1 && sh->emb_rows >= 2 && sh->hidden1 >= 1 && sh->hidden2 >= 1, "bad local-branch shape");
  return MURAL_OK;
}

}  // namespace

extern "C" size_t mural_snv_train_workspace_bytes(const MuralSnvShape* shape, int64_t B) {
  if (check_shape(shape) || B <= 0) return 0;
  Plan P;
  if (make_plan(*shape, B, nullptr, &P)) return 0;
  return P.total;
}

extern "C" int mural_snv_train_forward(const MuralSnvShape* shape, const MuralSnvParams* params, const int64_t* cat_x,
                                       const float* distal_x, const uint8_t* symbols, int64_t B, const float* dropout_p, const uint64_t* seeds,
                                       const uint64_t* seed_dev, float momentum, float* out, void* workspace, size_t workspace_bytes,
                                       int32_t* status, void* stream) {
  if (int rc = check_shape(shape)) return rc;
  MURAL_REQUIRE(params && out && dropout_p && seeds, "NULL argument");
  MURAL_REQUIRE(B >= 2, "a batch-statistics BatchNorm needs at least two rows (training.py:415 skips batches of one)");
  Plan P;
  if (int rc = make_plan(*shape, B, workspace, &P)) return rc;
  if (!workspace || workspace_bytes < P.total) {
    set_error("workspace too small: need %zu bytes, got %zu", P.total, workspace_bytes);
    return MURAL_E_WORKSPACE;
  }
  Ctx c{shape, params, nullptr, &P, momentum, stream};
  MURAL_HIP_CHECK(hipMemsetAsync(P.acc_begin, 0, P.acc_bytes, (hipStream_t)stream));
  const int m = shape->model_no;
  if (m != 1) MURAL_REQUIRE(cat_x, "cat_x is NULL");
  if (m == 0) {        // raw logits, model_snv.py:93
    if (int rc = local_f(c, cat_x, dropout_p, seeds, seed_dev)) return rc;
    MURAL_HIP_CHECK(hipMemcpyAsync(out, P.loc.logits, (size_t)B * shape->n_class * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MURAL_OK;
  }
  MURAL_REQUIRE(distal_x || symbols, "distal_x and symbols are both NULL");
  // three streams: large tower (caller's) | mid tower | local branch.  The local branch needs nothing of what the towers wait for (symbol
  // histograms, first-layer tables, weight fragments: ~35 us of small launches in a row), so it starts beside them
  SideStreamHold ss;      // holds the device's side streams until this call has joined them again
  if (int rc = ss.acquire()) return rc;
  static const int order = dev_env("MURAL_TRAIN_ORDER") ? atoi(dev_env("MURAL_TRAIN_ORDER")) : 0;      // experiment: 2 = local branch after the fork of the towers
  int rc_loc = MURAL_OK;
  if (order != 2 && m == 2) {
    if (int rc = ss->fork2((hipStream_t)stream)) return rc;
    c.stream = ss->side2;
    rc_loc = local_f(c, cat_x, dropout_p, seeds, seed_dev);
    c.stream = stream;
  }
  auto prepare = [&]() -> int {
  if (symbols) {
    MURAL_HIP_CHECK(hipMemcpyAsync(P.sym, symbols, (size_t)B * shape->distal_len, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  } else if (int rc = mural_op_dense_to_symbols(distal_x, B, shape->distal_len, P.sym, status, stream)) {
    return rc;
  }
  conv_weight_table(c);
  if (int rc = cw_wfrag_build(c.conv_w, 20, P.wfrag, (hipStream_t)stream)) return rc;     // (weights are the same in the backward of this step)
  if (!dev_env("MURAL_TRAIN_FIRST_SEPARATE")) {      // both towers' symbol histograms in one launch, both table sets in one launch
    const int col0[2] = {P.geo[1].col0, P.geo[0].col0}, L1[2] = {P.geo[1].L1, P.geo[0].L1};
    const MuralTower* T[2] = {&params->large, &params->mid};
    const float* gamma[2] = {T[0]->bn_in.weight, T[1]->bn_in.weight};
    const float* beta[2] = {T[0]->bn_in.bias, T[1]->bn_in.bias};
    const float* W[2] = {T[0]->conv_in.weight, T[1]->conv_in.weight};
    const float* bias[2] = {T[0]->conv_in.bias, T[1]->conv_in.bias};
    float* rmean[2] = {const_cast<float*>(T[0]->bn_in.running_mean), const_cast<float*>(T[1]->bn_in.running_mean)};
    float* rvar[2] = {const_cast<float*>(T[0]->bn_in.running_var), const_cast<float*>(T[1]->bn_in.running_var)};
    unsigned long long* counts[2] = {P.tw[1].counts, P.tw[0].counts};
    float* tab[2] = {P.tw[1].tab, P.tw[0].tab};
    if (P.geo[1].col0 == 0 && P.geo[1].L1 == P.Lwin) {
      if (int rc = train_first_prepare2(P.sym, B, P.Lwin, col0, L1, gamma, beta, W, bias, rmean, rvar, counts, tab, EPS, momentum, (hipStream_t)stream))
        return rc;
      c.first_prepared = true;
    }
  }
    return MURAL_OK;
  };
  int rc_prep = rc_loc ? MURAL_OK : prepare();
  // The large tower is the critical path, so it is enqueued first: the ~35 launches of the mid tower would otherwise hold its first
  // kernel back by their enqueue time
  int rc_large = MURAL_OK, rc_mid = MURAL_OK;
  if (!rc_prep && !rc_loc) {
    if (int rc = ss->fork((hipStream_t)stream, order == 2)) rc_prep = rc;
  }
  if (!rc_prep && !rc_loc) {
    rc_large = tower_f(c, 1, params->large, dropout_p[4], seeds[4], seed_dev);
    c.stream = ss->side;
    rc_mid = rc_large ? MURAL_OK : tower_f(c, 0, params->mid, dropout_p[3], seeds[3], seed_dev);
    c.stream = ss->side2;
    if (order == 2 && m == 2 && !rc_mid && !rc_large) rc_mid = local_f(c, cat_x, dropout_p, seeds, seed_dev);
    c.stream = stream;
  }
  if (int rc = ss->join((hipStream_t)stream, true)) return rc;     // also on an error: the side streams must not stay forked
  if (rc_loc) return rc_loc;
  if (rc_prep) return rc_prep;
  if (rc_mid) return rc_mid;
  if (rc_large) return rc_large;
  return mural_op_head_fwd(m == 2 ? P.loc.logits : nullptr, P.tw[0].logits, P.tw[1].logits, B, shape->n_class, out, stream);
}

extern "C" int mural_snv_train_backward(const MuralSnvShape* shape, const MuralSnvParams* params, const MuralSnvParams* grads,
                                        const int64_t* cat_x, const float* dout, int64_t B, const float* dropout_p,
                                        const uint64_t* seeds, const uint64_t* seed_dev, void* workspace, size_t workspace_bytes,
                                        void* stream) {
  if (int rc = check_shape(shape)) return rc;
  MURAL_REQUIRE(params && grads && dout && dropout_p && seeds, "NULL argument");
  Plan P;
  if (int rc = make_plan(*shape, B, workspace, &P)) return rc;
  if (!workspace || workspace_bytes < P.total) {
    set_error("workspace too small: need %zu bytes, got %zu", P.total, workspace_bytes);
    return MURAL_E_WORKSPACE;
  }
  Ctx c{shape, params, grads, &P, 0.f, stream};
  const int m = shape->model_no, nc = shape->n_class;
  if (m != 0) conv_weight_table(c);
  if (m == 0) return local_b(c, cat_x, dout, dropout_p, seeds, seed_dev);
  if (int rc = mural_op_head_bwd(m == 2 ? P.loc.logits : nullptr, P.tw[0].logits, P.tw[1].logits, dout, B, nc, m == 2 ? P.dlogit[0] : nullptr,
                                 P.dlogit[1], P.dlogit[2], stream)) return rc;
  SideStreamHold ss;      // holds the device's side streams until this call has joined them again
  if (int rc = ss.acquire()) return rc;
  if (int rc = ss->fork((hipStream_t)stream, true)) return rc;
  static const int order = dev_env("MURAL_TRAIN_ORDER") ? atoi(dev_env("MURAL_TRAIN_ORDER")) : 0;      // experiment: 1 = local branch first
  int rc_loc = MURAL_OK;
  if (order == 1 && m == 2) {
    c.stream = ss->side2;
    rc_loc = local_b(c, cat_x, P.dlogit[0], dropout_p, seeds, seed_dev);
    c.stream = stream;
  }
  int rc_large = tower_b(c, 1, params->large, grads->large, P.dlogit[2], dropout_p[4], seeds[4], seed_dev);   // critical path first
  c.stream = ss->side;
  int rc_mid = rc_large ? MURAL_OK : tower_b(c, 0, params->mid, grads->mid, P.dlogit[1], dropout_p[3], seeds[3], seed_dev);
  c.stream = ss->side2;
  if (order != 1 && m == 2 && !rc_mid && !rc_large) rc_mid = local_b(c, cat_x, P.dlogit[0], dropout_p, seeds, seed_dev);
  if (!rc_mid) rc_mid = rc_loc;
  c.stream = stream;
  if (int rc = ss->join((hipStream_t)stream, true)) return rc;
  if (rc_mid) return rc_mid;
  if (rc_large) return rc_large;
  return train_reduce_parts(c.job_part, c.job_nrow, c.job_dW, c.job_db, c.njobs, (hipStream_t)stream);
}

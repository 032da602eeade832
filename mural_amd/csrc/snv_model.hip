// Host side of the SNV path: eval-mode folding of the reference parameters into the device layout of
// snv.h, tile geometry, workspace carving, and the extern "C" forward entry points.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "snv.h"

namespace mural {
int launch_snv_towers(const MuralSnvModel* m, const SnvFwdArgs& a, size_t lds_bytes, hipStream_t stream);
int launch_snv_tower_wave_jobs(const SnvFwdArgs* jobs, const size_t* lds_bytes, int n, hipStream_t stream);   // snv_tower_wave.hip
int launch_snv_stage1(const Stage1Args& a, bool packed, size_t lds_bytes, hipStream_t stream);
size_t plan_wave_geometry(SnvFwdArgs& a, int Lwin, int Pw, int n_class, int tower, int phase);   // snv_tower_wave.hip
int launch_snv_local(const LocalDev& L, const int64_t* cat, int64_t n, float* out, hipStream_t stream);
int launch_dense_to_symbols(const float* x, int64_t n, int L, uint8_t* sym, int32_t* status, hipStream_t stream, int bad_code = -1);
bool stage1_small_batch(int64_t n);
bool local_mfma_plan(const LocalDev& L, LocalMfmaDims* d, size_t* lds_bytes);

namespace {

// forward-strand channel values (A,C,G,T) of the 15 IUPAC symbols, MuRaL/data/preprocessing.py:758-772
const double kSymVec[15][4] = {
    {1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}, {.25, .25, .25, .25},
    {.5, 0, .5, 0}, {0, .5, 0, .5}, {.5, .5, 0, 0}, {0, .5, .5, 0}, {.5, 0, 0, .5}, {0, 0, .5, .5},
    {0, 1. / 3, 1. / 3, 1. / 3}, {1. / 3, 0, 1. / 3, 1. / 3}, {1. / 3, 1. / 3, 0, 1. / 3}, {1. / 3, 1. / 3, 1. / 3, 0}};

struct Blob {
  std::vector<float> host;
  size_t alloc(size_t n) {
    size_t off = (host.size() + 63) & ~size_t(63);  // 256-byte aligned sections
    host.resize(off + n, 0.f);
    return off;
  }
};

void bn_affine(const MuralBN& bn, int C, float eps, double* s, double* t) {
  for (int c = 0; c < C; ++c) {
    // the one-hot symbol table is fp32 on the reference side: BN eval is (x-mean)/sqrt(var+eps)*w+b
    const double inv = 1.0 / std::sqrt((double)bn.running_var[c] + (double)eps);
    s[c] = (double)bn.weight[c] * inv;
    t[c] = (double)bn.bias[c] - (double)bn.running_mean[c] * s[c];
  }
}

bool bn_ok(const MuralBN& b) { return b.weight && b.bias && b.running_mean && b.running_var; }
bool aff_ok(const MuralAffine& a) { return a.weight && a.bias; }

// W[co][ci][t] (PyTorch Conv1d layout) -> MFMA A fragments [mblock][kstep][lane]
// kstep s = 8t + 4h + q feeds input channel ci = 16h + 4*(lane>>4) + q, output channel 16*mb + (lane&15)
void pack_wfrag(const float* W, float* dst) {
  for (int mb = 0; mb < 2; ++mb)
    for (int s = 0; s < SNV_KSTEPS; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int t = s / 8, h = (s % 8) / 4, q = s % 4;
        const int ci = 16 * h + 4 * (lane >> 4) + q;
        const int co = 16 * mb + (lane & 15);
        dst[(mb * SNV_KSTEPS + s) * 64 + lane] = W[(co * SNV_C + ci) * SNV_K + t];
      }
}

struct TowerOff { size_t lut4, lut, taps, bias0, wfrag, wfrag4, bias, post_s, post_t, ex_s, ex_t, fc_w, fc_b; };

int fold_tower(const MuralTower& T, const MuralSnvShape& sh, Blob& B, TowerOff& o) {
  const int C = SNV_C;
  MURAL_REQUIRE(bn_ok(T.bn_in) && aff_ok(T.conv_in) && bn_ok(T.bn_mid) && aff_ok(T.conv_mid) && bn_ok(T.bn_out) &&
                    aff_ok(T.conv_out) && bn_ok(T.fc_bn) && aff_ok(T.fc),
                "tower parameter pointer is NULL");
  for (int i = 0; i < 2; ++i) {
    MURAL_REQUIRE(bn_ok(T.rbs1[i].bn1) && aff_ok(T.rbs1[i].conv1) && bn_ok(T.rbs1[i].bn2) && aff_ok(T.rbs1[i].conv2) &&
                      bn_ok(T.rbs2[i].bn1) && aff_ok(T.rbs2[i].conv1) && bn_ok(T.rbs2[i].bn2) && aff_ok(T.rbs2[i].conv2),
                  "residual block parameter pointer is NULL");
  }
  o.lut = B.alloc(SNV_LUTBLK);          // lut | taps | bias0 in one block: the stage-1 kernel stages it with one copy
  o.taps = o.lut + SNV_LUT;
  o.bias0 = o.taps + SNV_TAPS;
  o.lut4 = B.alloc(SNV_LUT4);
  o.wfrag = B.alloc((size_t)SNV_NLAYER * SNV_WFRAG);
  o.wfrag4 = B.alloc((size_t)SNV_NLAYER * SNV_WFRAG);
  o.bias = B.alloc(SNV_NLAYER * C);
  o.post_s = B.alloc(SNV_NLAYER * C);
  o.post_t = B.alloc(SNV_NLAYER * C);
  o.ex_s = B.alloc(EX_COUNT * C);
  o.ex_t = B.alloc(EX_COUNT * C);
  o.fc_w = B.alloc((size_t)sh.n_class * C);
  o.fc_b = B.alloc(sh.n_class);

  // ---- first layer: BN(4) -> Conv1d(4->32,k3,pad 1) as per-tap symbol tables and a 3-mer table.
  // zero padding is applied AFTER the BN (model_snv.py:350-353), so the PAD symbol contributes exactly 0.
  double s4[4], t4[4];
  bn_affine(T.bn_in, 4, sh.bn_eps, s4, t4);
  std::vector<double> tap(3 * N_SYM * C, 0.0);
  for (int t = 0; t < 3; ++t)
    for (int sym = 0; sym < 15; ++sym)
      for (int co = 0; co < C; ++co) {
        double acc = 0.0;
        for (int ci = 0; ci < 4; ++ci) {
          // BN output as the reference computes it in fp32, then the conv product
          const float bnv = (float)(s4[ci] * kSymVec[sym][ci] + t4[ci]);
          acc += (double)T.conv_in.weight[(co * 4 + ci) * 3 + t] * (double)bnv;
        }
        tap[(t * N_SYM + sym) * C + co] = acc;
      }
  for (size_t i = 0; i < tap.size(); ++i) B.host[o.taps + i] = (float)tap[i];
  for (int co = 0; co < C; ++co) B.host[o.bias0 + co] = T.conv_in.bias[co];
  for (int l = 0; l < 5; ++l)
    for (int c = 0; c < 5; ++c)
      for (int r = 0; r < 5; ++r)
        for (int co = 0; co < C; ++co)
          B.host[o.lut + ((l * 25 + c * 5 + r) * C + co)] =
              (float)((double)T.conv_in.bias[co] + tap[(0 * N_SYM + l) * C + co] + tap[(1 * N_SYM + c) * C + co] +
                      tap[(2 * N_SYM + r) * C + co]);

  // pair table of the stage-1 kernel: two adjacent first-layer columns share three of their four bases, and the pooled value only
  // needs their maximum -- one 128-byte row read instead of two for windows of plain bases (exactly the maximum of the two rows)
  for (int a4 = 0; a4 < 4; ++a4)
    for (int b4 = 0; b4 < 4; ++b4)
      for (int c4 = 0; c4 < 4; ++c4)
        for (int d4 = 0; d4 < 4; ++d4)
          for (int co = 0; co < C; ++co) {
            const float u = B.host[o.lut + (size_t)(25 * a4 + 5 * b4 + c4) * C + co], v = B.host[o.lut + (size_t)(25 * b4 + 5 * c4 + d4) * C + co];
            B.host[o.lut4 + (size_t)(64 * a4 + 16 * b4 + 4 * c4 + d4) * C + co] = u > v ? u : v;
          }

  // ---- 32->32 convs
  const MuralAffine* convs[SNV_NLAYER] = {&T.rbs1[0].conv1, &T.rbs1[0].conv2, &T.rbs1[1].conv1, &T.rbs1[1].conv2, &T.conv_mid,
                                          &T.rbs2[0].conv1, &T.rbs2[0].conv2, &T.rbs2[1].conv1, &T.rbs2[1].conv2, &T.conv_out};
  // BN applied to relu(output of layer i) to form the next conv's input (nullptr: output kept raw)
  const MuralBN* post[SNV_NLAYER] = {&T.rbs1[0].bn2, &T.rbs1[1].bn1, &T.rbs1[1].bn2, nullptr, &T.rbs2[0].bn1,
                                     &T.rbs2[0].bn2, &T.rbs2[1].bn1, &T.rbs2[1].bn2, nullptr, nullptr};
  double s[SNV_C], t[SNV_C];
  for (int l = 0; l < SNV_NLAYER; ++l) {
    pack_wfrag(convs[l]->weight, &B.host[o.wfrag + (size_t)l * SNV_WFRAG]);
    // the same fragments with four consecutive k-steps of a lane side by side (one 16-byte load per lane brings them):
    // wfrag4[((mb * 6 + g) * 64 + lane) * 4 + j] = wfrag[(mb * 24 + 4 g + j) * 64 + lane]   (snv_tower_wave.hip)
    for (int mb = 0; mb < 2; ++mb)
      for (int g4 = 0; g4 < SNV_KSTEPS / 4; ++g4)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 4; ++j)
            B.host[o.wfrag4 + (size_t)l * SNV_WFRAG + ((size_t)(mb * (SNV_KSTEPS / 4) + g4) * 64 + lane) * 4 + j] =
                B.host[o.wfrag + (size_t)l * SNV_WFRAG + (size_t)(mb * SNV_KSTEPS + 4 * g4 + j) * 64 + lane];
    for (int c = 0; c < C; ++c) B.host[o.bias + l * C + c] = convs[l]->bias[c];
    if (post[l]) {
      bn_affine(*post[l], C, sh.bn_eps, s, t);
      for (int c = 0; c < C; ++c) {
        B.host[o.post_s + l * C + c] = (float)s[c];
        B.host[o.post_t + l * C + c] = (float)t[c];
      }
    } else {
      for (int c = 0; c < C; ++c) B.host[o.post_s + l * C + c] = 1.f;
    }
  }
  const MuralBN* ex[EX_COUNT] = {&T.rbs1[0].bn1, &T.bn_mid, &T.bn_out, &T.fc_bn};
  for (int e = 0; e < EX_COUNT; ++e) {
    bn_affine(*ex[e], C, sh.bn_eps, s, t);
    for (int c = 0; c < C; ++c) {
      B.host[o.ex_s + e * C + c] = (float)s[c];
      B.host[o.ex_t + e * C + c] = (float)t[c];
    }
  }
  // distal_fc = BN -> Dropout -> Linear on the global-max features: the BN folds exactly into the Linear
  bn_affine(T.fc_bn, C, sh.bn_eps, s, t);
  for (int k = 0; k < sh.n_class; ++k) {
    double b = T.fc.bias[k];
    for (int c = 0; c < C; ++c) {
      const double w = T.fc.weight[k * C + c];
      B.host[o.fc_w + k * C + c] = (float)(w * s[c]);
      b += w * t[c];
    }
    B.host[o.fc_b + k] = (float)b;
  }
  return MURAL_OK;
}

struct LocalOff { size_t emb, w1t, b1, w2t, b2, w3t, b3, frag, frag_floats; };

int fold_local(const MuralLocal& Lc, const MuralSnvShape& sh, Blob& B, LocalOff& o) {
  MURAL_REQUIRE(Lc.emb && aff_ok(Lc.lin[0]) && aff_ok(Lc.lin[1]) && bn_ok(Lc.bn[0]) && bn_ok(Lc.bn[1]) && aff_ok(Lc.out),
                "local-branch parameter pointer is NULL");
  const int in1 = 5 * sh.local_cols, h1 = sh.hidden1, h2 = sh.hidden2, nc = sh.n_class;
  const int xs = (in1 + 3) & ~3, h1s = (h1 + 3) & ~3, h2s = (h2 + 3) & ~3;
  o.emb = B.alloc((size_t)sh.emb_rows * 5);
  o.w1t = B.alloc((size_t)xs * h1);
  o.b1 = B.alloc(h1);
  o.w2t = B.alloc((size_t)h1s * h2);
  o.b2 = B.alloc(h2);
  o.w3t = B.alloc((size_t)h2s * nc);
  o.b3 = B.alloc(nc);
  for (int i = 0; i < sh.emb_rows * 5; ++i) B.host[o.emb + i] = Lc.emb[i];
  for (int h = 0; h < h1; ++h) {
    for (int k = 0; k < in1; ++k) B.host[o.w1t + (size_t)k * h1 + h] = Lc.lin[0].weight[(size_t)h * in1 + k];
    B.host[o.b1 + h] = Lc.lin[0].bias[h];
  }
  // order is Linear -> ReLU -> BN (model_snv.py:466-467): BN_i folds exactly into the next Linear
  std::vector<double> s(std::max(h1, h2)), t(std::max(h1, h2));
  bn_affine(Lc.bn[0], h1, sh.bn_eps, s.data(), t.data());
  for (int j = 0; j < h2; ++j) {
    double b = Lc.lin[1].bias[j];
    for (int h = 0; h < h1; ++h) {
      const double w = Lc.lin[1].weight[(size_t)j * h1 + h];
      B.host[o.w2t + (size_t)h * h2 + j] = (float)(w * s[h]);
      b += w * t[h];
    }
    B.host[o.b2 + j] = (float)b;
  }
  bn_affine(Lc.bn[1], h2, sh.bn_eps, s.data(), t.data());
  for (int k = 0; k < nc; ++k) {
    double b = Lc.out.bias[k];
    for (int j = 0; j < h2; ++j) {
      const double w = Lc.out.weight[(size_t)k * h2 + j];
      B.host[o.w3t + (size_t)j * nc + k] = (float)(w * s[j]);
      b += w * t[j];
    }
    B.host[o.b3 + k] = (float)b;
  }
  // the same three matrices in MFMA A-fragment order for snv_local_mlp_mfma (csrc/snv_local.hip):
  //   frag[((nb * J + j) * 64 + lane) * 4 + t] = W[f = 16 nb + lane % 16][k = 16 j + 4 (lane / 16) + t], zero outside
  const int K1p = (in1 + 15) & ~15, n1b = (h1 + 15) / 16, K2p = 16 * n1b, n2b = (h2 + 15) / 16, K3p = 16 * n2b;
  o.frag_floats = (size_t)256 * (n1b * (K1p / 16) + n2b * (K2p / 16) + K3p / 16);
  o.frag = B.alloc(o.frag_floats);
  size_t at = o.frag;
  auto emit = [&](size_t wt, int nb_count, int J, int K, int H) {
    for (int nb = 0; nb < nb_count; ++nb)
      for (int j = 0; j < J; ++j)
        for (int lane = 0; lane < 64; ++lane)
          for (int t = 0; t < 4; ++t) {
            const int f = 16 * nb + (lane & 15), k = 16 * j + 4 * (lane >> 4) + t;
            const float v = (k < K && f < H) ? B.host[wt + (size_t)k * H + f] : 0.f;   // read before the blob may grow: alloc is done
            B.host[at++] = v;
          }
  };
  emit(o.w1t, n1b, K1p / 16, in1, h1);
  emit(o.w2t, n2b, K2p / 16, h1, h2);
  emit(o.w3t, 1, K3p / 16, h2, nc);
  return MURAL_OK;
}

int pool_len(int L, int k, int s, int p) { return (L + 2 * p - k) / s + 1; }

// tile geometry for P positions per workgroup; returns LDS bytes (0 if a stage needs too many blocks per wave)
// `towers`: bit mask of the towers the launch runs; `phase`: 0 every stage, 1 first conv stage only, 2 the two short stages
// only (see SnvFwdArgs): only the columns of the stages a launch runs size its LDS buffers
size_t plan_geometry(SnvFwdArgs& a, int Lwin, int P, int n_class, int towers = 3, int phase = 0) {
  static const int pools[2][3][3] = {{{15, 15, 7}, {7, 7, 3}, {3, 3, 1}}, {{3, 3, 1}, {3, 3, 1}, {3, 3, 1}}};
  int maxcols = 0;
  for (int tw = 0; tw < 2; ++tw) {
    TowerGeom& g = a.geom[tw];
    g.L1 = tw == 0 ? Lwin : 2 * SNV_MID_HALF + 1;
    g.col0 = tw == 0 ? 0 : Lwin / 2 - SNV_MID_HALF;
    int L = g.L1;
    for (int i = 0; i < 3; ++i) {
      g.pk[i] = pools[tw][i][0];
      g.ps[i] = pools[tw][i][1];
      g.pp[i] = pools[tw][i][2];
      L = pool_len(L, g.pk[i], g.ps[i], g.pp[i]);
      if (L < 1) return 0;
      g.L[i] = L;
      g.Sc[i] = L + 1;
      g.NC[i] = 1 + P * g.Sc[i];
      g.nb[i] = (g.NC[i] + 15) / 16;
      g.dL[i] = FastDiv::make((uint32_t)L);
      g.dSc[i] = FastDiv::make((uint32_t)g.Sc[i]);
      if (!((towers >> tw) & 1)) continue;
      if ((phase == 1 && i != 0) || (phase == 2 && i == 0)) continue;
      if ((g.nb[i] + 1) / 2 > SNV_NB2MAX) return 0;
      maxcols = std::max(maxcols, 16 * g.nb[i] + 2);
    }
  }
  a.P = P;
  a.Lwin = Lwin;
  a.tw_first = (towers & 1) ? 0 : 1;
  a.tw_last = (towers & 2) ? 1 : 0;
  a.phase = phase;
  a.x0_cols = a.geom[0].L[0] + a.geom[1].L[0];
  a.nbuf = maxcols * SNV_C;
  const size_t par = (size_t)2 * (2 * EX_COUNT * SNV_C + n_class * SNV_C + SNV_MAXCLASS);   // resident small parameters
  return (size_t)2 * a.nbuf * 4 + (size_t)(2 * P * SNV_C + 3 * P * SNV_MAXCLASS) * 4 + par * 4;
}

constexpr size_t kLdsTwoPerCu = 80 * 1024;
constexpr size_t kLdsMax = 160 * 1024;

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Workspace { float* local_logits; int64_t* cat; uint8_t* symbols; float* x0; float* xlogit; float* s3[2]; int* counters;
                   float *vx0A, *vx0B, *vs3A, *vs3B; };      // long windows: the segments' inputs and pooled outputs

// Sites whose short-stage launches are deferred into ONE launch per tower (run_towers): with the units at a fixed stride a launch takes
// ceil(units / 2048 waves) unit-times, and a 131 072-site chunk gives the short stages 10.67 / 12.8 units per wave (3 % / 1.5 % of
// tail); four chunks' worth are 42.7 / 51.2.  Long windows and models without the split launches keep one chunk.
constexpr int SNV_SUPER = 4;
int64_t super_chunk_sites(const MuralSnvModel* m) {
  const bool off = dev_env("MURAL_SNV_DEFER_SHORT") && atoi(dev_env("MURAL_SNV_DEFER_SHORT")) == 0;
  return (m->split && !m->longwin && !off) ? SNV_SUPER * m->chunk : m->chunk;
}

// one_chunk: the layout whose short stages run per chunk (s3 / xlogit of one chunk: the smallest workspace a call accepts)
size_t carve(const MuralSnvModel* m, int64_t n, bool dense, void* base, Workspace* w, bool one_chunk = false) {
  size_t off = 0;
  const size_t guard = ws_guard_bytes();      // 0 outside the validation tests (common.h)
  ws_layout_reset();
  auto take = [&](size_t bytes) {
    size_t o = off;
    ws_layout_add(o, bytes);
    off = align_up(off + bytes + guard, 256);
    return o;
  };
  const size_t o_ll = take((size_t)n * m->shape.n_class * 4);
  const size_t o_cat = take((size_t)n * std::max(m->shape.local_cols, 1) * 8);
  const size_t o_sym = take(dense ? (size_t)n * m->shape.distal_len : 16);
  const size_t o_x0 = take((size_t)std::min<int64_t>(n, m->chunk) * std::max(m->args.x0_cols, 1) * SNV_C * 4);
  // (the short-stage launches run once per SUPER-chunk of up to SNV_SUPER chunks: their inputs and the mid logits are kept that long)
  const int64_t s3_sites = std::min<int64_t>(n, one_chunk ? m->chunk : super_chunk_sites(m));
  const size_t o_xl = take((size_t)s3_sites * SNV_MAXCLASS * 4);
  const size_t o_s3l = take((size_t)s3_sites * std::max(m->args.geom[0].L[1], 1) * SNV_C * 4);
  const size_t o_s3m = take((size_t)s3_sites * std::max(m->args.geom[1].L[1], 1) * SNV_C * 4);
  const size_t o_cnt = take(64);      // unit counters of the wave-private launches of a chunk (SnvFwdArgs::unit_counter)
  size_t o_vxa = 0, o_vxb = 0, o_vsa = 0, o_vsb = 0;
  if (m->longwin) {
    const size_t cn = (size_t)std::min<int64_t>(n, m->chunk);
    (void)o_vxa;      // (until round 6 the segments were copied out of x0 with their halo: 0.6 GB per chunk of 8192 sites at R = 4000)
    (void)o_vxb;
    o_vsa = take(cn * m->lw_nA * m->args_lwA.geom[0].L[1] * SNV_C * 4);
    o_vsb = take(cn * m->args_lwB.geom[0].L[1] * SNV_C * 4);
  }
  if (w) {
    char* b = static_cast<char*>(base);
    w->local_logits = reinterpret_cast<float*>(b + o_ll);
    w->cat = reinterpret_cast<int64_t*>(b + o_cat);
    w->symbols = reinterpret_cast<uint8_t*>(b + o_sym);
    w->x0 = reinterpret_cast<float*>(b + o_x0);
    w->xlogit = reinterpret_cast<float*>(b + o_xl);
    w->s3[0] = reinterpret_cast<float*>(b + o_s3l);
    w->s3[1] = reinterpret_cast<float*>(b + o_s3m);
    w->counters = reinterpret_cast<int*>(b + o_cnt);
    w->vx0A = reinterpret_cast<float*>(b + o_vxa);
    w->vx0B = reinterpret_cast<float*>(b + o_vxb);
    w->vs3A = reinterpret_cast<float*>(b + o_vsa);
    w->vs3B = reinterpret_cast<float*>(b + o_vsb);
  }
  return off;
}

// ---- long windows: segments of the large tower's pooled first-stage row (MuralSnvModel::longwin) -------------------------------
// columns are 32 floats = 128 bytes: a thread moves 16 bytes, 8 threads a column
// s3[site][j][32] for j in [j_lo, j_hi): from segment k = (j - jbase) / nj (clamped to nseg - 1), local pooled column jl = j - k * nj
__global__ void lw_scatter_kernel(const float* __restrict__ vs, int64_t n, int nseg, int Lp /* pooled columns of a segment */, int nj, int j_lo,
                                  int j_hi, int jl_shift, int L3, float* __restrict__ s3) {
  const int span = j_hi - j_lo;
  const int64_t total = n * span * 8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(i & 7);
    const int64_t col = i >> 3;
    const int j = j_lo + (int)(col % span);
    const int64_t site = col / span;
    // segment 0 owns outputs 0 .. nj (its left edge is the row's), segment k >= 1 owns k nj + 1 .. k nj + nj
    int k = j <= nj ? 0 : (j - 1) / nj;
    if (k > nseg - 1) k = nseg - 1;
    const int jl = j - k * nj + jl_shift;
    const float4 v = reinterpret_cast<const float4*>(vs + (((size_t)site * nseg + k) * Lp + jl) * SNV_C)[q];
    reinterpret_cast<float4*>(s3 + ((size_t)site * L3 + j) * SNV_C)[q] = v;
  }
}

// the launch-independent part of a tower geometry (what stage 1 and the workspace need)
void fill_tower_lengths(SnvFwdArgs& a, int Lwin) {
  static const int pools[2][3][3] = {{{15, 15, 7}, {7, 7, 3}, {3, 3, 1}}, {{3, 3, 1}, {3, 3, 1}, {3, 3, 1}}};
  for (int tw = 0; tw < 2; ++tw) {
    TowerGeom& g = a.geom[tw];
    g.L1 = tw == 0 ? Lwin : 2 * SNV_MID_HALF + 1;
    g.col0 = tw == 0 ? 0 : Lwin / 2 - SNV_MID_HALF;
    int L = g.L1;
    for (int i = 0; i < 3; ++i) {
      g.pk[i] = pools[tw][i][0]; g.ps[i] = pools[tw][i][1]; g.pp[i] = pools[tw][i][2];
      L = pool_len(L, g.pk[i], g.ps[i], g.pp[i]);
      g.L[i] = L;
    }
  }
  a.Lwin = Lwin;
  a.x0_cols = a.geom[0].L[0] + a.geom[1].L[0];
}

}  // namespace
}  // namespace mural

using namespace mural;

extern "C" int mural_snv_model_create(const MuralSnvShape* shape, const MuralSnvParams* hp, MuralSnvModel** out) {
  MURAL_REQUIRE(shape && hp && out, "NULL argument");
  const MuralSnvShape& sh = *shape;
  MURAL_REQUIRE(sh.model_no >= 0 && sh.model_no <= 2, "model_no for snv must be one of [0, 1, 2], got %d", sh.model_no);
  MURAL_REQUIRE(sh.n_class >= 1 && sh.n_class <= SNV_MAXCLASS, "n_class must be in [1,%d], got %d", SNV_MAXCLASS, sh.n_class);
  const bool has_local = sh.model_no != 1, has_towers = sh.model_no != 0;
  if (has_towers) {
    MURAL_REQUIRE(sh.channels == SNV_C && sh.ksize == SNV_K,
                  "the gfx950 tower kernel is built for CNN_out_channels=32, CNN_kernel_size=3 (got %d, %d)", sh.channels,
                  sh.ksize);
    MURAL_REQUIRE(sh.distal_len > 200, "Error: distal seq len must be >200");   // model_snv.py:470
  }
  if (has_local) {
    MURAL_REQUIRE(sh.local_cols >= 1 && sh.emb_rows >= 2 && sh.hidden1 >= 1 && sh.hidden2 >= 1, "bad local-branch shape");
  }
  MuralSnvModel* m = new MuralSnvModel();
  std::memset(m, 0, sizeof(*m));
  m->shape = sh;
  Blob B;
  TowerOff toff[2];
  LocalOff loff;
  int rc = MURAL_OK;
  if (has_towers) {
    rc = fold_tower(hp->large, sh, B, toff[0]);
    if (!rc) rc = fold_tower(hp->mid, sh, B, toff[1]);
    if (!rc) {
      // largest P that keeps two workgroups per CU; fall back to one workgroup per CU for long windows
      size_t lds = 0;
      int P = 0;
      for (int cand = 16; cand >= 1 && !P; --cand) {
        SnvFwdArgs tmp;
        const size_t need = plan_geometry(tmp, sh.distal_len, cand, sh.n_class);
        if (need && need <= kLdsTwoPerCu) { P = cand; lds = need; }
      }
      for (int cand = 16; cand >= 1 && !P; --cand) {
        SnvFwdArgs tmp;
        const size_t need = plan_geometry(tmp, sh.distal_len, cand, sh.n_class);
        if (need && need <= kLdsMax) { P = cand; lds = need; }
      }
      // (tower, phase) launch q of the split mode: the largest workgroup tile that keeps two workgroups per CU, then the wave-private
      // form where it applies
      auto plan_part = [&](int q) -> bool {
        const int towers = (q & 1) ? 2 : 1, phase = q < 2 ? 1 : 2;
        int Pq = 0, Pmax = 32;
        if (const char* e = dev_env("MURAL_DEBUG_SPLIT_P")) {   // diagnostic: "P0,P1,P2,P3" caps the tile sizes
          int v[4] = {32, 32, 32, 32};
          sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]);
          if (v[q] >= 1) Pmax = v[q];
        }
        for (int cand = Pmax; cand >= 1 && !Pq; --cand) {
          SnvFwdArgs tmp;
          const size_t need = plan_geometry(tmp, sh.distal_len, cand, sh.n_class, towers, phase);
          if (need && need <= kLdsTwoPerCu) { Pq = cand; m->lds_split[q] = need; }
        }
        if (!Pq) return false;
        if (phase == 2) {   // prefer a tile whose long stage splits into full block pairs for both waves (nb % 4 == 0)
          for (int cand = Pq; cand >= Pq - 2 && cand >= 1; --cand) {
            SnvFwdArgs tmp;
            const size_t need = plan_geometry(tmp, sh.distal_len, cand, sh.n_class, towers, phase);
            if (need && tmp.geom[q & 1].nb[1] % 4 == 0) { Pq = cand; m->lds_split[q] = need; break; }
          }
        }
        plan_geometry(m->args_split[q], sh.distal_len, Pq, sh.n_class, towers, phase);
        // Wave-private form of this launch (snv_tower_wave.hip): the most sites per wave that keep a wave within its nine
        // blocks and two four-wave workgroups on a CU.  MURAL_DEBUG_TOWER_WAVE = bit mask of the launches that may take it
        // (default: all four); 0 keeps the workgroup-tile kernel everywhere (A/B runs).
        int wave_mask = 15;
        if (const char* e = dev_env("MURAL_DEBUG_TOWER_WAVE")) wave_mask = atoi(e);
        if ((wave_mask >> q) & 1) {
          for (int cand = 31; cand >= 1; --cand) {
            SnvFwdArgs tmp;
            std::memset(&tmp, 0, sizeof(tmp));
            const size_t need = plan_wave_geometry(tmp, sh.distal_len, cand, sh.n_class, q & 1, phase);
            if (need && need <= kLdsTwoPerCu) {
              m->args_split[q] = tmp;
              m->lds_split[q] = need;
              break;
            }
          }
        }
        return true;
      };
      m->chunk = SNV_CHUNK;
      if (const char* e = dev_env("MURAL_SNV_CHUNK")) {      // experiment: smaller chunks keep x0 in the 256 MB Infinity Cache
        const long v = atol(e);
        if (v >= 1024 && v <= SNV_CHUNK) m->chunk = v;
      }
      m->longwin = false;
      if (!dev_env("MURAL_DEBUG_NO_LONGWIN")) {      // (tried first: a row of 143 .. ~270 columns also fits ONE workgroup tile per CU, slowly)
        // Long window: the large tower's pooled first-stage row (L2 columns) does not fit a wave's image.  Its first conv stage --
        // four k=3 convs, then a 7-wide pool -- runs on segments with 4 halo columns: lw_nA segments of 7 nj + 8 columns starting at
        // 0, 7 nj, 14 nj, ... (a start that is a multiple of the pool stride keeps the segment's pool windows on the row's; the first
        // pooled column of a later segment is halo and dropped) and one that ends with the row.  Everything behind -- the short stages
        // of the large tower on whole rows, the mid tower, the head -- runs as in the split mode.
        SnvFwdArgs probe;
        std::memset(&probe, 0, sizeof(probe));
        fill_tower_lengths(probe, sh.distal_len);
        const int L2 = probe.geom[0].L[0], nj = 19, LA = 7 * nj + 8;
        if (L2 > LA && probe.geom[0].pk[1] == 7 && probe.geom[0].ps[1] == 7 && probe.geom[0].pp[1] == 3) {
          int nA = 0;
          while (7 * nj * nA + LA <= L2) ++nA;
          const int SB = 7 * ((L2 - 142 + 6) / 7), LB = L2 - SB;      // 136 .. 142 columns, start <= 7 nj nA
          bool ok = nA >= 1 && LB >= 127 && LB <= 142 && SB <= 7 * nj * nA && SB >= 0;
          if (ok) {
            std::memset(&m->args_lwA, 0, sizeof(SnvFwdArgs));
            std::memset(&m->args_lwB, 0, sizeof(SnvFwdArgs));
            m->lds_lwA = plan_wave_geometry(m->args_lwA, 15 * (LA - 1) + 1, 1, sh.n_class, 0, 1);
            m->lds_lwB = plan_wave_geometry(m->args_lwB, 15 * (LB - 1) + 1, 1, sh.n_class, 0, 1);
            ok = m->lds_lwA && m->lds_lwA <= kLdsTwoPerCu && m->lds_lwB && m->lds_lwB <= kLdsTwoPerCu &&
                 m->args_lwA.geom[0].L[0] == LA && m->args_lwB.geom[0].L[0] == LB;
          }
          // the short stages (second / third conv stage of the large tower on whole pooled rows, the mid tower, the head) must fit
          // their kernels too.  Where only they do not (R >= ~15000: the second-stage row is longer than two waves' nine blocks) the
          // model is FRONT-ONLY: the segmented first stage -- 86 % of the work -- still runs here (mural_snv_forward_front hands out
          // the pooled second-stage input) and the caller finishes per layer.
          bool part_ok[4] = {ok, false, false, false};
          for (int q = 1; q < 4 && ok; ++q) part_ok[q] = plan_part(q);
          const bool short_ok = ok && part_ok[1] && part_ok[2] && part_ok[3];
          if (ok) {
            m->longwin = true;
            m->split = true;
            m->front_only = !short_ok;
            m->front_mid = !short_ok && part_ok[1] && part_ok[3];      // (the large tower's short stages are what does not fit)
            m->lw_nA = nA; m->lw_LA = LA; m->lw_LB = LB; m->lw_SB = SB; m->lw_nj = nj;
            m->args_lwA.x0_cols = LA;
            m->args_lwB.x0_cols = LB;
            fill_tower_lengths(m->args, sh.distal_len);
            m->lds_bytes = 0;
            m->chunk = L2 > 1100 ? 4096 : 8192;      // x0 + segment scratch: ~150 KB per site at R = 4000
            P = -1;
          }
        }
      }
      if (!P) {
        set_error("distal_radius %d is too long for the LDS-resident tower kernel", (sh.distal_len - 1) / 2);
        rc = MURAL_E_INVALID;
      } else {
        if (P > 0) {
        plan_geometry(m->args, sh.distal_len, P, sh.n_class);
        m->lds_bytes = lds;
        // split mode: (tower, phase) pairs in their own launches, each with the largest tile that keeps two workgroups per
        // CU: the mid tower and above all the short stages then run layers that are many blocks wide
        m->split = false;
        if (lds <= kLdsTwoPerCu && !dev_env("MURAL_DEBUG_NO_TOWER_SPLIT")) {
          bool ok = true;
          for (int q = 0; q < 4 && ok; ++q) ok = plan_part(q);
          m->split = ok;
        }
        }
        m->lds_small = m->longwin ? 0 : plan_geometry(m->args_small, sh.distal_len, 1, sh.n_class);
        Stage1Args& s1 = m->s1;
        for (int tw = 0; tw < 2; ++tw) {
          const TowerGeom& g = m->args.geom[tw];
          s1.tw[tw] = Stage1Tower{g.L1, g.col0, g.L[0], g.pk[0], g.ps[0], g.pp[0]};
        }
        s1.Lwin = sh.distal_len;
        s1.cw = (sh.distal_len + 2 + 15) & ~15;
        s1.wave_bytes = (s1.cw + s1.tw[0].L2 * 16 + s1.tw[1].L2 * 4 + 15) & ~15;
        s1.x0_cols = m->args.x0_cols;
        s1.nwords = (sh.distal_len + 15) / 16 + 1;
        s1.radius = (sh.distal_len - 1) / 2;
        m->s1_lds_bytes = (size_t)2 * SNV_LUTBLK * 4 + (size_t)16 * s1.wave_bytes;
        if (m->longwin) {      // one workgroup per site (snv_stage1_site_kernel): a single copy of the window and its index tables
          s1.site_mode = 1;
          m->s1_lds_bytes = (size_t)2 * SNV_LUTBLK * 4 + (size_t)s1.wave_bytes + 64;
        }
        // the large tower's pair table rides along where it fits (15-wide pools: the shipped first max-pool)
        m->s1_pair = !s1.site_mode && s1.tw[0].pk == 15 && s1.tw[0].ps == 15 && m->s1_lds_bytes + (size_t)SNV_LUT4 * 4 <= kLdsMax &&
                     !dev_env("MURAL_DEBUG_NO_PAIR_TABLE");
        if (m->s1_pair) m->s1_lds_bytes += (size_t)SNV_LUT4 * 4;
        if (m->s1_lds_bytes > kLdsMax || s1.tw[0].pk > 16 || s1.tw[1].pk > 4) {
          set_error("distal_radius %d is too long for the stage-1 kernel's LDS window", (sh.distal_len - 1) / 2);
          rc = MURAL_E_INVALID;
        }
      }
    }
  }
  if (!rc && has_local) rc = fold_local(hp->local, sh, B, loff);
  if (rc) { delete m; return rc; }
  m->blob_floats = B.host.size();
  if (hipError_t e = hipMalloc(&m->blob, std::max<size_t>(m->blob_floats, 64) * 4); e != hipSuccess) {
    set_error("hipMalloc of %zu bytes for the folded weights failed: %s", m->blob_floats * 4, hipGetErrorString(e));
    delete m;
    return MURAL_E_RUNTIME;
  }
  if (m->blob_floats && hipMemcpy(m->blob, B.host.data(), m->blob_floats * 4, hipMemcpyHostToDevice) != hipSuccess) {
    set_error("hipMemcpy of the folded weights failed");
    (void)hipFree(m->blob);
    delete m;
    return MURAL_E_RUNTIME;
  }
  if (has_towers) {
    for (int tw = 0; tw < 2; ++tw) {
      TowerDev& d = m->args.tw[tw];
      const TowerOff& o = toff[tw];
      d.lut = m->blob + o.lut; d.lut4 = m->blob + o.lut4; d.taps = m->blob + o.taps; d.bias0 = m->blob + o.bias0; d.wfrag = m->blob + o.wfrag; d.wfrag4 = m->blob + o.wfrag4;
      d.bias = m->blob + o.bias; d.post_s = m->blob + o.post_s; d.post_t = m->blob + o.post_t;
      d.ex_s = m->blob + o.ex_s; d.ex_t = m->blob + o.ex_t; d.fc_w = m->blob + o.fc_w; d.fc_b = m->blob + o.fc_b;
    }
    m->args.n_class = sh.n_class;
    m->args.has_local = sh.model_no == 2;
    for (SnvFwdArgs& a2 : m->args_split) {
      a2.tw[0] = m->args.tw[0];
      a2.tw[1] = m->args.tw[1];
      a2.n_class = sh.n_class;
      a2.has_local = m->args.has_local;
    }
    for (SnvFwdArgs* a3 : {&m->args_lwA, &m->args_lwB}) {
      a3->tw[0] = m->args.tw[0];
      a3->tw[1] = m->args.tw[1];
      a3->n_class = sh.n_class;
      a3->has_local = m->args.has_local;
    }
    m->args_small.tw[0] = m->args.tw[0];
    m->args_small.tw[1] = m->args.tw[1];
    m->args_small.n_class = sh.n_class;
    m->args_small.has_local = m->args.has_local;
    m->s1.lut[0] = m->args.tw[0].lut;
    m->s1.lut4 = m->s1_pair ? m->args.tw[0].lut4 : nullptr;
    m->s1.lut[1] = m->args.tw[1].lut;
  }
  if (has_local) {
    LocalDev& L = m->local;
    L.emb = m->blob + loff.emb; L.w1t = m->blob + loff.w1t; L.b1 = m->blob + loff.b1; L.w2t = m->blob + loff.w2t;
    L.b2 = m->blob + loff.b2; L.w3t = m->blob + loff.w3t; L.b3 = m->blob + loff.b3;
    L.frag = m->blob + loff.frag; L.frag_floats = (int)loff.frag_floats;
    L.cols = sh.local_cols; L.emb_rows = sh.emb_rows; L.in1 = 5 * sh.local_cols; L.h1 = sh.hidden1; L.h2 = sh.hidden2;
    L.n_class = sh.n_class;
    m->loc_fused = has_towers && local_mfma_plan(L, &m->loc_d, &m->loc_lds) && !dev_env("MURAL_DEBUG_NO_LOCAL_FUSE");
  }
  *out = m;
  return MURAL_OK;
}

extern "C" void mural_snv_model_destroy(MuralSnvModel* m) {
  if (!m) return;
  if (m->blob) (void)hipFree(m->blob);
  delete m;
}

extern "C" size_t mural_snv_workspace_bytes(const MuralSnvModel* m, int64_t n, int32_t dense) {
  if (!m || n <= 0) return 256;
  return carve(m, n, dense != 0, nullptr, nullptr);
}

extern "C" size_t mural_snv_workspace_bytes_min(const MuralSnvModel* m, int64_t n, int32_t dense) {
  if (!m || n <= 0) return 256;
  return carve(m, n, dense != 0, nullptr, nullptr, true);
}

extern "C" int mural_snv_tap_layout(const MuralSnvModel* m, int32_t* o) {
  MURAL_REQUIRE(m && o, "NULL argument");
  std::memset(o, 0, 16 * sizeof(int32_t));
  o[0] = m->args.P;
  o[1] = m->args.nbuf;
  for (int i = 0; i < 3; ++i) { o[2 + i] = m->args.geom[0].L[i]; o[5 + i] = m->args.geom[1].L[i]; }
  o[8] = 13;
  o[9] = (int32_t)m->lds_bytes;
  o[10] = m->front_only ? 1 : 0;
  o[11] = (int32_t)m->chunk;
  o[12] = m->front_mid ? 1 : 0;
  return MURAL_OK;
}

extern "C" const char* mural_snv_kernel_name(void) { return "snv_tower_wave"; }

namespace mural { int profile_begin(); int profile_end(double*, int64_t*); }
extern "C" int mural_profile_begin(void) { return mural::profile_begin(); }
extern "C" int mural_profile_end(double* total_ms, int64_t* launches) { return mural::profile_end(total_ms, launches); }

namespace mural { unsigned long long* g_tower_stamps = nullptr; }      // diagnostic (debug flavour: mural_debug_set_stamps)

// stage-1 kernel + tower kernel over chunks of SNV_CHUNK sites (the x0 scratch holds one chunk)
constexpr int64_t SNV_SMALL_BATCH = 256;   // up to here a call is latency-bound: single launch with one-site tiles

// front_out != nullptr (front-only models): per chunk only the stage-1 kernel and the segmented first stage of the large tower run; the
// pooled second-stage input goes to front_out [n][L[1]][32] instead of the workspace
static int run_towers(const MuralSnvModel* m, Stage1Args s1, SnvFwdArgs a, bool packed, int64_t n, const Workspace& w,
                      float* out, float* taps, const int32_t* status, hipStream_t stream, bool one_chunk = false, float* front_out = nullptr) {
  const int nc = m->shape.n_class;
  const bool small = taps == nullptr && n <= SNV_SMALL_BATCH && m->lds_small > 0 && !dev_env("MURAL_DEBUG_NO_SMALL_BATCH");
  // the short-stage launches of up to SNV_SUPER chunks as ONE launch per tower (super_chunk_sites): the chunks' first-stage launches leave
  // their pooled rows side by side in s3
  const bool defer = m->split && taps == nullptr && !small && !m->longwin && super_chunk_sites(m) > m->chunk && n > m->chunk && !one_chunk;
  const int64_t super = defer ? super_chunk_sites(m) : m->chunk;
  const size_t s3l_site = (size_t)std::max(m->args.geom[0].L[1], 1) * SNV_C, s3m_site = (size_t)std::max(m->args.geom[1].L[1], 1) * SNV_C;
  for (int64_t sc0 = 0; sc0 < n; sc0 += super) {
  const int64_t sn = std::min<int64_t>(super, n - sc0);
  for (int64_t c0 = sc0; c0 < sc0 + sn; c0 += m->chunk) {
    const int64_t cn = std::min<int64_t>(m->chunk, sc0 + sn - c0);
    const int64_t rel = defer ? c0 - sc0 : 0;      // this chunk's place in the super-chunk's s3 / xlogit
    Stage1Args s = s1;
    s.n = cn;
    s.x0 = w.x0;
    s.zero = small ? reinterpret_cast<int*>(w.s3[1]) : nullptr;
    if (packed) { s.pos = s1.pos + c0; s.strand = s1.strand + c0; }
    else s.codes = s1.codes + c0 * m->shape.distal_len;
    if (int rc = launch_snv_stage1(s, packed, s.loc_on ? std::max(m->s1_lds_bytes, m->loc_lds) : m->s1_lds_bytes, stream)) return rc;
    const bool split = m->split && taps == nullptr && !small;   // the debug dump wants both towers in one tile geometry
    // (the unit counters are read only with MURAL_TOWER_DYNAMIC_UNITS=1: the fill is a 4 us launch per chunk otherwise wasted)
    if (split && dev_env("MURAL_TOWER_DYNAMIC_UNITS") && atoi(dev_env("MURAL_TOWER_DYNAMIC_UNITS")) != 0)
      MURAL_HIP_CHECK(hipMemsetAsync(w.counters, 0, 64, stream));
    bool lw_mid_done = false;      // (long windows: the mid tower's first stage rode in the segments' launch)
    for (int part = 0; part < (split ? (defer ? 2 : 4) : 1); ++part) {
      if (part == 1 && lw_mid_done) continue;
      if (m->longwin && part == 0) {
        // the large tower's first conv stage on segments of the pooled row: gather (with halo) -> two wave-private launches (the
        // lw_nA equal segments of every site, then the one that ends with the row) -> scatter of the pooled columns into s3[0]
        const int LpA = m->args_lwA.geom[0].L[1], LpB = m->args_lwB.geom[0].L[1], L3 = m->args.geom[0].L[1], nj = m->lw_nj;
        const int x0_cols = m->args.x0_cols;
        auto grid_of = [](int64_t items) { return dim3((unsigned)std::min<int64_t>((items + 255) / 256, 65536)); };
        // (round 6: the segments are read in place from x0 -- SnvFwdArgs::seg_n -- instead of being copied out with their halo first:
        // 2 x 70 MB per 512 windows at R = 4000 and two launches less)
        // the two segment launches and the mid tower's first-stage launch are three jobs of one kernel instance: ONE launch (each launch of
        // its own pays ~14 us of start-up next to 22 us per unit; MURAL_DEBUG_LW_SEPARATE: three launches, same results)
        SnvFwdArgs lw_jobs[3];
        size_t lw_lds[3];
        const bool lw_merge = m->args_lwA.wave && m->args_lwB.wave && m->args_split[1].wave && !dev_env("MURAL_DEBUG_LW_SEPARATE") &&
                              (!front_out || m->front_mid);
        for (int kind = 0; kind < 2; ++kind) {
          SnvFwdArgs t = kind == 0 ? m->args_lwA : m->args_lwB;
          t.n = kind == 0 ? cn * m->lw_nA : cn;
          t.x0 = w.x0;
          t.x0_cols = x0_cols;
          t.seg_n = kind == 0 ? m->lw_nA : 1;
          t.seg_step = kind == 0 ? 7 * nj : 0;
          t.seg_col0 = kind == 0 ? 0 : m->lw_SB;
          t.s3[0] = kind == 0 ? w.vs3A : w.vs3B;
          t.s3[1] = w.s3[1];
          t.xlogit = w.xlogit;
          t.local_logits = w.local_logits + c0 * nc;
          t.out = out ? out + c0 * nc : nullptr;
          t.taps = nullptr;
          t.stamps = nullptr;
          t.status = status;
          t.unit_counter = w.counters + 4 + kind;
          if (lw_merge) {
            lw_jobs[kind] = t;
            lw_lds[kind] = kind == 0 ? m->lds_lwA : m->lds_lwB;
          } else if (int rc = launch_snv_towers(m, t, kind == 0 ? m->lds_lwA : m->lds_lwB, stream)) {
            return rc;
          }
        }
        if (lw_merge) {
          SnvFwdArgs t = m->args_split[1];      // the mid tower's first conv stage (what part 1 of this loop launches otherwise)
          t.s3[0] = w.s3[0];
          t.s3[1] = w.s3[1];
          t.n = cn;
          t.x0 = w.x0;
          t.xlogit = w.xlogit;
          t.local_logits = w.local_logits + c0 * nc;
          t.out = out ? out + c0 * nc : nullptr;
          t.taps = nullptr;
          t.tap_stride = a.nbuf;
          t.stamps = nullptr;
          t.status = status;
          t.unit_counter = nullptr;
          lw_jobs[2] = t;
          lw_lds[2] = m->lds_split[1];
          if (int rc = launch_snv_tower_wave_jobs(lw_jobs, lw_lds, 3, stream)) return rc;
          lw_mid_done = true;
        }
        const int jA = m->lw_nA * nj + 1;      // outputs 0 .. nA nj come from the equal segments, the rest from the last one
        float* s3_dst = front_out ? front_out + (size_t)c0 * L3 * SNV_C : w.s3[0];
        hipLaunchKernelGGL(lw_scatter_kernel, grid_of(cn * jA * 8), dim3(256), 0, stream, w.vs3A, cn, m->lw_nA, LpA, nj, 0, jA, 0, L3, s3_dst);
        if (L3 > jA)
          hipLaunchKernelGGL(lw_scatter_kernel, grid_of(cn * (L3 - jA) * 8), dim3(256), 0, stream, w.vs3B, cn, 1, LpB, nj, jA, L3,
                             -(m->lw_SB / 7), L3, s3_dst);
        MURAL_HIP_CHECK(hipGetLastError());
        if (front_out && !m->front_mid) break;
        continue;
      }
      if (front_out && part > 1) break;      // (front-only models: part 1 = the mid tower's first stage into s3[1]; mural_snv_forward_finish does the rest)
      SnvFwdArgs t = split ? m->args_split[part] : (small ? m->args_small : a);
      t.s3[0] = w.s3[0] + (size_t)rel * s3l_site;
      t.s3[1] = w.s3[1] + (size_t)rel * s3m_site;
      t.n = cn;
      t.x0 = w.x0;
      t.xlogit = w.xlogit + (size_t)rel * SNV_MAXCLASS;
      if (small) {   // one workgroup per (site, tower): the split launches' s3 scratch carries the mid logits and the arrival counters
        t.par = 1;
        t.xlogit2 = w.s3[0];
        t.tile_count = reinterpret_cast<int*>(w.s3[1]);
      }
      t.local_logits = w.local_logits + c0 * nc;
      t.out = out ? out + c0 * nc : nullptr;
      t.taps = c0 == 0 ? taps : nullptr;
      t.tap_stride = a.nbuf;
      t.stamps = packed ? mural::g_tower_stamps : nullptr;
      t.status = status;
      t.unit_counter = split && t.wave ? w.counters + part : nullptr;
      size_t lds = split ? m->lds_split[part] : (small ? m->lds_small : m->lds_bytes);
      // A workgroup-tile short-stage launch (long windows) packs P sites into a tile so that two workgroups fill a CU -- and then a call of
      // a few hundred windows is a few dozen workgroups on 256 CUs (R = 4000, 512 windows: 171 and 52).  Such a call takes the tile
      // with the fewest sites that still gives every CU two workgroups: same arithmetic per site, a third of the serial work per workgroup.
      if (split && !t.wave && part >= 2 && t.P > 1 && !dev_env("MURAL_DEBUG_NO_CALL_P")) {
        const int want = (int)std::max<int64_t>(1, std::min<int64_t>(t.P, (cn + 511) / 512));
        if (want < t.P) {
          SnvFwdArgs t2 = t;
          const size_t need = plan_geometry(t2, m->shape.distal_len, want, m->shape.n_class, (part & 1) ? 2 : 1, 2);
          if (need && need <= lds) {
            t = t2;
            lds = need;
          }
        }
      }
      if (int rc = launch_snv_towers(m, t, lds, stream)) return rc;
    }
  }
  if (defer) {
    for (int part = 2; part < 4; ++part) {
      SnvFwdArgs t = m->args_split[part];
      t.s3[0] = w.s3[0];
      t.s3[1] = w.s3[1];
      t.n = sn;
      t.x0 = w.x0;
      t.xlogit = w.xlogit;
      t.local_logits = w.local_logits + sc0 * nc;
      t.out = out + sc0 * nc;
      t.taps = nullptr;
      t.tap_stride = a.nbuf;
      t.stamps = packed ? mural::g_tower_stamps : nullptr;
      t.status = status;
      t.unit_counter = t.wave ? w.counters + part : nullptr;
      if (int rc = launch_snv_towers(m, t, m->lds_split[part], stream)) return rc;
    }
  }
  }
  return MURAL_OK;
}

// Network2 at small batch sizes: the local branch is one more workgroup of the first-stage launch instead of a launch of its own
static bool local_rides(const MuralSnvModel* m, int64_t n) { return m->shape.model_no == 2 && m->loc_fused && stage1_small_batch(n); }
static void ride_local(const MuralSnvModel* m, Stage1Args* s1, const int64_t* cat, float* logits) {
  s1->loc_on = 1;
  s1->loc = m->local;
  s1->loc_d = m->loc_d;
  s1->loc_cat = cat;
  s1->loc_out = logits;
}

// symbols != nullptr: the windows arrive as one MURAL_SYM_* byte per column (dev uint8 [n][distal_len]) instead of distal_x
static int forward_dense_impl(const MuralSnvModel* m, const int64_t* cat_x, const float* distal_x, int64_t n, float* out,
                              void* workspace, size_t ws_bytes, int32_t* status, float* taps, size_t taps_floats,
                              void* stream_, const uint8_t* symbols = nullptr) {
  MURAL_REQUIRE(m, "model handle is NULL");
  MURAL_REQUIRE(!m->front_only, "distal_radius %d: only the first conv stage of this model runs fused (mural_snv_forward_front)", (m->shape.distal_len - 1) / 2);
  MURAL_REQUIRE(n >= 0, "negative batch");
  if (n == 0) return MURAL_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const MuralSnvShape& sh = m->shape;
  MURAL_REQUIRE(out, "out is NULL");
  // a workspace of mural_snv_workspace_bytes holds the pooled rows of four chunks (the short stages as one launch per tower, + 1 %); a
  // caller short of memory may pass mural_snv_workspace_bytes_min: the short stages then run per chunk, same results bit for bit
  const bool own_symbols = symbols == nullptr;      // (the symbol region of the workspace is the dense entry's)
  const bool one_chunk = ws_bytes < carve(m, n, own_symbols, nullptr, nullptr);
  if (ws_bytes < carve(m, n, own_symbols, nullptr, nullptr, true) || !workspace) {
    set_error("workspace too small: need %zu bytes, got %zu", carve(m, n, own_symbols, nullptr, nullptr, true), ws_bytes);
    return MURAL_E_WORKSPACE;
  }
  Workspace w;
  carve(m, n, own_symbols, workspace, &w, one_chunk);
  if (sh.model_no == 0) {
    MURAL_REQUIRE(cat_x, "cat_x is NULL");
    return launch_snv_local(m->local, cat_x, n, out, stream);   // raw logits, model_snv.py:93
  }
  MURAL_REQUIRE(distal_x || symbols, "distal_x is NULL");
  if (sh.model_no == 2) {
    MURAL_REQUIRE(cat_x, "cat_x is NULL");
    if (!local_rides(m, n))
      if (int rc = launch_snv_local(m->local, cat_x, n, w.local_logits, stream)) return rc;
  }
  // a small batch is one launch per stage: its first-stage kernel classifies the dense columns itself
  const bool direct = stage1_small_batch(n) && !symbols;
  if (!direct && !symbols)
    if (int rc = launch_dense_to_symbols(distal_x, n, sh.distal_len, w.symbols, status, stream)) return rc;
  if (taps) MURAL_REQUIRE(!m->longwin, "the layer dump is not available for long windows (segmented first stage)");
  if (taps) MURAL_REQUIRE(taps_floats >= (size_t)13 * m->args.nbuf, "taps buffer too small (need %zu floats)", (size_t)13 * m->args.nbuf);
  Stage1Args s1 = m->s1;
  s1.codes = symbols ? symbols : w.symbols;
  if (direct) {
    s1.dense = distal_x;
    s1.status = status;
  }
  if (local_rides(m, n)) ride_local(m, &s1, cat_x, w.local_logits);
  return run_towers(m, s1, m->args, /*packed=*/false, n, w, out, taps, status, stream, one_chunk);
}

extern "C" int mural_snv_forward_dense(const MuralSnvModel* m, const int64_t* cat_x, const float* distal_x, int64_t n,
                                       float* out, void* workspace, size_t workspace_bytes, int32_t* status, void* stream) {
  return forward_dense_impl(m, cat_x, distal_x, n, out, workspace, workspace_bytes, status, nullptr, 0, stream);
}

extern "C" int mural_snv_forward_symbols(const MuralSnvModel* m, const int64_t* cat_x, const uint8_t* symbols, int64_t n, float* out,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  MURAL_REQUIRE(m && (m->shape.model_no == 0 || symbols), "symbols is NULL");
  return forward_dense_impl(m, cat_x, nullptr, n, out, workspace, workspace_bytes, nullptr, nullptr, 0, stream, symbols);
}

extern "C" int mural_snv_debug_taps(const MuralSnvModel* m, const int64_t* cat_x, const float* distal_x, int64_t n,
                                    float* out, void* workspace, size_t workspace_bytes, float* taps, size_t taps_floats,
                                    void* stream) {
  MURAL_REQUIRE(taps, "taps is NULL");
  return forward_dense_impl(m, cat_x, distal_x, n, out, workspace, workspace_bytes, nullptr, taps, taps_floats, stream);
}

extern "C" int mural_snv_forward_packed(const MuralSnvModel* m, const MuralGenome* g, const int64_t* pos,
                                        const uint8_t* strand, int64_t n, int32_t local_radius, int32_t local_order,
                                        float* out, void* workspace, size_t workspace_bytes, void* stream_) {
  MURAL_REQUIRE(m, "model handle is NULL");
  MURAL_REQUIRE(!m->front_only, "distal_radius %d: only the first conv stage of this model runs fused (mural_snv_forward_front)", (m->shape.distal_len - 1) / 2);
  MURAL_REQUIRE(g && g->packed2 && g->nmask, "genome pointers must not be NULL");
  MURAL_REQUIRE(g->n_amb == 0 || (g->amb_pos && g->amb_sym), "genome: n_amb > 0 needs amb_pos and amb_sym");
  MURAL_REQUIRE(n >= 0, "negative batch");
  if (n == 0) return MURAL_OK;
  MURAL_REQUIRE(pos && strand && out, "pos/strand/out must not be NULL");
  hipStream_t stream = (hipStream_t)stream_;
  const MuralSnvShape& sh = m->shape;
  const bool one_chunk = workspace_bytes < carve(m, n, false, nullptr, nullptr);      // (see forward_dense_impl)
  if (workspace_bytes < carve(m, n, false, nullptr, nullptr, true) || !workspace) {
    set_error("workspace too small: need %zu bytes, got %zu", carve(m, n, false, nullptr, nullptr, true), workspace_bytes);
    return MURAL_E_WORKSPACE;
  }
  Workspace w;
  carve(m, n, false, workspace, &w, one_chunk);
  if (sh.model_no != 1) {
    const int ncol = 2 * local_radius + 1 - (local_order - 1);
    MURAL_REQUIRE(ncol == sh.local_cols, "local_radius/local_order give %d k-mer columns, model has %d", ncol, sh.local_cols);
    int64_t sentinel = 1;
    for (int i = 0; i < local_order; ++i) sentinel *= 4;
    MURAL_REQUIRE(sentinel + 1 == sh.emb_rows, "local_order %d does not match the embedding table (%d rows)", local_order,
                  sh.emb_rows);
    if (int rc = mural_encode_kmer(g, pos, strand, n, local_radius, local_order, 0, w.cat, stream_)) return rc;
    float* dst = sh.model_no == 0 ? out : w.local_logits;
    if (!local_rides(m, n))
      if (int rc = launch_snv_local(m->local, w.cat, n, dst, stream)) return rc;
    if (sh.model_no == 0) return MURAL_OK;
  }
  Stage1Args s1 = m->s1;
  if (local_rides(m, n)) ride_local(m, &s1, w.cat, w.local_logits);
  s1.genome = *g;
  s1.pos = pos;
  s1.strand = strand;
  return run_towers(m, s1, m->args, /*packed=*/true, n, w, out, nullptr, nullptr, stream, one_chunk);
}

// Long windows whose short stages fit no kernel here (front-only models; R from ~15000 up to the stage-1 kernel's LDS limit): window
// decode + first layer + 15-wide pool (snv_stage1_site_kernel) and the large tower's first conv stage on halo'd segments of the
// pooled row -- 86 % of the model's arithmetic -- for the sites (pos, strand) of a packed genome.  s3_out: dev float
// [n][L3][32], the pooled second-stage input of the large tower (RBs1 + max-pool of model_snv.py:473-481, channel-last), L3 =
// mural_snv_tap_layout()[3].  The caller finishes per layer (mural_amd/model/generic_eval.py: conv2 .. fc of the large tower, the mid
// tower, the local branch, the head).  Works on every long-window model (the fused ones included).
extern "C" int mural_snv_forward_front(const MuralSnvModel* m, const MuralGenome* g, const int64_t* pos, const uint8_t* strand, int64_t n,
                                       int32_t local_radius, int32_t local_order, float* s3_out, void* workspace, size_t workspace_bytes,
                                       void* stream_) {
  MURAL_REQUIRE(m, "model handle is NULL");
  MURAL_REQUIRE(m->longwin, "mural_snv_forward_front serves long-window models (segmented first stage) only");
  MURAL_REQUIRE(!m->front_mid || n <= m->chunk, "mural_snv_forward_front: at most %lld sites per call (the workspace carries the mid tower to "
                "mural_snv_forward_finish)", (long long)m->chunk);
  MURAL_REQUIRE(g && g->packed2 && g->nmask, "genome pointers must not be NULL");
  MURAL_REQUIRE(g->n_amb == 0 || (g->amb_pos && g->amb_sym), "genome: n_amb > 0 needs amb_pos and amb_sym");
  MURAL_REQUIRE(n >= 0, "negative batch");
  if (n == 0) return MURAL_OK;
  MURAL_REQUIRE(pos && strand && s3_out, "pos/strand/s3_out must not be NULL");
  if (workspace_bytes < carve(m, n, false, nullptr, nullptr, true) || !workspace) {
    set_error("workspace too small: need %zu bytes, got %zu", carve(m, n, false, nullptr, nullptr, true), workspace_bytes);
    return MURAL_E_WORKSPACE;
  }
  Workspace w;
  carve(m, n, false, workspace, &w, true);
  const MuralSnvShape& sh = m->shape;
  if (m->front_mid && sh.model_no == 2) {      // the local branch's logits wait in the workspace for the head
    const int ncol = 2 * local_radius + 1 - (local_order - 1);
    MURAL_REQUIRE(ncol == sh.local_cols, "local_radius/local_order give %d k-mer columns, model has %d", ncol, sh.local_cols);
    int64_t sentinel = 1;
    for (int i = 0; i < local_order; ++i) sentinel *= 4;
    MURAL_REQUIRE(sentinel + 1 == sh.emb_rows, "local_order %d does not match the embedding table (%d rows)", local_order, sh.emb_rows);
    if (int rc = mural_encode_kmer(g, pos, strand, n, local_radius, local_order, 0, w.cat, stream_)) return rc;
    if (int rc = launch_snv_local(m->local, w.cat, n, w.local_logits, (hipStream_t)stream_)) return rc;
  }
  Stage1Args s1 = m->s1;
  s1.genome = *g;
  s1.pos = pos;
  s1.strand = strand;
  return run_towers(m, s1, m->args, /*packed=*/true, n, w, nullptr, nullptr, nullptr, (hipStream_t)stream_, true, s3_out);
}

// The second half of a front-only model with a fused mid tower (mural_snv_tap_layout()[12] == 1): the mid tower's short stages and the
// head (model_snv.py:496-513) around the large tower's logits the caller computed per layer from mural_snv_forward_front's s3_out.
// large_logits: dev float [n][n_class] (the large tower's fc output); workspace: THE SAME buffer the front call of these n sites used
// (it holds the mid tower's pooled row and the local branch's logits); out: dev float [n][n_class] log-probabilities.
extern "C" int mural_snv_forward_finish(const MuralSnvModel* m, const float* large_logits, int64_t n, float* out, void* workspace,
                                        size_t workspace_bytes, void* stream_) {
  MURAL_REQUIRE(m, "model handle is NULL");
  MURAL_REQUIRE(m->front_mid, "mural_snv_forward_finish serves front-only models with a fused mid tower only");
  MURAL_REQUIRE(n >= 0 && n <= m->chunk, "mural_snv_forward_finish: 0 .. %lld sites per call", (long long)m->chunk);
  if (n == 0) return MURAL_OK;
  MURAL_REQUIRE(large_logits && out, "large_logits/out must not be NULL");
  if (workspace_bytes < carve(m, n, false, nullptr, nullptr, true) || !workspace) {
    set_error("workspace too small: need %zu bytes, got %zu", carve(m, n, false, nullptr, nullptr, true), workspace_bytes);
    return MURAL_E_WORKSPACE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  Workspace w;
  carve(m, n, false, workspace, &w, true);
  const int nc = m->shape.n_class;
  MURAL_HIP_CHECK(hipMemcpy2DAsync(w.xlogit, SNV_MAXCLASS * 4, large_logits, (size_t)nc * 4, (size_t)nc * 4, (size_t)n, hipMemcpyDeviceToDevice, stream));
  SnvFwdArgs t = m->args_split[3];
  t.s3[0] = w.s3[0];
  t.s3[1] = w.s3[1];
  t.n = n;
  t.x0 = w.x0;
  t.xlogit = w.xlogit;
  t.local_logits = w.local_logits;
  t.out = out;
  t.taps = nullptr;
  t.tap_stride = m->args.nbuf;
  t.stamps = nullptr;
  t.status = nullptr;
  t.unit_counter = t.wave ? w.counters + 3 : nullptr;
  return launch_snv_towers(m, t, m->lds_split[3], stream);
}

// INDEL model (UNet_Small) eval-mode forward: host-side folding + layer program over the generic conv1d kernel.
// Reference: MuRaL/model/model_indel.py:6-19 (ConvBlock), :21-176 (UNet_Small).
//   * every BatchNorm follows its conv, so in eval mode it folds exactly into that conv's weights and bias;
//   * the strand-symmetrising input layer conv(x) + flip_L(conv(flip_{C,L}(x))) (:154-155) is ONE conv with weights
//     W[o][c][k] + W[o][3-c][K-1-k] (the second term's channel/length flips move onto the weights);
//   * out_fc = BN -> Dropout -> Linear -> Softplus on the global max: the BN folds into the Linear.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "conv1d.h"

namespace mural {
namespace {

static_assert(sizeof(MuralIndelShape) > 0, "header");
constexpr int INDEL_LEVELS = 6;
// positions per pass of the layer program (1.8 MB of activation scratch each, two passes in flight: 15 GB).  4096 instead of 2048
// (round 6): 964 -> 989 k positions/s -- half as many launches, each twice as long, and the largest tensor (4096 x 8 x 8000 floats) still
// within the 2^31-byte reach of the barrier-free kernels' buffer offsets.  (MURAL_INDEL_CHUNK: A/B switch of the debug flavour, 256 .. 4096)
constexpr int INDEL_CHUNK_DEFAULT = 4096;
static int indel_chunk() {
  const char* e = dev_env("MURAL_INDEL_CHUNK");
  const int c = e ? atoi(e) : INDEL_CHUNK_DEFAULT;
  return c >= 256 && c <= 4096 ? c : INDEL_CHUNK_DEFAULT;
}
#define INDEL_CHUNK indel_chunk()

struct FoldedConv { size_t w, b; int Cin, Cout, K; };   // offsets into the blob; w laid out [Cin][K][Cout]

struct Blob {
  std::vector<float> host;
  size_t alloc(size_t n) {
    size_t off = (host.size() + 63) & ~size_t(63);
    host.resize(off + n, 0.f);
    return off;
  }
};

void bn_scale_shift(const MuralBN& bn, int C, float eps, std::vector<double>& s, std::vector<double>& t) {
  s.resize(C);
  t.resize(C);
  for (int c = 0; c < C; ++c) {
    s[c] = (double)bn.weight[c] / std::sqrt((double)bn.running_var[c] + (double)eps);
    t[c] = (double)bn.bias[c] - (double)bn.running_mean[c] * s[c];
  }
}

// conv (weight [Cout][Cin][K], optional bias) followed by an optional BN -> [Cin][K][Cout] weights + bias
FoldedConv fold_conv(Blob& B, const float* W, const float* bias, const MuralBN* bn, int Cout, int Cin, int K, float eps,
                     bool symmetrise = false) {
  FoldedConv f{B.alloc((size_t)Cin * K * Cout), B.alloc(Cout), Cin, Cout, K};
  std::vector<double> s(Cout, 1.0), t(Cout, 0.0);
  if (bn) bn_scale_shift(*bn, Cout, eps, s, t);
  for (int co = 0; co < Cout; ++co) {
    for (int ci = 0; ci < Cin; ++ci)
      for (int k = 0; k < K; ++k) {
        double w = W[((size_t)co * Cin + ci) * K + k];
        if (symmetrise) w += W[((size_t)co * Cin + (Cin - 1 - ci)) * K + (K - 1 - k)];
        B.host[f.w + ((size_t)ci * K + k) * Cout + co] = (float)(w * s[co]);
      }
    const double b = bias ? bias[co] : 0.0;
    B.host[f.b + co] = (float)(symmetrise ? 2.0 * (b * s[co] + t[co]) : b * s[co] + t[co]);
  }
  return f;
}

bool bn_ok(const MuralBN& b) { return b.weight && b.bias && b.running_mean && b.running_var; }

}  // namespace
}  // namespace mural

using namespace mural;

struct MuralIndelModel {
  MuralIndelShape shape;
  int ch[INDEL_LEVELS], len[INDEL_LEVELS];
  FoldedConv sym, up_l[INDEL_LEVELS], up5[INDEL_LEVELS], up1[INDEL_LEVELS];
  FoldedConv dn_l[INDEL_LEVELS - 1], dn5[INDEL_LEVELS - 1], dn1[INDEL_LEVELS - 1], out1, out2;
  FoldedConv dn_lp[INDEL_LEVELS - 1];     // polyphase form of dn_l (conv of the upsampled tensor), K = 0: not built
  int dn_lp_pad[INDEL_LEVELS - 1];
  size_t front_pw;                        // level-0 decoder front in the block kernel's polyphase layout [4][Cf][3][C]; 0: not built
  // the layer in front of the U-Net per input symbol (ConvBlockArgs::symtab): [15][sym_taps][4] + bias [4]; the folded
  // strand-symmetrising conv with use_reverse (sym_taps = its kernel size), the one-hot columns themselves otherwise (sym_taps = 1)
  size_t symtab, sym_bias;
  int sym_taps;
  size_t e0_t3, e0_t1, e0_bias;   // the composed front of the first level's persistent kernel (indel_enc0_compose); e0_t3 == 0: not built
  size_t fc_w, fc_b;     // [n_class][C0] with the BN folded, [n_class]
  float* blob;
  size_t blob_floats;
  size_t per_pos_floats;  // activation scratch per position
};

namespace mural {
// softplus(fc(max features)) per (row, class)
// feat: [n][parts][C] partial maxima over positions (parts = 1: already the row maximum)
__global__ void indel_head_kernel(const float* __restrict__ feat, int64_t n, int parts, int C, int n_class,
                                  const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n * n_class) return;
  const int64_t row = i / n_class;
  const int k = (int)(i - row * n_class);
  float acc = b[k];
  for (int c = 0; c < C; ++c) {
    float m = feat[(row * parts) * C + c];
    for (int p = 1; p < parts; ++p) m = fmaxf(m, feat[(row * parts + p) * C + c]);
    acc = fmaf(w[k * C + c], m, acc);
  }
  out[i] = acc > 20.f ? acc : log1pf(expf(acc));
}

// The same with one wave per row (C a power of two <= 64, n_class <= 64): the row's parts x C partial maxima are one coalesced sweep
// (lane l always meets channel l % C), the lanes of a channel fold by shuffles, lane k < n_class finishes class k.  The thread-per-
// (row, class) form above walks 33 x 8 strided loads per thread on 64 workgroups: 37 us per 2048 rows of the shipped geometry.
__global__ __launch_bounds__(256) void indel_head_wave_kernel(const float* __restrict__ feat, int64_t n, int parts, int C, int n_class,
                                                              const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* f = feat + row * parts * C;
  const int total = parts * C;
  float m = -INFINITY;
  for (int i = lane; i < total; i += 64) m = fmaxf(m, f[i]);
  for (int off = 32; off >= C; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));      // lanes l, l + C, l + 2 C, ...: one channel
  float acc = lane < n_class ? b[lane] : 0.f;
  for (int c = 0; c < C; ++c) {
    const float mc = __shfl(m, c, 64);
    if (lane < n_class) acc = fmaf(w[lane * C + c], mc, acc);
  }
  if (lane < n_class) out[row * n_class + lane] = acc > 20.f ? acc : log1pf(expf(acc));
}
}  // namespace mural

extern "C" int mural_indel_model_create(const MuralIndelShape* shape, const MuralIndelParams* hp, MuralIndelModel** out) {
  MURAL_REQUIRE(shape && hp && out, "NULL argument");
  const MuralIndelShape& sh = *shape;
  MURAL_REQUIRE(sh.n_class >= 1 && sh.n_class <= 64, "n_class out of range: %d", sh.n_class);
  MURAL_REQUIRE(sh.channels >= 4 && sh.channels % 4 == 0, "CNN_out_channels must be a positive multiple of 4, got %d", sh.channels);
  MURAL_REQUIRE(sh.ksize >= 1 && (sh.ksize & 1), "CNN_kernel_size must be odd, got %d", sh.ksize);
  MURAL_REQUIRE(sh.length >= 1, "bad input length %d", sh.length);
  MuralIndelModel* m = new MuralIndelModel();
  std::memset(m, 0, sizeof(*m));
  m->shape = sh;
  const int K = sh.ksize, pad = (K - 1) / 2;
  int L = sh.length;
  for (int i = 0; i < INDEL_LEVELS; ++i) {
    MURAL_REQUIRE(sh.down[i] >= 1, "down_list entries must be >= 1");
    m->ch[i] = sh.channels * (i + 1);
    L = (L + 2 * pad - K) / sh.down[i] + 1;
    m->len[i] = L;
  }
  for (int i = INDEL_LEVELS - 1; i >= 1; --i) {   // nn.Upsample(scale_factor=down[i]) must land on the skip's length
    if (m->len[i] * sh.down[i] != m->len[i - 1]) {
      set_error("input length %d is not compatible with down_list: level %d upsamples %d x %d != %d (the reference fails "
                "at model_indel.py:170)", sh.length, i, m->len[i], sh.down[i], m->len[i - 1]);
      delete m;
      return MURAL_E_INVALID;
    }
  }
  const float eps = sh.bn_eps;
  Blob B;
  bool ok = true;
  if (sh.use_reverse) {
    ok = ok && hp->sym.conv.weight && hp->sym.conv.bias && bn_ok(hp->sym.bn);
    if (ok) m->sym = fold_conv(B, hp->sym.conv.weight, hp->sym.conv.bias, &hp->sym.bn, 4, 4, K, eps, /*symmetrise=*/true);
  }
  for (int i = 0; i < INDEL_LEVELS && ok; ++i) {
    const int cin = i == 0 ? 4 : m->ch[i - 1], c = m->ch[i];
    ok = hp->up_l[i].conv.weight && hp->up_l[i].conv.bias && bn_ok(hp->up_l[i].bn) && hp->up_b[i].conv5_w &&
         bn_ok(hp->up_b[i].bn1) && hp->up_b[i].conv1_w && bn_ok(hp->up_b[i].bn2);
    if (!ok) break;
    m->up_l[i] = fold_conv(B, hp->up_l[i].conv.weight, hp->up_l[i].conv.bias, &hp->up_l[i].bn, c, cin, K, eps);
    m->up5[i] = fold_conv(B, hp->up_b[i].conv5_w, nullptr, &hp->up_b[i].bn1, 2 * c, c, 5, eps);
    m->up1[i] = fold_conv(B, hp->up_b[i].conv1_w, nullptr, &hp->up_b[i].bn2, c, 2 * c, 1, eps);
  }
  for (int j = 0; j < INDEL_LEVELS - 1 && ok; ++j) {
    const int cin = m->ch[INDEL_LEVELS - 1 - j], c = m->ch[INDEL_LEVELS - 2 - j];
    ok = hp->down_l[j].conv.weight && hp->down_l[j].conv.bias && bn_ok(hp->down_l[j].bn) && hp->down_b[j].conv5_w &&
         bn_ok(hp->down_b[j].bn1) && hp->down_b[j].conv1_w && bn_ok(hp->down_b[j].bn2);
    if (!ok) break;
    m->dn_l[j] = fold_conv(B, hp->down_l[j].conv.weight, hp->down_l[j].conv.bias, &hp->down_l[j].bn, c, cin, K, eps);
    {   // Upsample(scale) + Conv1d(k): each output phase only sees 3 (scale 4 / 5) or 5 (scale 2) source columns -- the taps that
        // share a source column are summed once here (conv1d_phase_weights) and the layer runs as a GEMM with Cout * scale rows
      const int up = sh.down[INDEL_LEVELS - 1 - j];
      if (up > 1) {
        std::vector<float> pw;
        int kj = 0, padj = 0;
        conv1d_phase_weights(B.host.data() + m->dn_l[j].w, cin, K, c, up, &pw, &kj, &padj);
        m->dn_lp[j] = FoldedConv{B.alloc(pw.size()), m->dn_l[j].b, cin, c, kj};
        std::copy(pw.begin(), pw.end(), B.host.begin() + m->dn_lp[j].w);
        m->dn_lp_pad[j] = padj;
        if (j == INDEL_LEVELS - 2 && up == 4 && kj == 3 && padj == 1) {   // the fused front of the level-0 block kernel
          m->front_pw = B.alloc((size_t)4 * cin * 3 * c);
          for (int p = 0; p < 4; ++p)
            for (int ci = 0; ci < cin; ++ci)
              for (int d = 0; d < 3; ++d)
                for (int co = 0; co < c; ++co)
                  B.host[m->front_pw + (((size_t)p * cin + ci) * 3 + d) * c + co] = pw[((size_t)ci * 3 + d) * c * 4 + (size_t)co * 4 + p];
        }
      }
    }
    m->dn5[j] = fold_conv(B, hp->down_b[j].conv5_w, nullptr, &hp->down_b[j].bn1, 2 * c, c, 5, eps);
    m->dn1[j] = fold_conv(B, hp->down_b[j].conv1_w, nullptr, &hp->down_b[j].bn2, c, 2 * c, 1, eps);
  }
  const int C0 = m->ch[0];
  ok = ok && hp->out1.weight && hp->out1.bias && bn_ok(hp->out_bn) && hp->out2.weight && hp->out2.bias && bn_ok(hp->fc_bn) &&
       hp->fc.weight && hp->fc.bias;
  if (!ok) {
    set_error("INDEL parameter pointer is NULL");
    delete m;
    return MURAL_E_INVALID;
  }
  m->out1 = fold_conv(B, hp->out1.weight, hp->out1.bias, &hp->out_bn, C0, C0, 1, eps);
  m->out2 = fold_conv(B, hp->out2.weight, hp->out2.bias, nullptr, C0, C0, 1, eps);
  {
    std::vector<double> s, t;
    bn_scale_shift(hp->fc_bn, C0, eps, s, t);
    m->fc_w = B.alloc((size_t)sh.n_class * C0);
    m->fc_b = B.alloc(sh.n_class);
    for (int k = 0; k < sh.n_class; ++k) {
      double b = hp->fc.bias[k];
      for (int c = 0; c < C0; ++c) {
        const double w = hp->fc.weight[k * C0 + c];
        B.host[m->fc_w + k * C0 + c] = (float)(w * s[c]);
        b += w * t[c];
      }
      B.host[m->fc_b + k] = (float)b;
    }
  }
  {
    // symbol table of the input layer: column value of channel c for symbol s = 1 / |set(s)| if c is in the symbol's base set
    // (preprocessing.py:758-772: one-hot, pairs 0.5, triples 1/3, N 0.25)
    static const uint8_t kSet[15] = {1, 2, 4, 8, 15, 5, 10, 3, 6, 9, 12, 14, 13, 11, 7};   // A C G T N R Y M S W K B D H V (bit c = base c)
    const int taps = sh.use_reverse ? K : 1;
    m->sym_taps = taps;
    m->symtab = B.alloc((size_t)15 * taps * 4);
    m->sym_bias = B.alloc(4);
    for (int sy = 0; sy < 15; ++sy) {
      const int members = __builtin_popcount(kSet[sy]);
      const float v = members == 1 ? 1.0f : members == 2 ? 0.5f : members == 3 ? (float)(1.0 / 3.0) : 0.25f;
      for (int k = 0; k < taps; ++k)
        for (int o = 0; o < 4; ++o) {
          double acc = 0.0;
          for (int c = 0; c < 4; ++c) {
            if (!((kSet[sy] >> c) & 1)) continue;
            acc += sh.use_reverse ? (double)v * B.host[m->sym.w + ((size_t)c * K + k) * 4 + o] : (c == o ? (double)v : 0.0);
          }
          B.host[m->symtab + ((size_t)sy * taps + k) * 4 + o] = (float)acc;
        }
    }
    for (int o = 0; o < 4; ++o) B.host[m->sym_bias + o] = sh.use_reverse ? B.host[m->sym.b + o] : 0.f;
    if (sh.down[0] == 1 && C0 == 8 && K == 7) {
      std::vector<float> t3, t1, bias;
      indel_enc0_compose(B.host.data() + m->up_l[0].w, B.host.data() + m->up_l[0].b, B.host.data() + m->symtab, B.host.data() + m->sym_bias,
                         taps, &t3, &t1, &bias);
      m->e0_t3 = B.alloc(t3.size());
      m->e0_t1 = B.alloc(t1.size());
      m->e0_bias = B.alloc(bias.size());
      std::copy(t3.begin(), t3.end(), B.host.begin() + m->e0_t3);
      std::copy(t1.begin(), t1.end(), B.host.begin() + m->e0_t1);
      std::copy(bias.begin(), bias.end(), B.host.begin() + m->e0_bias);
    }
  }
  // activation scratch per position: S | E_0..E_5 | T1 | T2 | H | SP | M | X (one-hot window of the packed entry's fallback)
  size_t per = (size_t)4 * sh.length + (size_t)4 * sh.length;
  for (int i = 0; i < INDEL_LEVELS; ++i) per += (size_t)m->ch[i] * m->len[i];
  const size_t big = (size_t)m->ch[0] * m->len[0];
  size_t tmax = 0, hmax = 0;
  for (int i = 0; i < INDEL_LEVELS; ++i) {
    tmax = std::max(tmax, (size_t)m->ch[i] * m->len[i]);
    hmax = std::max(hmax, (size_t)2 * m->ch[i] * m->len[i]);
  }
  per += 2 * tmax + hmax + big + (size_t)C0 * convblock_tiles(m->len[0], true);   // M: per-tile maxima
  m->per_pos_floats = per;
  m->blob_floats = B.host.size();
  if (hipError_t e = hipMalloc(&m->blob, m->blob_floats * 4); e != hipSuccess) {
    set_error("hipMalloc of %zu bytes for the folded INDEL weights failed: %s", m->blob_floats * 4, hipGetErrorString(e));
    delete m;
    return MURAL_E_RUNTIME;
  }
  if (hipMemcpy(m->blob, B.host.data(), m->blob_floats * 4, hipMemcpyHostToDevice) != hipSuccess) {
    set_error("hipMemcpy of the folded INDEL weights failed");
    (void)hipFree(m->blob);
    delete m;
    return MURAL_E_RUNTIME;
  }
  *out = m;
  return MURAL_OK;
}

extern "C" void mural_indel_model_destroy(MuralIndelModel* m) {
  if (!m) return;
  if (m->blob) (void)hipFree(m->blob);
  delete m;
}

// chunks in flight: even / odd chunks on two streams; MURAL_INDEL_LANES=3 (experiment): a third one on the second side stream
static int indel_lanes(int64_t n) {
  static const int want = dev_env("MURAL_INDEL_LANES") ? atoi(dev_env("MURAL_INDEL_LANES")) : 2;
  const int64_t chunks = (n + INDEL_CHUNK - 1) / INDEL_CHUNK;
  return (int)std::max<int64_t>(1, std::min<int64_t>(std::min(want, 3), chunks));
}
constexpr size_t INDEL_WS_REGIONS = INDEL_LEVELS + 7;   // S | E[levels] | T1 | T2 | H | SP | M | X

extern "C" size_t mural_indel_workspace_bytes(const MuralIndelModel* m, int64_t n) {
  if (!m || n <= 0) return 256;
  // two chunks in flight (one per stream, see mural_indel_forward_dense) once there is more than one
  const size_t lanes = (size_t)indel_lanes(n);
  // (+ the validation guard behind each of a lane's INDEL_WS_REGIONS regions: 0 outside the tests, common.h)
  return lanes * ((size_t)std::min<int64_t>(n, INDEL_CHUNK) * m->per_pos_floats * 4 + INDEL_WS_REGIONS * ws_guard_bytes()) + 4096;
}

static int run_conv(const MuralIndelModel* m, const FoldedConv& f, const float* in, int B, int Lin, float* out, int Lout,
                    int stride, int up, int act, const float* res1, const float* res2, hipStream_t stream) {
  Conv1dArgs a;
  std::memset(&a, 0, sizeof(a));
  a.in = in; a.wt = m->blob + f.w; a.bias = m->blob + f.b; a.out = out;
  a.B = B; a.Cin = f.Cin; a.Lin = Lin; a.Cout = f.Cout; a.Lout = Lout;
  a.K = f.K; a.stride = stride; a.pad = (f.K - 1) / 2; a.up = up;
  a.act = act; a.res1 = res1; a.res2 = res2;
  return launch_conv1d(a, stream);
}

// decoder front: Conv1d(k) of the nearest-neighbour upsampled tensor; polyphase GEMM when built and served by the MFMA kernel
static int run_upconv(const MuralIndelModel* m, int j, const float* in, int B, int Lin, float* out, int Lout, int up, hipStream_t stream) {
  const FoldedConv& fp = m->dn_lp[j];
  if (fp.K > 0 && Lin * up == Lout) {
    Conv1dArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = in; a.wt = m->blob + fp.w; a.bias = m->blob + fp.b; a.out = out;
    a.B = B; a.Cin = fp.Cin; a.Lin = Lin; a.Cout = fp.Cout; a.Lout = Lout;
    a.K = fp.K; a.stride = 1; a.pad = m->dn_lp_pad[j]; a.up = 1; a.phases = up;
    a.act = ACT_NONE;
    static const bool use_direct = !(dev_env("MURAL_CONV1D_DIRECT") && atoi(dev_env("MURAL_CONV1D_DIRECT")) == 0);
    if (use_direct && (int64_t)B * Lin >= 32768 && conv1d_direct_poly_supported(a)) return launch_conv1d_direct_poly(a, stream);
    if (conv1d_mfma_supported(a)) return launch_conv1d_mfma(a, stream);
  }
  return run_conv(m, m->dn_l[j], in, B, Lin, out, Lout, 1, up, ACT_NONE, nullptr, nullptr, stream);
}

// ConvBlock: x + BN(1x1(SiLU(BN(k5(x))))) [+ skip]; fused kernel when instantiated for the channel count
static bool block_fusable(const FoldedConv& f5, const FoldedConv& f1, int L) {
  // (a lane of the fused kernel owns one position: rows shorter than half a workgroup keep the per-layer kernels)
  return L >= 128 && convblock_supported(f5.Cin) && f5.K == 5 && f1.K == 1 && f5.Cout == 2 * f5.Cin && f1.Cout == f5.Cin;
}

// front: the stride-1 k=7 conv (with nearest-neighbour upsampling `up` of its input fin [B][ff->Cin][Lf]) that produces the
// block input x; when fusable, x is never materialised (pass x = nullptr then)
static bool front_fusable(const FoldedConv& ff, int up, int C) {
  // measured (rocprofv3, 2048 positions): a win at 8 channels (594 us vs 491 + 450 us for the level-0 encoder), a loss at
  // 16 / 24 channels where the block kernel's LDS footprint leaves too few waves to cover the front conv's latencies
  return C == 8 && ff.K == 7 && ff.Cout == C && up >= 1 && ff.Cin * (262 / up + 3) <= 2048;
}

struct GenomeSrc {      // packed-genome source of the first level's front input (ConvBlockArgs::symtab)
  const MuralGenome* g = nullptr;
  const int64_t* pos = nullptr;
  const uint8_t* strand = nullptr;
  int off = 0;
  const uint8_t* sym = nullptr;      // or: one symbol byte per column of the dense windows (ConvBlockArgs::sym_in; the front input is the dense tensor)
};

static int run_block(const MuralIndelModel* m, const FoldedConv& f5, const FoldedConv& f1, const float* x, int B, int L,
                     float* H, float* out, const float* skip, hipStream_t stream, float* tail_max = nullptr,
                     const FoldedConv* ff = nullptr, const float* fin = nullptr, int up = 1, const GenomeSrc* gs = nullptr,
                     int* tiles_out = nullptr, const FoldedConv* down = nullptr, float* down_out = nullptr, int down_L = 0) {
  if (block_fusable(f5, f1, L)) {
    ConvBlockArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x = x; a.w5 = m->blob + f5.w; a.b5 = m->blob + f5.b; a.w1 = m->blob + f1.w; a.b1 = m->blob + f1.b;
    a.res2 = skip; a.out = out; a.B = B; a.C = f5.Cin; a.L = L;
    if (ff) {
      a.f_in = fin; a.f_w = m->blob + ff->w; a.f_b = m->blob + ff->b; a.Cf = ff->Cin; a.Lf = L / up; a.f_up = up;
      if (up == 4 && m->front_pw && ff == &m->dn_l[INDEL_LEVELS - 2]) a.f_pw = m->blob + m->front_pw;
      if (gs && (gs->g || gs->sym)) {
        if (gs->g) { a.genome = *gs->g; a.g_pos = gs->pos; a.g_strand = gs->strand; a.g_off = gs->off; }
        a.sym_in = gs->sym;
        a.symtab = m->blob + m->symtab; a.sym_bias = m->blob + m->sym_bias; a.sym_taps = m->sym_taps;
        if (m->e0_t3) {
          a.e0_t3 = m->blob + m->e0_t3; a.e0_t1 = m->blob + m->e0_t1; a.e0_bias = m->blob + m->e0_bias;
          if (down) {
            a.d_w = m->blob + down->w; a.d_b = m->blob + down->b; a.d_out = down_out; a.d_L = down_L;
          }
        }
      }
    }
    if (tail_max) {   // out_conv (1x1, BN, ReLU, 1x1, Softplus) + max over positions ride on the last decoder block
      a.ta_w = m->blob + m->out1.w; a.ta_b = m->blob + m->out1.b; a.tb_w = m->blob + m->out2.w; a.tb_b = m->blob + m->out2.b;
      a.tail_max = tail_max;
    }
    if (tiles_out) *tiles_out = convblock_tiles_of(a);
    return launch_convblock(a, stream);
  }
  MURAL_REQUIRE(!tail_max && !ff, "internal: front / tail fusion requested for an unfusable block");
  if (f5.K == 5 && f1.K == 1 && f5.Cout == 2 * f5.Cin && f1.Cout == f5.Cin) {      // short rows: one row per workgroup pass (convblock_deep.hip)
    ConvBlockArgs a;
    std::memset(&a, 0, sizeof(a));
    a.x = x; a.w5 = m->blob + f5.w; a.b5 = m->blob + f5.b; a.w1 = m->blob + f1.w; a.b1 = m->blob + f1.b;
    a.res2 = skip; a.out = out; a.B = B; a.C = f5.Cin; a.L = L;
    if (convblock_deep_supported(a)) return launch_convblock_deep(a, stream);
  }
  if (int rc = run_conv(m, f5, x, B, L, H, L, 1, 1, ACT_SILU, nullptr, nullptr, stream)) return rc;
  return run_conv(m, f1, H, B, L, out, L, 1, 1, ACT_NONE, x, skip, stream);
}

extern "C" int mural_encode_onehot(const MuralGenome* g, const int64_t* pos, const uint8_t* strand, int64_t n, int32_t radius,
                                   int32_t indel, float* out, void* stream);
namespace mural { int launch_dense_to_symbols(const float* x, int64_t n, int L, uint8_t* sym, int32_t* status, hipStream_t stream, int bad_code = -1); }

// distal_x != nullptr: dense entry; otherwise the windows come from the packed genome (genome, pos, strand; window
// [pos - radius + 1, pos + radius], preprocessing.py:564-566)
static int indel_forward_impl(const MuralIndelModel* m, const float* distal_x, const MuralGenome* genome, const int64_t* pos,
                              const uint8_t* strand, int64_t n, float* out, void* workspace, size_t workspace_bytes, void* stream_) {
  MURAL_REQUIRE(m, "model handle is NULL");
  MURAL_REQUIRE(n >= 0, "negative batch");
  if (n == 0) return MURAL_OK;
  MURAL_REQUIRE((distal_x || (genome && pos && strand)) && out, "input/out is NULL");
  if (!workspace || workspace_bytes < mural_indel_workspace_bytes(m, n)) {
    set_error("workspace too small: need %zu bytes, got %zu", mural_indel_workspace_bytes(m, n), workspace_bytes);
    return MURAL_E_WORKSPACE;
  }
  hipStream_t main_stream = (hipStream_t)stream_;
  const MuralIndelShape& sh = m->shape;
  const int Lx = sh.length, C0 = m->ch[0];
  // Chunks are independent, and the ~45 launches of one chunk include a dozen on the deep levels (rows of 80 / 16 / 8 columns) that
  // leave most of the chip idle: even chunks run on the caller's stream, odd ones on a side stream with their own half of the
  // workspace, so the small kernels of one chunk fill the gaps of the other.
  const int lanes = indel_lanes(n);
  SideStreamHold ss;      // holds the device's side streams until this call has joined them again
  if (lanes >= 2) {
    if (int rc = ss.acquire()) return rc;
    if (int rc = ss->fork(main_stream, lanes == 3)) return rc;
  }
  int rc_all = MURAL_OK;
  int64_t chunk_no = 0;
  for (int64_t c0 = 0; c0 < n && rc_all == MURAL_OK; c0 += INDEL_CHUNK, ++chunk_no) {
    const int B = (int)std::min<int64_t>(INDEL_CHUNK, n - c0);
    const int lane = (int)(chunk_no % lanes);
    hipStream_t stream = lane == 0 ? main_stream : lane == 1 ? ss->side : ss->side2;
    const size_t guard_floats = ws_guard_bytes() / 4;
    float* p = static_cast<float*>(workspace) + (size_t)lane * (INDEL_CHUNK * m->per_pos_floats + INDEL_WS_REGIONS * guard_floats);
    if (chunk_no < lanes) {
      if (chunk_no == 0) ws_layout_reset();
    }
    rc_all = [&]() -> int {
    auto take = [&](size_t per_pos) {
      float* r = p;
      if (chunk_no < lanes) ws_layout_add((size_t)(r - static_cast<float*>(workspace)) * 4, per_pos * (size_t)B * 4);
      p += per_pos * (size_t)B + guard_floats;
      return r;
    };
    float* S = take((size_t)4 * Lx);
    float* E[INDEL_LEVELS];
    for (int i = 0; i < INDEL_LEVELS; ++i) E[i] = take((size_t)m->ch[i] * m->len[i]);
    size_t tmax = 0, hmax = 0;
    for (int i = 0; i < INDEL_LEVELS; ++i) {
      tmax = std::max(tmax, (size_t)m->ch[i] * m->len[i]);
      hmax = std::max(hmax, (size_t)2 * m->ch[i] * m->len[i]);
    }
    float* T1 = take(tmax);
    float* T2 = take(tmax);
    float* H = take(hmax);
    float* SP = take((size_t)C0 * m->len[0]);
    int mparts = 1;
    float* M = take((size_t)C0 * convblock_tiles(m->len[0], true));
    float* X = take((size_t)4 * Lx);
    int rc = MURAL_OK;
    // packed entry: the first level's block kernel decodes its span of the window and evaluates the input layer per symbol
    // (ConvBlockArgs::symtab) when that kernel applies; otherwise the window is materialised first
    GenomeSrc gs;
    const bool first_fused = sh.down[0] == 1 && block_fusable(m->up5[0], m->up1[0], m->len[0]) && front_fusable(m->up_l[0], 1, m->ch[0]);
    const float* x = nullptr;
    if (distal_x) {
      x = distal_x + (size_t)c0 * 4 * Lx;
      // dense one-hot windows (what the reference's loader yields): classified into one symbol byte per column and taken by the same
      // persistent table-driven first level as the packed entry; columns that are no MuRaL symbol are evaluated from their floats there
      // (MURAL_INDEL_DENSE_SYMBOLS=0: the input layer and the first level as launches of their own on the dense tensor)
      const bool sym_off = (dev_env("MURAL_INDEL_DENSE_SYMBOLS") && atoi(dev_env("MURAL_INDEL_DENSE_SYMBOLS")) == 0) ||
                           (dev_env("MURAL_INDEL_ENC0") && atoi(dev_env("MURAL_INDEL_ENC0")) == 0) || dev_env("MURAL_DEBUG_CONVBLOCK_VALU") ||
                           dev_env("MURAL_CONVBLOCK8_VALU");
      if (first_fused && m->e0_t3 && (Lx & 3) == 0 && !sym_off) {
        if ((rc = launch_dense_to_symbols(x, B, Lx, reinterpret_cast<uint8_t*>(X), nullptr, stream, 16))) return rc;
        gs.sym = reinterpret_cast<const uint8_t*>(X);
      }
    } else if (first_fused && !dev_env("MURAL_DEBUG_INDEL_NO_GENOME_FRONT")) {
      gs.g = genome; gs.pos = pos + c0; gs.strand = strand + c0; gs.off = -(Lx / 2) + 1;
    } else {
      if ((rc = mural_encode_onehot(genome, pos + c0, strand + c0, B, Lx / 2, 1, X, stream))) return rc;
      x = X;
    }
    const float* cur = x;
    int Lcur = Lx;
    if (sh.use_reverse && !gs.g && !gs.sym) {
      if ((rc = run_conv(m, m->sym, cur, B, Lcur, S, Lcur, 1, 1, ACT_NONE, nullptr, nullptr, stream))) return rc;
      cur = S;
    }
    // the genome-fed first level also emits the second level's strided conv (8 -> 16, k = 7, stride 4) where the shapes are the
    // persistent kernel's (indel_level0.hip); MURAL_INDEL_ENC0_DOWN=0: the conv as a launch of its own
    const bool down_off = (dev_env("MURAL_INDEL_ENC0_DOWN") && atoi(dev_env("MURAL_INDEL_ENC0_DOWN")) == 0) ||
                                 (dev_env("MURAL_INDEL_ENC0") && atoi(dev_env("MURAL_INDEL_ENC0")) == 0);
    const bool emit_down = (gs.g || gs.sym) && m->e0_t3 && !down_off && sh.down[1] == 4 && m->up_l[1].K == 7 && m->up_l[1].Cin == 8 && m->up_l[1].Cout == 16 &&
                           (m->len[0] & 3) == 0 && m->len[1] == (m->len[0] - 1) / 4 + 1 && !dev_env("MURAL_DEBUG_CONVBLOCK_VALU") &&
                           !dev_env("MURAL_CONVBLOCK8_VALU");
    for (int i = 0; i < INDEL_LEVELS; ++i) {     // encoder: strided conv+BN, then ConvBlock (x + BN(1x1(SiLU(BN(k5)))))
      const int Li = m->len[i];
      if (sh.down[i] == 1 && block_fusable(m->up5[i], m->up1[i], Li) && front_fusable(m->up_l[i], 1, m->ch[i])) {
        if ((rc = run_block(m, m->up5[i], m->up1[i], nullptr, B, Li, H, E[i], nullptr, stream, nullptr, &m->up_l[i], cur, 1,
                            i == 0 ? &gs : nullptr, nullptr, (i == 0 && emit_down) ? &m->up_l[1] : nullptr, T1, m->len[1])))
          return rc;
      } else {
        // short rows of 32 channels (the fourth level): the strided k = 7 conv is the front of the block's own launch (convblock_deep.hip)
        ConvBlockArgs da;
        std::memset(&da, 0, sizeof(da));
        const FoldedConv &fd = m->up_l[i], &f5 = m->up5[i], &f1 = m->up1[i];
        da.f_in = cur; da.f_w = m->blob + fd.w; da.f_b = m->blob + fd.b; da.Cf = fd.Cin; da.Lf = Lcur; da.f_up = 1; da.f_stride = sh.down[i];
        da.w5 = m->blob + f5.w; da.b5 = m->blob + f5.b; da.w1 = m->blob + f1.w; da.b1 = m->blob + f1.b;
        da.out = E[i]; da.B = B; da.C = f5.Cin; da.L = Li;
        if (i >= 2 && fd.K == 7 && fd.Cout == f5.Cin && f5.K == 5 && f1.K == 1 && f5.Cout == 2 * f5.Cin && f1.Cout == f5.Cin &&
            convblock_deep_supported(da)) {
          if ((rc = launch_convblock_deep(da, stream))) return rc;
        } else {
        if (!(i == 1 && emit_down))      // (else: T1 already holds this level's strided conv)
          if ((rc = run_conv(m, m->up_l[i], cur, B, Lcur, T1, Li, sh.down[i], 1, ACT_NONE, nullptr, nullptr, stream))) return rc;
        if ((rc = run_block(m, m->up5[i], m->up1[i], T1, B, Li, H, E[i], nullptr, stream))) return rc;
        }
      }
      cur = E[i];
      Lcur = Li;
    }
    float* dec = T2;
    bool tail_done = false;
    for (int j = 0; j < INDEL_LEVELS - 1; ++j) {   // decoder: upsample, conv+BN, ConvBlock, + encoder skip
      const int lvl = INDEL_LEVELS - 2 - j;
      const int Li = m->len[lvl];
      const bool fuse_tail = lvl == 0 && block_fusable(m->dn5[j], m->dn1[j], Li);
      const int up = sh.down[lvl + 1];
      if (block_fusable(m->dn5[j], m->dn1[j], Li) && front_fusable(m->dn_l[j], up, m->ch[lvl]) && Lcur * up == Li) {
        int tiles = 0;
        if ((rc = run_block(m, m->dn5[j], m->dn1[j], nullptr, B, Li, H, dec, E[lvl], stream, fuse_tail ? M : nullptr, &m->dn_l[j], cur,
                            up, nullptr, &tiles)))
          return rc;
        if (fuse_tail) mparts = tiles;
      } else {
        if ((rc = run_upconv(m, j, cur, B, Lcur, T1, Li, up, stream))) return rc;
        int tiles = 0;
        if ((rc = run_block(m, m->dn5[j], m->dn1[j], T1, B, Li, H, dec, E[lvl], stream, fuse_tail ? M : nullptr, nullptr, nullptr, 1, nullptr, &tiles)))
          return rc;
        if (fuse_tail) mparts = tiles;
      }
      tail_done = fuse_tail;
      cur = dec;
      Lcur = Li;
      dec = (cur == T2) ? SP : T2;   // ping-pong between two level-0-sized buffers
    }
    if (!tail_done) {
      if ((rc = run_conv(m, m->out1, cur, B, Lcur, H, Lcur, 1, 1, ACT_RELU, nullptr, nullptr, stream))) return rc;
      float* sp = (cur == SP) ? T2 : SP;
      if ((rc = run_conv(m, m->out2, H, B, Lcur, sp, Lcur, 1, 1, ACT_SOFTPLUS, nullptr, nullptr, stream))) return rc;
      if ((rc = launch_rowmax(sp, (int64_t)B * C0, Lcur, M, stream))) return rc;
    }
    const int64_t total = (int64_t)B * sh.n_class;
    if (C0 <= 64 && (C0 & (C0 - 1)) == 0 && sh.n_class <= 64)
      hipLaunchKernelGGL(indel_head_wave_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, M, (int64_t)B, tail_done ? mparts : 1, C0,
                         sh.n_class, m->blob + m->fc_w, m->blob + m->fc_b, out + c0 * sh.n_class);
    else
    hipLaunchKernelGGL(indel_head_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, M, (int64_t)B,
                       tail_done ? mparts : 1, C0,
                       sh.n_class, m->blob + m->fc_w, m->blob + m->fc_b, out + c0 * sh.n_class);
    MURAL_HIP_CHECK(hipGetLastError());
    return MURAL_OK;
    }();
  }
  if (lanes >= 2)
    if (int rc = ss->join(main_stream, lanes == 3)) return rc;     // also on an error: the side stream must not stay forked
  return rc_all;
}

// Replaces UNet_Small.forward(distal_input) (model_indel.py:151-176): distal_x dev float [n][4][length] -> out
// dev float [n][n_class] (positive Softplus scores; callers apply softmax / cross-entropy, run_predict.py:214).
extern "C" int mural_indel_forward_dense(const MuralIndelModel* m, const float* distal_x, int64_t n, float* out,
                                         void* workspace, size_t workspace_bytes, void* stream_) {
  MURAL_REQUIRE(n <= 0 || distal_x, "distal_x is NULL");
  return indel_forward_impl(m, distal_x, nullptr, nullptr, nullptr, n, out, workspace, workspace_bytes, stream_);
}

// The same from the packed genome: replaces seq_ohe_encoder (MuRaL/data/preprocessing.py:756-816, indel window :564-566) +
// UNet_Small.forward for the sites (pos, strand).  With the shipped geometry (down_list[0] == 1, 8 channels) the window is decoded
// inside the first level's kernel and the strand-symmetrising input conv is evaluated there per symbol: neither the one-hot
// window (128 KB per position at L = 8000) nor that conv's output exist in HBM.  The model's `length` must be even (2 * radius).
extern "C" int mural_indel_forward_packed(const MuralIndelModel* m, const MuralGenome* genome, const int64_t* pos, const uint8_t* strand,
                                          int64_t n, float* out, void* workspace, size_t workspace_bytes, void* stream_) {
  MURAL_REQUIRE(m, "model handle is NULL");
  MURAL_REQUIRE(genome && genome->packed2 && genome->nmask, "genome pointers must not be NULL");
  MURAL_REQUIRE(genome->n_amb == 0 || (genome->amb_pos && genome->amb_sym), "genome: n_amb > 0 needs amb_pos and amb_sym");
  MURAL_REQUIRE((m->shape.length & 1) == 0, "the indel window is 2 * distal_radius wide: model length %d is odd", m->shape.length);
  MURAL_REQUIRE(n <= 0 || (pos && strand), "pos/strand is NULL");
  return indel_forward_impl(m, nullptr, genome, pos, strand, n, out, workspace, workspace_bytes, stream_);
}

// Cross-position reuse for dense same-strand site lists (the reference has no counterpart beyond the
// window sharing of its encoders, MuRaL/data/preprocessing.py:602-610, :808-814).
//
// For two sites p, p' on the same strand the first conv stage of a tower (model_snv.py:473-479 mid, :496-499 large: conv1 ->
// maxpool1 -> RBs1 + skip) evaluates the SAME function of the genome wherever the windows overlap, as long as the pooled
// columns are phase-aligned (p = p' mod 15 for the large tower, mod 3 for the mid tower) and the column is far enough from a
// window edge: maxpool1's column q is centred on base (window start + stride * q), a k=3 conv on pooled columns is a conv with
// dilation `stride` on the base axis, and zero padding / -inf padding at the window edges reaches one pooled column further
// per conv layer.  So, per strand and genomic chunk:
//
//   A  reuse_rows_stage1_kernel   every base b: the pooled first-layer column centred on b (F), and the three window-specific
//                                 variants a window edge produces (first pooled column of a window starting at b, last pooled
//                                 column of a window ending at b) -- all from the same 3-mer tables as snv_stage1_kernel
//   B  reuse_rows_conv_kernel     the four ResBlock convs as dilated convs over the [base][32] rows (fp32 MFMA), 4 launches/tower
//   S  reuse_rows_pool_kernel     sliding maxpool2 + BN over the rows: S[b] = pooled column centred on b
//   E  snv_edge_kernel            per site: the 9 -> 5 pooled columns next to each window edge are recomputed exactly through
//                                 the four convs (same conv code, same tile layout as the per-window kernel: 18 columns per
//                                 site instead of 134 / 67), the interior of maxpool2 is gathered from S, and the pooled tile
//                                 s3 goes to the unchanged short-stage launches (snv_towers_fused<2>)
//
// Every arithmetic step is the per-window kernel's own (same tables, same A fragments, same k order, same epilogue maps), so
// the results agree to rounding of identical operation sequences; the parity tests hold both paths to 1e-5 on probabilities.
// Work per site: 2 x 19 columns x 4 layers instead of (135 + 68) x 4 -> 5.3x fewer MFMAs in the first conv stage.
#include <algorithm>
#include <cstdlib>

#include "snv_tower_conv.h"

namespace mural {
int launch_snv_towers(const MuralSnvModel* m, const SnvFwdArgs& a, size_t lds_bytes, hipStream_t stream);
int launch_snv_local(const LocalDev& L, const int64_t* cat, int64_t n, float* out, hipStream_t stream);

namespace {

constexpr int64_t RU_CHUNK_SPAN = 1 << 21;   // bases per row chunk (11 row arrays x 128 B x span = 2.9 GB)

// --------------------------------------------------------------------------------------------------------------- kernel A
struct RowsS1Args {
  MuralGenome genome;
  int neg;                  // 1: rows run along the reverse-complement strand
  int64_t t0;               // oriented coordinate of row 0 (oriented t = g on '+', length - 1 - g on '-')
  int64_t nb;               // rows
  const float* lut[2];      // lut | taps | bias0 blocks: 0 large, 1 mid
  int er_n;                 // interior conv columns of the large tower's last pooled column (12 at R = 1000)
  float *FL, *FM, *ELl, *ELr, *EMl;   // [nb][32]
};

constexpr int RS1_THREADS = 1024;
constexpr int RS1_TB = 1024;             // bases per tile
constexpr int RS1_HALO = 16;

__device__ __forceinline__ f32x4 rows_y(const float* lutS, const uint8_t* sym, const uint8_t* idx, int j, int cg) {
  const uint32_t k = idx[j];
  if (k != 255u) return ld4(lutS + k * 32u + 4u * cg);
  const float* tapS = lutS + SNV_LUT;
  f32x4 v = ld4(tapS + SNV_TAPS + 4 * cg);                      // same summation order as snv_stage1_kernel's per-tap path
  v += ld4(tapS + (0 * N_SYM + sym[j - 1]) * 32 + 4 * cg);
  v += ld4(tapS + (1 * N_SYM + sym[j]) * 32 + 4 * cg);
  v += ld4(tapS + (2 * N_SYM + sym[j + 1]) * 32 + 4 * cg);
  return v;
}
__device__ __forceinline__ f32x4 rows_y_pad(const float* lutS, uint32_t sl, uint32_t sc, uint32_t sr, int cg) {
  const float* tapS = lutS + SNV_LUT;
  f32x4 v = ld4(tapS + SNV_TAPS + 4 * cg);
  v += ld4(tapS + (0 * N_SYM + sl) * 32 + 4 * cg);
  v += ld4(tapS + (1 * N_SYM + sc) * 32 + 4 * cg);
  v += ld4(tapS + (2 * N_SYM + sr) * 32 + 4 * cg);
  return v;
}

__global__ __launch_bounds__(RS1_THREADS) void reuse_rows_stage1_kernel(const RowsS1Args a) {
  extern __shared__ __attribute__((aligned(16))) float rsmem[];
  float* lutL = rsmem;
  float* lutM = rsmem + SNV_LUTBLK;
  uint8_t* sym = reinterpret_cast<uint8_t*>(rsmem + 2 * SNV_LUTBLK);      // [TB + 2 HALO]
  uint8_t* idx = sym + RS1_TB + 2 * RS1_HALO;                              // [TB + 2 HALO]
  const int tid = threadIdx.x;
  for (int i = tid * 4; i < SNV_LUTBLK; i += RS1_THREADS * 4) {
    st4(lutL + i, ld4(a.lut[0] + i));
    st4(lutM + i, ld4(a.lut[1] + i));
  }
  const int64_t n_tiles = (a.nb + RS1_TB - 1) / RS1_TB;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t b0 = tile * RS1_TB;
    __syncthreads();                                                       // previous tile's readers are done (and the LUT landed)
    for (int j = tid; j < RS1_TB + 2 * RS1_HALO; j += RS1_THREADS) {
      const int64_t t = a.t0 + b0 + j - RS1_HALO;
      uint32_t s;
      if (a.neg) s = sym_complement(genome_sym_iupac(a.genome, a.genome.length - 1 - t));
      else s = genome_sym_iupac(a.genome, t);
      sym[j] = (uint8_t)s;
    }
    __syncthreads();
    for (int j = tid; j < RS1_TB + 2 * RS1_HALO; j += RS1_THREADS) {
      uint32_t k = 255u;
      if (j >= 1 && j + 1 < RS1_TB + 2 * RS1_HALO) {
        const uint32_t l = sym[j - 1], c = sym[j], r = sym[j + 1];
        if (l <= 4u && c <= 4u && r <= 4u) k = 25u * l + 5u * c + r;
      }
      idx[j] = (uint8_t)k;
    }
    __syncthreads();
    for (int task = tid; task < RS1_TB * 8; task += RS1_THREADS) {
      const int jb = task >> 3, cg = task & 7;
      const int64_t b = b0 + jb;
      if (b >= a.nb) continue;
      const int j = jb + RS1_HALO;
      // large tower: 15-wide window centred on b, and the two edge variants
      f32x4 lo7 = rows_y(lutL, sym, idx, j - 7, cg), hi7 = rows_y(lutL, sym, idx, j + 1, cg);
#pragma unroll
      for (int d = 2; d <= 7; ++d) {
        lo7 = max4(lo7, rows_y(lutL, sym, idx, j - 8 + d, cg));           // j-6 .. j-1
        hi7 = max4(hi7, rows_y(lutL, sym, idx, j + d, cg));               // j+2 .. j+7
      }
      const f32x4 yc = rows_y(lutL, sym, idx, j, cg);
      const size_t o = (size_t)b * 32 + 4 * cg;
      st4(a.FL + o, max4(max4(lo7, yc), hi7));
      st4(a.ELl + o, max4(rows_y_pad(lutL, SYM_PAD, sym[j], sym[j + 1], cg), hi7));
      f32x4 er = rows_y_pad(lutL, sym[j - 1], sym[j], SYM_PAD, cg);
      for (int d = 1; d <= a.er_n; ++d) er = max4(er, rows_y(lutL, sym, idx, j - d, cg));
      st4(a.ELr + o, er);
      // mid tower: 3-wide window, and the first pooled column of a crop starting at b
      const f32x4 m0 = rows_y(lutM, sym, idx, j - 1, cg), m1 = rows_y(lutM, sym, idx, j, cg), m2 = rows_y(lutM, sym, idx, j + 1, cg);
      st4(a.FM + o, max4(max4(m0, m1), m2));
      st4(a.EMl + o, max4(rows_y_pad(lutM, SYM_PAD, sym[j], sym[j + 1], cg), m2));
    }
  }
}

// --------------------------------------------------------------------------------------------------------------- kernel B
struct RowsConvArgs {
  const float* x;           // [nb][32] raw input rows; the conv sees BN(ReLU(x)) with (pre_s, pre_t)
  const float* pre_s;
  const float* pre_t;
  const float* wfrag;       // A fragments of this layer [mblock][kstep][lane]
  const float* bias;
  const float* res;         // nullable: accumulator starts from bias + res
  float* y;                 // raw output rows
  const float* zadd;        // nullable (with z): z = y + zadd
  float* z;
  int64_t nb;
  int D;                    // dilation on the base axis = stride of maxpool1
};

constexpr int RC_TN = 256;  // output columns per workgroup

template <int D>
__global__ __launch_bounds__(256) void reuse_rows_conv_kernel(const RowsConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float rcmem[];     // image [(TN + 2 D) columns + 1][32], swizzled
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mb = wave & 1, cgp = wave >> 1;
  const int n16 = lane & 15, kk = lane >> 4;
  const int chv = 16 * mb + 4 * kk;
  float af[SNV_KSTEPS];
  {
    const float* wf = a.wfrag + (size_t)mb * SNV_KSTEPS * 64 + lane;
#pragma unroll
    for (int s = 0; s < SNV_KSTEPS; ++s) af[s] = wf[s * 64];
  }
  const f32x4 pb = ld4(a.bias + chv);
  const int cg = tid & 7;
  const f32x4 ps = ld4(a.pre_s + 4 * cg), pt = ld4(a.pre_t + 4 * cg);
  const int64_t n_tiles = (a.nb + RC_TN - 1) / RC_TN;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t c0 = tile * RC_TN;
    __syncthreads();
    for (int task = tid; task < (RC_TN + 2 * D) * 8; task += 256) {     // image column pc <-> row c0 - D + pc
      const int pc = task >> 3;
      const int64_t b = c0 - D + pc;
      f32x4 v = splat(0.f);
      if (b >= 0 && b < a.nb) v = relu_bn(ld4(a.x + (size_t)b * 32 + 4 * cg), ps, pt);
      st4(rcmem + lds_off(pc, cg), v);
    }
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i < RC_TN / 32; i += 2) {                            // two 16-column blocks per pass: two accumulator chains
      const int blk0 = cgp + 2 * i, blk1 = cgp + 2 * (i + 1);
      const int64_t col0 = c0 + 16 * blk0 + n16, col1 = c0 + 16 * blk1 + n16;
      f32x4 r0 = splat(0.f), r1 = splat(0.f);
      if (a.res) {
        if (col0 < a.nb) r0 = ld4(a.res + (size_t)col0 * 32 + chv);
        if (col1 < a.nb) r1 = ld4(a.res + (size_t)col1 * 32 + chv);
      }
      f32x4 acc0 = f32x4{fmaf(r0.x, 1.f, pb.x), fmaf(r0.y, 1.f, pb.y), fmaf(r0.z, 1.f, pb.z), fmaf(r0.w, 1.f, pb.w)};
      f32x4 acc1 = f32x4{fmaf(r1.x, 1.f, pb.x), fmaf(r1.y, 1.f, pb.y), fmaf(r1.z, 1.f, pb.z), fmaf(r1.w, 1.f, pb.w)};
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        f32x4 B0[2], B1[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          B0[h] = ld4(rcmem + lds_off(16 * blk0 + n16 + t * D, 4 * h + kk));
          B1[h] = ld4(rcmem + lds_off(16 * blk1 + n16 + t * D, 4 * h + kk));
        }
        mfma_tap<true, SNV_KSTEPS>(af, t, B0, B1, acc0, acc1);
      }
      if (col0 < a.nb) {
        st4(a.y + (size_t)col0 * 32 + chv, acc0);
        if (a.z) {
          const f32x4 q = ld4(a.zadd + (size_t)col0 * 32 + chv);
          st4(a.z + (size_t)col0 * 32 + chv, f32x4{fmaf(acc0.x, 1.f, q.x), fmaf(acc0.y, 1.f, q.y), fmaf(acc0.z, 1.f, q.z), fmaf(acc0.w, 1.f, q.w)});
        }
      }
      if (col1 < a.nb) {
        st4(a.y + (size_t)col1 * 32 + chv, acc1);
        if (a.z) {
          const f32x4 q = ld4(a.zadd + (size_t)col1 * 32 + chv);
          st4(a.z + (size_t)col1 * 32 + chv, f32x4{fmaf(acc1.x, 1.f, q.x), fmaf(acc1.y, 1.f, q.y), fmaf(acc1.z, 1.f, q.z), fmaf(acc1.w, 1.f, q.w)});
        }
      }
    }
  }
}

// --------------------------------------------------------------------------------------------------------------- kernel S
// S[b] = BN(max over the maxpool2 window centred on row b): columns b + D * d, |d| <= half
__global__ __launch_bounds__(256) void reuse_rows_pool_kernel(const float* __restrict__ r, float* __restrict__ s, int64_t nb, int D,
                                                              int half, const float* __restrict__ bn_s, const float* __restrict__ bn_t) {
  const int cg = threadIdx.x & 7;
  const f32x4 ps = ld4(bn_s + 4 * cg), pt = ld4(bn_t + 4 * cg);
  const int64_t total = nb * 8;
  for (int64_t task = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; task < total; task += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = task >> 3;
    f32x4 m = ld4(r + (size_t)b * 32 + 4 * cg);
    for (int d = 1; d <= half; ++d) {
      const int64_t lo = b - (int64_t)D * d, hi = b + (int64_t)D * d;
      if (lo >= 0) m = max4(m, ld4(r + (size_t)lo * 32 + 4 * cg));
      if (hi < nb) m = max4(m, ld4(r + (size_t)hi * 32 + 4 * cg));
    }
    st4(s + (size_t)b * 32 + 4 * cg, f32x4{fmaf(ps.x, m.x, pt.x), fmaf(ps.y, m.y, pt.y), fmaf(ps.z, m.z, pt.z), fmaf(ps.w, m.w, pt.w)});
  }
}

// --------------------------------------------------------------------------------------------------------------- kernel E
// stage-2 column of edge-tile column j (0 .. 17): the 9 leftmost, then the 9 rightmost
__device__ __forceinline__ int edge_q(int j, int L2) { return j < RU_EC ? j : L2 - RU_L + j; }

// row index of a site's first input column | strand << 62, or -1 when the site lies outside the rows
__device__ __forceinline__ int64_t edge_wstart(const EdgeArgs& args, int64_t row) {
  if (row >= args.n) return -1;
  const int64_t gp = args.pos[row];
  const int neg = args.strand[row] != 0;
  const int64_t t = neg ? args.glen - 1 - gp : gp;
  const int64_t w = t + args.woff - args.t0[neg];
  if (w < 0 || w + args.L1 > args.nb || args.F[neg] == nullptr) return -1;   // the caller's bounds / strand mask were wrong
  return w | ((int64_t)neg << 62);
}

// x0 of the lane's edge columns of one tile, raw, in MFMA accumulator layout (requested one tile ahead)
__device__ __forceinline__ void edge_request_x0(const EdgeArgs& args, const uint32_t (&plan)[SNV_NB2MAX], const int64_t* wst, int nbw,
                                                int chv, f32x4 (&xres)[SNV_NB2MAX], uint32_t& live) {
  live = 0;
#pragma unroll
  for (int i = 0; i < SNV_NB2MAX; ++i) {
    xres[i] = splat(0.f);
    if (i < nbw && plan[i] != ~0u) {
      const int p = (int)(plan[i] >> 16), q = (int)(plan[i] & 0xFFFFu);
      const int64_t ws = wst[p];
      if (ws >= 0) {
        const int set = (int)(ws >> 62);
        const int64_t w = ws & ((1ll << 62) - 1);
        const float* src = args.F[set] + (size_t)(w + (int64_t)args.D * q) * 32;
        if (q == 0) src = args.El[set] + (size_t)w * 32;
        else if (q == args.L2 - 1 && args.right_pad) src = args.Er[set] + (size_t)(w + args.L1 - 1) * 32;
        xres[i] = ld4(src + chv);
        live |= 1u << i;
      }
    }
  }
}

constexpr int RU_POOL_SLOTS = 9;      // interior pooled (site, column, channel group) gathers per thread: P * n_int * 8 / 256 <= 9

__global__ __launch_bounds__(SNV_THREADS, 2) void snv_edge_kernel(const EdgeArgs args) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mb = wave & 1, cgp = wave >> 1;
  const int n16 = lane & 15, kk = lane >> 4;
  const int chv = 16 * mb + 4 * kk;
  const int P = args.P;
  float* bufA = smem;
  float* bufB = smem + args.nbuf;
  int64_t* wbuf = reinterpret_cast<int64_t*>(smem + 2 * args.nbuf);     // [2][P]: this tile's and the next tile's window starts
  const TowerGeom& g = args.ge;
  const TowerDev& tw = args.tw;
  const int64_t n_tiles = (args.n + P - 1) / P;
  const StageAddr sa = stage_setup(g, 0, P, n16, kk, mb, cgp);
  const int nb = g.nb[0];
  const int nbw = nb > cgp ? (nb - cgp + 1) / 2 : 0;
  // per-lane plan of the entry gather: (site p, stage-2 column q) of the lane's column in each owned block
  uint32_t plan[SNV_NB2MAX];
#pragma unroll
  for (int i = 0; i < SNV_NB2MAX; ++i) {
    plan[i] = ~0u;
    const int c = 16 * (cgp + 2 * i) + n16;
    if (i < nbw && c >= 1) {
      const uint32_t u = (uint32_t)(c - 1);
      const uint32_t p = g.dSc[0].div(u);
      const uint32_t j = u - p * (uint32_t)RU_SC;
      if (p < (uint32_t)P && j < (uint32_t)RU_L) plan[i] = (p << 16) | (uint32_t)edge_q((int)j, args.L2);
    }
  }
  const f32x4 es = ld4(tw.ex_s + EX_RB1_ENTRY * 32 + chv), et = ld4(tw.ex_t + EX_RB1_ENTRY * 32 + chv);
  const int cgq = tid & 7;
  const f32x4 pool_s = ld4(tw.ex_s + EX_BN_MID * 32 + 4 * cgq), pool_t = ld4(tw.ex_t + EX_BN_MID * 32 + 4 * cgq);
  float a_cur[SNV_KSTEPS];
  {
    const float* wf = tw.wfrag + (size_t)mb * SNV_KSTEPS * 64 + lane;
#pragma unroll
    for (int s = 0; s < SNV_KSTEPS; ++s) a_cur[s] = wf[s * 64];
  }
  f32x4 xres[SNV_NB2MAX];
  uint32_t live = 0;
  int cur = 0;
  if (tid < P) wbuf[tid] = edge_wstart(args, (int64_t)blockIdx.x * P + tid);
  __syncthreads();
  edge_request_x0(args, plan, wbuf, nbw, chv, xres, live);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * P;
    const int64_t* wst = wbuf + cur * P;
    // ---- entry: BN(ReLU(x0)) image of the 18 edge columns into bufA; the raw values stay in the residual registers
    {
      char* A = reinterpret_cast<char*>(bufA);
#pragma unroll
      for (int i = 0; i < SNV_NB2MAX; ++i)
        if (i < nbw) lds_st4(A, sa.wr + 4096u * i, ((live >> i) & 1u) ? relu_bn(xres[i], es, et) : splat(0.f));
    }
    // window starts of the next tile: the position load hides under the convs
    if (tid < P) wbuf[(cur ^ 1) * P + tid] = edge_wstart(args, (tile + gridDim.x) * P + tid);
    __syncthreads();
    // ---- the four ResBlock convs on the edge tile
    for (int layer = 0; layer < 4; ++layer) {
      const LayerK lk = layer_consts(layer_mode(layer));
      const bool in_is_a = ((0xA5u >> layer) & 1u) != 0;
      const char* in = reinterpret_cast<const char*>(in_is_a ? bufA : bufB);
      char* out = reinterpret_cast<char*>(in_is_a ? bufB : bufA);
      const f32x4 pb = ld4(tw.bias + layer * 32 + chv), ps = ld4(tw.post_s + layer * 32 + chv), pt = ld4(tw.post_t + layer * 32 + chv);
      __builtin_amdgcn_sched_barrier(0);
      float a_nxt[SNV_KSTEPS];
      {
        const float* wfn = tw.wfrag + (size_t)(layer < 3 ? layer + 1 : 0) * SNV_WFRAG + (size_t)mb * SNV_KSTEPS * 64 + lane;
#pragma unroll
        for (int s = 0; s < SNV_KSTEPS; ++s) a_nxt[s] = wfn[s * 64];
      }
      conv_layer(in, out, sa, nbw, lk, a_cur, pb, ps, pt, xres);
      __syncthreads();
#pragma unroll
      for (int s = 0; s < SNV_KSTEPS; ++s) a_cur[s] = a_nxt[s];
    }
    // the residual registers are dead: the next tile's x0 gathers fly under the pooling
    if (tile + gridDim.x < n_tiles) edge_request_x0(args, plan, wbuf + (cur ^ 1) * P, nbw, chv, xres, live);
    // ---- maxpool2 + BN -> s3.  Interior windows (columns u_lo .. u_hi) are gathers from S, all issued before the first store;
    //      the few windows that touch the edge pyramids read bufA (+ R rows when mixed) and get the BN here.
    {
      const int n_int = args.u_hi - args.u_lo + 1;
      const int total = P * n_int * 8;
      f32x4 sv[RU_POOL_SLOTS];
#pragma unroll
      for (int k = 0; k < RU_POOL_SLOTS; ++k) {
        const int task = tid + k * SNV_THREADS;
        sv[k] = splat(__uint_as_float(0x7FC00000u));
        if (task < total) {
          const int pj = task >> 3;
          const int p = pj / n_int, u = args.u_lo + (pj - p * n_int);
          const int64_t ws = wst[p];
          if (ws >= 0) {
            const int set = (int)(ws >> 62);
            const int64_t w = ws & ((1ll << 62) - 1);
            sv[k] = ld4(args.S[set] + (size_t)(w + (int64_t)args.D * args.ps2 * u) * 32 + 4 * cgq);
          }
        }
      }
      // edge / mixed windows while the gathers fly
      const int n_edge = args.L3 - n_int;
#pragma unroll 1
      for (int task = tid; task < P * n_edge * 8; task += SNV_THREADS) {
        const int pj = task >> 3;
        const int p = pj / n_edge, e = pj - p * n_edge;
        const int u = e < args.u_lo ? e : args.u_hi + 1 + (e - args.u_lo);
        if (row0 + p >= args.n) continue;
        const int64_t ws = wst[p];
        f32x4 m = splat(__uint_as_float(0x7FC00000u));
        if (ws >= 0) {
          const int set = (int)(ws >> 62);
          const int64_t w = ws & ((1ll << 62) - 1);
          const int jlo = u * args.ps2 - args.pp2;
          const int lo = jlo < 0 ? 0 : jlo;
          const int hi = (jlo + args.pk2 - 1) < (args.L2 - 1) ? (jlo + args.pk2 - 1) : (args.L2 - 1);
          f32x4 rv[7];
#pragma unroll
          for (int d = 0; d < 7; ++d) {           // the model's second pools are 7 / 3 wide: every row read in flight together
            rv[d] = splat(-INFINITY);
            const int q = lo + d;
            if (q <= hi) {
              if (q < RU_EV) rv[d] = ld4(bufA + lds_off(1 + p * RU_SC + q + 1, cgq));
              else if (q > args.L2 - 1 - RU_EV) rv[d] = ld4(bufA + lds_off(1 + p * RU_SC + (q - (args.L2 - RU_L)) + 1, cgq));
              else rv[d] = ld4(args.R[set] + (size_t)(w + (int64_t)args.D * q) * 32 + 4 * cgq);
            }
          }
          m = max4(max4(max4(rv[0], rv[1]), max4(rv[2], rv[3])), max4(max4(rv[4], rv[5]), rv[6]));
          for (int q = lo + 7; q <= hi; ++q) {     // wider pools (not in the model): plain loop
            f32x4 v;
            if (q < RU_EV) v = ld4(bufA + lds_off(1 + p * RU_SC + q + 1, cgq));
            else if (q > args.L2 - 1 - RU_EV) v = ld4(bufA + lds_off(1 + p * RU_SC + (q - (args.L2 - RU_L)) + 1, cgq));
            else v = ld4(args.R[set] + (size_t)(w + (int64_t)args.D * q) * 32 + 4 * cgq);
            m = max4(m, v);
          }
          m = f32x4{fmaf(pool_s.x, m.x, pool_t.x), fmaf(pool_s.y, m.y, pool_t.y), fmaf(pool_s.z, m.z, pool_t.z),
                    fmaf(pool_s.w, m.w, pool_t.w)};
        }
        st4(args.s3 + ((size_t)(row0 + p) * args.L3 + u) * 32 + 4 * cgq, m);
      }
#pragma unroll
      for (int k = 0; k < RU_POOL_SLOTS; ++k) {
        const int task = tid + k * SNV_THREADS;
        if (task < total) {
          const int pj = task >> 3;
          const int p = pj / n_int, u = args.u_lo + (pj - p * n_int);
          if (row0 + p < args.n) st4(args.s3 + ((size_t)(row0 + p) * args.L3 + u) * 32 + 4 * cgq, sv[k]);
        }
      }
    }
    lds_barrier();          // LDS-only hand-off: a full barrier would drain the x0 gathers of the next tile
    cur ^= 1;
  }
}

int pool_len(int L, int k, int s, int p) { return (L + 2 * p - k) / s + 1; }
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct ReuseWs {
  float* rows[2][11];       // per strand: FL FM ELl ELr EMl | T1 T2 T3 large | T1 T2 T3 mid
  int64_t* cat;
  float* local_logits;
  float* xlogit;
  float* s3[2];
  int* counters;
};

constexpr int RU_SUPER = 4;
int64_t reuse_super_sites() {
  const bool off = dev_env("MURAL_SNV_DEFER_SHORT") && atoi(dev_env("MURAL_SNV_DEFER_SHORT")) == 0;
  return off ? (int64_t)SNV_CHUNK : (int64_t)RU_SUPER * SNV_CHUNK;
}

size_t carve_reuse(const MuralSnvModel* m, int64_t n, int64_t span, int strands, void* base, ReuseWs* w) {
  const int64_t nb = std::min<int64_t>(span, RU_CHUNK_SPAN) + m->shape.distal_len;
  size_t off = 0;
  const size_t guard = ws_guard_bytes();      // 0 outside the validation tests (common.h)
  ws_layout_reset();
  auto take = [&](size_t bytes) {
    size_t o = off;
    ws_layout_add(o, bytes);
    off = align_up(off + bytes + guard, 256);
    return o;
  };
  size_t o_rows[2][11];
  for (int st = 0; st < 2; ++st)
    for (int i = 0; i < 11; ++i) o_rows[st][i] = ((strands >> st) & 1) ? take((size_t)nb * 32 * 4) : 0;
  // (the short-stage launches run once per RU_SUPER chunks, as in the per-window path: snv_model.hip, super_chunk_sites)
  const int64_t ns = std::min<int64_t>(std::max<int64_t>(n, 1), reuse_super_sites());
  const size_t o_cat = take((size_t)ns * std::max(m->shape.local_cols, 1) * 8);
  const size_t o_ll = take((size_t)ns * m->shape.n_class * 4);
  const size_t o_xl = take((size_t)ns * SNV_MAXCLASS * 4);
  const size_t o_s3l = take((size_t)ns * std::max(m->args.geom[0].L[1], 1) * SNV_C * 4);
  const size_t o_s3m = take((size_t)ns * std::max(m->args.geom[1].L[1], 1) * SNV_C * 4);
  const size_t o_cnt = take(64);      // unit counters of the wave-private launches (SnvFwdArgs::unit_counter)
  if (w) {
    char* b = static_cast<char*>(base);
    for (int st = 0; st < 2; ++st)
      for (int i = 0; i < 11; ++i) w->rows[st][i] = ((strands >> st) & 1) ? reinterpret_cast<float*>(b + o_rows[st][i]) : nullptr;
    w->cat = reinterpret_cast<int64_t*>(b + o_cat);
    w->local_logits = reinterpret_cast<float*>(b + o_ll);
    w->xlogit = reinterpret_cast<float*>(b + o_xl);
    w->s3[0] = reinterpret_cast<float*>(b + o_s3l);
    w->s3[1] = reinterpret_cast<float*>(b + o_s3m);
    w->counters = reinterpret_cast<int*>(b + o_cnt);
  }
  return off;
}

// edge tile: as many sites as keep <= SNV_NB2MAX blocks per wave (two workgroups per CU fit by far)
int reuse_edge_sites() {
  for (int cand = 15; cand >= 1; --cand) {
    const int NC = 1 + cand * RU_SC, nbk = (NC + 15) / 16;
    if ((nbk + 1) / 2 <= SNV_NB2MAX) return cand;
  }
  return 1;
}

// interior pooled-column range [u_lo, u_hi] of the second pool for a tower geometry (empty range: u_lo = L3, u_hi = L3 - 1)
void reuse_interior(const TowerGeom& gg, int* u_lo, int* u_hi) {
  const int pk2 = gg.pk[1], ps2 = gg.ps[1], pp2 = gg.pp[1], L2 = gg.L[0], L3 = gg.L[1];
  int lo = (RU_EV + pp2 + ps2 - 1) / ps2;
  int hi = (L2 - 1 - RU_EV - (pk2 - 1) + pp2) / ps2;
  if (hi > L3 - 1) hi = L3 - 1;
  if (hi < lo) { lo = L3; hi = L3 - 1; }
  *u_lo = lo;
  *u_hi = hi;
}

// Every geometry condition the reuse launches REQUIRE is evaluated here, so that a caller who asks first (HipShardForward does)
// falls back to the per-window kernels instead of meeting an error half-way through a shard.  Long-window models
// (MuralSnvModel::longwin: the segmented first stage) are never served: their pooled interior does not fit the edge tile.
bool reuse_supported(const MuralSnvModel* m) {
  if (m->shape.model_no == 0 || !m->split || m->longwin) return false;
  const TowerGeom& gl = m->args.geom[0];
  const TowerGeom& gm = m->args.geom[1];
  if (gl.L[0] < RU_L || gm.L[0] < RU_L) return false;
  if (!(gl.pk[0] == 15 && gl.ps[0] == 15 && gl.pp[0] == 7 && gm.pk[0] == 3 && gm.ps[0] == 3 && gm.pp[0] == 1)) return false;
  const int last_lo = gl.ps[0] * (gl.L[0] - 1) - gl.pp[0];
  const int er_n = std::min(last_lo + gl.pk[0] - 1, gl.L1 - 1) - last_lo;
  if (er_n < 0 || er_n > 14) return false;
  const int P = reuse_edge_sites();
  for (int t = 0; t < 2; ++t) {
    const TowerGeom& gg = m->args.geom[t];
    if (!(gg.pk[1] == 2 * gg.pp[1] + 1 && gg.ps[1] == gg.pk[1])) return false;
    const int l_lo = gg.ps[0] * (gg.L[0] - 1) - gg.pp[0];
    if (t == 1 && l_lo + gg.pk[0] - 1 >= gg.L1 - 1) return false;      // the mid crop's last pooled column must be interior
    int u_lo, u_hi;
    reuse_interior(gg, &u_lo, &u_hi);
    if ((int64_t)P * (u_hi - u_lo + 1) * 8 > (int64_t)RU_POOL_SLOTS * SNV_THREADS) return false;
  }
  return true;
}

int launch_rows_conv(const RowsConvArgs& a, hipStream_t stream) {
  const int64_t n_tiles = (a.nb + RC_TN - 1) / RC_TN;
  const int grid = (int)std::min<int64_t>(n_tiles, 4096);
  const size_t lds = (size_t)(RC_TN + 2 * a.D + 1) * 32 * 4;
  if (a.D == 15) hipLaunchKernelGGL(reuse_rows_conv_kernel<15>, dim3(grid), dim3(256), lds, stream, a);
  else if (a.D == 3) hipLaunchKernelGGL(reuse_rows_conv_kernel<3>, dim3(grid), dim3(256), lds, stream, a);
  else {
    set_error("reuse: pooling stride %d not built", a.D);
    return MURAL_E_INVALID;
  }
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

}  // namespace
}  // namespace mural

using namespace mural;

extern "C" int mural_snv_reuse_supported(const MuralSnvModel* m) { return m && reuse_supported(m) ? 1 : 0; }

extern "C" int64_t mural_snv_reuse_chunk_span(void) { return RU_CHUNK_SPAN; }

extern "C" size_t mural_snv_reuse_workspace_bytes(const MuralSnvModel* m, int64_t n, int64_t span, int32_t strands) {
  if (!m || n <= 0 || span <= 0) return 256;
  return carve_reuse(m, n, span, strands & 3, nullptr, nullptr);
}

extern "C" int mural_snv_forward_packed_reuse(const MuralSnvModel* m, const MuralGenome* g, const int64_t* pos,
                                              const uint8_t* strand, int64_t n, int32_t strands, int64_t pos_min, int64_t pos_max,
                                              int32_t local_radius, int32_t local_order, float* out, void* workspace,
                                              size_t workspace_bytes, void* stream_) {
  MURAL_REQUIRE(m, "model handle is NULL");
  MURAL_REQUIRE(g && g->packed2 && g->nmask, "genome pointers must not be NULL");
  MURAL_REQUIRE(g->n_amb == 0 || (g->amb_pos && g->amb_sym), "genome: n_amb > 0 needs amb_pos and amb_sym");
  MURAL_REQUIRE(reuse_supported(m), "cross-position reuse needs a tower model whose pooled rows hold at least %d columns", RU_L);
  MURAL_REQUIRE(n >= 0, "negative batch");
  if (n == 0) return MURAL_OK;
  MURAL_REQUIRE(pos && strand && out, "pos/strand/out must not be NULL");
  MURAL_REQUIRE(strands >= 1 && strands <= 3, "strands must be 1 ('+' sites only), 2 ('-' only) or 3 (both)");
  MURAL_REQUIRE(pos_min <= pos_max && pos_max - pos_min + 1 <= RU_CHUNK_SPAN,
                "reuse: sites must span at most %lld bases per call", (long long)RU_CHUNK_SPAN);
  const int64_t span = pos_max - pos_min + 1;
  const size_t need = carve_reuse(m, n, span, strands, nullptr, nullptr);
  if (workspace_bytes < need || !workspace) {
    set_error("workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    return MURAL_E_WORKSPACE;
  }
  hipStream_t stream = (hipStream_t)stream_;
  const MuralSnvShape& sh = m->shape;
  ReuseWs w;
  carve_reuse(m, n, span, strands, workspace, &w);
  const int R = (sh.distal_len - 1) / 2;
  const int64_t nb = span + 2 * (int64_t)R;
  const TowerGeom& gl = m->args.geom[0];
  const TowerGeom& gm = m->args.geom[1];
  MURAL_REQUIRE(gl.pk[0] == 15 && gl.ps[0] == 15 && gl.pp[0] == 7 && gm.pk[0] == 3 && gm.ps[0] == 3 && gm.pp[0] == 1,
                "reuse: unexpected maxpool1 geometry");
  int64_t t0s[2] = {0, 0};
  const float* Rrows[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [strand][tower]
  const float* Srows[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  for (int neg = 0; neg < 2; ++neg) {
    if (!((strands >> neg) & 1)) continue;
    // oriented coordinates: t = g on '+', length - 1 - g on '-'; rows cover [c0 - R, c1 + R]
    const int64_t c0 = neg ? g->length - 1 - pos_max : pos_min;
    const int64_t t0 = c0 - R;
    t0s[neg] = t0;
    float* const* rows = w.rows[neg];
    float *FL = rows[0], *FM = rows[1], *ELl = rows[2], *ELr = rows[3], *EMl = rows[4];
    // ---- A: pooled first-layer rows
    {
      RowsS1Args a;
      a.genome = *g;
      a.neg = neg;
      a.t0 = t0;
      a.nb = nb;
      a.lut[0] = m->args.tw[0].lut;
      a.lut[1] = m->args.tw[1].lut;
      const int last_lo = gl.ps[0] * (gl.L[0] - 1) - gl.pp[0];
      const int last_hi = std::min(last_lo + gl.pk[0] - 1, gl.L1 - 1);
      a.er_n = last_hi - last_lo;                     // interior columns in front of the zero-padded last conv column
      MURAL_REQUIRE(a.er_n >= 0 && a.er_n <= 14, "reuse: unexpected last pooled column");
      a.FL = FL; a.FM = FM; a.ELl = ELl; a.ELr = ELr; a.EMl = EMl;
      const size_t lds = (size_t)2 * SNV_LUTBLK * 4 + 2 * (RS1_TB + 2 * RS1_HALO);
      static DynLdsOnce big_lds;
      if (int rc = big_lds.ensure(&reuse_rows_stage1_kernel)) return rc;
      const int64_t n_tiles = (nb + RS1_TB - 1) / RS1_TB;
      hipLaunchKernelGGL(reuse_rows_stage1_kernel, dim3((unsigned)std::min<int64_t>(n_tiles, 512)), dim3(RS1_THREADS), lds, stream, a);
      MURAL_HIP_CHECK(hipGetLastError());
    }
    // ---- B + S per tower: the four ResBlock convs as dilated convs over the rows, then the sliding maxpool2
    for (int t = 0; t < 2; ++t) {
      const TowerDev& tw = m->args.tw[t];
      const TowerGeom& gg = m->args.geom[t];
      const float* x0 = t == 0 ? FL : FM;
      float *T1 = rows[5 + 3 * t], *T2 = rows[6 + 3 * t], *T3 = rows[7 + 3 * t];
      const int D = gg.ps[0];
      RowsConvArgs a{};
      a.nb = nb;
      a.D = D;
      // RBs1[0].conv1 on BN(ReLU(x0))
      a.x = x0; a.pre_s = tw.ex_s + EX_RB1_ENTRY * 32; a.pre_t = tw.ex_t + EX_RB1_ENTRY * 32;
      a.wfrag = tw.wfrag + 0 * (size_t)SNV_WFRAG; a.bias = tw.bias + 0 * 32; a.res = nullptr; a.y = T1; a.z = nullptr; a.zadd = nullptr;
      if (int rc = launch_rows_conv(a, stream)) return rc;
      // RBs1[0].conv2: x1 = x0 + conv(...), z = x1 + x0 (outer skip, model_snv.py:477-479)
      a.x = T1; a.pre_s = tw.post_s + 0 * 32; a.pre_t = tw.post_t + 0 * 32;
      a.wfrag = tw.wfrag + 1 * (size_t)SNV_WFRAG; a.bias = tw.bias + 1 * 32; a.res = x0; a.y = T2; a.z = T3; a.zadd = x0;
      if (int rc = launch_rows_conv(a, stream)) return rc;
      // RBs1[1].conv1
      a.x = T2; a.pre_s = tw.post_s + 1 * 32; a.pre_t = tw.post_t + 1 * 32;
      a.wfrag = tw.wfrag + 2 * (size_t)SNV_WFRAG; a.bias = tw.bias + 2 * 32; a.res = nullptr; a.y = T1; a.z = nullptr; a.zadd = nullptr;
      if (int rc = launch_rows_conv(a, stream)) return rc;
      // RBs1[1].conv2 on top of z: RBs1(x0) + x0
      a.x = T1; a.pre_s = tw.post_s + 2 * 32; a.pre_t = tw.post_t + 2 * 32;
      a.wfrag = tw.wfrag + 3 * (size_t)SNV_WFRAG; a.bias = tw.bias + 3 * 32; a.res = T3; a.y = T2;
      if (int rc = launch_rows_conv(a, stream)) return rc;
      MURAL_REQUIRE(gg.pk[1] == 2 * gg.pp[1] + 1 && gg.ps[1] == gg.pk[1], "reuse: unexpected maxpool2 geometry");
      const int64_t tasks = nb * 8;
      hipLaunchKernelGGL(reuse_rows_pool_kernel, dim3((unsigned)std::min<int64_t>((tasks + 255) / 256, 8192)), dim3(256), 0, stream, T2,
                         T3, nb, D, gg.pp[1], tw.ex_s + EX_BN_MID * 32, tw.ex_t + EX_BN_MID * 32);
      MURAL_HIP_CHECK(hipGetLastError());
      Rrows[neg][t] = T2;
      Srows[neg][t] = T3;
    }
  }
  // ---- per batch of sites: local branch, edge kernels, short-stage launches
  const int nc = sh.n_class;
  const int P = reuse_edge_sites();
  TowerGeom ge{};
  ge.L[0] = RU_L; ge.Sc[0] = RU_SC; ge.NC[0] = 1 + P * RU_SC; ge.nb[0] = (ge.NC[0] + 15) / 16;
  ge.dL[0] = FastDiv::make(RU_L); ge.dSc[0] = FastDiv::make(RU_SC);
  const int nbuf = (16 * ge.nb[0] + 2) * SNV_C;
  const size_t lds_edge = (size_t)2 * nbuf * 4 + (size_t)2 * P * 8 + 64;

  static DynLdsOnce edge_lds;
  if (int rc = edge_lds.ensure(&snv_edge_kernel)) return rc;
  const bool edge_wave = !dev_env("MURAL_DEBUG_EDGE_TILE");
  const int64_t super = reuse_super_sites();
  const size_t s3_site[2] = {(size_t)std::max(m->args.geom[0].L[1], 1) * SNV_C, (size_t)std::max(m->args.geom[1].L[1], 1) * SNV_C};
  for (int64_t u0 = 0; u0 < n; u0 += super) {
  const int64_t un = std::min<int64_t>(super, n - u0);
  for (int64_t s0 = u0; s0 < u0 + un; s0 += SNV_CHUNK) {
    const int64_t sn = std::min<int64_t>(SNV_CHUNK, u0 + un - s0);
    const int64_t rel = s0 - u0;      // this chunk's place in the super-chunk's s3 / logits
    if (sh.model_no == 2) {
      const int ncol = 2 * local_radius + 1 - (local_order - 1);
      MURAL_REQUIRE(ncol == sh.local_cols, "local_radius/local_order give %d k-mer columns, model has %d", ncol, sh.local_cols);
      int64_t sentinel = 1;
      for (int i = 0; i < local_order; ++i) sentinel *= 4;
      MURAL_REQUIRE(sentinel + 1 == sh.emb_rows, "local_order %d does not match the embedding table (%d rows)", local_order, sh.emb_rows);
      if (int rc = mural_encode_kmer(g, pos + s0, strand + s0, sn, local_radius, local_order, 0, w.cat, stream_)) return rc;
      if (int rc = launch_snv_local(m->local, w.cat, sn, w.local_logits + rel * nc, stream)) return rc;
    }
    if (dev_env("MURAL_TOWER_DYNAMIC_UNITS") && atoi(dev_env("MURAL_TOWER_DYNAMIC_UNITS")) != 0)
      MURAL_HIP_CHECK(hipMemsetAsync(w.counters, 0, 64, stream));      // unit counters of this chunk's four wave-private launches (read with that switch only)
    for (int t = 0; t < 2; ++t) {
      const TowerGeom& gg = m->args.geom[t];
      EdgeArgs e{};
      e.ge = ge;
      e.tw = m->args.tw[t];
      e.P = P;
      e.nbuf = nbuf;
      e.n = sn;
      e.pos = pos + s0;
      e.strand = strand + s0;
      e.glen = g->length;
      e.t0[0] = t0s[0];
      e.t0[1] = t0s[1];
      e.nb = nb;
      e.woff = -R + gg.col0;
      e.L1 = gg.L1;
      e.L2 = gg.L[0];
      e.L3 = gg.L[1];
      e.D = gg.ps[0];
      e.pk2 = gg.pk[1]; e.ps2 = gg.ps[1]; e.pp2 = gg.pp[1];
      const int last_lo = gg.ps[0] * (gg.L[0] - 1) - gg.pp[0];
      e.right_pad = (last_lo + gg.pk[0] - 1 >= gg.L1 - 1) ? 1 : 0;
      MURAL_REQUIRE(t == 0 || !e.right_pad, "reuse: the mid crop's last pooled column is expected to be interior");
      reuse_interior(gg, &e.u_lo, &e.u_hi);        // no interior window: every pooled column from the edge loop
      MURAL_REQUIRE((int64_t)P * (e.u_hi - e.u_lo + 1) * 8 <= (int64_t)RU_POOL_SLOTS * SNV_THREADS, "reuse: pooled tile too large");
      for (int neg = 0; neg < 2; ++neg) {
        e.F[neg] = w.rows[neg][t == 0 ? 0 : 1];
        e.El[neg] = w.rows[neg][t == 0 ? 2 : 4];
        e.Er[neg] = w.rows[neg][3];
        e.R[neg] = Rrows[neg][t];
        e.S[neg] = Srows[neg][t];
      }
      e.s3 = w.s3[t] + (size_t)rel * s3_site[t];
      if (edge_wave) {      // wave-private form (snv_tower_wave.hip); MURAL_DEBUG_EDGE_TILE=1 keeps the workgroup-tile kernel (A/B runs)
        if (int rc = launch_snv_edge_wave(e, w.counters + t, stream)) return rc;
      } else {
        const int64_t n_tiles = (sn + P - 1) / P;
        hipLaunchKernelGGL(snv_edge_kernel, dim3((unsigned)std::min<int64_t>(n_tiles, 2048)), dim3(SNV_THREADS), lds_edge, stream, e);
        MURAL_HIP_CHECK(hipGetLastError());
      }
    }
  }
    for (int part = 2; part < 4; ++part) {      // the short stages and the head: one launch per tower for the whole super-chunk
      SnvFwdArgs t = m->args_split[part];
      t.s3[0] = w.s3[0];
      t.s3[1] = w.s3[1];
      t.n = un;
      t.x0 = nullptr;
      t.xlogit = w.xlogit;
      t.local_logits = w.local_logits;
      t.out = out + u0 * nc;
      t.taps = nullptr;
      t.tap_stride = 0;
      t.stamps = nullptr;
      t.status = nullptr;
      t.unit_counter = t.wave ? w.counters + part : nullptr;
      if (int rc = launch_snv_towers(m, t, m->lds_split[part], stream)) return rc;
    }
  }
  return MURAL_OK;
}

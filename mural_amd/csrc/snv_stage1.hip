// Stage 1 of both SNV towers: window decode -> BN(4)+Conv1d(4->32,k3) -> MaxPool1d, as 3-mer table lookups.
//
// Reference semantics: MuRaL/model/model_snv.py:473-475 (mid tower: centre crop, conv1, maxpool1) and :496-497
// (large tower), fed by the encoders of MuRaL/data/preprocessing.py:756-816.  On a one-hot input the first layer at
// column j depends only on the 3-mer (j-1, j, j+1): a 125 x 32 table per tower (A,C,G,T,N)^3; columns that touch an
// IUPAC code, the conv zero padding or the crop edge take per-tap tables instead.  The (32 x 2001) first-layer
// activation is never materialised: each lane owns 4 channels of one pooled column and maxes its 15 (large) or 3
// (mid) lookups in registers.  Output: pooled activations x0[row][134 + 67 columns][32] (25.7 KB per site at R=1000).
//
// Mapping: ONE WAVE PER SITE, 16 waves per workgroup sharing the two towers' tables in LDS (staged once per
// workgroup, persistent grid).  No workgroup barrier in the site loop: a wave's LDS writes are consumed only by
// itself, so four waves per SIMD hide each other's LDS latency.  Bound: LDS reads (256 KB of table rows per site).
#include <algorithm>
#include <cstdlib>

#include "dense_symbol.h"
#include "snv.h"
#include "snv_local_mfma.h"

namespace mural {

using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f32x4 s1_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 s1_max4(f32x4 a, f32x4 b) {
  return f32x4{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)};
}

constexpr int S1_WAVES = 16;
constexpr int S1_THREADS = 64 * S1_WAVES;

// all LDS traffic of this wave issued so far is complete and visible to its other lanes
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// 3-mer index of tower column j (255 = needs the per-tap path: IUPAC code, zero padding or crop edge)
__device__ __forceinline__ uint32_t kmer_index(const uint8_t* cb, int j, int L1) {
  if (j <= 0 || j >= L1 - 1) return 255u;           // cb[j + 1] is the symbol of tower column j
  const uint32_t l = cb[j], c = cb[j + 1], r = cb[j + 2];
  return (l <= 4u && c <= 4u && r <= 4u) ? (25u * l + 5u * c + r) : 255u;
}

// window-major 3-mer indices: kw[j2][SLOT], entry w = index of tower column ps*j2 - pp + w (255 beyond pk / range)
// `lane` / `nl`: index and count of the lanes that share one site (64 = one wave per site; 1024 = the whole workgroup)
template <int SLOT>
__device__ __forceinline__ void build_kwin(const Stage1Tower& g, int lane, const uint8_t* cb0, uint8_t* kw, int nl = 64) {
  const uint8_t* cb = cb0 + g.col0;
  const int ndw = g.L2 * (SLOT / 4);
  for (int t = lane; t < ndw; t += nl) {
    const int j2 = t / (SLOT / 4);
    const int w0 = 4 * (t % (SLOT / 4));
    uint32_t packed = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int w = w0 + q;
      const uint32_t idx = (w < g.pk) ? kmer_index(cb, j2 * g.ps - g.pp + w, g.L1) : 255u;
      packed |= idx << (8 * q);
    }
    *reinterpret_cast<uint32_t*>(kw + (size_t)t * 4) = packed;
  }
}

template <int SLOT>
__device__ __forceinline__ void pooled_lookup(const Stage1Tower& g, int lane, const float* lutS, const uint8_t* cb0,
                                              const uint8_t* kw, float* __restrict__ out /* [L2][32] */, int nl = 64) {
  const float* tapS = lutS + SNV_LUT;
  const float* b0S = tapS + SNV_TAPS;
  const uint8_t* cb = cb0 + g.col0;
  const int cg = lane & 7;
  const int total = g.L2 * 8;
  for (int task = lane; task < total; task += nl) {
    const int jt = task >> 3;
    const int j2 = jt == 0 ? g.L2 - 1 : jt - 1;      // (last pooled column first: both edge columns in one round, see pooled_lookup_pair)
    uint32_t d[SLOT / 4];
    if (SLOT == 16) {
      const uint4 q = *reinterpret_cast<const uint4*>(kw + (size_t)j2 * 16);
      d[0] = q.x; d[1] = q.y; d[2] = q.z; d[SLOT / 4 - 1] = q.w;
    } else {
      d[0] = *reinterpret_cast<const uint32_t*>(kw + (size_t)j2 * 4);
    }
    // bytes 0 .. SLOT-2 are table indices (< 128) unless a column needs the per-tap path (255)
    uint32_t hi = d[SLOT / 4 - 1] & 0x00FFFFFFu;
#pragma unroll
    for (int r = 0; r + 1 < SLOT / 4; ++r) hi |= d[r];
    const bool fast = ((hi & 0x80808080u) == 0u) && (g.pk == SLOT - 1);
    f32x4 m = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (fast) {
#pragma unroll
      for (int w = 0; w < SLOT - 1; ++w) {
        const uint32_t idx = (d[w >> 2] >> (8 * (w & 3))) & 0xFFu;
        m = s1_max4(m, s1_ld4(lutS + idx * 32u + 4u * cg));
      }
    } else {
      const int jlo = j2 * g.ps - g.pp;
      for (int w = 0; w < g.pk; ++w) {
        const int j = jlo + w;
        if (j < 0 || j >= g.L1) continue;      // MaxPool1d pads with -inf
        const uint32_t idx = kmer_index(cb, j, g.L1);
        f32x4 v;
        if (idx != 255u) {
          v = s1_ld4(lutS + idx * 32u + 4u * cg);
        } else {                               // zero padding is applied after the BN: PAD rows are exactly 0
          const uint32_t sl = (j == 0) ? (uint32_t)SYM_PAD : cb[j];
          const uint32_t sc = cb[j + 1];
          const uint32_t sr = (j == g.L1 - 1) ? (uint32_t)SYM_PAD : cb[j + 2];
          v = s1_ld4(b0S + 4 * cg);
          v += s1_ld4(tapS + (0 * N_SYM + sl) * 32 + 4 * cg);
          v += s1_ld4(tapS + (1 * N_SYM + sc) * 32 + 4 * cg);
          v += s1_ld4(tapS + (2 * N_SYM + sr) * 32 + 4 * cg);
        }
        m = s1_max4(m, v);
      }
    }
    *reinterpret_cast<f32x4*>(out + (size_t)j2 * 32 + 4 * cg) = m;   // 8 lanes write one 128-byte column
  }
}

// ---- pair-table form of the large tower's pooled lookup (15-wide pools).  Slot of pooled column j2 (16 bytes): two halves of
// 8 bytes, written by two lanes -- [pair codes of window columns (0,1) (2,3) (4,5) (6,7) | flag | 3 pad] and [pair codes of
// (8,9) (10,11) (12,13) | 3-mer index of column 14 | flag | 3 pad]; a pair code is the 4-mer of the bases under both columns
// (64 a + 16 b + 4 c + d), valid only when every base is A/C/G/T and every column is interior: the flag is 1 otherwise and the
// whole pooled column takes the per-column path of pooled_lookup.
__device__ __forceinline__ void build_kpair(const Stage1Tower& g, int lane, const uint8_t* cb0, uint8_t* kw) {
  const uint8_t* cb = cb0 + g.col0;
  for (int t = lane; t < 2 * g.L2; t += 64) {
    const int j2 = t >> 1, half = t & 1;
    const int j0 = j2 * g.ps - g.pp + 8 * half;      // first window column of this half
    const int ncol = half ? 7 : 8;                   // columns j0 .. j0 + ncol - 1; symbols cb[j0] .. cb[j0 + ncol + 1]
    uint32_t sy[10];
    uint32_t bad = (j0 < 1 || j0 + ncol - 1 > g.L1 - 2) ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int idx = j0 + i;                        // cb[idx] = symbol of tower column idx - 1
      const bool in = idx >= 0 && idx <= g.L1 + 1 && i < ncol + 2;
      sy[i] = in ? cb[idx] : 0u;
      if (i < ncol + 2 && (!in || sy[i] > 3u)) bad = 1u;
    }
    uint32_t lo = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t code;
      if (half && q == 3) code = 25u * sy[6] + 5u * sy[7] + sy[8];                     // single column 14: 3-mer index (no N here)
      else code = 64u * sy[2 * q] + 16u * sy[2 * q + 1] + 4u * sy[2 * q + 2] + sy[2 * q + 3];
      lo |= (code & 0xFFu) << (8 * q);
    }
    uint32_t* dst = reinterpret_cast<uint32_t*>(kw + (size_t)j2 * 16 + 8 * half);
    dst[0] = lo;
    dst[1] = bad;
  }
}

__device__ __forceinline__ void pooled_lookup_pair(const Stage1Tower& g, int lane, const float* lutS, const float* lut4S, const uint8_t* cb0,
                                                   const uint8_t* kw, float* __restrict__ out /* [L2][32] */) {
  const float* tapS = lutS + SNV_LUT;
  const float* b0S = tapS + SNV_TAPS;
  const uint8_t* cb = cb0 + g.col0;
  const int cg = lane & 7;
  const int total = g.L2 * 8;
  for (int task = lane; task < total; task += 64) {
    // (the row's LAST pooled column first: the two columns whose windows hang over the row's ends then share a round, and the
    // fifteen-iteration per-column path they send their wave through runs once per site instead of twice)
    const int jt = task >> 3;
    const int j2 = jt == 0 ? g.L2 - 1 : jt - 1;
    const uint4 q = *reinterpret_cast<const uint4*>(kw + (size_t)j2 * 16);
    f32x4 m;
    if (((q.y | q.w) & 0xFFu) == 0u) {      // eight table rows instead of fifteen
      m = s1_ld4(lut4S + (q.x & 0xFFu) * 32u + 4u * cg);
      m = s1_max4(m, s1_ld4(lut4S + ((q.x >> 8) & 0xFFu) * 32u + 4u * cg));
      m = s1_max4(m, s1_ld4(lut4S + ((q.x >> 16) & 0xFFu) * 32u + 4u * cg));
      m = s1_max4(m, s1_ld4(lut4S + (q.x >> 24) * 32u + 4u * cg));
      m = s1_max4(m, s1_ld4(lut4S + (q.z & 0xFFu) * 32u + 4u * cg));
      m = s1_max4(m, s1_ld4(lut4S + ((q.z >> 8) & 0xFFu) * 32u + 4u * cg));
      m = s1_max4(m, s1_ld4(lut4S + ((q.z >> 16) & 0xFFu) * 32u + 4u * cg));
      m = s1_max4(m, s1_ld4(lutS + (q.z >> 24) * 32u + 4u * cg));
    } else {                                 // a column with N / IUPAC codes, zero padding or the crop edge: column by column
      m = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      const int jlo = j2 * g.ps - g.pp;
      for (int w = 0; w < g.pk; ++w) {
        const int j = jlo + w;
        if (j < 0 || j >= g.L1) continue;      // MaxPool1d pads with -inf
        const uint32_t idx = kmer_index(cb, j, g.L1);
        f32x4 v;
        if (idx != 255u) {
          v = s1_ld4(lutS + idx * 32u + 4u * cg);
        } else {                               // zero padding is applied after the BN: PAD rows are exactly 0
          const uint32_t sl = (j == 0) ? (uint32_t)SYM_PAD : cb[j];
          const uint32_t sc = cb[j + 1];
          const uint32_t sr = (j == g.L1 - 1) ? (uint32_t)SYM_PAD : cb[j + 2];
          v = s1_ld4(b0S + 4 * cg);
          v += s1_ld4(tapS + (0 * N_SYM + sl) * 32 + 4 * cg);
          v += s1_ld4(tapS + (1 * N_SYM + sc) * 32 + 4 * cg);
          v += s1_ld4(tapS + (2 * N_SYM + sr) * 32 + 4 * cg);
        }
        m = s1_max4(m, v);
      }
    }
    *reinterpret_cast<f32x4*>(out + (size_t)j2 * 32 + 4 * cg) = m;   // 8 lanes write one 128-byte column
  }
}

template <int SRC>  // 0: symbol rows in HBM (dense path), 1: packed genome
__global__ __launch_bounds__(S1_THREADS) void snv_stage1_kernel(const Stage1Args args) {
  extern __shared__ __attribute__((aligned(16))) float s1mem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* lutL = s1mem;
  float* lutM = s1mem + SNV_LUTBLK;
  const bool pair = args.lut4 != nullptr;      // large tower through the pair table (launch_snv_stage1 sized the LDS for it)
  float* lut4 = s1mem + 2 * SNV_LUTBLK;
  uint8_t* wbase = reinterpret_cast<uint8_t*>(s1mem + 2 * SNV_LUTBLK + (pair ? SNV_LUT4 : 0)) + (size_t)wave * args.wave_bytes;
  uint8_t* cb = wbase;                        // [CW] symbols, PAD at both ends
  uint8_t* kwL = cb + args.cw;                // [L2 large][16]
  uint8_t* kwM = kwL + args.tw[0].L2 * 16;    // [L2 mid][4]

  for (int i = tid * 4; i < SNV_LUTBLK; i += S1_THREADS * 4) {
    *reinterpret_cast<f32x4*>(lutL + i) = s1_ld4(args.lut[0] + i);
    *reinterpret_cast<f32x4*>(lutM + i) = s1_ld4(args.lut[1] + i);
  }
  if (pair)
    for (int i = tid * 4; i < SNV_LUT4; i += S1_THREADS * 4) *reinterpret_cast<f32x4*>(lut4 + i) = s1_ld4(args.lut4 + i);
  __syncthreads();

  const int Lwin = args.Lwin;
  const int64_t stride = (int64_t)gridDim.x * S1_WAVES;
  int64_t row = (int64_t)blockIdx.x * S1_WAVES + wave;
  int64_t pos_pre = 0;
  uint32_t neg_pre = 0;
  if (SRC == 1 && row < args.n) {
    pos_pre = args.pos[row];
    neg_pre = args.strand[row];
  }
  for (; row < args.n; row += stride) {
    // ------------------------------------------------------------------ symbols of the window -> LDS
    if (SRC == 1) {
      const int64_t ws = pos_pre - args.radius;
      const bool neg = neg_pre != 0;
      const int64_t glen = args.genome.length;
      if (row + stride < args.n) {            // site of the next iteration: requested a whole iteration ahead
        pos_pre = args.pos[row + stride];
        neg_pre = args.strand[row + stride];
      }
      if (lane == 0) {
        cb[0] = SYM_PAD;
        cb[Lwin + 1] = SYM_PAD;
      }
      for (int wi = lane; wi < args.nwords; wi += 64) {   // one lane per 16-base word of the packed genome
        const int64_t w = (ws >> 4) + wi;
        uint32_t word = 0, mword = 0;
        if (w >= 0 && 16 * w < glen) {
          word = args.genome.packed2[w];
          mword = (args.genome.nmask[w >> 1] >> (16u * (uint32_t)(w & 1))) & 0xFFFFu;
        }
        // the word's first base as a window column on the plus strand (32 bits: |16 w - ws| < 16 nwords); bases outside the chromosome
        // read as N: their mask bits are set here, once per word (the per-base form did this in 64-bit position arithmetic, ~25 vector
        // instructions per base; this kernel is bound by vector-instruction issue -- PMC: 2800 per site, 70 % of the issue slots)
        const int j0 = (int)(16 * w - ws);
        if (w < 0 || 16 * w >= glen) mword = 0xFFFFu;
        else if (16 * w + 15 >= glen) mword |= 0xFFFFu << (uint32_t)(glen - 16 * w);
        const uint32_t cw = neg ? ~word : word;              // complement: 3 - base in every 2-bit field
        const int jb = neg ? Lwin - 1 - j0 - 15 : j0;        // window column of the lowest-addressed byte this word writes
        uint8_t* dst = cb + jb + 1;
        // (ONE form for every word: the window's two end words sit in some lane of every round, so a separate fast form for the
        // interior words would only be executed in addition -- the wave runs both sides of a divergent branch)
        const bool nfree = (mword & 0xFFFFu) == 0u;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const int o = neg ? 15 - k : k;
          uint32_t sym = (cw >> (2 * k)) & 3u;
          if (!nfree) sym = ((mword >> k) & 1u) ? (uint32_t)SYM_N : sym;
          if ((unsigned)(jb + o) < (unsigned)Lwin) dst[o] = (uint8_t)sym;
        }
      }
      if (args.genome.n_amb > 0) {            // IUPAC codes of the side table overwrite the N the mask produced
        const int64_t e0 = amb_lower_bound_wave(args.genome.amb_pos, args.genome.n_amb, ws, lane);
        for (int64_t e = e0 + lane; e < args.genome.n_amb; e += 64) {
          const int64_t gpos = args.genome.amb_pos[e];
          if (gpos >= ws + Lwin) break;
          uint32_t sym = args.genome.amb_sym[e];
          if (neg) sym = sym_complement(sym);
          cb[(int)(neg ? (ws + Lwin - 1 - gpos) : (gpos - ws)) + 1] = (uint8_t)sym;
        }
      }
    } else {
      const uint8_t* src = args.codes + row * Lwin;
      for (int jj = lane; jj < Lwin + 2; jj += 64) {
        const int j = jj - 1;
        cb[jj] = (j >= 0 && j < Lwin) ? src[j] : (uint8_t)SYM_PAD;
      }
    }
    wave_lds_fence();
    if (pair) build_kpair(args.tw[0], lane, cb, kwL);
    else build_kwin<16>(args.tw[0], lane, cb, kwL);
    build_kwin<4>(args.tw[1], lane, cb, kwM);
    wave_lds_fence();
    float* out = args.x0 + (size_t)(args.dbg_alias ? (row & 63) : row) * args.x0_cols * 32;
    if (pair) pooled_lookup_pair(args.tw[0], lane, lutL, lut4, cb, kwL, out);
    else pooled_lookup<16>(args.tw[0], lane, lutL, cb, kwL, out);
    pooled_lookup<4>(args.tw[1], lane, lutM, cb, kwM, out + (size_t)args.tw[0].L2 * 32);
    wave_lds_fence();   // the next iteration overwrites cb / kw
  }
}

// Small batches (the reference's default predict call has 16 sites): ONE WORKGROUP PER SITE -- the 16 waves split the window decode
// and the pooled columns of a single site, so the latency of the launch is a sixteenth of the wave-per-site kernel's.
template <int SRC>
__global__ __launch_bounds__(S1_THREADS) void snv_stage1_site_kernel(const Stage1Args args) {
  extern __shared__ __attribute__((aligned(16))) float s1mem[];
  const int tid = threadIdx.x;
  if (args.loc_on && (int64_t)blockIdx.x == args.n) {   // the extra workgroup: local branch of the whole (small) batch, 4 of the 16 waves
    if (tid < LOC_THREADS) local_mlp_mfma_body(args.loc, args.loc_cat, args.n, args.loc_out, args.loc_d, s1mem, 0, 1);
    return;
  }
  float* lutL = s1mem;
  float* lutM = s1mem + SNV_LUTBLK;
  uint8_t* cb = reinterpret_cast<uint8_t*>(s1mem + 2 * SNV_LUTBLK);
  uint8_t* kwL = cb + args.cw;
  uint8_t* kwM = kwL + args.tw[0].L2 * 16;
  for (int i = tid * 4; i < SNV_LUTBLK; i += S1_THREADS * 4) {
    *reinterpret_cast<f32x4*>(lutL + i) = s1_ld4(args.lut[0] + i);
    *reinterpret_cast<f32x4*>(lutM + i) = s1_ld4(args.lut[1] + i);
  }
  const int Lwin = args.Lwin;
  if (args.zero != nullptr && blockIdx.x == 0)
    for (int64_t i = tid; i < args.n; i += S1_THREADS) args.zero[i] = 0;
  for (int64_t row = blockIdx.x; row < args.n; row += gridDim.x) {
    __syncthreads();                         // previous site's readers are done (first pass: nothing to wait for but the LUT writers)
    if (SRC == 1) {
      const int64_t ws = args.pos[row] - args.radius;
      const bool neg = args.strand[row] != 0;
      if (tid == 0) {
        cb[0] = SYM_PAD;
        cb[Lwin + 1] = SYM_PAD;
      }
      for (int j = tid; j < Lwin; j += S1_THREADS) {
        const int64_t gpos = neg ? ws + Lwin - 1 - j : ws + j;
        uint32_t sym = genome_sym_iupac(args.genome, gpos);
        if (neg) sym = sym_complement(sym);
        cb[j + 1] = (uint8_t)sym;
      }
    } else if (SRC == 2) {                   // the dense window itself: 4 channel loads per column, up to 2 columns per thread in flight
      const float* src = args.dense + (size_t)row * 4 * Lwin;
      if (tid == 0) {
        cb[0] = SYM_PAD;
        cb[Lwin + 1] = SYM_PAD;
      }
      bool bad = false;
      for (int j0 = tid; j0 < Lwin; j0 += 2 * S1_THREADS) {
        const int j1 = j0 + S1_THREADS;
        const int j1c = j1 < Lwin ? j1 : j0;
        float v[2][4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          v[0][ch] = src[(size_t)ch * Lwin + j0];
          v[1][ch] = src[(size_t)ch * Lwin + j1c];
        }
        int s0 = dense_symbol(v[0][0], v[0][1], v[0][2], v[0][3]);
        int s1 = dense_symbol(v[1][0], v[1][1], v[1][2], v[1][3]);
        bad |= (s0 < 0) | (s1 < 0);
        cb[j0 + 1] = (uint8_t)(s0 < 0 ? SYM_N : s0);
        if (j1 < Lwin) cb[j1 + 1] = (uint8_t)(s1 < 0 ? SYM_N : s1);
      }
      if (bad && args.status != nullptr) atomicOr(args.status, (int)MURAL_E_ENCODING);
    } else {
      const uint8_t* src = args.codes + row * Lwin;
      for (int jj = tid; jj < Lwin + 2; jj += S1_THREADS) {
        const int j = jj - 1;
        cb[jj] = (j >= 0 && j < Lwin) ? src[j] : (uint8_t)SYM_PAD;
      }
    }
    __syncthreads();
    build_kwin<16>(args.tw[0], tid, cb, kwL, S1_THREADS);
    build_kwin<4>(args.tw[1], tid, cb, kwM, S1_THREADS);
    __syncthreads();
    float* out = args.x0 + (size_t)row * args.x0_cols * 32;
    pooled_lookup<16>(args.tw[0], tid, lutL, cb, kwL, out, S1_THREADS);
    pooled_lookup<4>(args.tw[1], tid, lutM, cb, kwM, out + (size_t)args.tw[0].L2 * 32, S1_THREADS);
  }
}

// ------------------------------------------------------------------------------------------------ training mode
// The same table formulation serves the training step (MuRaL/training.py:424-433 through model_snv.py:473-475,496-497
// with BatchNorm1d(4) on batch statistics): the tables are rebuilt on the device every step (train_ops.hip,
// first_tables_kernel), the forward keeps the window offset of every pooled maximum (one byte per output, channel-last)
// and the backward scatters the pooled gradient into a gradient TABLE (d lut | d taps | d bias) that the caller folds back
// into conv weight / BN(4) gradients.  Activations leave in the training layout [B][32][L2] (NCL).

template <int SLOT>
__device__ __forceinline__ void pooled_lookup_train(const Stage1Tower& g, int lane, const float* lutS, const uint8_t* cb0,
                                                    const uint8_t* kw, float* __restrict__ y /* [32][L2] or [L2][32] */,
                                                    uint8_t* __restrict__ arg /* [L2][32] */, int cl, f32x4& sum1, f32x4& sum2) {
  const float* tapS = lutS + SNV_LUT;
  const float* b0S = tapS + SNV_TAPS;
  const uint8_t* cb = cb0 + g.col0;
  const int cg = lane & 7;
  const int total = g.L2 * 8;
  for (int task = lane; task < total; task += 64) {
    const int j2 = task >> 3;
    uint32_t d[SLOT / 4];
    if (SLOT == 16) {
      const uint4 q = *reinterpret_cast<const uint4*>(kw + (size_t)j2 * 16);
      d[0] = q.x; d[1] = q.y; d[2] = q.z; d[SLOT / 4 - 1] = q.w;
    } else {
      d[0] = *reinterpret_cast<const uint32_t*>(kw + (size_t)j2 * 4);
    }
    uint32_t hi = d[SLOT / 4 - 1] & 0x00FFFFFFu;
#pragma unroll
    for (int r = 0; r + 1 < SLOT / 4; ++r) hi |= d[r];
    const bool fast = ((hi & 0x80808080u) == 0u) && (g.pk == SLOT - 1);
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    uint32_t am[4] = {0u, 0u, 0u, 0u};                 // first maximum wins, like MaxPool1d(return_indices)
    if (fast) {
#pragma unroll
      for (int w = 0; w < SLOT - 1; ++w) {
        const uint32_t idx = (d[w >> 2] >> (8 * (w & 3))) & 0xFFu;
        const f32x4 v = s1_ld4(lutS + idx * 32u + 4u * cg);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool gt = v[q] > m[q];
          m[q] = gt ? v[q] : m[q];
          am[q] = gt ? (uint32_t)w : am[q];
        }
      }
    } else {
      const int jlo = j2 * g.ps - g.pp;
      for (int w = 0; w < g.pk; ++w) {
        const int j = jlo + w;
        if (j < 0 || j >= g.L1) continue;
        const uint32_t idx = kmer_index(cb, j, g.L1);
        f32x4 v;
        if (idx != 255u) {
          v = s1_ld4(lutS + idx * 32u + 4u * cg);
        } else {
          const uint32_t sl = (j == 0) ? (uint32_t)SYM_PAD : cb[j];
          const uint32_t sc = cb[j + 1];
          const uint32_t sr = (j == g.L1 - 1) ? (uint32_t)SYM_PAD : cb[j + 2];
          v = s1_ld4(b0S + 4 * cg);
          v += s1_ld4(tapS + (0 * N_SYM + sl) * 32 + 4 * cg);
          v += s1_ld4(tapS + (1 * N_SYM + sc) * 32 + 4 * cg);
          v += s1_ld4(tapS + (2 * N_SYM + sr) * 32 + 4 * cg);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool gt = v[q] > m[q];
          m[q] = gt ? v[q] : m[q];
          am[q] = gt ? (uint32_t)w : am[q];
        }
      }
    }
    if (cl) {
      *reinterpret_cast<f32x4*>(y + (size_t)j2 * 32 + 4 * cg) = f32x4{m[0], m[1], m[2], m[3]};
      // batch sums of relu(y) for the BatchNorm of the first ResBlock (lanes with the same lane & 7 own the same four channels)
      const f32x4 r = f32x4{fmaxf(m[0], 0.f), fmaxf(m[1], 0.f), fmaxf(m[2], 0.f), fmaxf(m[3], 0.f)};
      sum1 += r;
      sum2 += f32x4{r.x * r.x, r.y * r.y, r.z * r.z, r.w * r.w};
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) y[(size_t)(4 * cg + q) * g.L2 + j2] = m[q];
    }
    *reinterpret_cast<uint32_t*>(arg + (size_t)j2 * 32 + 4 * cg) = am[0] | (am[1] << 8) | (am[2] << 16) | (am[3] << 24);
  }
}

// gradient scatter: acc = workgroup-private (d lut | d taps | d bias) block in LDS
template <int SLOT>
__device__ __forceinline__ void pooled_scatter(const Stage1Tower& g, int lane, float* acc, const uint8_t* cb0,
                                               const uint8_t* kw, const float* __restrict__ dy /* [32][L2] or [L2][32] */,
                                               const uint8_t* __restrict__ arg /* [L2][32] */, int cl) {
  float* tapA = acc + SNV_LUT;
  float* b0A = tapA + SNV_TAPS;
  const uint8_t* cb = cb0 + g.col0;
  const int cg = lane & 7;
  const int total = g.L2 * 8;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  // a wave owns one row: its ~17 rounds would each wait a full global round trip for their gradient, so the loads of NT rounds are
  // issued together before any of them is scattered
  constexpr int NT = 6;
  for (int t0 = lane; t0 < total; t0 += 64 * NT) {
    uint32_t awv[NT];
    f32x4 gv[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int task = t0 + 64 * u;
      const int j2 = (task < total ? task : t0) >> 3;
      awv[u] = *reinterpret_cast<const uint32_t*>(arg + (size_t)j2 * 32 + 4 * cg);
      if (cl) {
        gv[u] = *reinterpret_cast<const f32x4*>(dy + (size_t)j2 * 32 + 4 * cg);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) gv[u][q] = dy[(size_t)(4 * cg + q) * g.L2 + j2];
      }
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int task = t0 + 64 * u;
      if (task >= total) break;
      const int j2 = task >> 3;
      const uint32_t aw = awv[u];
      const int jlo = j2 * g.ps - g.pp;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 4 * cg + q;
        const float gq = gv[u][q];
        const int w = (int)((aw >> (8 * q)) & 0xFFu);
        const uint32_t idx = kw[(size_t)j2 * SLOT + w];
        bsum[q] += gq;
        if (idx != 255u) {
          atomicAdd(&acc[idx * 32u + c], gq);
        } else {
          const int j = jlo + w;
          const uint32_t sl = (j == 0) ? (uint32_t)SYM_PAD : cb[j];
          const uint32_t sc = cb[j + 1];
          const uint32_t sr = (j == g.L1 - 1) ? (uint32_t)SYM_PAD : cb[j + 2];
          atomicAdd(&tapA[(0 * N_SYM + sl) * 32 + c], gq);
          atomicAdd(&tapA[(1 * N_SYM + sc) * 32 + c], gq);
          atomicAdd(&tapA[(2 * N_SYM + sr) * 32 + c], gq);
        }
      }
    }
  }
  // the table entries contain the conv bias; its gradient is the plain sum of the pooled gradients.  Lanes with the same
  // channel group (lane & 7) meet through shuffles before one LDS atomic per channel.
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v = bsum[q];
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (lane < 8) atomicAdd(&b0A[4 * cg + q], v);
  }
}

#define FT_STAMP(K) do { if (a.stamps && tid == 0) a.stamps[8 * blockIdx.x + (K)] = __builtin_amdgcn_s_memrealtime(); } while (0)


// The same scatter for the channel-last gradient of the composed step, one CHANNEL per lane: lane = (parity of the pooled column,
// channel).  The 32 lanes of a half-wave add into 32 different LDS banks (table rows are 32 floats), and a wave's gradient load is 256
// contiguous bytes.  The form above -- lane = (column, four channels) -- puts the eight columns of a wave instruction on the same
// eight banks: eight serial read-modify-writes per bank, 75 of the launch's 100 us at batch 4096 (tools/phase_stamps_first.py).
template <int SLOT>
__device__ __forceinline__ void pooled_scatter_cl(const Stage1Tower& g, int lane, float* acc, const uint8_t* cb0, const uint8_t* kw,
                                                  const float* __restrict__ dy /* [L2][32] */, const uint8_t* __restrict__ arg /* [L2][32] */,
                                                  int dbg) {
  float* tapA = acc + SNV_LUT;
  float* b0A = tapA + SNV_TAPS;
  const uint8_t* cb = cb0 + g.col0;
  const int c = lane & 31, h = lane >> 5;
  const int npair = (g.L2 + 1) >> 1;
  float bsum = 0.f;
  // the row through range-checked descriptors (a column pair behind the row reads as zero), one lane offset, the pair in the immediate
  const __amdgpu_buffer_rsrc_t gd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, (dbg & 2) ? 0 : g.L2 * 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t ad = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(arg), 0, (dbg & 2) ? 0 : g.L2 * 32, 0x00020000);
  constexpr int NT = 16;                       // column pairs in flight per lane (a row of 134 pooled columns: five round trips)
  for (int p0 = 0; p0 < npair; p0 += NT) {
    float gv[NT];
    uint32_t awv[NT];
    const uint32_t go = 4u * (uint32_t)lane + 256u * (uint32_t)p0, ao = (uint32_t)lane + 64u * (uint32_t)p0;
    int hh = h;                                // opaque per batch: the per-pair address pieces (three per pair) are made here, not hoisted
    asm volatile("" : "+v"(hh));               // out of the batch loop into 48 registers
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      awv[u] = __builtin_amdgcn_raw_buffer_load_b8(ad, ao + 64u * u, 0, 0);
      gv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gd, go + 256u * u, 0, 0));
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int j2 = 2 * (p0 + u) + hh;
      if (j2 < g.L2) {
        const float gq = gv[u];
        const int w = (int)(awv[u] & 0xFFu);
        const uint32_t idx = (dbg & 4) ? (uint32_t)(j2 & 63) : kw[j2 * SLOT + w];
        bsum += gq;
        if (dbg & 1) continue;
        // one add on the common path; a 3-mer with a symbol outside ACGTN (idx 255) goes to the three per-tap tables instead
        const int j = j2 * g.ps - g.pp + w;
        const uint32_t sl = (j == 0) ? (uint32_t)SYM_PAD : cb[j];
        const bool tab = idx != 255u;
        atomicAdd(tab ? &acc[idx * 32u + c] : &tapA[(0 * N_SYM + sl) * 32 + c], gq);
        if (!tab) {
          const uint32_t sc = cb[j + 1];
          const uint32_t sr = (j == g.L1 - 1) ? (uint32_t)SYM_PAD : cb[j + 2];
          atomicAdd(&tapA[(1 * N_SYM + sc) * 32 + c], gq);
          atomicAdd(&tapA[(2 * N_SYM + sr) * 32 + c], gq);
        }
      }
      if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (left alone the scheduler hoists all sixteen index reads: spills)
    }
  }
  bsum += __shfl_xor(bsum, 32);
  if (lane < 32) atomicAdd(&b0A[c], bsum);
}

template <int SLOT, bool BWD>
__global__ __launch_bounds__(S1_THREADS) void first_train_kernel(const FirstTrainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s1mem[];
  const int tid = threadIdx.x;
  FT_STAMP(0);
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* blk = s1mem;                         // forward: lut | taps | bias; backward: their gradient accumulators
  uint8_t* cb = reinterpret_cast<uint8_t*>(s1mem + SNV_LUTBLK) + (size_t)wave * a.wave_bytes;
  uint8_t* kw = cb + a.cw;
  for (int i = tid * 4; i < SNV_LUTBLK; i += S1_THREADS * 4)
    *reinterpret_cast<f32x4*>(blk + i) = BWD ? f32x4{0.f, 0.f, 0.f, 0.f} : s1_ld4(a.lutblk + i);
  __syncthreads();
  FT_STAMP(1);
  const Stage1Tower g = a.tw;
  const int Lwin = a.Lwin;
  f32x4 sum1 = f32x4{0.f, 0.f, 0.f, 0.f}, sum2 = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int64_t row = (int64_t)blockIdx.x * S1_WAVES + wave; row < a.B; row += (int64_t)gridDim.x * S1_WAVES) {
    const uint8_t* src = a.sym + row * Lwin;
    constexpr int UN = 8;                     // byte loads in flight per lane (one wave loads the whole row)
    for (int jj0 = lane; jj0 < Lwin + 2; jj0 += 64 * UN) {
      uint8_t v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int j = jj0 + 64 * u - 1;
        v[u] = (j >= 0 && j < Lwin) ? src[j] : (uint8_t)SYM_PAD;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (jj0 + 64 * u < Lwin + 2) cb[jj0 + 64 * u] = v[u];
    }
    wave_lds_fence();
    FT_STAMP(2);
    build_kwin<SLOT>(g, lane, cb, kw);
    wave_lds_fence();
    FT_STAMP(3);
    const size_t o = (size_t)row * 32 * g.L2;
    if (BWD) pooled_scatter<SLOT>(g, lane, blk, cb, kw, a.dy + o, a.arg + o, a.cl);
    else pooled_lookup_train<SLOT>(g, lane, blk, cb, kw, a.y + o, a.arg + o, a.cl, sum1, sum2);
    wave_lds_fence();
    FT_STAMP(4);
  }
  if (!BWD && a.stat) {                       // 16 waves -> 64 sums through LDS -> one double atomic per sum and workgroup
    __syncthreads();                          // every wave is done with the tables: their LDS carries the partial sums
    float* red = blk;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v1 = sum1[q], v2 = sum2[q];
#pragma unroll
      for (int off = 8; off < 64; off <<= 1) {
        v1 += __shfl_xor(v1, off);
        v2 += __shfl_xor(v2, off);
      }
      if (lane < 8) {
        red[wave * 64 + 4 * lane + q] = v1;
        red[wave * 64 + 32 + 4 * lane + q] = v2;
      }
    }
    __syncthreads();
    if (tid < 64) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < S1_WAVES; ++w) v += red[w * 64 + tid];
      atomicAdd(&a.stat[(size_t)(blockIdx.x % MURAL_BN_SLOTS) * 64 + tid], (double)v);
    }
  }
  if (BWD) {                                  // one partial block per workgroup; first_grad_fold_kernel sums them in order
    __syncthreads();
    FT_STAMP(5);
    float* dst = a.dpart + (size_t)blockIdx.x * SNV_LUTBLK;
    for (int i = tid * 4; i < SNV_LUTBLK; i += S1_THREADS * 4) *reinterpret_cast<f32x4*>(dst + i) = s1_ld4(blk + i);
  }
  FT_STAMP(6);
}


// ---- backward of the channel-last form (the composed training step) ---------------------------------------------------------------
// The scatter above is bound by the LDS float atomics themselves: a ds_add_f32 wave instruction occupies the LDS pipe for ~128
// cycles whatever its bank pattern (tools/phase_stamps_first.py with MURAL_DEBUG_FIRST: 100 -> 22 us per batch of 4096 without the
// adds, the same 100 us with one channel per lane, i.e. conflict-free) -- 4288 adds per row is 75 us of a 100 us launch.  This kernel
// has none on its common path.  The 3-mer table is a sum of three per-tap symbol tables (lut[l,m,r] = bias + tap0[l] + tap1[m] +
// tap2[r]; first_param_grad_kernel folds d lut back into d taps anyway), and a tap sees one of FOUR common symbols: a lane owns one
// channel and keeps d tap[t][A C G T] of it in twelve registers; a pooled gradient goes to the three registers its window position's
// symbols select (compare / select / add, no memory).  Symbols outside A C G T (N, IUPAC codes, the zero padding) take an LDS atomic
// on a shared per-tap table.  Partial tables meet per workgroup in a fixed order (bitwise reproducible; the atomics were not).
// FOLD: the pooled gradient is made from the BatchNorm-backward apply of the layer behind (FirstFold, snv.h) instead of read: four
// 4-byte loads per element against the one of a gradient tensor that a pass of its own (four reads, one write) would have had to make.
constexpr int FB_ROWS = 13;                  // twelve per-tap sums + the bias gradient
template <bool FOLD>
__global__ __launch_bounds__(S1_THREADS) void first_bwd_cl_kernel(const FirstTrainArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s1mem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  FT_STAMP(0);
  float* rare = s1mem;                                          // [3][N_SYM][32]
  float* red = rare + SNV_TAPS;                                 // [S1_WAVES][FB_ROWS][32]
  uint8_t* cb0 = reinterpret_cast<uint8_t*>(red + S1_WAVES * FB_ROWS * 32) + (size_t)wave * a.cw;
  for (int i = tid; i < SNV_TAPS; i += S1_THREADS) rare[i] = 0.f;
  const int c = lane & 31, h = lane >> 5;
  float k0 = 0.f, m1 = 0.f, m2 = 0.f, mu = 0.f, is = 0.f;      // FOLD: constants of this lane's channel
  if constexpr (FOLD) {
    float* fc = red;                                            // [5][32], read before the partial sums take the region
    if (tid < 32) {
      double s1 = 0.0, s2 = 0.0;
      for (int k = 0; k < MURAL_BN_SLOTS; ++k) {
        s1 += a.fold.acc[((size_t)k * 2 + 0) * 32 + tid];
        s2 += a.fold.acc[((size_t)k * 2 + 1) * 32 + tid];
      }
      fc[tid] = a.fold.gamma[tid] * a.fold.state[3 * 32 + tid];
      fc[32 + tid] = (float)(s1 / a.fold.n);
      fc[64 + tid] = (float)(s2 / a.fold.n);
      fc[96 + tid] = a.fold.state[2 * 32 + tid];
      fc[128 + tid] = a.fold.state[3 * 32 + tid];
      if (blockIdx.x == 0) {
        a.fold.dgamma[tid] = (float)s2;
        a.fold.dbeta[tid] = (float)s1;
      }
    }
    __syncthreads();
    k0 = fc[c]; m1 = fc[32 + c]; m2 = fc[64 + c]; mu = fc[96 + c]; is = fc[128 + c];
  }
  __syncthreads();
  FT_STAMP(1);
  const Stage1Tower g = a.tw;
  const int Lwin = a.Lwin;
  const int npair = (g.L2 + 1) >> 1;
  const uint8_t* cb = cb0 + g.col0;                             // cb[j + 1] is the symbol of tower column j
  float acc[3][4];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[t][q] = 0.f;
  float bsum = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * S1_WAVES + wave; row < a.B; row += (int64_t)gridDim.x * S1_WAVES) {
    const uint8_t* src = a.sym + row * Lwin;
    constexpr int UN = 8;                     // byte loads in flight per lane (one wave loads the whole row)
    for (int jj0 = lane; jj0 < Lwin + 2; jj0 += 64 * UN) {
      uint8_t v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int j = jj0 + 64 * u - 1;
        v[u] = (j >= 0 && j < Lwin) ? src[j] : (uint8_t)SYM_PAD;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (jj0 + 64 * u < Lwin + 2) cb0[jj0 + 64 * u] = v[u];
    }
    wave_lds_fence();
    FT_STAMP(2);
    const size_t o = (size_t)row * 32 * g.L2;
    // the row through range-checked descriptors (a column pair behind the row reads as zero), one lane offset, the pair in the immediate
    const int gbytes = g.L2 * 128;
    const __amdgpu_buffer_rsrc_t gd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((FOLD ? a.fold.dz : a.dy) + o), 0, gbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ad = __builtin_amdgcn_make_buffer_rsrc(a.arg + o, 0, g.L2 * 32, 0x00020000);
    // (an absent tensor: a descriptor of zero bytes over any valid address -- every load through it returns 0)
    auto opt = [&](const float* p) __attribute__((always_inline)) {
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p ? p + o : a.dy), 0, p ? gbytes : 0, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t xd = opt(FOLD ? a.fold.x : nullptr), r1d = opt(FOLD ? a.fold.add1 : nullptr), r2d = opt(FOLD ? a.fold.add2 : nullptr);
    constexpr int NT = FOLD ? 8 : 16;         // column pairs in flight per lane (a row of 134 pooled columns: five / nine round trips)
    for (int p0 = 0; p0 < npair; p0 += NT) {
      float gv[NT];
      uint32_t awv[NT];
      const uint32_t go = 4u * (uint32_t)lane + 256u * (uint32_t)p0, ao = (uint32_t)lane + 64u * (uint32_t)p0;
      int hh = h;                             // opaque per batch: the per-pair address pieces are made here, not hoisted out of the
      asm volatile("" : "+v"(hh));            // batch loop into 48 registers
      if constexpr (FOLD) {
        float xv[NT], r1[NT], r2[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          awv[u] = __builtin_amdgcn_raw_buffer_load_b8(ad, ao + 64u * u, 0, 0);
          gv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gd, go + 256u * u, 0, 0));
          xv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xd, go + 256u * u, 0, 0));
          r1[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1d, go + 256u * u, 0, 0));
          r2[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r2d, go + 256u * u, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < NT; ++u) {        // (a pair behind the row: x = 0 is masked like every x <= 0, the residuals read 0)
          const float xh = (xv[u] - mu) * is;
          const float gq = k0 * ((gv[u] - m1) - xh * m2);
          gv[u] = ((xv[u] > 0.f ? gq : 0.f) + r1[u]) + r2[u];
        }
      } else {
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          awv[u] = __builtin_amdgcn_raw_buffer_load_b8(ad, ao + 64u * u, 0, 0);
          gv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gd, go + 256u * u, 0, 0));
        }
      }
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int j2 = 2 * (p0 + u) + hh;
        // a pair behind the row (odd row length) loaded zeros: it adds 0 at the last column's symbols -- in-range table addresses
        const float gq = j2 < g.L2 ? gv[u] : 0.f;
        const int j = (j2 < g.L2 ? j2 : g.L2 - 1) * g.ps - g.pp + (int)(awv[u] & 0xFFu);
        uint32_t sy[3] = {cb[j], cb[j + 1], cb[j + 2]};
        sy[0] = j == 0 ? (uint32_t)SYM_PAD : sy[0];
        sy[2] = j == g.L1 - 1 ? (uint32_t)SYM_PAD : sy[2];
        bsum += gq;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[t][q] += sy[t] == (uint32_t)q ? gq : 0.f;
        if ((sy[0] | sy[1] | sy[2]) > 3u) {                     // rare: N / IUPAC / padding at one of the three taps
#pragma unroll
          for (int t = 0; t < 3; ++t)
            if (sy[t] > 3u) atomicAdd(&rare[(t * N_SYM + sy[t]) * 32 + c], gq);
        }
        if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
    wave_lds_fence();
    FT_STAMP(4);
  }
  // the two halves of a wave meet through a shuffle, the sixteen waves through LDS in a fixed order
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float v = acc[t][q] + __shfl_xor(acc[t][q], 32);
      if (lane < 32) red[(wave * FB_ROWS + 4 * t + q) * 32 + c] = v;
    }
  {
    const float v = bsum + __shfl_xor(bsum, 32);
    if (lane < 32) red[(wave * FB_ROWS + 12) * 32 + c] = v;
  }
  __syncthreads();
  FT_STAMP(5);
  // one partial block per workgroup in the layout of the forward's tables (lut | taps | bias): the 3-mer part stays zero
  float* dst = a.dpart + (size_t)blockIdx.x * SNV_LUTBLK;
  for (int i = tid * 4; i < SNV_LUT; i += S1_THREADS * 4) *reinterpret_cast<f32x4*>(dst + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < SNV_TAPS + 32; i += S1_THREADS) {
    const int t = i / (N_SYM * 32), r = i - t * N_SYM * 32, sym = r >> 5, cc = r & 31;
    const int k = i >= SNV_TAPS ? 12 : (sym < 4 ? 4 * t + sym : -1);
    float v = i < SNV_TAPS ? rare[i] : 0.f;
    if (k >= 0) {
      const int ch = i >= SNV_TAPS ? i - SNV_TAPS : cc;
#pragma unroll
      for (int w = 0; w < S1_WAVES; ++w) v += red[(w * FB_ROWS + k) * 32 + ch];
    }
    dst[SNV_LUT + i] = v;
  }
  FT_STAMP(6);
}

#undef FT_STAMP

unsigned long long* g_first_stamps = nullptr;      // diagnostic (mural_debug_first_set_stamps)

int first_train_grid(int64_t B) {
  const int64_t want = (B + S1_WAVES - 1) / S1_WAVES;
  return (int)(want < 1 ? 1 : (want < FIRST_TRAIN_MAXGRID ? want : FIRST_TRAIN_MAXGRID));
}

bool first_train_supported(int C, int pk) { return C == SNV_C && pk >= 1 && pk <= 15; }

int launch_first_train(FirstTrainArgs a, bool bwd, hipStream_t stream) {
  if (a.B == 0) return MURAL_OK;
  MURAL_REQUIRE(first_train_supported(SNV_C, a.tw.pk), "first layer: pool window %d not supported by the table kernel", a.tw.pk);
  const int slot = a.tw.pk <= 3 ? 4 : 16;
  a.cw = (a.Lwin + 2 + 15) & ~15;
  a.wave_bytes = a.cw + ((a.tw.L2 * slot + 15) & ~15);
  const size_t lds = (size_t)SNV_LUTBLK * 4 + (size_t)S1_WAVES * a.wave_bytes;
  MURAL_REQUIRE(lds <= 160 * 1024, "first layer: window of %d columns does not fit the LDS working set", a.Lwin);
  using KernelFn = void (*)(const FirstTrainArgs);
  MURAL_REQUIRE(!a.fold.dz || (bwd && a.cl), "first layer: the folded BatchNorm-backward apply belongs to the channel-last backward");
  if (bwd && a.cl && (a.fold.dz || !dev_env("MURAL_DEBUG_FIRST_SCATTER"))) {      // the composed step's backward: register sums, no kmer windows in LDS
    const size_t lds_b = (size_t)(SNV_TAPS + S1_WAVES * FB_ROWS * 32) * 4 + (size_t)S1_WAVES * a.cw;
    MURAL_REQUIRE(lds_b <= 160 * 1024, "first layer: window of %d columns does not fit the LDS working set", a.Lwin);
    static DynLdsOnce big_b;
    if (int rc = big_b.ensure(&first_bwd_cl_kernel<false>, &first_bwd_cl_kernel<true>)) return rc;
    a.stamps = g_first_stamps;
    if (a.fold.dz) hipLaunchKernelGGL(first_bwd_cl_kernel<true>, dim3(first_train_grid(a.B)), dim3(S1_THREADS), lds_b, stream, a);
    else hipLaunchKernelGGL(first_bwd_cl_kernel<false>, dim3(first_train_grid(a.B)), dim3(S1_THREADS), lds_b, stream, a);
    MURAL_HIP_CHECK(hipGetLastError());
    return MURAL_OK;
  }
  KernelFn fn = slot == 4 ? (bwd ? first_train_kernel<4, true> : first_train_kernel<4, false>)
                          : (bwd ? first_train_kernel<16, true> : first_train_kernel<16, false>);
  static DynLdsOnce big_lds[4];                             // once per instantiation and device (never inside a graph capture)
  if (int rc = big_lds[(slot == 4 ? 0 : 2) + (bwd ? 1 : 0)].ensure(fn)) return rc;
  a.stamps = g_first_stamps;
  static const int dbg = dev_env("MURAL_DEBUG_FIRST") ? atoi(dev_env("MURAL_DEBUG_FIRST")) : 0;
  a.dbg = dbg;
  hipLaunchKernelGGL(fn, dim3(first_train_grid(a.B)), dim3(S1_THREADS), lds, stream, a);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

bool stage1_small_batch(int64_t n) { return n <= 256 && !dev_env("MURAL_DEBUG_NO_SMALL_BATCH"); }

int launch_snv_stage1(const Stage1Args& a, bool packed, size_t lds_bytes, hipStream_t stream) {
  if (a.n == 0) return MURAL_OK;
  MURAL_REQUIRE(a.dense == nullptr || (!packed && stage1_small_batch(a.n)), "stage 1: the dense source is the small-batch kernel's");
  MURAL_REQUIRE(!a.loc_on || stage1_small_batch(a.n), "stage 1: the local-branch workgroup is the small-batch kernel's");
  static DynLdsOnce big_lds;
  if (int rc = big_lds.ensure(&snv_stage1_kernel<0>, &snv_stage1_kernel<1>, &snv_stage1_site_kernel<0>, &snv_stage1_site_kernel<1>,
                              &snv_stage1_site_kernel<2>))
    return rc;
  if (stage1_small_batch(a.n) || a.site_mode) {     // latency-bound call, or a window too long for sixteen per-wave copies: one
                                                    // workgroup per site (+ one for the local branch of a small batch)
    const unsigned grid = stage1_small_batch(a.n) ? (unsigned)a.n + (a.loc_on ? 1u : 0u) : (unsigned)std::min<int64_t>(a.n, 16384);
    if (!packed && a.dense != nullptr)
      hipLaunchKernelGGL(snv_stage1_site_kernel<2>, dim3(grid), dim3(S1_THREADS), lds_bytes, stream, a);
    else if (packed) hipLaunchKernelGGL(snv_stage1_site_kernel<1>, dim3(grid), dim3(S1_THREADS), lds_bytes, stream, a);
    else hipLaunchKernelGGL(snv_stage1_site_kernel<0>, dim3(grid), dim3(S1_THREADS), lds_bytes, stream, a);
    MURAL_HIP_CHECK(hipGetLastError());
    return MURAL_OK;
  }
  const int64_t want = (a.n + S1_WAVES - 1) / S1_WAVES;
  const int grid = (int)(want < 256 ? want : 256);   // one 16-wave workgroup per CU, persistent
  Stage1Args b = a;
  static const int alias = dev_env("MURAL_DEBUG_S1_ALIAS") ? 1 : 0;
  b.dbg_alias = alias;
  if (packed)
    hipLaunchKernelGGL(snv_stage1_kernel<1>, dim3(grid), dim3(S1_THREADS), lds_bytes, stream, b);
  else
    hipLaunchKernelGGL(snv_stage1_kernel<0>, dim3(grid), dim3(S1_THREADS), lds_bytes, stream, b);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

}  // namespace mural

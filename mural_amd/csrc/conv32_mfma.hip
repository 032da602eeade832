// fp32-MFMA kernels for the 32->32, k=3 convolutions of the SNV towers on HBM-resident [B][32][L] tensors: forward /
// input gradient (same kernel, different weight fragments) and weight + bias gradient.  Used by the training step
// (MuRaL/training.py:424-427 autograd over nn.Conv1d, here one explicit kernel per direction).
//
// These layers are HBM-bound at batch 4096 (3.4 GFLOP against 140-280 MB per layer), so the kernels are organised around
// the memory streams: R batch rows (contiguous in memory) form one tile; every global access is a coalesced float4
// stream over that contiguous block - the input, the residuals, the output - and the layout change happens in LDS.
//   LDS image [32 ch][pitch]: the R rows share one flattened column axis with zero separator columns between them (= the
//   conv's zero padding); pitch = 4 mod 32 makes every MFMA operand read conflict-free (v_mfma_f32_16x16x4_f32 operands
//   are one float per lane: lanes = 16 columns x 4 channels or 16 channels x 4 columns).
//   forward : D[16 cout][16 col] += W[cout][k] act[k][col], K = 3 taps x 32 ch; wave = (M-block, column parity).  The
//             accumulators of all blocks stay in registers until every wave is done with the image, then overwrite it
//             in place; the tile leaves through a streaming pass that adds the residuals and, on request, takes the
//             per-channel sums the next BatchNorm (forward) or this BatchNorm's backward needs.
//   wgrad   : dW[cout][(tap, cin)] += dy[cout][col] * act[cin][col + tap - 1]: M = cout, N = (tap, cin) = 6 blocks of 16,
//             K = columns in steps of 4; each wave reduces a quarter of the tile's columns into 12 accumulator tiles
#include <cstdlib>
#include <cstring>

#include "mfma_tile.h"

namespace mural {

constexpr int C32 = 32;
constexpr int C32_KSTEPS = 24;
constexpr int C32_NB2MAX = 9;          // <= 18 blocks of 16 columns per tile
constexpr int C32_MAXCOLS = 16 * 2 * C32_NB2MAX;
constexpr int C32_AUX = 128;           // floats after the image: pre_s | pre_t | stat_mean | stat_invstd

// logical column c (0 = leading separator) lives at image index c + 1; index 0 is the guard read by tap 0 of column 0.
// One compile-time pitch (>= 16 * 18 + 2, = 4 mod 32) for every geometry: operand addresses are one per-lane base + immediates.
constexpr int C32_PITCH = 292;

struct Conv32Args {
  const float* x;        // [B][32][L]
  float* y;              // [B][32][L]
  const float* W;        // PyTorch [32][32][3]; dgrad: use the transposed, tap-flipped filter
  int dgrad;
  // per-channel sums taken while the tile streams out (the producer of a tensor knows its values: no extra pass over HBM)
  int stat_mode;         // 0 none | 1: sum, sum of squares of act(y) (batch statistics of the NEXT BatchNorm)
                         //        | 2: sum(y), sum(y * xhat), xhat = (act(stat_x) - mean) * invstd (BatchNorm backward of y = dz)
  int stat_relu;         // act = relu
  const float* stat_x;   // mode 2: [B][32][L]
  const float* stat_mean;
  const float* stat_invstd;
  double* stat_out;      // accumulator block [MURAL_BN_SLOTS][2][32] (zeroed by the caller), copy picked by workgroup index
  const float* bias;     // [32] or nullptr
  const float* pre_s;    // [32] or nullptr: x' = pre_s * act(x) + pre_t, act = relu if pre_relu
  const float* pre_t;
  const float* res1;     // optional residuals [B][32][L]
  const float* res2;
  int pre_relu, post_relu;
  int B, L, R, Sc, NC, nb;
  FastDiv dSc, dL;
  // optional: finalize the BatchNorm of the conv input here instead of in its own launch.  fin_acc = accumulator block with the
  // batch sums of act(x); every workgroup derives scale / shift from it, workgroup 0 also writes the state the backward reads
  // (scale | shift | mean | invstd) and updates the running statistics (what bn_finalize_kernel does).
  const double* fin_acc;
  double fin_n;
  const float* fin_gamma;
  const float* fin_beta;
  float fin_eps, fin_momentum;
  float* fin_running_mean;
  float* fin_running_var;
  float* fin_state;      // [4][32]
};

// stage the tile's rows (contiguous in memory) into the image, applying the BN(+ReLU) affine; separator columns = 0.
// `aff` is an LDS copy of pre_s | pre_t (64 floats) or nullptr.  zero_tail: also clear the guard and every column behind the
// last staged row up to image index `ncols` (kernels that reduce over columns need exact zeros there).
// The two halves of stage_rows for callers that keep a whole tile's loads in flight (issued one tile ahead): NV float4 per thread
// cover the largest tile (32 channels x 288 columns / 256 threads).
constexpr int C32_NV = 9;
__device__ __forceinline__ void load_rows(const float* __restrict__ x, int64_t b0, int B, int L, int R, int tid, f32x4 (&v4)[C32_NV]) {
  const int rows = (int)((B - b0) < R ? (B - b0) : R);
  const int total = rows > 0 ? rows * C32 * L : 0;
  const float* src = x + (size_t)b0 * C32 * L;
#pragma unroll
  for (int q = 0; q < C32_NV; ++q) {
    const int i0 = tid * 4 + q * 256 * 4;
    v4[q] = i0 < total ? ld4(src + i0) : splat(0.f);
  }
}

__device__ __forceinline__ void scatter_rows(const f32x4 (&v4)[C32_NV], int64_t b0, int B, int L, int R, int Sc, const FastDiv& dL,
                                             const float* aff, int pre_relu, float* img, int tid, int ncols) {
  constexpr int pitch = C32_PITCH;
  const int rows = (int)((B - b0) < R ? (B - b0) : R);
  const int total = rows * C32 * L;
#pragma unroll
  for (int q = 0; q < C32_NV; ++q) {
    const int i0 = tid * 4 + q * 256 * 4;
    if (i0 >= total) break;
    uint32_t rc = dL.div((uint32_t)i0);
    int l = i0 - (int)rc * L;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (l >= L) { l -= L; ++rc; }
      const int r = (int)rc >> 5, ci = (int)rc & 31;
      float v = v4[q][e];
      if (pre_relu) v = fmaxf(v, 0.f);
      if (aff) v = fmaf(aff[ci], v, aff[C32 + ci]);
      img[ci * pitch + 2 + r * Sc + l] = v;
      ++l;
    }
  }
  for (int i = tid; i < (R + 1) * C32; i += 256) img[(i & 31) * pitch + 1 + (i >> 5) * Sc] = 0.f;   // separators
  const int first = 2 + rows * Sc;                       // exact zeros behind the last staged row (kernels that reduce over columns)
  const int n = ncols - first + 2;
  for (int i = tid; i < n * C32; i += 256) {
    const int k = i >> 5;
    img[(i & 31) * pitch + (k == 0 ? 0 : first + k - 1)] = 0.f;
  }
}

template <int STAGE_DEPTH = 8>
__device__ __forceinline__ void stage_rows(const float* __restrict__ x, int64_t b0, int B, int L, int R, int Sc,
                                           const FastDiv& dL, const float* aff, int pre_relu, float* img, int tid,
                                           bool zero_tail, int ncols) {
  constexpr int pitch = C32_PITCH;
  const int rows = (int)((B - b0) < R ? (B - b0) : R);
  const int total = rows * C32 * L;                      // a multiple of 4 (32 channels)
  const float* src = x + (size_t)b0 * C32 * L;           // 128-byte aligned: float4 loads
  // The layer is HBM-bound and a workgroup is only 256 threads: keep STAGE_DEPTH 16-byte loads per thread in flight before the first
  // LDS write (a dependent load-use pair per iteration leaves ~16 KB in flight per CU, a quarter of what the HBM pipe needs).
  for (int base = tid * 4; base < total; base += 256 * 4 * STAGE_DEPTH) {
    f32x4 v4[STAGE_DEPTH];
#pragma unroll
    for (int q = 0; q < STAGE_DEPTH; ++q) {
      const int i0 = base + q * 256 * 4;
      v4[q] = i0 < total ? ld4(src + i0) : splat(0.f);
    }
#pragma unroll
    for (int q = 0; q < STAGE_DEPTH; ++q) {
      const int i0 = base + q * 256 * 4;
      if (i0 >= total) break;
      uint32_t rc = dL.div((uint32_t)i0);
      int l = i0 - (int)rc * L;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (l >= L) { l -= L; ++rc; }                      // next (row, channel)
        const int r = (int)rc >> 5, ci = (int)rc & 31;
        float v = v4[q][e];
        if (pre_relu) v = fmaxf(v, 0.f);
        if (aff) v = fmaf(aff[ci], v, aff[C32 + ci]);
        img[ci * pitch + 2 + r * Sc + l] = v;
        ++l;
      }
    }
  }
  for (int i = tid; i < (R + 1) * C32; i += 256) img[(i & 31) * pitch + 1 + (i >> 5) * Sc] = 0.f;   // separators
  if (zero_tail) {
    const int first = 2 + rows * Sc;                     // image index behind the closing separator of the last row
    const int n = ncols - first + 2;                     // first .. ncols, + the guard
    for (int i = tid; i < n * C32; i += 256) {
      const int k = i >> 5;
      img[(i & 31) * pitch + (k == 0 ? 0 : first + k - 1)] = 0.f;
    }
  }
}

template <int STAT>   // = Conv32Args::stat_mode, compile-time so that the plain forward does not carry the sums
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv32_mfma_kernel(const Conv32Args a) {
  extern __shared__ __attribute__((aligned(16))) float img[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mb = wave & 1, cgp = wave >> 1;
  const int n16 = lane & 15, kk = lane >> 4;
  constexpr int pitch = C32_PITCH;
  float af[C32_KSTEPS];   // A fragments: k-step s = 8 tap + 4 half + q holds W[cout = 16 mb + n16][cin = 16 half + 4 kk + q][tap]
#pragma unroll
  for (int s = 0; s < C32_KSTEPS; ++s) {
    const int t = s / 8, h = (s % 8) / 4, q = s % 4;
    const int cin = 16 * h + 4 * kk + q;               // channel of the tensor being convolved
    const int cout = 16 * mb + n16;                    // channel of the tensor being produced
    af[s] = a.W[a.dgrad ? (cin * C32 + cout) * 3 + (2 - t) : (cout * C32 + cin) * 3 + t];
  }
  const int chv = 16 * mb + 4 * kk;
  const f32x4 bias = a.bias ? ld4(a.bias + chv) : splat(0.f);
  float* aux = img + C32 * pitch;                        // pre_s | pre_t | stat_mean | stat_invstd
  if (a.fin_acc) {
    // 64 sums over the 32 accumulator copies: all 256 threads take part (thread = channel x copy group, its eight loads in flight
    // together) and meet in LDS -- 32 threads walking the copies one dependent load after the other cost 17 us per launch
    double* red = reinterpret_cast<double*>(img);          // [8 groups][2][32] doubles, the image is not staged yet
    {
      const int c = tid & 31, grp = tid >> 5;
      double v[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[2 * q] = a.fin_acc[((size_t)(4 * grp + q) * 2 + 0) * C32 + c];
        v[2 * q + 1] = a.fin_acc[((size_t)(4 * grp + q) * 2 + 1) * C32 + c];
      }
      red[(grp * 2 + 0) * C32 + c] = (v[0] + v[2]) + (v[4] + v[6]);
      red[(grp * 2 + 1) * C32 + c] = (v[1] + v[3]) + (v[5] + v[7]);
    }
    __syncthreads();
    if (tid < C32) {
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int g = 0; g < MURAL_BN_SLOTS / 4; ++g) {
        s1 += red[(g * 2 + 0) * C32 + tid];
        s2 += red[(g * 2 + 1) * C32 + tid];
      }
      const double mean = s1 / a.fin_n;
      double var = s2 / a.fin_n - mean * mean;
      if (var < 0.0) var = 0.0;
      const double invstd = 1.0 / sqrt(var + (double)a.fin_eps);
      const float sc = (float)(a.fin_gamma[tid] * invstd);
      const float sh = (float)(a.fin_beta[tid] - mean * a.fin_gamma[tid] * invstd);
      aux[tid] = sc;
      aux[C32 + tid] = sh;
      if (blockIdx.x == 0) {
        a.fin_state[tid] = sc;
        a.fin_state[C32 + tid] = sh;
        a.fin_state[2 * C32 + tid] = (float)mean;
        a.fin_state[3 * C32 + tid] = (float)invstd;
        if (a.fin_running_mean) {
          const double unbiased = a.fin_n > 1.0 ? var * a.fin_n / (a.fin_n - 1.0) : var;
          a.fin_running_mean[tid] = (float)((1.0 - a.fin_momentum) * a.fin_running_mean[tid] + a.fin_momentum * mean);
          a.fin_running_var[tid] = (float)((1.0 - a.fin_momentum) * a.fin_running_var[tid] + a.fin_momentum * unbiased);
        }
      }
    }
  } else if (a.pre_s && tid < 2 * C32) {
    aux[tid] = tid < C32 ? a.pre_s[tid] : a.pre_t[tid - C32];
  }
  if (STAT == 2 && tid < 2 * C32) aux[2 * C32 + tid] = tid < C32 ? a.stat_mean[tid] : a.stat_invstd[tid - C32];
  const bool has_pre = a.pre_s != nullptr || a.fin_acc != nullptr;
  const int nbw = a.nb > cgp ? (a.nb - cgp + 1) / 2 : 0;
  // operand reads: lane (n16, kk) takes act[cin = 16 h + 4 kk + q][column 16 blk + n16 + tap - 1] = image index (.. + tap)
  const float* rd = img + 4 * kk * pitch + n16;
  float sv[4] = {0.f, 0.f, 0.f, 0.f};                    // STAT 2: sum of the outputs, in accumulator layout
  float racc1 = 0.f, racc2 = 0.f;                        // channel-major sums: thread = (channel tid / 8, column residue tid % 8)
  const int64_t ntiles = ((int64_t)a.B + a.R - 1) / a.R;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t b0 = tile * a.R;
    const int rows = (int)((a.B - b0) < a.R ? (a.B - b0) : a.R);
    __syncthreads();
    stage_rows<4>(a.x, b0, a.B, a.L, a.R, a.Sc, a.dL, has_pre ? aux : nullptr, a.pre_relu, img, tid, false, 0);
    __syncthreads();
    f32x4 acc[C32_NB2MAX + 1];
#pragma unroll
    for (int ip = 0; ip < (C32_NB2MAX + 1) / 2; ++ip) {
      const int i0 = 2 * ip, i1 = 2 * ip + 1;
      if (i0 < nbw) {
        const bool dual = i1 < nbw;
        const float* p0 = rd + 16 * (cgp + 2 * i0);
        const float* p1 = rd + 16 * (cgp + 2 * (dual ? i1 : i0));
        f32x4 a0 = bias, a1 = bias;                      // two independent accumulator chains (40-cycle dependent latency)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          float b0v[8], b1v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            b0v[j] = p0[(16 * (j >> 2) + (j & 3)) * pitch + t];
            b1v[j] = p1[(16 * (j >> 2) + (j & 3)) * pitch + t];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[8 * t + j], b0v[j], a0, 0, 0, 0);
            if (dual) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[8 * t + j], b1v[j], a1, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);             // one tap's operands in flight at a time: the layer is HBM-bound
        }
        acc[i0] = a0;
        acc[i1] = a1;
      }
    }
    __syncthreads();                                     // every wave is done reading the image: overwrite it with the outputs
#pragma unroll
    for (int i = 0; i < C32_NB2MAX; ++i) {
      if (i < nbw) {
        const int c = 16 * (cgp + 2 * i) + n16;          // logical column
        f32x4 v = acc[i];
        if (a.post_relu) v = max4(v, splat(0.f));
#pragma unroll
        for (int q = 0; q < 4; ++q) img[(chv + q) * pitch + 1 + c] = v[q];
        if (STAT == 2 && c >= 1) {
          const uint32_t u = (uint32_t)(c - 1);
          const uint32_t r = a.dSc.div(u);
          if ((int)r < rows && (int)(u - r * (uint32_t)a.Sc) < a.L) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sv[q] += v[q];
          }
        }
      }
    }
    __syncthreads();
    // the tile leaves as one contiguous float4 stream (+ residuals); values the sums need go back into the image
    const int total = rows * C32 * a.L;
    const size_t base = (size_t)b0 * C32 * a.L;
    constexpr int OUT_DEPTH = 2;                          // residual / statistics loads of two float4 groups in flight together
#pragma unroll 1
    for (int ib = tid * 4; ib < total; ib += 256 * 4 * OUT_DEPTH) {
      f32x4 r1v[OUT_DEPTH], r2v[OUT_DEPTH], sxv[OUT_DEPTH];
#pragma unroll
      for (int q = 0; q < OUT_DEPTH; ++q) {
        const int i0 = ib + q * 256 * 4;
        const bool live = i0 < total;
        r1v[q] = (a.res1 && live) ? ld4(a.res1 + base + i0) : splat(0.f);
        r2v[q] = (a.res2 && live) ? ld4(a.res2 + base + i0) : splat(0.f);
        sxv[q] = (STAT == 2 && live) ? ld4(a.stat_x + base + i0) : splat(0.f);
      }
#pragma unroll
      for (int q = 0; q < OUT_DEPTH; ++q) {
        const int i0 = ib + q * 256 * 4;
        if (i0 >= total) break;
        const f32x4 r1 = r1v[q], r2 = r2v[q], sx = sxv[q];
        uint32_t rc = a.dL.div((uint32_t)i0);
        int l = i0 - (int)rc * a.L;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (l >= a.L) { l -= a.L; ++rc; }
          const int r = (int)rc >> 5, ci = (int)rc & 31;
          const int idx = ci * pitch + 2 + r * a.Sc + l;
          const float v = (img[idx] + r1[e]) + r2[e];
          o[e] = v;
          if (STAT == 1) img[idx] = a.stat_relu ? fmaxf(v, 0.f) : v;
          if (STAT == 2) {
            const float xr = a.stat_relu ? fmaxf(sx[e], 0.f) : sx[e];
            img[idx] = v * ((xr - aux[2 * C32 + ci]) * aux[3 * C32 + ci]);
          }
          ++l;
        }
        st4(a.y + base + i0, o);
      }
    }
    if (STAT) {
      __syncthreads();
      const int ch = tid >> 3, p8 = tid & 7;
      float s1 = 0.f, s2 = 0.f;
      for (int r = 0; r < rows; ++r) {
        const float* row = img + ch * pitch + 2 + r * a.Sc;
        for (int l = p8; l < a.L; l += 8) {
          const float t = row[l];
          s1 += t;
          if (STAT == 1) s2 += t * t;
        }
      }
      racc1 += s1;
      racc2 += s2;
    }
  }
  if (STAT) {
    double* slot = a.stat_out + (size_t)(blockIdx.x % MURAL_BN_SLOTS) * 2 * C32;
    racc1 += __shfl_xor(racc1, 1); racc1 += __shfl_xor(racc1, 2); racc1 += __shfl_xor(racc1, 4);
    racc2 += __shfl_xor(racc2, 1); racc2 += __shfl_xor(racc2, 2); racc2 += __shfl_xor(racc2, 4);
    if ((tid & 7) == 0) {
      const int ch = tid >> 3;
      if (STAT == 1) {
        atomicAdd(&slot[ch], (double)racc1);
        atomicAdd(&slot[C32 + ch], (double)racc2);
      } else {
        atomicAdd(&slot[C32 + ch], (double)racc1);      // sum(dz * xhat)
      }
    }
    if (STAT == 2) {                                     // sum(dz): lanes of one kk group hold 4 channels for 16 columns
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) sv[q] += __shfl_xor(sv[q], off);
        if (n16 == 0) atomicAdd(&slot[chv + q], (double)sv[q]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------- weight gradient
struct Wgrad32Args {
  const float* dy;       // [B][32][L]
  const float* x;        // [B][32][L] pre-activation saved by the forward
  const float* pre_s;    // input of the conv was pre_s * act(x) + pre_t
  const float* pre_t;
  int pre_relu;
  int B, L, R, Sc, NC, nb;
  FastDiv dL;
  float* part;           // [grid][32*32*3 + 32] per-workgroup partial sums
};

__global__ __launch_bounds__(256) void wgrad32_mfma_kernel(const Wgrad32Args a) {
  extern __shared__ __attribute__((aligned(16))) float wimg[];
  constexpr int pitch = C32_PITCH;
  float* gimg = wimg;                       // dy tile
  float* aimg = wimg + C32 * pitch;         // BN(act(x)) tile
  float* aff = aimg + C32 * pitch;          // pre_s | pre_t
  if (a.pre_s && threadIdx.x < 2 * C32) aff[threadIdx.x] = threadIdx.x < C32 ? a.pre_s[threadIdx.x] : a.pre_t[threadIdx.x - C32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kk = lane >> 4;
  f32x4 acc[2][3][2];                       // [M-block][tap][cin half]
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h) acc[m][t][h] = splat(0.f);
  float bacc = 0.f;
  const int nk = 4 * a.nb;                  // k-steps of 4 columns
  const int k_lo = wave * nk / 4, k_hi = (wave + 1) * nk / 4;
  const float* gp = gimg + n16 * pitch + 1 + kk;     // dy[cout = 16 m + n16][logical column 4 s + kk]
  const float* ap = aimg + n16 * pitch + kk;         // act[cin = 16 h + n16][logical column 4 s + kk + tap - 1]
  const int64_t ntiles = ((int64_t)a.B + a.R - 1) / a.R;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t b0 = tile * a.R;
    __syncthreads();
    stage_rows(a.dy, b0, a.B, a.L, a.R, a.Sc, a.dL, nullptr, 0, gimg, tid, true, 16 * a.nb + 1);
    stage_rows(a.x, b0, a.B, a.L, a.R, a.Sc, a.dL, a.pre_s ? aff : nullptr, a.pre_relu, aimg, tid, true, 16 * a.nb + 1);
    __syncthreads();
    for (int s = k_lo; s < k_hi; ++s) {
      float g[2], bv[3][2];
#pragma unroll
      for (int m = 0; m < 2; ++m) g[m] = gp[16 * m * pitch + 4 * s];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) bv[t][h] = ap[16 * h * pitch + 4 * s + t];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int h = 0; h < 2; ++h) acc[m][t][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[m], bv[t][h], acc[m][t][h], 0, 0, 0);
    }
    {   // bias gradient: thread = (cout tid/8, column residue tid%8)
      const int co = tid >> 3, p8 = tid & 7;
      const float* row = gimg + co * pitch + 1;
      float sum = 0.f;
      for (int c = p8; c < 16 * a.nb; c += 8) sum += row[c];
      bacc += sum;
    }
  }
  bacc += __shfl_xor(bacc, 1);
  bacc += __shfl_xor(bacc, 2);
  bacc += __shfl_xor(bacc, 4);
  // D[row = cout 4kk+r][col = cin n16] of tile (m, t, h) -> dW[16m + 4kk + r][16h + n16][t].  The four waves (column
  // quarters) meet in LDS in a fixed order; one partial row per workgroup goes to HBM.
  constexpr int NW = C32 * C32 * 3;
  __syncthreads();                    // the images are dead: reuse them as [4][NW + 32]
  float* mine = wimg + (size_t)wave * (NW + C32);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[((16 * m + 4 * kk + r) * C32 + 16 * h + n16) * 3 + t] = acc[m][t][h][r];
  // each wave holds the bias sums of its own 8 output channels (in lanes 0, 8, ..); the other 24 entries of its row are 0
  const float sv = __shfl(bacc, (lane & 7) * 8);
  if (lane < C32) mine[NW + lane] = ((lane >> 3) == wave) ? sv : 0.f;
  __syncthreads();
  float* dst = a.part + (size_t)blockIdx.x * (NW + C32);
  for (int i = tid; i < NW + C32; i += 256)
    dst[i] = (wimg[i] + wimg[(NW + C32) + i]) + (wimg[2 * (NW + C32) + i] + wimg[3 * (NW + C32) + i]);
}

// ------------------------------------------------------------------------------------------------- fused backward
// One pass over a layer's backward: weight / bias gradient partials (as wgrad32_mfma_kernel), the input gradient dz = dgrad(dy)
// and the BatchNorm-backward sums of dz (as conv32_mfma_kernel<2>) from ONE staging of the dy tile: dy is read once
// instead of twice and a launch disappears.  LDS: dy image | BN(act(x)) image | aux.
struct Bwd32Args {
  const float* dy;       // [B][32][L]
  const float* x;        // [B][32][L] pre-activation saved by the forward
  const float* W;        // PyTorch [32][32][3]
  const float* pre_s;    // conv input was pre_s * act(x) + pre_t
  const float* pre_t;
  const float* mean;     // batch statistics of act(x) (BatchNorm backward)
  const float* invstd;
  int pre_relu;
  int B, L, R, Sc, NC, nb;
  FastDiv dL, dSc;
  float* part;           // [grid][32*32*3 + 32]
  float* dz;             // [B][32][L]
  double* stat_out;      // accumulator block: sum(dz), sum(dz * xhat)
  int dbg;               // timing experiments only (MURAL_DEBUG_BWD32): 1 no x re-read, 2 no wgrad MFMA, 4 no dgrad MFMA, 8 no dz store
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void bwd32_mfma_kernel(const Bwd32Args a) {
  extern __shared__ __attribute__((aligned(16))) float wimg[];
  constexpr int pitch = C32_PITCH;
  float* gimg = wimg;                       // dy tile
  float* aimg = wimg + C32 * pitch;         // BN(act(x)) tile, later dz / dz * xhat
  float* aux = aimg + C32 * pitch;          // pre_s | pre_t | mean | invstd
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mb = wave & 1, cgp = wave >> 1;
  const int n16 = lane & 15, kk = lane >> 4;
  if (tid < 2 * C32) {
    aux[tid] = a.pre_s ? (tid < C32 ? a.pre_s[tid] : a.pre_t[tid - C32]) : (tid < C32 ? 1.f : 0.f);
    aux[2 * C32 + tid] = tid < C32 ? a.mean[tid] : a.invstd[tid - C32];
  }
  float af[C32_KSTEPS];                     // input-gradient filter fragments (transposed, tap-flipped)
#pragma unroll
  for (int s = 0; s < C32_KSTEPS; ++s) {
    const int t = s / 8, h = (s % 8) / 4, q = s % 4;
    const int cin = 16 * h + 4 * kk + q, cout = 16 * mb + n16;
    af[s] = a.W[(cin * C32 + cout) * 3 + (2 - t)];
  }
  const int chv = 16 * mb + 4 * kk;
  f32x4 wacc[2][3][2];                      // weight gradient [M-block][tap][cin half]
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h) wacc[m][t][h] = splat(0.f);
  float bacc = 0.f, racc = 0.f;
  float sv[4] = {0.f, 0.f, 0.f, 0.f};
  const int nk = 4 * a.nb;
  const int k_lo = wave * nk / 4, k_hi = (wave + 1) * nk / 4;
  const float* gp = gimg + n16 * pitch + 1 + kk;
  const float* ap = aimg + n16 * pitch + kk;
  const int nbw = a.nb > cgp ? (a.nb - cgp + 1) / 2 : 0;
  const float* rd = gimg + 4 * kk * pitch + n16;
  const int64_t ntiles = ((int64_t)a.B + a.R - 1) / a.R;
  // both tensors of a tile are requested as one batch of loads, one tile ahead: they fly under the MFMA and stream-out phases of
  // the previous tile (four dependent load round trips per tile cost more than everything else in this kernel)
  f32x4 pdy[C32_NV], px[C32_NV];
  load_rows(a.dy, (int64_t)blockIdx.x * a.R, a.B, a.L, a.R, tid, pdy);
  load_rows(a.x, (int64_t)blockIdx.x * a.R, a.B, a.L, a.R, tid, px);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t b0 = tile * a.R;
    const int rows = (int)((a.B - b0) < a.R ? (a.B - b0) : a.R);
    __syncthreads();
    // (the LDS addresses of a thread's elements do not depend on the tile: without this opaque zero the compiler hoists all 72 of
    // them out of the tile loop and spills the prefetched tile instead)
    int tidv = tid;
    asm volatile("" : "+v"(tidv));
    if (!(a.dbg & 16)) {
      scatter_rows(pdy, b0, a.B, a.L, a.R, a.Sc, a.dL, nullptr, 0, gimg, tidv, 16 * a.nb + 1);
      scatter_rows(px, b0, a.B, a.L, a.R, a.Sc, a.dL, a.pre_s ? aux : nullptr, a.pre_relu, aimg, tidv, 16 * a.nb + 1);
    } else if (pdy[0].x == 12345.f && px[1].y == 54321.f) gimg[tid] = pdy[3].z + px[5].w;
    __syncthreads();
    // ---- weight gradient: this wave's quarter of the columns
    for (int s = k_lo; s < ((a.dbg & 2) ? k_lo : k_hi); ++s) {
      float g[2], bv[3][2];
#pragma unroll
      for (int m = 0; m < 2; ++m) g[m] = gp[16 * m * pitch + 4 * s];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) bv[t][h] = ap[16 * h * pitch + 4 * s + t];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int h = 0; h < 2; ++h) wacc[m][t][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[m], bv[t][h], wacc[m][t][h], 0, 0, 0);
    }
    if (!(a.dbg & 64)) {   // bias gradient: thread = (cout tid/8, column residue tid%8)
      const int co = tid >> 3, p8 = tid & 7;
      const float* row = gimg + co * pitch + 1;
      float sum = 0.f;
      for (int c = p8; c < 16 * a.nb; c += 8) sum += row[c];
      bacc += sum;
    }
    __syncthreads();                                     // the BN(act(x)) image is dead: the input gradient is written over it
    // ---- input gradient: this wave's M-block against every second column block of the dy image, straight into the dead image
#pragma unroll
    for (int ip = 0; ip < (C32_NB2MAX + 1) / 2; ++ip) {
      const int i0 = 2 * ip, i1 = 2 * ip + 1;
      if (i0 < nbw && !(a.dbg & 4)) {
        const bool dual = i1 < nbw;
        const float* p0 = rd + 16 * (cgp + 2 * i0);
        const float* p1 = rd + 16 * (cgp + 2 * (dual ? i1 : i0));
        f32x4 a0 = splat(0.f), a1 = splat(0.f);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          float b0v[8], b1v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            b0v[j] = p0[(16 * (j >> 2) + (j & 3)) * pitch + t];
            b1v[j] = p1[(16 * (j >> 2) + (j & 3)) * pitch + t];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[8 * t + j], b0v[j], a0, 0, 0, 0);
            if (dual) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[8 * t + j], b1v[j], a1, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (u == 1 && !dual) break;
          const f32x4 v = u == 0 ? a0 : a1;
          const int c = 16 * (cgp + 2 * (i0 + u)) + n16;
#pragma unroll
          for (int q = 0; q < 4; ++q) aimg[(chv + q) * pitch + 1 + c] = v[q];
          if (c >= 1) {
            const uint32_t uu = (uint32_t)(c - 1);
            const uint32_t r = a.dSc.div(uu);
            if ((int)r < rows && (int)(uu - r * (uint32_t)a.Sc) < a.L) {
#pragma unroll
              for (int q = 0; q < 4; ++q) sv[q] += v[q];
            }
          }
        }
      }
    }
    __syncthreads();
    // the MFMA phases are over and their registers free: request the next tile now, it lands while this one streams out
    if (tile + gridDim.x < ntiles) {
      load_rows(a.dy, (tile + gridDim.x) * a.R, a.B, a.L, a.R, tid, pdy);
      load_rows(a.x, (tile + gridDim.x) * a.R, a.B, a.L, a.R, tid, px);
    }
    const int total = rows * C32 * a.L;
    const size_t base = (size_t)b0 * C32 * a.L;
    constexpr int OUT_DEPTH = 2;
#pragma unroll 1
    for (int ib = tid * 4; ib < ((a.dbg & 32) ? 0 : total); ib += 256 * 4 * OUT_DEPTH) {
      f32x4 sxv[OUT_DEPTH];
#pragma unroll
      for (int q = 0; q < OUT_DEPTH; ++q) {
        const int i0 = ib + q * 256 * 4;
        sxv[q] = (i0 < total && !(a.dbg & 1)) ? ld4(a.x + base + i0) : splat(0.f);
      }
#pragma unroll
      for (int q = 0; q < OUT_DEPTH; ++q) {
        const int i0 = ib + q * 256 * 4;
        if (i0 >= total) break;
        const f32x4 sx = sxv[q];
        uint32_t rc = a.dL.div((uint32_t)i0);
        int l = i0 - (int)rc * a.L;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (l >= a.L) { l -= a.L; ++rc; }
          const int r = (int)rc >> 5, ci = (int)rc & 31;
          const int idx = ci * pitch + 2 + r * a.Sc + l;
          const float v = aimg[idx];
          o[e] = v;
          const float xr = a.pre_relu ? fmaxf(sx[e], 0.f) : sx[e];
          aimg[idx] = v * ((xr - aux[2 * C32 + ci]) * aux[3 * C32 + ci]);
          ++l;
        }
        if (!(a.dbg & 8)) st4(a.dz + base + i0, o);
      }
    }
    __syncthreads();
    {
      const int ch = tid >> 3, p8 = tid & 7;
      float s1 = 0.f;
      for (int r = 0; r < ((a.dbg & 64) ? 0 : rows); ++r) {
        const float* row = aimg + ch * pitch + 2 + r * a.Sc;
        for (int l = p8; l < a.L; l += 8) s1 += row[l];
      }
      racc += s1;
    }
  }
  // BatchNorm-backward sums
  {
    double* slot = a.stat_out + (size_t)(blockIdx.x % MURAL_BN_SLOTS) * 2 * C32;
    racc += __shfl_xor(racc, 1); racc += __shfl_xor(racc, 2); racc += __shfl_xor(racc, 4);
    if ((tid & 7) == 0) atomicAdd(&slot[C32 + (tid >> 3)], (double)racc);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) sv[q] += __shfl_xor(sv[q], off);
      if (n16 == 0) atomicAdd(&slot[chv + q], (double)sv[q]);
    }
  }
  // weight / bias gradient partial row of this workgroup
  bacc += __shfl_xor(bacc, 1);
  bacc += __shfl_xor(bacc, 2);
  bacc += __shfl_xor(bacc, 4);
  constexpr int NW = C32 * C32 * 3;
  __syncthreads();
  float* mine = wimg + (size_t)wave * (NW + C32);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[((16 * m + 4 * kk + r) * C32 + 16 * h + n16) * 3 + t] = wacc[m][t][h][r];
  const float svb = __shfl(bacc, (lane & 7) * 8);
  if (lane < C32) mine[NW + lane] = ((lane >> 3) == wave) ? svb : 0.f;
  __syncthreads();
  float* dst = a.part + (size_t)blockIdx.x * (NW + C32);
  for (int i = tid; i < NW + C32; i += 256)
    dst[i] = (wimg[i] + wimg[(NW + C32) + i]) + (wimg[2 * (NW + C32) + i] + wimg[3 * (NW + C32) + i]);
}

// 64 outputs x 16 slices of the partial rows per workgroup; fixed summation order -> reproducible gradients
__global__ __launch_bounds__(1024) void part_reduce_kernel(const float* __restrict__ part, int nrow, int nW, int nB,
                                                           float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float sh[16][64];
  const int o = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  float s = 0.f;
  if (i < nW + nB) {
    // eight rows in flight per thread (the loop is latency-bound: 49 workgroups read 6 MB); the order of the adds is fixed
    const size_t rs = (size_t)(nW + nB);
    const float* p = part + i;
    int b = slice;
    float s0 = 0.f, s1 = 0.f;
    for (; b + 7 * 16 < nrow; b += 8 * 16) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = p[(size_t)(b + 16 * q) * rs];
      s0 += (v[0] + v[1]) + (v[2] + v[3]);
      s1 += (v[4] + v[5]) + (v[6] + v[7]);
    }
    for (; b < nrow; b += 16) s0 += p[(size_t)b * rs];
    s = s0 + s1;
  }
  sh[slice][o] = s;
  __syncthreads();
  if (slice == 0 && i < nW + nB) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sh[q][o];
    if (i < nW) dW[i] = t;
    else if (db) db[i - nW] = t;
  }
}

// the same reduction for up to PR_MAXJOBS layers in one launch (blockIdx.y = layer): the backward kernels of a step leave their
// partial rows in separate regions and one launch at the end of the backward turns them into dW / db
constexpr int PR_MAXJOBS = 24;
struct PartJobs {
  const float* part[PR_MAXJOBS];
  float* dW[PR_MAXJOBS];
  float* db[PR_MAXJOBS];
  int nrow[PR_MAXJOBS];
};

__global__ __launch_bounds__(1024) void part_reduce_multi_kernel(const PartJobs jobs, int nW, int nB) {
  __shared__ float sh[16][64];
  const int job = blockIdx.y;
  const float* __restrict__ part = jobs.part[job];
  const int nrow = jobs.nrow[job];
  const int o = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  float s = 0.f;
  if (i < nW + nB) {
    const size_t rs = (size_t)(nW + nB);
    const float* p = part + i;
    int b = slice;
    float s0 = 0.f, s1 = 0.f;
    for (; b + 7 * 16 < nrow; b += 8 * 16) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = p[(size_t)(b + 16 * q) * rs];
      s0 += (v[0] + v[1]) + (v[2] + v[3]);
      s1 += (v[4] + v[5]) + (v[6] + v[7]);
    }
    for (; b < nrow; b += 16) s0 += p[(size_t)b * rs];
    s = s0 + s1;
  }
  sh[slice][o] = s;
  __syncthreads();
  if (slice == 0 && i < nW + nB) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sh[q][o];
    if (i < nW) jobs.dW[job][i] = t;
    else if (jobs.db[job]) jobs.db[job][i - nW] = t;
  }
}

static bool tile_geometry(int B, int L, int* R, int* Sc, int* NC, int* nb) {
  *Sc = L + 1;
  int r = (C32_MAXCOLS - 1) / *Sc;
  if (r < 1) return false;
  if (r > B) r = B;
  // large batches of short rows: prefer >= 1024 tiles (4 per CU) over the tallest tile.  (The tile height only regroups
  // the fp32 partial sums of the fused BatchNorm statistics: results differ in the last bits, nothing else.)
  const int balanced = (B + 1023) / 1024;
  if (B >= 1024 && r > balanced) r = balanced;
  if (const char* e = dev_env("MURAL_DEBUG_CONV32_R")) {   // debugging aid (tools/gpu_debug_train_diff.py)
    const int v = atoi(e);
    if (v >= 1 && v <= r) r = v;
  }
  *R = r;
  *NC = 1 + r * *Sc;
  *nb = (*NC + 15) / 16;
  return *nb <= 2 * C32_NB2MAX;
}

// ---- entry points of the composed training step (snv_train.hip) ----------------------------------------------------------
// forward of BN(batch statistics) -> conv32 with the BatchNorm finalisation folded into the conv launch
int train_conv32_fwd(const float* x, int64_t B, int L, int pre_relu, const double* acc, const float* gamma, const float* beta, float eps,
                     float momentum, float* running_mean, float* running_var, float* state, const float* W, const float* bias,
                     int post_relu, const float* res1, const float* res2, double* acc_out, int out_relu, float* y, hipStream_t stream) {
  if (B == 0 || L == 0) return MURAL_OK;
  Conv32Args a;
  std::memset(&a, 0, sizeof(a));
  MURAL_REQUIRE(tile_geometry((int)B, L, &a.R, &a.Sc, &a.NC, &a.nb), "conv32: L = %d does not fit the LDS tile", L);
  a.x = x; a.y = y; a.W = W; a.bias = bias; a.res1 = res1; a.res2 = res2;
  a.pre_relu = pre_relu; a.post_relu = post_relu; a.B = (int)B; a.L = L;
  a.stat_mode = acc_out ? 1 : 0; a.stat_relu = out_relu; a.stat_out = acc_out;
  a.fin_acc = acc; a.fin_n = (double)B * L; a.fin_gamma = gamma; a.fin_beta = beta; a.fin_eps = eps; a.fin_momentum = momentum;
  a.fin_running_mean = running_mean; a.fin_running_var = running_var; a.fin_state = state;
  a.dSc = FastDiv::make((uint32_t)a.Sc);
  a.dL = FastDiv::make((uint32_t)L);
  const size_t lds = (size_t)(C32 * C32_PITCH + C32_AUX) * 4;
  const int64_t ntiles = (B + a.R - 1) / a.R;
  const int grid = (int)(ntiles < 1024 ? ntiles : 1024);
  if (a.stat_mode) hipLaunchKernelGGL(conv32_mfma_kernel<1>, dim3(grid), dim3(256), lds, stream, a);
  else hipLaunchKernelGGL(conv32_mfma_kernel<0>, dim3(grid), dim3(256), lds, stream, a);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

size_t train_conv32_part_floats() { return (size_t)512 * (C32 * C32 * 3 + C32); }

// backward kernel only: the partial rows stay in `part` (rows = workgroups launched, returned in *nrow) for reduce_parts()
int train_conv32_bwd(const float* dy, const float* x, const float* W, int64_t B, int L, const float* state, int pre_relu, float* dz,
                     double* stat_out, float* part, int* nrow, hipStream_t stream) {
  Bwd32Args a;
  std::memset(&a, 0, sizeof(a));
  MURAL_REQUIRE(tile_geometry((int)B, L, &a.R, &a.Sc, &a.NC, &a.nb), "conv32_bwd: L = %d does not fit the LDS tile", L);
  a.dy = dy; a.x = x; a.W = W; a.pre_s = state; a.pre_t = state + C32; a.pre_relu = pre_relu; a.mean = state + 2 * C32;
  a.invstd = state + 3 * C32;
  a.B = (int)B; a.L = L; a.part = part; a.dz = dz; a.stat_out = stat_out;
  if (const char* e = dev_env("MURAL_DEBUG_BWD32")) a.dbg = atoi(e);
  a.dL = FastDiv::make((uint32_t)L);
  a.dSc = FastDiv::make((uint32_t)a.Sc);
  const int64_t ntiles = (B + a.R - 1) / a.R;
  const int grid = (int)(ntiles < 512 ? ntiles : 512);
  size_t lds = (size_t)(2 * C32 * C32_PITCH + C32_AUX) * 4;
  const size_t lds_red = (size_t)4 * (C32 * C32 * 3 + C32) * 4;
  lds = lds > lds_red ? lds : lds_red;
  static DynLdsOnce big_lds;
  if (int rc = big_lds.ensure(&bwd32_mfma_kernel)) return rc;
  hipLaunchKernelGGL(bwd32_mfma_kernel, dim3(grid), dim3(256), lds, stream, a);
  MURAL_HIP_CHECK(hipGetLastError());
  *nrow = grid;
  return MURAL_OK;
}

int train_reduce_parts(const float* const* part, const int* nrow, float* const* dW, float* const* db, int njobs, hipStream_t stream) {
  for (int j0 = 0; j0 < njobs; j0 += PR_MAXJOBS) {
    PartJobs jobs;
    std::memset(&jobs, 0, sizeof(jobs));
    const int n = njobs - j0 < PR_MAXJOBS ? njobs - j0 : PR_MAXJOBS;
    // validation only (tests/test_gpu_train.py): MURAL_DEBUG_DROP_PART_ROW=<job> leaves the last partial row of that job out of
    // its sum -- the fault the parity tests of the training step must be able to see
    int drop_job = -1;
    if (const char* e = dev_env("MURAL_DEBUG_DROP_PART_ROW")) drop_job = atoi(e);
    for (int j = 0; j < n; ++j) {
      jobs.part[j] = part[j0 + j];
      jobs.nrow[j] = nrow[j0 + j] - ((j0 + j == drop_job && nrow[j0 + j] > 1) ? 1 : 0);
      jobs.dW[j] = dW[j0 + j];
      jobs.db[j] = db[j0 + j];
    }
    hipLaunchKernelGGL(part_reduce_multi_kernel, dim3((C32 * C32 * 3 + C32 + 63) / 64, n), dim3(1024), 0, stream, jobs, C32 * C32 * 3, C32);
    MURAL_HIP_CHECK(hipGetLastError());
  }
  return MURAL_OK;
}

}  // namespace mural

using namespace mural;

extern "C" int mural_op_conv32_supported(int32_t L) { return (L + 2) <= C32_MAXCOLS ? 1 : 0; }

// y = conv32(pre(x)) [+bias] [relu] [+res1 +res2]; W is the PyTorch [32][32][3] weight (dgrad != 0: input-gradient
// filter, i.e. y = dL/dx for x = dL/dy).  stat_mode / stat_*: per-channel sums of the output taken while it streams out
// (see Conv32Args); stat_out: accumulator block double[MURAL_BN_SLOTS][2][32] zeroed by the caller.
extern "C" int mural_op_conv32(const float* x, const float* W, const float* bias, float* y, int64_t B, int32_t L, int32_t dgrad,
                               const float* pre_s, const float* pre_t, int32_t pre_relu, int32_t post_relu, const float* res1,
                               const float* res2, int32_t stat_mode, int32_t stat_relu, const float* stat_x,
                               const float* stat_mean, const float* stat_invstd, double* stat_out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B == 0 || L == 0) return MURAL_OK;
  Conv32Args a;
  std::memset(&a, 0, sizeof(a));
  MURAL_REQUIRE(tile_geometry((int)B, L, &a.R, &a.Sc, &a.NC, &a.nb), "conv32: L = %d does not fit the LDS tile", L);
  MURAL_REQUIRE(stat_mode >= 0 && stat_mode <= 2 && (stat_mode == 0 || stat_out), "conv32: stat_mode %d needs stat_out", stat_mode);
  MURAL_REQUIRE(stat_mode != 2 || (stat_x && stat_mean && stat_invstd && !res1 && !res2 && !bias && !post_relu),
                "conv32: stat_mode 2 needs stat_x / mean / invstd and a plain convolution");
  a.x = x; a.y = y; a.W = W; a.dgrad = dgrad; a.bias = bias; a.pre_s = pre_s; a.pre_t = pre_t; a.res1 = res1; a.res2 = res2;
  a.pre_relu = pre_relu; a.post_relu = post_relu; a.B = (int)B; a.L = L;
  a.stat_mode = stat_mode; a.stat_relu = stat_relu; a.stat_x = stat_x; a.stat_mean = stat_mean; a.stat_invstd = stat_invstd;
  a.stat_out = stat_out;
  a.dSc = FastDiv::make((uint32_t)a.Sc);
  a.dL = FastDiv::make((uint32_t)L);
  const size_t lds = (size_t)(C32 * C32_PITCH + C32_AUX) * 4;
  const int64_t ntiles = (B + a.R - 1) / a.R;
  const int grid = (int)(ntiles < 1024 ? ntiles : 1024);
  if (stat_mode == 0) hipLaunchKernelGGL(conv32_mfma_kernel<0>, dim3(grid), dim3(256), lds, stream, a);
  else if (stat_mode == 1) hipLaunchKernelGGL(conv32_mfma_kernel<1>, dim3(grid), dim3(256), lds, stream, a);
  else hipLaunchKernelGGL(conv32_mfma_kernel<2>, dim3(grid), dim3(256), lds, stream, a);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

// dW [32][32][3], db [32] of y = conv32(pre(x)); part: float scratch of at least mural_op_conv32_wgrad_scratch() floats
extern "C" size_t mural_op_conv32_wgrad_scratch(void) { return (size_t)512 * (C32 * C32 * 3 + C32); }

extern "C" int mural_op_conv32_wgrad(const float* dy, const float* x, int64_t B, int32_t L, const float* pre_s, const float* pre_t,
                                     int32_t pre_relu, float* dW, float* db, float* part, size_t part_floats, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B == 0 || L == 0) return MURAL_OK;
  Wgrad32Args a;
  std::memset(&a, 0, sizeof(a));
  MURAL_REQUIRE(tile_geometry((int)B, L, &a.R, &a.Sc, &a.NC, &a.nb), "conv32_wgrad: L = %d does not fit the LDS tile", L);
  a.dy = dy; a.x = x; a.pre_s = pre_s; a.pre_t = pre_t; a.pre_relu = pre_relu; a.B = (int)B; a.L = L;
  a.dL = FastDiv::make((uint32_t)L);
  a.part = part;
  const int64_t ntiles = (B + a.R - 1) / a.R;
  const int grid = (int)(ntiles < 512 ? ntiles : 512);
  MURAL_REQUIRE(part && part_floats >= (size_t)grid * (C32 * C32 * 3 + C32), "conv32_wgrad: partial-sum scratch too small");
  size_t lds = (size_t)(2 * C32 * C32_PITCH + 2 * C32) * 4;
  const size_t lds_red = (size_t)4 * (C32 * C32 * 3 + C32) * 4;     // the in-workgroup reduction reuses the image space
  lds = lds > lds_red ? lds : lds_red;
  static DynLdsOnce big_lds;
  if (int rc = big_lds.ensure(&wgrad32_mfma_kernel)) return rc;
  hipLaunchKernelGGL(wgrad32_mfma_kernel, dim3(grid), dim3(256), lds, stream, a);
  hipLaunchKernelGGL(part_reduce_kernel, dim3((C32 * C32 * 3 + C32 + 63) / 64), dim3(1024), 0, stream, part, grid,
                     C32 * C32 * 3, C32, dW, db);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

// Whole backward of y = conv32(pre_s * act(x) + pre_t) through the conv and its BatchNorm statistics in one pass over dy:
// dW [32][32][3], db [32], dz = dL/d(conv input) [B][32][L] and the BatchNorm-backward sums of dz (accumulator block
// stat_out, zeroed by the caller).  part: mural_op_conv32_wgrad_scratch() floats.
extern "C" int mural_op_conv32_bwd(const float* dy, const float* x, const float* W, int64_t B, int32_t L, const float* pre_s,
                                   const float* pre_t, int32_t pre_relu, const float* mean, const float* invstd, float* dW,
                                   float* db, float* dz, double* stat_out, float* part, size_t part_floats, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B == 0 || L == 0) return MURAL_OK;
  Bwd32Args a;
  std::memset(&a, 0, sizeof(a));
  MURAL_REQUIRE(tile_geometry((int)B, L, &a.R, &a.Sc, &a.NC, &a.nb), "conv32_bwd: L = %d does not fit the LDS tile", L);
  MURAL_REQUIRE(dy && x && W && mean && invstd && dW && db && dz && stat_out, "conv32_bwd: NULL argument");
  a.dy = dy; a.x = x; a.W = W; a.pre_s = pre_s; a.pre_t = pre_t; a.pre_relu = pre_relu; a.mean = mean; a.invstd = invstd;
  a.B = (int)B; a.L = L; a.part = part; a.dz = dz; a.stat_out = stat_out;
  if (const char* e = dev_env("MURAL_DEBUG_BWD32")) a.dbg = atoi(e);
  a.dL = FastDiv::make((uint32_t)L);
  a.dSc = FastDiv::make((uint32_t)a.Sc);
  const int64_t ntiles = (B + a.R - 1) / a.R;
  const int grid = (int)(ntiles < 512 ? ntiles : 512);
  MURAL_REQUIRE(part && part_floats >= (size_t)grid * (C32 * C32 * 3 + C32), "conv32_bwd: partial-sum scratch too small");
  size_t lds = (size_t)(2 * C32 * C32_PITCH + C32_AUX) * 4;
  const size_t lds_red = (size_t)4 * (C32 * C32 * 3 + C32) * 4;
  lds = lds > lds_red ? lds : lds_red;
  static DynLdsOnce big_lds;
  if (int rc = big_lds.ensure(&bwd32_mfma_kernel)) return rc;
  hipLaunchKernelGGL(bwd32_mfma_kernel, dim3(grid), dim3(256), lds, stream, a);
  hipLaunchKernelGGL(part_reduce_kernel, dim3((C32 * C32 * 3 + C32 + 63) / 64), dim3(1024), 0, stream, part, grid,
                     C32 * C32 * 3, C32, dW, db);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

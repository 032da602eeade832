// Training-mode building blocks of the SNV models (forward with batch-statistics BatchNorm + every backward).
// Reference semantics: MuRaL/training.py:404-450 (one step = forward, CE-sum loss, backward through every op) over
// MuRaL/model/model_snv.py:439-525; torch defaults for BatchNorm1d (momentum 0.1, eps 1e-5, biased variance for the
// normalisation, unbiased for running_var), MaxPool1d (-inf padding), Dropout (scale 1/(1-p)).
// Layout: activations [B][C][L] fp32 (the reference's NCL).  Every entry point enqueues on the caller's stream.
// Correctness-first kernels (vector ALU, atomics for cross-workgroup sums); the conv forward / input-gradient reuse the
// generic conv1d kernel.  The Python autograd glue lives in mural_amd/model/train_ops.py.
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "conv1d.h"
#include "snv.h"

namespace mural {
namespace {

using f32x4_t = __attribute__((ext_vector_type(4))) float;

constexpr int BN_SLOTS = MURAL_BN_SLOTS;

__device__ __forceinline__ uint64_t mix64(uint64_t z);

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ------------------------------------------------------------------------------------------- weight re-layouts
// W[Cout][Cin][K] -> wt[Cin][K][Cout]  (forward)         or -> wt[Cout][K][Cin] with taps flipped (input gradient)
__global__ void relayout_kernel(const float* __restrict__ W, float* __restrict__ wt, int Cout, int Cin, int K, int dgrad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cout * Cin * K) return;
  const int co = i / (Cin * K), r = i - co * Cin * K, ci = r / K, k = r - ci * K;
  if (dgrad) wt[((size_t)co * K + (K - 1 - k)) * Cin + ci] = W[i];
  else wt[((size_t)ci * K + k) * Cout + co] = W[i];
}

// ------------------------------------------------------------------------------------------- BatchNorm (train)
// sums over (B, L) of a(x) and a(x)^2 per channel, a = relu or identity
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int B, int C, int L, int relu,
                                                       double* __restrict__ acc) {
  const int c = blockIdx.x;
  double s = 0.0, q = 0.0;
  const int64_t per = (int64_t)B * L;
  if ((L & 3) == 0 && per * C < (int64_t(1) << 31)) {
    // four consecutive positions per thread: 16-byte loads and 32-bit index math (a 64-bit division per element made the pass
    // ALU-bound on the long rows of the U-Net's first levels)
    const uint32_t nq = (uint32_t)(per >> 2), qstep = gridDim.y * blockDim.x;
    for (uint32_t qi = blockIdx.y * blockDim.x + threadIdx.x; qi < nq; qi += qstep) {
      const uint32_t e = qi << 2, b = e / (uint32_t)L, l = e - b * (uint32_t)L;
      const float4 v4 = *reinterpret_cast<const float4*>(x + (size_t)(b * (uint32_t)C + (uint32_t)c) * L + l);
      const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float v = relu ? fmaxf(vv[t], 0.f) : vv[t];
        s += v;
        q += (double)v * v;
      }
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.y * blockDim.x) {
      const int64_t b = i / L;
      const int l = (int)(i - b * L);
      float v = x[(b * C + c) * L + l];
      if (relu) v = fmaxf(v, 0.f);
      s += v;
      q += (double)v * v;
    }
  }
  __shared__ double sh[2][256];
  sh[0][threadIdx.x] = s;
  sh[1][threadIdx.x] = q;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + off];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {   // accumulator copy by workgroup: same-address atomics serialise in L2
    double* slot = acc + (size_t)(blockIdx.y % BN_SLOTS) * 2 * C;
    atomicAdd(&slot[c], sh[0][0]);
    atomicAdd(&slot[C + c], sh[1][0]);
  }
}

// scale / shift of y = gamma * (a(x) - mean) * invstd + beta, running statistics update
__device__ __forceinline__ double slot_sum(const double* __restrict__ acc, int C, int which, int c) {
  double t = 0.0;
  for (int k = 0; k < BN_SLOTS; ++k) t += acc[((size_t)k * 2 + which) * C + c];
  return t;
}

__global__ void bn_finalize_kernel(const double* __restrict__ acc, double n, int C,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
                                   float* __restrict__ invstd_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mean = slot_sum(acc, C, 0, c) / n;
  double var = slot_sum(acc, C, 1, c) / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  scale[c] = (float)(gamma[c] * invstd);
  shift[c] = (float)(beta[c] - mean * gamma[c] * invstd);
  mean_out[c] = (float)mean;
  invstd_out[c] = (float)invstd;
  if (running_mean) {
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}

// y = scale * a(x) + shift (materialised form, used on 2-D tensors)
__global__ void bn_apply_kernel(const float* __restrict__ x, int64_t total, int C, int L, int relu,
                                const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ y) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / L) % C);
    float v = x[i];
    if (relu) v = fmaxf(v, 0.f);
    y[i] = fmaf(scale[c], v, shift[c]);
  }
}

// ---- small fused forms used by the composed training step (snv_train.hip) on (B, C) feature tensors ----------------------
// y = dropout(scale * a(x) + shift) with scale / shift derived from the batch sums in `acc` by every workgroup; workgroup 0 also
// writes the state (scale | shift | mean | invstd, [4][C]) and the running statistics: BatchNorm finalisation, affine map and
// dropout of `Linear -> ReLU -> BatchNorm1d -> Dropout` (model_snv.py:466-467) and of distal_fc (:383-385) in one launch
__global__ __launch_bounds__(256) void bn2d_apply_dropout_kernel(const float* __restrict__ x, int64_t B, int C, int relu,
                                                                 const double* __restrict__ acc, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps, float momentum,
                                                                 float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                 float* __restrict__ state, float p, uint64_t seed,
                                                                 const uint64_t* __restrict__ seed_dev, float* __restrict__ y_bn,
                                                                 float* __restrict__ y) {
  extern __shared__ float cst[];      // [C][2] scale, shift
  const double n = (double)B;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double s1 = 0.0, s2 = 0.0;
    double v1[BN_SLOTS], v2[BN_SLOTS];
#pragma unroll
    for (int k = 0; k < BN_SLOTS; ++k) {
      v1[k] = acc[((size_t)k * 2 + 0) * C + c];
      v2[k] = acc[((size_t)k * 2 + 1) * C + c];
    }
#pragma unroll
    for (int k = 0; k < BN_SLOTS; ++k) {
      s1 += v1[k];
      s2 += v2[k];
    }
    const double mean = s1 / n;
    double var = s2 / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float sc = (float)(gamma[c] * invstd), sh = (float)(beta[c] - mean * gamma[c] * invstd);
    cst[2 * c] = sc;
    cst[2 * c + 1] = sh;
    if (blockIdx.x == 0) {
      state[c] = sc;
      state[C + c] = sh;
      state[2 * C + c] = (float)mean;
      state[3 * C + c] = (float)invstd;
      if (running_mean) {
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
      }
    }
  }
  __syncthreads();
  if (seed_dev) seed += *seed_dev;
  const float keep_scale = 1.f / (1.f - p);
  const int64_t total = B * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    float v = x[i];
    if (relu) v = fmaxf(v, 0.f);
    v = fmaf(cst[2 * c], v, cst[2 * c + 1]);
    if (y_bn) y_bn[i] = v;
    if (p > 0.f) {
      const uint64_t r = mix64(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1));
      const float u = (float)(r >> 40) * (1.f / 16777216.f);
      v = u >= p ? v * keep_scale : 0.f;
    }
    y[i] = v;
  }
}

// dc3[b][c][l] = (l == arg[b][c] && c3[b][c][l] > 0) ? dfeat[b][c] : 0: backward of the global max over columns and of the ReLU
// in front of it (model_snv.py:487-489) in one pass
__global__ void gmax_relu_bwd_kernel(const float* __restrict__ dfeat, const int32_t* __restrict__ arg, const float* __restrict__ c3,
                                     int64_t rows, int L, float* __restrict__ dx) {
  const int64_t total = rows * L;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / L;
    const int l = (int)(i - r * L);
    dx[i] = (arg[r] == l && c3[i] > 0.f) ? dfeat[r] : 0.f;
  }
}

// sums over (B, L) of dz and dz * xhat per channel
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dz, const float* __restrict__ x, int B,
                                                            int C, int L, int relu, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, double* __restrict__ acc) {
  const int c = blockIdx.x;
  const float mu = mean[c], is = invstd[c];
  double a = 0.0, bq = 0.0;
  const int64_t per = (int64_t)B * L;
  for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.y * blockDim.x) {
    const int64_t b = i / L;
    const int l = (int)(i - b * L);
    const size_t o = (size_t)(b * C + c) * L + l;
    float v = x[o];
    if (relu) v = fmaxf(v, 0.f);
    const float g = dz[o];
    a += g;
    bq += (double)g * ((v - mu) * is);
  }
  __shared__ double sh[2][256];
  sh[0][threadIdx.x] = a;
  sh[1][threadIdx.x] = bq;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + off];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double* slot = acc + (size_t)(blockIdx.y % BN_SLOTS) * 2 * C;
    atomicAdd(&slot[c], sh[0][0]);
    atomicAdd(&slot[C + c], sh[1][0]);
  }
}

// dx = a'(x) * gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)) [+ add1 + add2];  workgroup 0 also writes
// dgamma = sum(dz * xhat), dbeta = sum(dz).  Per-channel constants are prepared once per workgroup in LDS.
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ x, int64_t total,
                                                           int C, int L, int relu, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const double* __restrict__ acc, double n,
                                                           const float* __restrict__ add1, const float* __restrict__ add2,
                                                           float* __restrict__ dx, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta) {
  extern __shared__ float cst[];      // [C][4]: gamma * invstd, mean(dz), mean(dz * xhat), mean
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const double s1 = slot_sum(acc, C, 0, c), s2 = slot_sum(acc, C, 1, c);
    cst[4 * c + 0] = gamma[c] * invstd[c];
    cst[4 * c + 1] = (float)(s1 / n);
    cst[4 * c + 2] = (float)(s2 / n);
    cst[4 * c + 3] = mean[c];
    if (blockIdx.x == 0) {
      dgamma[c] = (float)s2;
      dbeta[c] = (float)s1;
    }
  }
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / L) % C);
    const float raw = x[i];
    const float v = relu ? fmaxf(raw, 0.f) : raw;
    const float xh = (v - cst[4 * c + 3]) * invstd[c];
    float g = cst[4 * c + 0] * (dz[i] - cst[4 * c + 1] - xh * cst[4 * c + 2]);
    if (relu && raw <= 0.f) g = 0.f;
    if (add1) g += add1[i];
    if (add2) g += add2[i];
    dx[i] = g;
  }
}

// ------------------------------------------------------------------------------------------- conv weight / bias gradients
// dW[co][ci][k] = sum_{b,l} dy[b][co][l] * a[b][ci][l + k - pad], a = scale[ci] * act(x) + shift[ci] (zero outside [0, L))
// db[co] = sum dy.  One workgroup walks (b, 64-column tile) items; thread t owns 4 (co, ci) pairs x K taps.
template <int C, int K>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, int B, int L,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         int pre_relu, float* __restrict__ part /*[grid][C*C*K + C]*/) {
  constexpr int TL = 64, PAD = (K - 1) / 2, TW = TL + K - 1;
  __shared__ float sdy[C][TL + 1];
  __shared__ float sa[C][TW + 3];   // +1 column read by the two-column inner loop; odd stride: conflict-free rows
  const int tid = threadIdx.x;
  float acc[(C * C) / 256][K];
#pragma unroll
  for (int p = 0; p < (C * C) / 256; ++p)
#pragma unroll
    for (int k = 0; k < K; ++k) acc[p][k] = 0.f;
  float bacc = 0.f;
  const int ntile = (L + TL - 1) / TL;
  const int64_t items = (int64_t)B * ntile;
  for (int64_t it = blockIdx.x; it < items; it += gridDim.x) {
    const int b = (int)(it / ntile), l0 = (int)(it - (int64_t)b * ntile) * TL;
    __syncthreads();
    for (int i = tid; i < C * TL; i += 256) {
      const int c = i / TL, j = i - c * TL;
      sdy[c][j] = (l0 + j < L) ? dy[((size_t)b * C + c) * L + l0 + j] : 0.f;
    }
    for (int i = tid; i < C * (TW + 1); i += 256) {
      const int c = i / (TW + 1), j = i - c * (TW + 1);
      const int l = l0 + j - PAD;
      float v = 0.f;
      if (l >= 0 && l < L) {
        v = x[((size_t)b * C + c) * L + l];
        if (pre_relu) v = fmaxf(v, 0.f);
        v = fmaf(scale[c], v, shift[c]);
      }
      sa[c][j] = v;
    }
    __syncthreads();
    {
      // thread = (input channel ci, group of (C*C)/256 consecutive output channels): the K+1 input values of two
      // neighbouring columns are reused across the group's output channels
      constexpr int G = (C * C) / 256;
      const int ci = tid % C, co0 = (tid / C) * G;
      for (int j = 0; j < TL; j += 2) {
        float av[K + 1];
#pragma unroll
        for (int k = 0; k < K + 1; ++k) av[k] = sa[ci][j + k];
#pragma unroll
        for (int p = 0; p < G; ++p) {
          const float g0 = sdy[co0 + p][j], g1 = sdy[co0 + p][j + 1];
#pragma unroll
          for (int k = 0; k < K; ++k) acc[p][k] = fmaf(g1, av[k + 1], fmaf(g0, av[k], acc[p][k]));
        }
      }
    }
    {   // bias gradient: thread = (output channel tid/8, 8-column slice tid%8)
      const int co = tid >> 3, part8 = tid & 7;
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < TL / 8; ++q) s += sdy[co][part8 * (TL / 8) + q];
      bacc += s;
    }
  }
  bacc += __shfl_xor(bacc, 1);
  bacc += __shfl_xor(bacc, 2);
  bacc += __shfl_xor(bacc, 4);
  // per-workgroup partial sums (one plain store each); wgrad_reduce_kernel adds them up in a fixed order, so the
  // gradient is bitwise reproducible and no float atomics contend on the 3 K-entry tensor
  float* mine = part + (size_t)blockIdx.x * (C * C * K + C);
#pragma unroll
  for (int p = 0; p < (C * C) / 256; ++p) {
    const int ci = tid % C, co = (tid / C) * ((C * C) / 256) + p;
#pragma unroll
    for (int k = 0; k < K; ++k) mine[((size_t)co * C + ci) * K + k] = acc[p][k];
  }
  if ((tid & 7) == 0) mine[C * C * K + (tid >> 3)] = bacc;
}

// 64 outputs x 16 slices of the partial rows per workgroup; fixed summation order -> reproducible gradients
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ part, int nblk, int nW, int nB,
                                                            float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float sh[16][64];
  const int o = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  float s = 0.f;
  if (i < nW + nB)
    for (int b = slice; b < nblk; b += 16) s += part[(size_t)b * (nW + nB) + i];
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

// ------------------------------------------------------------------------------------------- pooling
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, int64_t rows, int L, int Lout, int k, int s, int p,
                                   float* __restrict__ y, int32_t* __restrict__ arg /* optional */) {
  const int64_t total = rows * Lout;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / Lout;
    const int lo = (int)(i - r * Lout);
    const float* xr = x + r * L;
    float m = -INFINITY;
    int am = -1;
    if (k <= 16) {
      // the window's loads together (index clamped, validity applied afterwards): a load behind `if (in range)` carries a full wait of
      // its own -- fifteen serial round trips per output on the towers' first pool
      float v[16];
      const int l0 = lo * s - p;
#pragma unroll
      for (int w = 0; w < 16; ++w) {
        const int l = l0 + w;
        const bool ok = (w < k) & (l >= 0) & (l < L);
        v[w] = xr[ok ? l : 0];
      }
#pragma unroll
      for (int w = 0; w < 16; ++w) {
        const int l = l0 + w;
        const bool ok = (w < k) & (l >= 0) & (l < L);
        if (ok && (v[w] > m || am < 0)) { m = v[w]; am = l; }      // first maximum wins, like torch
      }
    } else {
      for (int w = 0; w < k; ++w) {
        const int l = lo * s - p + w;
        if (l < 0 || l >= L) continue;
        const float v = xr[l];
        if (v > m || am < 0) { m = v; am = l; }    // first maximum wins, like torch
      }
    }
    y[i] = m;
    if (arg) arg[i] = am;
  }
}

// wide windows (the global max over a long row): one wave per output, lanes stride over the window, then a shuffle reduction
// that keeps the first maximum
__global__ __launch_bounds__(256) void maxpool_fwd_wide_kernel(const float* __restrict__ x, int64_t rows, int L, int Lout, int k, int s,
                                                               int p, float* __restrict__ y, int32_t* __restrict__ arg) {
  const int64_t total = rows * Lout;
  const int lane = threadIdx.x & 63;
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < total; i += (int64_t)gridDim.x * 4) {
    const int64_t r = i / Lout;
    const int lo = (int)(i - r * Lout);
    const float* xr = x + r * L;
    float m = -INFINITY;
    int am = INT_MAX;
    for (int w = lane; w < k; w += 64) {
      const int l = lo * s - p + w;
      if (l < 0 || l >= L) continue;
      const float v = xr[l];
      if (v > m || am == INT_MAX) { m = v; am = l; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float om = __shfl_xor(m, off, 64);
      const int oa = __shfl_xor(am, off, 64);
      if (oa != INT_MAX && (am == INT_MAX || om > m || (om == m && oa < am))) { m = om; am = oa; }
    }
    if (lane == 0) {
      y[i] = m;
      if (arg) arg[i] = am == INT_MAX ? -1 : am;
    }
  }
}

__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ arg, int64_t rows, int L, int Lout,
                                   float* __restrict__ dx) {
  const int64_t total = rows * Lout;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / Lout;
    atomicAdd(&dx[r * L + arg[i]], dy[i]);
  }
}

// non-overlapping windows (stride >= kernel, the model's pools): every input column belongs to at most one window, so
// the gradient is a gather - no atomics, and dx needs no zero fill
__global__ void maxpool_bwd_gather_kernel(const float* __restrict__ dy, const int32_t* __restrict__ arg, int64_t rows, int L,
                                          int Lout, int k, int s, int p, float* __restrict__ dx) {
  const int64_t total = rows * L;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / L;
    const int j = (int)(i - r * L);
    const int w = (j + p) / s;                       // the only window that can contain column j
    float g = 0.f;
    if (w < Lout && j + p - w * s < k && arg[r * Lout + w] == j) g = dy[r * Lout + w];
    dx[i] = g;
  }
}

// ------------------------------------------------------------------------------------------- first layer (one-hot input)
// symbol histogram of the tower's input columns.  A wave counts its 64 symbols per round with one ballot per base (A/C/G/T are all
// but a handful of the symbols) into scalar counters; anything else takes the LDS-atomic path.  One global atomic per symbol and
// workgroup at the end.
__global__ __launch_bounds__(256) void sym_hist_kernel(const uint8_t* __restrict__ sym, int64_t B, int Lwin, int col0, int L1,
                                                       unsigned long long* __restrict__ counts) {
  __shared__ unsigned int h[N_SYM];
  if (threadIdx.x < N_SYM) h[threadIdx.x] = 0;
  __syncthreads();
  unsigned int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  // a workgroup walks whole rows, its threads the row's columns, UN byte loads of a lane in flight; one workgroup per CU: the
  // final global atomics all land on the same four addresses and serialise (8192 of them cost 28 us whatever the input size)
  constexpr int UN = 4;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const uint8_t* row = sym + b * Lwin + col0;
    for (int j0 = 0; j0 < L1; j0 += 256 * UN) {                    // uniform trip count per wave: the ballots see every lane
      int v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int j = j0 + 256 * u + (int)threadIdx.x;
        v[u] = j < L1 ? (row[j] & 15) : -1;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        c0 += __popcll(__ballot(v[u] == 0));
        c1 += __popcll(__ballot(v[u] == 1));
        c2 += __popcll(__ballot(v[u] == 2));
        c3 += __popcll(__ballot(v[u] == 3));
        if (v[u] > 3) atomicAdd(&h[v[u]], 1u);
      }
    }
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&h[0], c0);
    atomicAdd(&h[1], c1);
    atomicAdd(&h[2], c2);
    atomicAdd(&h[3], c3);
  }
  __syncthreads();
  if (threadIdx.x < N_SYM && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// The two towers' histograms in ONE launch for the composed step (the large tower sees every byte of the symbol buffer, the mid tower a
// crop of every row): workgroups [0, HIST2_FLAT) walk the buffer as 16-byte pieces -- A C G T are counted with three bit tricks and a
// population count per dword, a piece that holds any other symbol goes byte by byte through LDS atomics --, the rest count the crop with
// wave ballots like sym_hist_kernel.  1024-thread workgroups: the final global atomics of a launch all land on the same few addresses.
constexpr int HIST2_FLAT = 64, HIST2_CROP = 64;
__global__ __launch_bounds__(1024) void sym_hist2_kernel(const uint8_t* __restrict__ sym, int64_t B, int Lwin, int col0, int L1,
                                                         unsigned long long* __restrict__ counts_full, unsigned long long* __restrict__ counts_crop) {
  __shared__ unsigned int h[N_SYM];
  const int tid = threadIdx.x;
  if (tid < N_SYM) h[tid] = 0;
  __syncthreads();
  unsigned int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  if ((int)blockIdx.x < HIST2_FLAT) {
    const int64_t n = B * Lwin, n16 = n >> 4;
    const uint4* p = reinterpret_cast<const uint4*>(sym);
    constexpr int UN = 4;
    const int64_t stride = (int64_t)HIST2_FLAT * 1024;
    for (int64_t i0 = (int64_t)blockIdx.x * 1024 + tid; i0 < n16; i0 += stride * UN) {
      uint4 v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) v[u] = i0 + stride * u < n16 ? p[i0 + stride * u] : make_uint4(0xFFFFFFFFu, 0, 0, 0);      // (0xFF: skipped below)
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (i0 + stride * u >= n16) continue;
        const uint32_t w[4] = {v[u].x & 0x0F0F0F0Fu, v[u].y & 0x0F0F0F0Fu, v[u].z & 0x0F0F0F0Fu, v[u].w & 0x0F0F0F0Fu};
        if (((w[0] | w[1] | w[2] | w[3]) & 0x0C0C0C0Cu) == 0u) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t b0 = w[q] & 0x01010101u, b1 = (w[q] >> 1) & 0x01010101u;
            c3 += __popc(b0 & b1);
            c2 += __popc(b1 & ~b0);
            c1 += __popc(b0 & ~b1);
          }
          c0 += 16;      // (made exact below: c0 -= c1 + c2 + c3)
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < 4; ++b) atomicAdd(&h[(w[q] >> (8 * b)) & 15u], 1u);
        }
      }
    }
    if (blockIdx.x == 0 && tid == 0)
      for (int64_t j = n16 << 4; j < n; ++j) atomicAdd(&h[sym[j] & 15], 1u);
    c0 -= c1 + c2 + c3;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      c0 += __shfl_xor(c0, off); c1 += __shfl_xor(c1, off); c2 += __shfl_xor(c2, off); c3 += __shfl_xor(c3, off);
    }
  } else {
    // a wave takes RW rows per round, every lane UN bytes of each: RW * UN byte loads in flight (a row per round was a round trip per row)
    const int wave = tid >> 6, lane = tid & 63;
    const int nw = HIST2_CROP * 16;
    constexpr int UN = 4, RW = 4;
    for (int64_t b0 = ((int64_t)(blockIdx.x - HIST2_FLAT) * 16 + wave) * RW; b0 < B; b0 += (int64_t)nw * RW) {
      for (int j0 = 0; j0 < L1; j0 += 64 * UN) {
        int v[RW][UN];
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
          for (int u = 0; u < UN; ++u) {
            const int j = j0 + 64 * u + lane;
            const int64_t b = b0 + r;
            const int raw = sym[(b < B ? b : 0) * Lwin + col0 + (j < L1 ? j : 0)] & 15;      // (clamped address, unconditional load)
            v[r][u] = (j < L1 && b < B) ? raw : -1;
          }
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
          for (int u = 0; u < UN; ++u) {
            c0 += __popcll(__ballot(v[r][u] == 0));
            c1 += __popcll(__ballot(v[r][u] == 1));
            c2 += __popcll(__ballot(v[r][u] == 2));
            c3 += __popcll(__ballot(v[r][u] == 3));
            if (v[r][u] > 3) atomicAdd(&h[v[r][u]], 1u);
          }
      }
    }
  }
  if ((tid & 63) == 0) {
    atomicAdd(&h[0], c0);
    atomicAdd(&h[1], c1);
    atomicAdd(&h[2], c2);
    atomicAdd(&h[3], c3);
  }
  __syncthreads();
  unsigned long long* counts = (int)blockIdx.x < HIST2_FLAT ? counts_full : counts_crop;
  if (tid < N_SYM && h[tid]) atomicAdd(&counts[tid], (unsigned long long)h[tid]);
}

__constant__ float kSymFrac[15][4] = {
    {1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}, {.25f, .25f, .25f, .25f},
    {.5f, 0, .5f, 0}, {0, .5f, 0, .5f}, {.5f, .5f, 0, 0}, {0, .5f, .5f, 0}, {.5f, 0, 0, .5f}, {0, 0, .5f, .5f},
    {0, 1.f / 3, 1.f / 3, 1.f / 3}, {1.f / 3, 0, 1.f / 3, 1.f / 3}, {1.f / 3, 1.f / 3, 0, 1.f / 3}, {1.f / 3, 1.f / 3, 1.f / 3, 0}};

__host__ __device__ __forceinline__ int first_tab_floats(int C) { return 3 * N_SYM * C + 2 * N_SYM * 4 + 8; }
__host__ __device__ __forceinline__ float* first_lutblk(float* tab, int C) { return tab + first_tab_floats(C); }

// batch statistics of the one-hot tensor from the histogram -> BN output per (symbol, channel), per-tap tables
// tab: [0..3*16*C) taps[t][sym][co] | bnval[16][4] | xhat[16][4] | mean[4] | invstd[4] | (with_lut: lut | taps | bias block
// in the layout of the prediction path's stage-1 tables, snv.h SNV_LUTBLK)
__device__ __forceinline__ void first_tables_body(const unsigned long long* __restrict__ counts, int C, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, const float* __restrict__ W /*[C][4][3]*/,
                                                  const float* __restrict__ bias, float eps, float momentum,
                                                  float* __restrict__ running_mean, float* __restrict__ running_var,
                                                  float* __restrict__ tab, int with_lut) {
  __shared__ float bnv[N_SYM][4], xh[N_SYM][4];
  const int tid = threadIdx.x;
  float* taps = tab;
  float* bnval = tab + 3 * N_SYM * C;
  float* xhat = bnval + N_SYM * 4;
  float* stat = xhat + N_SYM * 4;
  if (tid < 4) {
    double n = 0, s = 0, q = 0;
    for (int sym = 0; sym < 15; ++sym) {
      const double cnt = (double)counts[sym];
      n += cnt;
      s += cnt * kSymFrac[sym][tid];
      q += cnt * (double)kSymFrac[sym][tid] * kSymFrac[sym][tid];
    }
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0) var = 0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    stat[tid] = (float)mean;
    stat[4 + tid] = (float)invstd;
    for (int sym = 0; sym < N_SYM; ++sym) {
      const float h = sym < 15 ? (float)((kSymFrac[sym][tid] - mean) * invstd) : 0.f;
      xh[sym][tid] = h;
      bnv[sym][tid] = sym < 15 ? fmaf(gamma[tid], h, beta[tid]) : 0.f;   // PAD: zero padding after the BN
    }
    if (running_mean) {
      const double unbiased = n > 1 ? var * n / (n - 1) : var;
      running_mean[tid] = (float)((1.0 - momentum) * running_mean[tid] + momentum * mean);
      running_var[tid] = (float)((1.0 - momentum) * running_var[tid] + momentum * unbiased);
    }
  }
  __syncthreads();
  for (int i = tid; i < N_SYM * 4; i += blockDim.x) {
    bnval[i] = bnv[i / 4][i % 4];
    xhat[i] = xh[i / 4][i % 4];
  }
  for (int i = tid; i < 3 * N_SYM * C; i += blockDim.x) {
    const int t = i / (N_SYM * C), r = i - t * N_SYM * C, sym = r / C, co = r - sym * C;
    float acc = 0.f;
    for (int ci = 0; ci < 4; ++ci) acc = fmaf(W[(co * 4 + ci) * 3 + t], bnv[sym][ci], acc);
    taps[i] = acc;
  }
  if (!with_lut) return;
  __syncthreads();                       // taps[] (global) written by this workgroup are visible to it
  float* blk = first_lutblk(tab, C);
  for (int i = tid; i < 125 * C; i += blockDim.x) {
    const int e = i / C, co = i - e * C, l = e / 25, m = (e / 5) % 5, r = e % 5;
    // same association as the per-tap path of the lookup kernels: ((bias + left) + centre) + right
    blk[i] = ((bias[co] + taps[(0 * N_SYM + l) * C + co]) + taps[(1 * N_SYM + m) * C + co]) + taps[(2 * N_SYM + r) * C + co];
  }
  for (int i = tid; i < 3 * N_SYM * C; i += blockDim.x) blk[125 * C + i] = taps[i];
  for (int i = tid; i < C; i += blockDim.x) blk[125 * C + 3 * N_SYM * C + i] = bias[i];
}

__global__ void first_tables_kernel(const unsigned long long* __restrict__ counts, int C, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ W /*[C][4][3]*/,
                                    const float* __restrict__ bias, float eps, float momentum,
                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                    float* __restrict__ tab, int with_lut) {
  first_tables_body(counts, C, gamma, beta, W, bias, eps, momentum, running_mean, running_var, tab, with_lut);
}

// both towers' tables in one launch (blockIdx.x = tower)
struct FirstTablesJob {
  const unsigned long long* counts;
  const float* gamma;
  const float* beta;
  const float* W;
  const float* bias;
  float* running_mean;
  float* running_var;
  float* tab;
};
struct FirstTablesJobs { FirstTablesJob j[2]; };
__global__ void first_tables2_kernel(const FirstTablesJobs jobs, int C, float eps, float momentum) {
  const FirstTablesJob& j = jobs.j[blockIdx.x];
  first_tables_body(j.counts, C, j.gamma, j.beta, j.W, j.bias, eps, momentum, j.running_mean, j.running_var, j.tab, 1);
}

// conv1 (via per-tap tables) + maxpool1 with argmax: one thread per (b, co, pooled column)
__global__ __launch_bounds__(256) void first_pool_fwd_kernel(const uint8_t* __restrict__ sym, int64_t B, int Lwin, int col0,
                                                             int L1, int C, int L2, int pk, int ps, int pp,
                                                             const float* __restrict__ tab, const float* __restrict__ bias,
                                                             float* __restrict__ y, int32_t* __restrict__ arg) {
  extern __shared__ float staps[];   // [3][16][C]
  for (int i = threadIdx.x; i < 3 * N_SYM * C; i += blockDim.x) staps[i] = tab[i];
  __syncthreads();
  const int64_t total = B * C * L2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int l2 = (int)(i % L2);
    const int co = (int)((i / L2) % C);
    const int64_t b = i / ((int64_t)L2 * C);
    const uint8_t* s = sym + b * Lwin + col0;
    float m = -INFINITY;
    int am = -1;
    for (int w = 0; w < pk; ++w) {
      const int j = l2 * ps - pp + w;
      if (j < 0 || j >= L1) continue;
      const int sl = j > 0 ? (s[j - 1] & 15) : SYM_PAD, sc = s[j] & 15, sr = j < L1 - 1 ? (s[j + 1] & 15) : SYM_PAD;
      const float v = bias[co] + staps[(0 * N_SYM + sl) * C + co] + staps[(1 * N_SYM + sc) * C + co] +
                      staps[(2 * N_SYM + sr) * C + co];
      if (v > m || am < 0) { m = v; am = j; }
    }
    y[i] = m;
    arg[i] = am;
  }
}

// dTap[t][sym][co] += dy at the arg-max column's three taps; dbias[co] += dy
__global__ __launch_bounds__(256) void first_pool_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ arg,
                                                             const uint8_t* __restrict__ sym, int64_t B, int Lwin, int col0,
                                                             int L1, int C, int L2, float* __restrict__ dtap,
                                                             float* __restrict__ dbias) {
  extern __shared__ float sd[];   // [3][16][C] + [C]
  const int ntab = 3 * N_SYM * C;
  for (int i = threadIdx.x; i < ntab + C; i += blockDim.x) sd[i] = 0.f;
  __syncthreads();
  const int64_t total = B * C * L2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)((i / L2) % C);
    const int64_t b = i / ((int64_t)L2 * C);
    const uint8_t* s = sym + b * Lwin + col0;
    const int j = arg[i];
    const float g = dy[i];
    const int sl = j > 0 ? (s[j - 1] & 15) : SYM_PAD, sc = s[j] & 15, sr = j < L1 - 1 ? (s[j + 1] & 15) : SYM_PAD;
    atomicAdd(&sd[(0 * N_SYM + sl) * C + co], g);
    atomicAdd(&sd[(1 * N_SYM + sc) * C + co], g);
    atomicAdd(&sd[(2 * N_SYM + sr) * C + co], g);
    atomicAdd(&sd[ntab + co], g);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ntab; i += blockDim.x)
    if (sd[i] != 0.f) atomicAdd(&dtap[i], sd[i]);
  for (int i = threadIdx.x; i < C; i += blockDim.x)
    if (sd[ntab + i] != 0.f) atomicAdd(&dbias[i], sd[ntab + i]);
}

// ordered sum of the per-workgroup gradient tables of first_train_kernel<.., true>: 64 columns x 16 row slices per workgroup (a
// slice walks every 16th table with 8 loads in flight, the slices meet in LDS in a fixed order) -- one thread per column summing
// 256 tables serially was 64 dependent rounds on 22 workgroups: 46 us at the tail of the backward
__global__ __launch_bounds__(1024) void first_part_reduce_kernel(const float* __restrict__ part, int nblk, int n, int pitch,
                                                                 float* __restrict__ red) {
  __shared__ float sh[16][64];
  const int o = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  float s = 0.f;
  if (i < n) {
    const float* p = part + i;
    int b = slice;
    float s0 = 0.f, s1 = 0.f;
    for (; b + 7 * 16 < nblk; b += 8 * 16) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = p[(size_t)(b + 16 * q) * pitch];
      s0 += (v[0] + v[1]) + (v[2] + v[3]);
      s1 += (v[4] + v[5]) + (v[6] + v[7]);
    }
    for (; b < nblk; b += 16) s0 += p[(size_t)b * pitch];
    s = s0 + s1;
  }
  sh[slice][o] = s;
  __syncthreads();
  if (slice == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sh[q][o];
    red[i] = t;
  }
}

// dW[co][ci][t] = sum_sym dTap[t][sym][co] * bnval[sym][ci];  dgamma / dbeta of the BN(4) through the BN outputs.
// dlut (optional, [125][C] + taps [3][16][C] + bias [C] as one block): gradient of the 3-mer table of the lookup kernels,
// folded into the per-tap gradients first (lut[l,m,r] = bias + tap0[l] + tap1[m] + tap2[r]); dbias is then written too.
__global__ __launch_bounds__(256) void first_param_grad_kernel(const float* __restrict__ dtap, const float* __restrict__ dlutblk,
                                                               const float* __restrict__ tab, int C,
                                                               const float* __restrict__ W, float* __restrict__ dW,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ dbias, int lut_is_zero) {
  extern __shared__ float sg[];          // dtap [3][16][C] | dbn [16][4]
  float* sdt = sg;
  float* sdbn = sg + 3 * N_SYM * C;
  const float* bnval = tab + 3 * N_SYM * C;
  const float* xhat = bnval + N_SYM * 4;
  const int tid = threadIdx.x;
  for (int i = tid; i < 3 * N_SYM * C; i += blockDim.x) {
    float acc = dlutblk ? dlutblk[125 * C + i] : dtap[i];
    if (dlutblk) {
      const int t = i / (N_SYM * C), r = i - t * N_SYM * C, sym = r / C, co = r - sym * C;
      if (sym < 5 && !lut_is_zero) {      // (the channel-last backward of the composed step sums per tap directly: no 3-mer part)
        for (int u = 0; u < 25; ++u) {     // the 25 table entries whose tap-t symbol is `sym`
          const int e = t == 0 ? 25 * sym + u : (t == 1 ? 25 * (u / 5) + 5 * sym + (u % 5) : 5 * u + sym);
          acc += dlutblk[e * C + co];
        }
      }
    }
    sdt[i] = acc;
  }
  if (dlutblk)
    for (int i = tid; i < C; i += blockDim.x) dbias[i] = dlutblk[125 * C + 3 * N_SYM * C + i];
  __syncthreads();
  for (int i = tid; i < C * 4 * 3; i += blockDim.x) {
    const int co = i / 12, r = i - co * 12, ci = r / 3, t = r - ci * 3;
    float acc = 0.f;
    for (int sym = 0; sym < 15; ++sym) acc = fmaf(sdt[(t * N_SYM + sym) * C + co], bnval[sym * 4 + ci], acc);
    dW[i] = acc;
  }
  for (int i = tid; i < 15 * 4; i += blockDim.x) {   // gradient w.r.t. the BN output of (sym, ci)
    const int sym = i / 4, ci = i - sym * 4;
    float dbn = 0.f;
    for (int t = 0; t < 3; ++t)
      for (int co = 0; co < C; ++co) dbn = fmaf(sdt[(t * N_SYM + sym) * C + co], W[(co * 4 + ci) * 3 + t], dbn);
    sdbn[i] = dbn;
  }
  __syncthreads();
  if (tid < 4) {
    float dg = 0.f, dbt = 0.f;
    for (int sym = 0; sym < 15; ++sym) {
      dg = fmaf(sdbn[sym * 4 + tid], xhat[sym * 4 + tid], dg);
      dbt += sdbn[sym * 4 + tid];
    }
    dgamma[tid] = dg;
    dbeta[tid] = dbt;
  }
}

// ------------------------------------------------------------------------------------------- small dense ops
__global__ void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ b,
                                  int64_t B, int I, int O, float* __restrict__ y) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= B * O) return;
  const int64_t r = i / O;
  const int o = (int)(i - r * O);
  float acc = b ? b[o] : 0.f;
  const float* xr = x + r * I;
  const float* w = W + (size_t)o * I;
  for (int k = 0; k < I; ++k) acc = fmaf(xr[k], w[k], acc);
  y[i] = acc;
}

__global__ void linear_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ W, int64_t B, int I, int O,
                                    float* __restrict__ dx) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= B * I) return;
  const int64_t r = i / I;
  const int k = (int)(i - r * I);
  float acc = 0.f;
  for (int o = 0; o < O; ++o) acc = fmaf(dy[r * O + o], W[(size_t)o * I + k], acc);
  dx[i] = acc;
}

// contiguous global -> LDS copy by a 256-thread workgroup (float4 when the source is 16-byte aligned), zero fill up to `pad`
// (eight loads of a thread in flight per round, index clamped instead of a branch around the load: the one-load-per-iteration loop
// waited out a global round trip per 4 KB -- 30 rounds for a 120 KB weight matrix, most of the dense kernels' 30-80 us)
__device__ __forceinline__ void copy_to_lds(float* dst, const float* __restrict__ src, int n, int pad, int tid) {
  constexpr int UN = 8;
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    const int n4 = n >> 2;
    for (int i0 = tid; i0 < n4; i0 += 256 * UN) {
      float4 v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = i0 + 256 * u;
        v[u] = reinterpret_cast<const float4*>(src)[i < n4 ? i : n4 - 1];
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = i0 + 256 * u;
        if (i < n4) reinterpret_cast<float4*>(dst)[i] = v[u];
      }
    }
    for (int i = 4 * n4 + tid; i < n; i += 256) dst[i] = src[i];
  } else {
    for (int i0 = tid; i0 < n; i0 += 256 * UN) {
      float v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = i0 + 256 * u;
        v[u] = src[i < n ? i : n - 1];
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = i0 + 256 * u;
        if (i < n) dst[i] = v[u];
      }
    }
  }
  for (int i = n + tid; i < pad; i += 256) dst[i] = 0.f;
}

// Dense layer on the fp32 matrix cores for 64 rows per workgroup: y[b][n] = bias[n] + sum_k x[b][k] * Wm(k, n), with
// Wm(k, n) = W[n][k] (trans = 0: forward, W is [N][K]) or W[k][n] (trans = 1: input gradient, W is [K][N]).  The whole weight
// matrix and the x tile are copied linearly into LDS; wave w owns rows 16w..16w+15 (A fragments in registers, K <= 256) and
// walks the 16-column output blocks: D[16 rows][16 outputs] += x[16][4] * Wm[4][16] per v_mfma_f32_16x16x4_f32.
constexpr int LIN_MAXK = 256, LIN_MAXNT = 16;
__global__ __launch_bounds__(256) void linear_mfma_kernel(const float* __restrict__ x, const float* __restrict__ W, int trans,
                                                          const float* __restrict__ bias, int64_t B, int K, int N,
                                                          float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  const int KN4 = (K * N + 3) & ~3;
  float* ws = lsm;                 // W as stored
  float* xs = lsm + KN4;           // [64][K] as stored
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kk = lane >> 4;
  const int64_t b0 = (int64_t)blockIdx.x * 64;
  const int rows = (int)((B - b0) < 64 ? (B - b0) : 64);
  copy_to_lds(ws, W, K * N, K * N, tid);
  copy_to_lds(xs, x + (size_t)b0 * K, rows * K, 64 * K, tid);
  __syncthreads();
  const int ksteps = (K + 3) / 4;
  float af[LIN_MAXK / 4];
#pragma unroll
  for (int s = 0; s < LIN_MAXK / 4; ++s) {
    const int k = 4 * s + kk;
    af[s] = (s < ksteps && k < K) ? xs[(16 * wave + n16) * K + k] : 0.f;
  }
  // the bias of every output block up front (N <= 16 x LIN_MAXNT columns; a load per block inside the loop is a round trip per block)
  float bvs[LIN_MAXNT];
#pragma unroll
  for (int t = 0; t < LIN_MAXNT; ++t) bvs[t] = 0.f;
  if (bias) {
#pragma unroll
    for (int t = 0; t < LIN_MAXNT; ++t) {
      const int n = 16 * t + n16;
      bvs[t] = bias[n < N ? n : 0];
    }
  }
#pragma unroll
  for (int nt = 0; nt < LIN_MAXNT; ++nt) {
    if (16 * nt >= N) break;
    const int n = 16 * nt + n16;
    const bool nv = n < N;
    const float bv = nv ? bvs[nt] : 0.f;
    f32x4_t acc = {bv, bv, bv, bv};
    const float* wcol = ws + (trans ? n : n * K);
    const int kstride = trans ? N : 1;
#pragma unroll
    for (int s = 0; s < LIN_MAXK / 4; ++s) {
      if (s < ksteps) {
        const int k = 4 * s + kk;
        const float bf = (nv && k < K) ? wcol[k * kstride] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf, acc, 0, 0, 0);
      }
    }
    if (nv) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * wave + 4 * kk + r;
        if (row < rows) y[(size_t)(b0 + row) * N + n] = acc[r];
      }
    }
  }
}

// dW[o][i] = sum_b dy[b][o] x[b][i], db[o] = sum_b dy[b][o] on the fp32 matrix cores: one workgroup of 16 waves per 16 x 16
// block of dW; wave w reduces rows [w B/16, (w+1) B/16) in k-steps of 4 rows with operands straight from HBM / L2
// (each 16-column strip of dy and x is read by the workgroups of one block row / column only), the 16 partial tiles meet
// in LDS in a fixed order: no atomics, reproducible.
__global__ __launch_bounds__(1024) void linear_wgrad_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 int64_t B, int I, int O, float* __restrict__ dW,
                                                                 float* __restrict__ db) {
  __shared__ float red[16][16 * 16 + 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kk = lane >> 4;
  const int nbi = (I + 15) / 16;
  const int ob = blockIdx.x / nbi, ib = blockIdx.x - ob * nbi;
  const int o = 16 * ob + n16, i = 16 * ib + n16;
  const bool ov = o < O, iv = i < I;
  const int64_t per = ((B + 15) / 16 + 3) & ~(int64_t)3;            // rows per wave, a multiple of 4
  const int64_t r0 = wave * per, r1 = (r0 + per < B) ? r0 + per : B;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  for (int64_t r = r0; r < r1; r += 32) {
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {                                      // 16 loads in flight per lane
      const int64_t row = r + 4 * u + kk;
      const bool rv = row < r1;
      a[u] = (rv && ov) ? dy[row * O + o] : 0.f;
      b[u] = (rv && iv) ? x[row * I + i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
      bsum += a[u];
    }
  }
  // D[row = o 4kk+r][col = i n16]
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][(4 * kk + r) * 16 + n16] = acc[r];
  bsum += __shfl_xor(bsum, 16);
  bsum += __shfl_xor(bsum, 32);
  if (lane < 16) red[wave][256 + lane] = bsum;
  __syncthreads();
  if (tid < 256 + 16) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][tid];
    if (tid < 256) {
      const int oo = 16 * ob + (tid >> 4), ii = 16 * ib + (tid & 15);
      if (oo < O && ii < I) dW[(size_t)oo * I + ii] = t;
    } else if (ib == 0 && db) {
      const int oo = 16 * ob + (tid - 256);
      if (oo < O) db[oo] = t;
    }
  }
}

__global__ void embedding_fwd_kernel(const int64_t* __restrict__ cat, const float* __restrict__ E, int64_t B, int cols, int rows,
                                     float* __restrict__ y) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= B * cols * 5) return;
  const int d = (int)(i % 5);
  const int64_t bc = i / 5;
  int64_t id = cat[bc];
  id = id < 0 ? 0 : (id >= rows ? rows - 1 : id);
  y[i] = E[id * 5 + d];
}

__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* __restrict__ cat, const float* __restrict__ dy,
                                                            int64_t B, int cols, int rows, float* __restrict__ dE) {
  extern __shared__ float se[];   // [rows][5]
  for (int i = threadIdx.x; i < rows * 5; i += blockDim.x) se[i] = 0.f;
  __syncthreads();
  const int64_t total = B * cols * 5;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % 5);
    int64_t id = cat[i / 5];
    id = id < 0 ? 0 : (id >= rows ? rows - 1 : id);
    atomicAdd(&se[id * 5 + d], dy[i]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < rows * 5; i += blockDim.x)
    if (se[i] != 0.f) atomicAdd(&dE[i], se[i]);
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// y = x * keep / (1 - p), keep ~ Bernoulli(1 - p) from a counter-based generator (seed, element index)
__global__ void dropout_kernel(const float* __restrict__ x, int64_t total, float p, uint64_t seed,
                               const uint64_t* __restrict__ seed_dev, float* __restrict__ y) {
  if (seed_dev) seed += *seed_dev;     // device-resident step counter: a captured graph draws a new mask on every replay
  const float scale = 1.f / (1.f - p);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t r = mix64(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1));
    const float u = (float)(r >> 40) * (1.f / 16777216.f);
    y[i] = u >= p ? x[i] * scale : 0.f;
  }
}

__global__ void scale_mask_kernel(const float* __restrict__ x, const float* __restrict__ ref, int64_t total, int mode,
                                  float* __restrict__ y) {
  // mode 0: y = x where ref > 0 else 0 (ReLU backward on the saved output)
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = ref[i] > 0.f ? x[i] : 0.f;
}

// head forward: out = log(clamp((sm(loc) + (sm(mid) + sm(large)) / 2) / 2, 1e-9)) ; Network1: without the local term
__global__ void head_fwd_kernel(const float* __restrict__ loc, const float* __restrict__ mid, const float* __restrict__ lar,
                                int64_t B, int nc, float* __restrict__ out) {
  const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* v[3] = {lar + b * nc, mid + b * nc, loc ? loc + b * nc : nullptr};
  float mx[3], sm[3];
  for (int t = 0; t < 3; ++t) {
    mx[t] = -INFINITY;
    sm[t] = 0.f;
    if (!v[t]) continue;
    for (int k = 0; k < nc; ++k) mx[t] = fmaxf(mx[t], v[t][k]);
    for (int k = 0; k < nc; ++k) sm[t] += expf(v[t][k] - mx[t]);
  }
  for (int k = 0; k < nc; ++k) {
    float p = (expf(v[1][k] - mx[1]) / sm[1] + expf(v[0][k] - mx[0]) / sm[0]) / 2.f;
    if (v[2]) p = (expf(v[2][k] - mx[2]) / sm[2] + p) / 2.f;
    out[b * nc + k] = logf(fmaxf(p, 1e-9f));
  }
}

// NC > 0: the class count at compile time (the shipped 4): every loop unrolls and the per-class arrays live in registers -- with a run-time
// count they are indexed dynamically and sit in scratch memory (16 us for 4096 rows against 5).  Same operations in the same order.
template <int NC>
__global__ void head_bwd_kernel(const float* __restrict__ loc, const float* __restrict__ mid, const float* __restrict__ lar,
                                const float* __restrict__ dout, int64_t B, int nc_rt, float* __restrict__ dloc,
                                float* __restrict__ dmid, float* __restrict__ dlar) {
  const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int nc = NC > 0 ? NC : nc_rt;
  constexpr int CAP = NC > 0 ? NC : SNV_MAXCLASS;
  const float* v[3] = {lar + b * nc, mid + b * nc, loc ? loc + b * nc : nullptr};
  float* dv[3] = {dlar + b * nc, dmid + b * nc, dloc ? dloc + b * nc : nullptr};
  const float wgt[3] = {loc ? 0.25f : 0.5f, loc ? 0.25f : 0.5f, 0.5f};
  float s[3][CAP], dp[CAP];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    if (!v[t]) continue;
    float mx = -INFINITY, sum = 0.f;
#pragma unroll
    for (int k = 0; k < CAP; ++k)
      if (k < nc) {
        s[t][k] = v[t][k];
        mx = fmaxf(mx, s[t][k]);
      }
#pragma unroll
    for (int k = 0; k < CAP; ++k)
      if (k < nc) {      // (each exponential once: the same values as exp / sum computed in two passes)
        s[t][k] = expf(s[t][k] - mx);
        sum += s[t][k];
      }
#pragma unroll
    for (int k = 0; k < CAP; ++k)
      if (k < nc) s[t][k] = s[t][k] / sum;
  }
#pragma unroll
  for (int k = 0; k < CAP; ++k)
    if (k < nc) {
      float p = (s[1][k] + s[0][k]) / 2.f;
      if (v[2]) p = (s[2][k] + p) / 2.f;
      dp[k] = p > 1e-9f ? dout[b * nc + k] / p : 0.f;      // clamp passes no gradient below its floor
    }
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    if (!v[t]) continue;
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < CAP; ++k)
      if (k < nc) dot += dp[k] * wgt[t] * s[t][k];
#pragma unroll
    for (int k = 0; k < CAP; ++k)
      if (k < nc) dv[t][k] = s[t][k] * (dp[k] * wgt[t] - dot);
  }
}

int grid_for(int64_t total, int block = 256, int cap = 16384) {
  const int64_t g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace
}  // namespace mural

using namespace mural;
#define STREAM ((hipStream_t)stream)
#define CHECK_LAUNCH() MURAL_HIP_CHECK(hipGetLastError()); return MURAL_OK

extern "C" int mural_op_relayout(const float* W, float* wt, int32_t Cout, int32_t Cin, int32_t K, int32_t dgrad, void* stream) {
  const int total = Cout * Cin * K;
  hipLaunchKernelGGL(relayout_kernel, dim3((total + 255) / 256), dim3(256), 0, STREAM, W, wt, Cout, Cin, K, dgrad);
  CHECK_LAUNCH();
}

// generic conv on [B][Cin][L] with re-laid-out weights wt [Cin][K][Cout]; pre-op BN(+ReLU) per input channel; optional
// bias / post ReLU / two residuals.  stride 1, pad (K-1)/2.
extern "C" int mural_op_conv1d(const float* in, const float* wt, const float* bias, float* out, int64_t B, int32_t Cin,
                               int32_t Cout, int32_t L, int32_t K, const float* pre_s, const float* pre_t, int32_t pre_relu,
                               int32_t post_relu, const float* res1, const float* res2, void* stream) {
  Conv1dArgs a;
  std::memset(&a, 0, sizeof(a));
  a.in = in; a.wt = wt; a.bias = bias; a.out = out;
  a.B = (int)B; a.Cin = Cin; a.Lin = L; a.Cout = Cout; a.Lout = L;
  a.K = K; a.stride = 1; a.pad = (K - 1) / 2; a.up = 1;
  a.pre_s = pre_s; a.pre_t = pre_t; a.pre_relu = pre_relu;
  a.act = post_relu ? ACT_RELU : ACT_NONE; a.res1 = res1; a.res2 = res2;
  return launch_conv1d(a, STREAM);
}



// validation hook (tests/test_gpu_indel.py): one fused ConvBlock launch (conv1d.hip / convblock_mfma.hip) with its optional front
// (k = 7 conv Cf -> C on the input upsampled f_up times; f_up < 0: strided by -f_up instead), skip tensor and tail.  form: 0 the 8-channel block entirely on the vector
// ALU, 1 its split form (convs on the matrix cores), -1 the library's choice.

// Batch sums live in an accumulator block acc = double[MURAL_BN_SLOTS][2][C] (zeroed by the caller): workgroups add into
// the copy picked by their index, readers sum the copies.  [k][0][c] = sum, [k][1][c] = sum of squares (forward) or
// sum(dz), sum(dz * xhat) (backward).
extern "C" int mural_op_bn_stats(const float* x, int64_t B, int32_t C, int32_t L, int32_t relu, double* acc, void* stream) {
  const int64_t per = B * L;
  int gy = (int)((per + 256 * 8 - 1) / (256 * 8));
  gy = gy < 1 ? 1 : (gy > 256 ? 256 : gy);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(C, gy), dim3(256), 0, STREAM, x, (int)B, C, L, relu, acc);
  CHECK_LAUNCH();
}

extern "C" int mural_op_bn_finalize(const double* acc, double n, int32_t C, const float* gamma,
                                    const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                    float* scale, float* shift, float* mean, float* invstd, void* stream) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, STREAM, acc, n, C, gamma, beta, eps, momentum,
                     running_mean, running_var, scale, shift, mean, invstd);
  CHECK_LAUNCH();
}

extern "C" int mural_op_bn_apply(const float* x, int64_t B, int32_t C, int32_t L, int32_t relu, const float* scale,
                                 const float* shift, float* y, void* stream) {
  const int64_t total = B * C * L;
  if (total == 0) return MURAL_OK;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(total)), dim3(256), 0, STREAM, x, total, C, L, relu, scale, shift, y);
  CHECK_LAUNCH();
}

// BatchNorm (batch statistics) backward.  acc: accumulator block; have_sums == 0: zeroed by the caller and reduced here,
// != 0: already holds sum(dz) / sum(dz * xhat) (taken by the producer of dz, mural_op_conv32 stat_mode 2).  Writes dx (+ the optional
// add1 / add2 tensors: gradients arriving at x through residual connections), dgamma, dbeta.
extern "C" int mural_op_bn_backward(const float* dz, const float* x, int64_t B, int32_t C, int32_t L, int32_t relu,
                                    const float* mean, const float* invstd, const float* gamma, double* acc,
                                    int32_t have_sums, const float* add1, const float* add2, float* dx, float* dgamma,
                                    float* dbeta, void* stream) {
  const int64_t per = B * L, total = B * C * L;
  if (total == 0) return MURAL_OK;
  if (!have_sums) {
    int gy = (int)((per + 256 * 8 - 1) / (256 * 8));
    gy = gy < 1 ? 1 : (gy > 256 ? 256 : gy);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, gy), dim3(256), 0, STREAM, dz, x, (int)B, C, L, relu, mean, invstd, acc);
  }
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total, 256, 2048)), dim3(256), (size_t)C * 16, STREAM, dz, x, total, C, L, relu, mean,
                     invstd, gamma, acc, (double)per, add1, add2, dx, dgamma, dbeta);
  CHECK_LAUNCH();
}

// dW [32][32][3] and db [32] of a 32->32 k=3 conv whose input was scale*act(x)+shift; part: float scratch of
// 1024 * 3104 floats for the per-workgroup partial sums
extern "C" int mural_op_conv_wgrad(const float* dy, const float* x, int64_t B, int32_t C, int32_t L, int32_t K,
                                   const float* scale, const float* shift, int32_t pre_relu, float* dW, float* db,
                                   float* part, size_t part_floats, void* stream) {
  MURAL_REQUIRE(C == 32 && K == 3, "conv_wgrad is built for 32 channels, 3 taps (got %d, %d)", C, K);
  const int64_t items = B * ((L + 63) / 64);
  const int grid = (int)(items < 1024 ? (items < 1 ? 1 : items) : 1024);
  MURAL_REQUIRE(part && part_floats >= (size_t)grid * (32 * 32 * 3 + 32), "conv_wgrad: partial-sum scratch too small");
  hipLaunchKernelGGL((conv_wgrad_kernel<32, 3>), dim3(grid), dim3(256), 0, STREAM, dy, x, (int)B, L, scale, shift, pre_relu,
                     part);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((32 * 32 * 3 + 32 + 63) / 64), dim3(1024), 0, STREAM, part, grid, 32 * 32 * 3, 32,
                     dW, db);
  CHECK_LAUNCH();
}

extern "C" int mural_op_maxpool_fwd(const float* x, int64_t rows, int32_t L, int32_t k, int32_t s, int32_t p, float* y,
                                    int32_t* arg, void* stream) {
  const int Lout = (L + 2 * p - k) / s + 1;
  const int64_t total = rows * Lout;
  if (total == 0) return MURAL_OK;
  if (k >= 64) {      // (a wave per window from one wave's width on: 28 -> 8 us for the 102-wide global max of 16 384 rows)
    const int64_t g = (total + 3) / 4;
    hipLaunchKernelGGL(maxpool_fwd_wide_kernel, dim3((unsigned)(g > 65536 ? 65536 : g)), dim3(256), 0, STREAM, x, rows, L, Lout, k, s,
                       p, y, arg);
    CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, STREAM, x, rows, L, Lout, k, s, p, y, arg);
  CHECK_LAUNCH();
}

// k, s, p: the forward's window.  s >= k (disjoint windows): dx is fully written; otherwise dx must be zeroed by the caller
// (mural_op_maxpool_bwd_needs_zero) and receives atomic adds
extern "C" int mural_op_maxpool_bwd_needs_zero(int32_t k, int32_t s) { return s >= k ? 0 : 1; }

extern "C" int mural_op_maxpool_bwd(const float* dy, const int32_t* arg, int64_t rows, int32_t L, int32_t Lout, int32_t k,
                                    int32_t s, int32_t p, float* dx, void* stream) {
  const int64_t total = rows * Lout;
  if (total == 0) return MURAL_OK;
  if (s >= k) {
    hipLaunchKernelGGL(maxpool_bwd_gather_kernel, dim3(grid_for(rows * L)), dim3(256), 0, STREAM, dy, arg, rows, L, Lout, k, s, p,
                       dx);
    CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, STREAM, dy, arg, rows, L, Lout, dx);
  CHECK_LAUNCH();
}

// first layer of a tower in training mode: sizes of the caller-allocated buffers for (C, pool window).  The 3-mer table
// kernels (snv_stage1.hip) serve C == 32 and windows up to 15; everything else takes the generic per-tap kernels.
extern "C" int mural_op_first_plan(int32_t C, int32_t pk, int64_t* tab_floats, int64_t* arg_bytes, int64_t* scratch_floats) {
  const bool fast = first_train_supported(C, pk);
  *tab_floats = first_tab_floats(C) + (fast ? SNV_LUTBLK : 0);
  *arg_bytes = fast ? 1 : 4;                       // per pooled output: window offset (uint8, [B][L2][C]) or column (int32, [B][C][L2])
  *scratch_floats = fast ? (int64_t)(FIRST_TRAIN_MAXGRID + 1) * SNV_LUTBLK : 3 * N_SYM * C;
  return MURAL_OK;
}

// counts: uint64[16] zeroed by the caller; tab / arg sized by mural_op_first_plan
static int first_fwd_impl(const uint8_t* sym, int64_t B, int32_t Lwin, int32_t col0, int32_t L1, int32_t C, int32_t pk,
                          int32_t ps, int32_t pp, const float* gamma, const float* beta, const float* W,
                          const float* bias, float eps, float momentum, float* running_mean, float* running_var,
                          unsigned long long* counts, float* tab, float* y, void* arg, int cl, double* stat, void* stream) {
  const int L2 = (L1 + 2 * pp - pk) / ps + 1;
  const bool fast = first_train_supported(C, pk);
  hipLaunchKernelGGL(sym_hist_kernel, dim3((unsigned)(B < 1024 ? B : 1024)), dim3(256), 0, STREAM, sym, B, Lwin, col0, L1, counts);
  hipLaunchKernelGGL(first_tables_kernel, dim3(1), dim3(256), 0, STREAM, counts, C, gamma, beta, W, bias, eps, momentum,
                     running_mean, running_var, tab, fast ? 1 : 0);
  if (fast) {
    FirstTrainArgs a;
    std::memset(&a, 0, sizeof(a));
    a.tw = Stage1Tower{L1, col0, L2, pk, ps, pp};
    a.Lwin = Lwin; a.B = B; a.sym = sym; a.lutblk = first_lutblk(tab, C); a.y = y; a.arg = static_cast<uint8_t*>(arg);
    a.cl = cl;
    a.stat = cl ? stat : nullptr;
    return launch_first_train(a, false, STREAM);
  }
  MURAL_REQUIRE(!cl, "first layer: the channel-last layout is served by the table kernels only");
  const int64_t total = B * C * L2;
  hipLaunchKernelGGL(first_pool_fwd_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), (size_t)3 * N_SYM * C * 4, STREAM, sym,
                     B, Lwin, col0, L1, C, L2, pk, ps, pp, tab, bias, y, static_cast<int32_t*>(arg));
  CHECK_LAUNCH();
}

// scratch sized by mural_op_first_plan (contents undefined on entry); writes dW [C][4][3], dbias[C], dgamma[4], dbeta[4]
extern "C" int mural_op_first_fwd(const uint8_t* sym, int64_t B, int32_t Lwin, int32_t col0, int32_t L1, int32_t C, int32_t pk,
                                  int32_t ps, int32_t pp, const float* gamma, const float* beta, const float* W,
                                  const float* bias, float eps, float momentum, float* running_mean, float* running_var,
                                  unsigned long long* counts, float* tab, float* y, void* arg, void* stream) {
  static const int cl = (dev_env("MURAL_DEBUG_FIRST_CL") && C == 32) ? 1 : 0;
  return first_fwd_impl(sym, B, Lwin, col0, L1, C, pk, ps, pp, gamma, beta, W, bias, eps, momentum, running_mean, running_var, counts, tab, y,
                        arg, cl, nullptr, stream);
}

static int first_bwd_impl(const float* dy, const void* arg, const uint8_t* sym, int64_t B, int32_t Lwin,
                          int32_t col0, int32_t L1, int32_t C, int32_t pk, int32_t ps, int32_t pp, const float* tab,
                          const float* W, float* scratch, float* dW, float* dbias, float* dgamma, float* dbeta, int cl,
                          const mural::FirstFold* fold, void* stream) {
  const int L2 = (L1 + 2 * pp - pk) / ps + 1;
  const size_t lds = (size_t)(3 * N_SYM * C + N_SYM * 4) * 4;
  MURAL_REQUIRE(!fold || (cl && first_train_supported(C, pk)), "first layer: the folded BatchNorm-backward apply needs the channel-last table kernels");
  if (first_train_supported(C, pk)) {
    FirstTrainArgs a;
    std::memset(&a, 0, sizeof(a));
    if (fold) a.fold = *fold;
    a.tw = Stage1Tower{L1, col0, L2, pk, ps, pp};
    a.Lwin = Lwin; a.B = B; a.sym = sym; a.dy = dy; a.arg = const_cast<uint8_t*>(static_cast<const uint8_t*>(arg));
    a.dpart = scratch;
    a.cl = cl;
    const int nblk = first_train_grid(B);
    float* red = scratch + (size_t)FIRST_TRAIN_MAXGRID * SNV_LUTBLK;
    if (int rc = launch_first_train(a, true, STREAM)) return rc;
    // (MURAL_DEBUG_FIRST_SCATTER=1 without a fold keeps the LDS-atomic scatter kernel, whose blocks carry a 3-mer part: launch_first_train)
    if (cl && (fold || !dev_env("MURAL_DEBUG_FIRST_SCATTER"))) {      // first_bwd_cl_kernel fills the per-tap and bias parts only (its 3-mer part is zero, not read)
      constexpr int N = SNV_TAPS + SNV_C;
      hipLaunchKernelGGL(first_part_reduce_kernel, dim3((N + 63) / 64), dim3(1024), 0, STREAM, scratch + SNV_LUT, B ? nblk : 0, N, SNV_LUTBLK,
                         red + SNV_LUT);
      hipLaunchKernelGGL(first_param_grad_kernel, dim3(1), dim3(256), lds, STREAM, nullptr, red, tab, C, W, dW, dgamma, dbeta, dbias, 1);
      CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(first_part_reduce_kernel, dim3((SNV_LUTBLK + 63) / 64), dim3(1024), 0, STREAM, scratch, B ? nblk : 0,
                       SNV_LUTBLK, SNV_LUTBLK, red);
    hipLaunchKernelGGL(first_param_grad_kernel, dim3(1), dim3(256), lds, STREAM, nullptr, red, tab, C, W, dW, dgamma, dbeta,
                       dbias, 0);
    CHECK_LAUNCH();
  }
  MURAL_REQUIRE(!cl, "first layer: the channel-last layout is served by the table kernels only");
  float* dtap = scratch;
  MURAL_HIP_CHECK(hipMemsetAsync(dtap, 0, (size_t)3 * N_SYM * C * 4, STREAM));
  MURAL_HIP_CHECK(hipMemsetAsync(dbias, 0, (size_t)C * 4, STREAM));
  const int64_t total = B * C * L2;
  hipLaunchKernelGGL(first_pool_bwd_kernel, dim3(grid_for(total, 256, 1024)), dim3(256), (size_t)(3 * N_SYM * C + C) * 4, STREAM,
                     dy, static_cast<const int32_t*>(arg), sym, B, Lwin, col0, L1, C, L2, dtap, dbias);
  hipLaunchKernelGGL(first_param_grad_kernel, dim3(1), dim3(256), lds, STREAM, dtap, nullptr, tab, C, W, dW, dgamma, dbeta,
                     dbias, 0);
  CHECK_LAUNCH();
}

extern "C" int mural_op_first_bwd(const float* dy, const void* arg, const uint8_t* sym, int64_t B, int32_t Lwin,
                                  int32_t col0, int32_t L1, int32_t C, int32_t pk, int32_t ps, int32_t pp, const float* tab,
                                  const float* W, float* scratch, float* dW, float* dbias, float* dgamma, float* dbeta,
                                  void* stream) {
  // (MURAL_DEBUG_FIRST_CL=1: tools/phase_stamps_first.py times the channel-last form of the composed step through this entry; the
  // buffers have the same sizes, only the element order of dy differs)
  static const int cl = (dev_env("MURAL_DEBUG_FIRST_CL") && C == 32) ? 1 : 0;
  return first_bwd_impl(dy, arg, sym, B, Lwin, col0, L1, C, pk, ps, pp, tab, W, scratch, dW, dbias, dgamma, dbeta, cl, nullptr, stream);
}

namespace mural {
// The composed step's first layers: ONE histogram launch and ONE table launch for both towers (on the caller's stream, in front of the
// fork), then train_first_fwd_cl_prepared per tower runs the lookup kernel alone.  jobs[0] = the tower that sees whole rows (col0 = 0,
// L1 = Lwin), jobs[1] = the crop.  sym: 16-byte aligned.
int train_first_prepare2(const uint8_t* sym, int64_t B, int Lwin, const int* col0, const int* L1, const float* const* gamma,
                         const float* const* beta, const float* const* W, const float* const* bias, float* const* running_mean,
                         float* const* running_var, unsigned long long* const* counts, float* const* tab, float eps, float momentum,
                         hipStream_t stream) {
  MURAL_REQUIRE(col0[0] == 0 && L1[0] == Lwin && (reinterpret_cast<uintptr_t>(sym) & 15) == 0, "first layers: tower 0 must see whole, aligned rows");
  if (B == 0) return MURAL_OK;
  hipLaunchKernelGGL(sym_hist2_kernel, dim3(HIST2_FLAT + HIST2_CROP), dim3(1024), 0, stream, sym, B, Lwin, col0[1], L1[1], counts[0], counts[1]);
  FirstTablesJobs jobs;
  for (int t = 0; t < 2; ++t) jobs.j[t] = FirstTablesJob{counts[t], gamma[t], beta[t], W[t], bias[t], running_mean[t], running_var[t], tab[t]};
  hipLaunchKernelGGL(first_tables2_kernel, dim3(2), dim3(256), 0, stream, jobs, 32, eps, momentum);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}
int train_first_fwd_cl_prepared(const uint8_t* sym, int64_t B, int Lwin, int col0, int L1, int pk, int ps, int pp, const float* tab, float* y,
                                void* arg, double* stat, hipStream_t stream) {
  MURAL_REQUIRE(first_train_supported(32, pk), "first layer: pool window %d not supported by the table kernel", pk);
  FirstTrainArgs a;
  std::memset(&a, 0, sizeof(a));
  a.tw = Stage1Tower{L1, col0, (L1 + 2 * pp - pk) / ps + 1, pk, ps, pp};
  a.Lwin = Lwin; a.B = B; a.sym = sym; a.lutblk = first_lutblk(const_cast<float*>(tab), 32); a.y = y; a.arg = static_cast<uint8_t*>(arg);
  a.cl = 1;
  a.stat = stat;
  return launch_first_train(a, false, stream);
}
int train_first_fwd_cl(const uint8_t* sym, int64_t B, int Lwin, int col0, int L1, int pk, int ps, int pp, const float* gamma, const float* beta,
                       const float* W, const float* bias, float eps, float momentum, float* running_mean, float* running_var,
                       unsigned long long* counts, float* tab, float* y, void* arg, double* stat, hipStream_t stream) {
  return first_fwd_impl(sym, B, Lwin, col0, L1, 32, pk, ps, pp, gamma, beta, W, bias, eps, momentum, running_mean, running_var, counts, tab, y,
                        arg, 1, stat, stream);
}
int train_first_bwd_cl(const float* dy, const void* arg, const uint8_t* sym, int64_t B, int Lwin, int col0, int L1, int pk, int ps, int pp,
                       const float* tab, const float* W, float* scratch, float* dW, float* dbias, float* dgamma, float* dbeta,
                       const FirstFold* fold, hipStream_t stream) {
  return first_bwd_impl(dy, arg, sym, B, Lwin, col0, L1, 32, pk, ps, pp, tab, W, scratch, dW, dbias, dgamma, dbeta, 1, fold, stream);
}
}  // namespace mural

static bool linear_tile_fits(int K, int N, size_t* lds) {
  *lds = ((size_t)((K * N + 3) & ~3) + (size_t)64 * K) * sizeof(float);
  return *lds <= 150 * 1024 && K <= LIN_MAXK && N <= 16 * LIN_MAXNT;
}

static int launch_linear_tile(const float* x, const float* W, int trans, const float* bias, int64_t B, int K, int N,
                              float* y, size_t lds, hipStream_t stream) {
  if (lds > 64 * 1024) {
    static DynLdsOnce big_lds;
    if (int rc = big_lds.ensure(&linear_mfma_kernel)) return rc;
  }
  hipLaunchKernelGGL(linear_mfma_kernel, dim3((unsigned)((B + 63) / 64)), dim3(256), lds, stream, x, W, trans, bias, B, K, N, y);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

extern "C" int mural_op_linear_fwd(const float* x, const float* W, const float* b, int64_t B, int32_t I, int32_t O, float* y,
                                   void* stream) {
  if (B == 0) return MURAL_OK;
  size_t lds;
  if (linear_tile_fits(I, O, &lds)) return launch_linear_tile(x, W, 0, b, B, I, O, y, lds, STREAM);   // Wm(k, n) = W[n][k]
  hipLaunchKernelGGL(linear_fwd_kernel, dim3((unsigned)((B * O + 255) / 256)), dim3(256), 0, STREAM, x, W, b, B, I, O, y);
  CHECK_LAUNCH();
}

extern "C" int mural_op_linear_bwd(const float* dy, const float* x, const float* W, int64_t B, int32_t I, int32_t O, float* dx,
                                   float* dW, float* db, void* stream) {
  if (B == 0) return MURAL_OK;
  size_t lds;
  if (dx) {
    if (linear_tile_fits(O, I, &lds)) {                            // dx[b][i] = sum_o dy[b][o] W[o][i]: Wm(k = o, n = i)
      if (int rc = launch_linear_tile(dy, W, 1, nullptr, B, O, I, dx, lds, STREAM)) return rc;
    } else {
      hipLaunchKernelGGL(linear_bwd_x_kernel, dim3((unsigned)((B * I + 255) / 256)), dim3(256), 0, STREAM, dy, W, B, I, O, dx);
    }
  }
  // dW / db are fully written (no accumulation into the caller's buffer)
  hipLaunchKernelGGL(linear_wgrad_mfma_kernel, dim3((unsigned)(((O + 15) / 16) * ((I + 15) / 16))), dim3(1024), 0, STREAM, dy, x, B,
                     I, O, dW, db);
  CHECK_LAUNCH();
}

extern "C" int mural_op_embedding_fwd(const int64_t* cat, const float* E, int64_t B, int32_t cols, int32_t rows, float* y,
                                      void* stream) {
  const int64_t total = B * cols * 5;
  if (total == 0) return MURAL_OK;
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, STREAM, cat, E, B, cols, rows, y);
  CHECK_LAUNCH();
}

// dE [rows][5] zeroed by the caller
extern "C" int mural_op_embedding_bwd(const int64_t* cat, const float* dy, int64_t B, int32_t cols, int32_t rows, float* dE,
                                      void* stream) {
  const int64_t total = B * cols * 5;
  if (total == 0) return MURAL_OK;
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(grid_for(total, 256, 256)), dim3(256), (size_t)rows * 5 * 4, STREAM, cat, dy, B,
                     cols, rows, dE);
  CHECK_LAUNCH();
}

extern "C" int mural_op_dropout(const float* x, int64_t total, float p, uint64_t seed, const uint64_t* seed_dev, float* y,
                                void* stream) {
  if (total == 0) return MURAL_OK;
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(total)), dim3(256), 0, STREAM, x, total, p, seed, seed_dev, y);
  CHECK_LAUNCH();
}

extern "C" int mural_op_relu_mask(const float* g, const float* ref, int64_t total, float* y, void* stream) {
  if (total == 0) return MURAL_OK;
  hipLaunchKernelGGL(scale_mask_kernel, dim3(grid_for(total)), dim3(256), 0, STREAM, g, ref, total, 0, y);
  CHECK_LAUNCH();
}

extern "C" int mural_op_head_fwd(const float* loc, const float* mid, const float* lar, int64_t B, int32_t nc, float* out,
                                 void* stream) {
  MURAL_REQUIRE(nc >= 1 && nc <= SNV_MAXCLASS, "n_class out of range");
  if (B == 0) return MURAL_OK;
  hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, STREAM, loc, mid, lar, B, nc, out);
  CHECK_LAUNCH();
}

extern "C" int mural_op_head_bwd(const float* loc, const float* mid, const float* lar, const float* dout, int64_t B, int32_t nc,
                                 float* dloc, float* dmid, float* dlar, void* stream) {
  MURAL_REQUIRE(nc >= 1 && nc <= SNV_MAXCLASS, "n_class out of range");
  if (B == 0) return MURAL_OK;
  // (64-thread workgroups: 4096 rows then occupy 64 CUs instead of 16)
  if (nc == 4) hipLaunchKernelGGL(head_bwd_kernel<4>, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, STREAM, loc, mid, lar, dout, B, nc, dloc, dmid, dlar);
  else hipLaunchKernelGGL(head_bwd_kernel<0>, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, STREAM, loc, mid, lar, dout, B, nc, dloc, dmid, dlar);
  CHECK_LAUNCH();
}

// The criterion of the training loops (training.py: nn.CrossEntropyLoss(reduction='sum') on the model output) in two launches instead of
// torch's four-to-six: loss = -sum_i (x[i][y_i] - logsumexp(x[i])), prob = softmax(x) kept for the backward, dx = g (prob - onehot(y)).
// ONE workgroup: the sum is taken in a fixed order (double accumulators), the loss is reproducible.  A label outside [0, nc) makes the
// loss NaN (torch raises a device assert there).
// NC > 0: the class count at compile time -- a thread's rows (B / 1024 of them, four at a time) are requested before the first one is
// used, a row's logits stay in registers and its probabilities are stored once (the run-time form reads a logit three times and makes
// one dependent round trip to memory per row: 17 us for 4096 x 4 logits against 6).  Same operations in the same order per row and
// the same order of the sum: the two forms give the same bits.
template <int NC>
__global__ __launch_bounds__(1024) void ce_sum_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ y, int64_t B, int nc_rt,
                                                          float* __restrict__ prob, float* __restrict__ loss) {
  __shared__ double red[16];
  double acc = 0.0;
  if constexpr (NC > 0) {
    constexpr int R = 4;
    for (int64_t i0 = threadIdx.x; i0 < B; i0 += 1024 * R) {
      float r[R][NC];
      int64_t t[R];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int64_t i = i0 + 1024 * u;
        const bool ok = i < B;
#pragma unroll
        for (int k = 0; k < NC; ++k) r[u][k] = ok ? x[i * NC + k] : 0.f;
        t[u] = ok ? y[i] : 0;
      }
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int64_t i = i0 + 1024 * u;
        if (i >= B) break;
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < NC; ++k) m = fmaxf(m, r[u][k]);
        float e[NC], s = 0.f;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          e[k] = expf(r[u][k] - m);
          s += e[k];
        }
        const float lse = m + logf(s);
        const float inv = 1.f / s;
        float rt = 0.f;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          prob[i * NC + k] = e[k] * inv;
          rt = t[u] == k ? r[u][k] : rt;
        }
        acc += (t[u] >= 0 && t[u] < NC) ? (double)(lse - rt) : __longlong_as_double(0x7ff8000000000000LL);      // (NaN by its bits: this file is built with -fno-honor-nans)
      }
    }
  } else {
    const int nc = nc_rt;
    for (int64_t i = threadIdx.x; i < B; i += 1024) {
      const float* r = x + i * nc;
      float m = -INFINITY;
      for (int k = 0; k < nc; ++k) m = fmaxf(m, r[k]);
      float s = 0.f;
      for (int k = 0; k < nc; ++k) {      // (each exponential once: parked in prob, scaled below)
        const float e = expf(r[k] - m);
        prob[i * nc + k] = e;
        s += e;
      }
      const float lse = m + logf(s);
      const float inv = 1.f / s;
      for (int k = 0; k < nc; ++k) prob[i * nc + k] *= inv;
      const int64_t t = y[i];
      acc += (t >= 0 && t < nc) ? (double)(lse - r[t]) : __longlong_as_double(0x7ff8000000000000LL);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += red[w];
    loss[0] = (float)t;
  }
}
__global__ void ce_sum_bwd_kernel(const float* __restrict__ prob, const int64_t* __restrict__ y, const float* __restrict__ g, int64_t total,
                                  int nc, float* __restrict__ dx) {
  const float gv = g[0];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nc;
    const int k = (int)(i - row * nc);
    dx[i] = gv * (prob[i] - (y[row] == k ? 1.f : 0.f));
  }
}
extern "C" int mural_op_ce_sum_fwd(const float* x, const int64_t* y, int64_t B, int32_t nc, float* prob, float* loss, void* stream) {
  MURAL_REQUIRE(x && y && prob && loss && nc >= 1 && B >= 0, "ce_sum_fwd: bad arguments");
  if (nc == 4) hipLaunchKernelGGL(ce_sum_fwd_kernel<4>, dim3(1), dim3(1024), 0, STREAM, x, y, B, nc, prob, loss);
  else hipLaunchKernelGGL(ce_sum_fwd_kernel<0>, dim3(1), dim3(1024), 0, STREAM, x, y, B, nc, prob, loss);
  CHECK_LAUNCH();
}
extern "C" int mural_op_ce_sum_bwd(const float* prob, const int64_t* y, const float* g, int64_t B, int32_t nc, float* dx, void* stream) {
  MURAL_REQUIRE(prob && y && g && dx && nc >= 1 && B >= 0, "ce_sum_bwd: bad arguments");
  if (B == 0) return MURAL_OK;
  hipLaunchKernelGGL(ce_sum_bwd_kernel, dim3(grid_for(B * nc)), dim3(256), 0, STREAM, prob, y, g, B * nc, nc, dx);
  CHECK_LAUNCH();
}

// torch.optim.Adam.step() (training.py:346-350 builds it, :432 steps it; amsgrad / maximize off) over flat buffers with ONE slot layout:
// parameters, gradients and the two moment buffers of every parameter at the same offsets (zero in the padding: it stays zero).  One
// elementwise launch instead of torch's multi-tensor sequence (a step-counter foreach + three launches over ~150 tensor quadruples: 40 us
// at the end of every step with nothing beside them).  The update is torch's fused kernel's, in float:
//   g' = g + weight_decay p;  m = m + (g' - m)(1 - beta1);  v = beta2 v + (1 - beta2) g' g';
//   p = p - (lr / (1 - beta1^t)) m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// (the bias corrections come from the host in double, rounded once).
__global__ __launch_bounds__(256) void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n4, float w1, float beta2, float w2, float step_size,
                                                        float bc2_sqrt, float eps, float wd) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pv = reinterpret_cast<float4*>(p)[i], mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float* pp = &pv.x; float* mp = &mv.x; float* vp = &vv.x;
    const float* gp = &gv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = gp[k] + wd * pp[k];
      mp[k] = mp[k] + (gk - mp[k]) * w1;
      vp[k] = beta2 * vp[k] + w2 * gk * gk;
      const float denom = sqrtf(vp[k]) / bc2_sqrt + eps;
      pp[k] = pp[k] - step_size * mp[k] / denom;
    }
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
}
extern "C" int mural_op_adam_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                                  double beta2, double eps, double weight_decay, int64_t step, void* stream) {
  MURAL_REQUIRE(param && grad && exp_avg && exp_avg_sq && n >= 0 && (n & 3) == 0, "adam_flat: NULL buffer or a length that is no multiple of 4");
  MURAL_REQUIRE(step >= 1 && lr >= 0 && beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0 && weight_decay >= 0,
                "adam_flat: step >= 1, lr >= 0, 0 <= beta < 1, eps >= 0, weight_decay >= 0");
  MURAL_REQUIRE(((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
                  reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0, "adam_flat: buffers must be 16-byte aligned");
  if (n == 0) return MURAL_OK;
  const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
  hipLaunchKernelGGL(adam_flat_kernel, dim3(grid_for(n / 4)), dim3(256), 0, STREAM, param, grad, exp_avg, exp_avg_sq, n / 4, (float)(1.0 - beta1),
                     (float)beta2, (float)(1.0 - beta2), (float)(lr / bc1), (float)std::sqrt(bc2), (float)eps, (float)weight_decay);
  CHECK_LAUNCH();
}

// torch.nn.utils.clip_grad_norm_(params, max_norm) (training.py:430) over ONE flat gradient buffer (zero between the parameters' slots) in
// two launches: sums of squares per workgroup (fixed order inside a workgroup and over the workgroups: reproducible), then every workgroup
// adds the CLIP_WGS partial sums in the same order, forms coef = min(max_norm / (norm + 1e-6), 1) -- torch's rule -- and scales its share;
// workgroup 0 writes the norm.  (torch's route: one single-workgroup reduction of 20 us and five scalar launches.)
constexpr int CLIP_WGS = 64;
__global__ __launch_bounds__(1024) void clip_norm_partial_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ part) {
  __shared__ double sh[16];
  const int64_t per = ((n + CLIP_WGS - 1) / CLIP_WGS + 3) & ~(int64_t)3;
  const int64_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double s = 0.0;
  for (int64_t i = lo + 4 * (int64_t)threadIdx.x; i < hi; i += 4 * 1024) {
    if (i + 4 <= hi) {
      const float4 v = *reinterpret_cast<const float4*>(g + i);
      s += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    } else {
      for (int64_t j = i; j < hi; ++j) s += (double)g[j] * g[j];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += sh[w];
    part[blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(1024) void clip_norm_scale_kernel(float* __restrict__ g, int64_t n, const double* __restrict__ part, float max_norm,
                                                               float* __restrict__ total) {
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < CLIP_WGS; ++w) t += part[w];
  const float norm = (float)sqrt(t);
  if (blockIdx.x == 0 && threadIdx.x == 0) *total = norm;
  // torch: grads *= clamp(max_norm / (norm + 1e-6), max = 1) -- a NaN norm (a diverged step) makes the coefficient NaN and every
  // gradient with it, an infinite norm makes it 0 (the infinite entries become NaN): that is what makes a diverged step visible, so it is
  // kept.  The NaN test is on the bits: this file is built with -fno-honor-nans, under which `coef >= 1` may be folded either way.
  const float coef = max_norm / (norm + 1e-6f);
  const bool coef_nan = (__float_as_uint(coef) & 0x7fffffffu) > 0x7f800000u;
  if (!coef_nan && coef >= 1.f) return;      // (torch multiplies by 1: the same values)
  const int64_t per = ((n + CLIP_WGS - 1) / CLIP_WGS + 3) & ~(int64_t)3;
  const int64_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  for (int64_t i = lo + 4 * (int64_t)threadIdx.x; i < hi; i += 4 * 1024) {
    if (i + 4 <= hi) {
      float4 v = *reinterpret_cast<float4*>(g + i);
      v.x *= coef; v.y *= coef; v.z *= coef; v.w *= coef;
      *reinterpret_cast<float4*>(g + i) = v;
    } else {
      for (int64_t j = i; j < hi; ++j) g[j] *= coef;
    }
  }
}
extern "C" int mural_op_clip_grad_norm(float* flat, int64_t n, float max_norm, double* scratch64, float* total, void* stream) {
  MURAL_REQUIRE(flat && scratch64 && total && n >= 0, "clip_grad_norm: bad arguments");
  MURAL_REQUIRE((reinterpret_cast<uintptr_t>(flat) & 15) == 0, "clip_grad_norm: the gradient buffer must be 16-byte aligned");
  hipLaunchKernelGGL(clip_norm_partial_kernel, dim3(CLIP_WGS), dim3(1024), 0, STREAM, flat, n, scratch64);
  hipLaunchKernelGGL(clip_norm_scale_kernel, dim3(CLIP_WGS), dim3(1024), 0, STREAM, flat, n, scratch64, max_norm, total);
  CHECK_LAUNCH();
}

namespace mural {
int train_bn2d_apply_dropout(const float* x, int64_t B, int C, int relu, const double* acc, const float* gamma, const float* beta, float eps,
                             float momentum, float* running_mean, float* running_var, float* state, float p, uint64_t seed,
                             const uint64_t* seed_dev, float* y_bn, float* y, hipStream_t stream) {
  const int64_t total = B * C;
  if (total == 0) return MURAL_OK;
  hipLaunchKernelGGL(bn2d_apply_dropout_kernel, dim3(grid_for(total, 256, 1024)), dim3(256), (size_t)C * 8, stream, x, B, C, relu, acc, gamma,
                     beta, eps, momentum, running_mean, running_var, state, p, seed, seed_dev, y_bn, y);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

int train_gmax_relu_bwd(const float* dfeat, const int32_t* arg, const float* c3, int64_t rows, int L, float* dx, hipStream_t stream) {
  if (rows * L == 0) return MURAL_OK;
  hipLaunchKernelGGL(gmax_relu_bwd_kernel, dim3(grid_for(rows * L)), dim3(256), 0, stream, dfeat, arg, c3, rows, L, dx);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}
}  // namespace mural

#include "snv_local_train.h"      // the local branch of the composed training step in three launches per direction
#include "snv_head_train.h"       // a tower's head in two launches per direction

namespace mural { int launch_dense_to_symbols(const float* x, int64_t n, int L, uint8_t* sym, int32_t* status, hipStream_t stream, int bad_code = -1); }
// dense (n,4,L) MuRaL encoding -> 1 symbol per column (status: see mural_snv_forward_dense)
extern "C" int mural_op_dense_to_symbols(const float* x, int64_t n, int32_t L, uint8_t* sym, int32_t* status, void* stream) {
  return mural::launch_dense_to_symbols(x, n, L, sym, status, STREAM);
}

// ------------------------------------------------------------------------------------------- composed layer calls
// One host call per BN -> conv32 layer and direction (the kernels are the ones above / in conv32_mfma.hip; composing them
// here keeps the Python glue at one ctypes transition per layer, which matters once the step is launch-bound).
extern "C" int mural_op_conv32(const float* x, const float* W, const float* bias, float* y, int64_t B, int32_t L, int32_t dgrad,
                               const float* pre_s, const float* pre_t, int32_t pre_relu, int32_t post_relu, const float* res1,
                               const float* res2, int32_t stat_mode, int32_t stat_relu, const float* stat_x,
                               const float* stat_mean, const float* stat_invstd, double* stat_out, void* stream);
extern "C" int mural_op_conv32_bwd(const float* dy, const float* x, const float* W, int64_t B, int32_t L, const float* pre_s,
                                   const float* pre_t, int32_t pre_relu, const float* mean, const float* invstd, float* dW,
                                   float* db, float* dz, double* stat_out, float* part, size_t part_floats, void* stream);

// forward: batch statistics (taken here unless acc already holds them) -> scale / shift / mean / invstd (state: float[4][32],
// kept for the backward) + running statistics -> y = conv32(scale * act(x) + shift) [+ bias] [relu] [+ res1 + res2],
// optionally with the batch sums of act'(y) for the next layer (acc_out)
extern "C" int mural_op_bnconv32_fwd(const float* x, int64_t B, int32_t L, int32_t pre_relu, double* acc, int32_t have_acc,
                                     const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                     float* running_var, float* state, const float* W, const float* bias, int32_t post_relu,
                                     const float* res1, const float* res2, double* acc_out, int32_t out_relu, float* y,
                                     void* stream) {
  if (!have_acc)
    if (int rc = mural_op_bn_stats(x, B, 32, L, pre_relu, acc, stream)) return rc;
  if (int rc = mural_op_bn_finalize(acc, (double)B * L, 32, gamma, beta, eps, momentum, running_mean, running_var, state,
                                    state + 32, state + 64, state + 96, stream))
    return rc;
  return mural_op_conv32(x, W, bias, y, B, L, 0, state, state + 32, pre_relu, post_relu, res1, res2, acc_out ? 1 : 0, out_relu,
                         nullptr, nullptr, nullptr, acc_out, stream);
}

// backward of the same layer: dW, db, then dx = BatchNorm backward of dz (+ add1 + add2), dgamma, dbeta.  dz: scratch [B][32][L]
extern "C" int mural_op_bnconv32_bwd(const float* dy, const float* x, int64_t B, int32_t L, int32_t pre_relu, const float* state,
                                     const float* gamma, const float* W, double* acc, float* part, size_t part_floats, float* dz,
                                     const float* add1, const float* add2, float* dW, float* db, float* dx, float* dgamma,
                                     float* dbeta, void* stream) {
  if (int rc = mural_op_conv32_bwd(dy, x, W, B, L, state, state + 32, pre_relu, state + 64, state + 96, dW, db, dz, acc, part,
                                   part_floats, stream))
    return rc;
  return mural_op_bn_backward(dz, x, B, 32, L, pre_relu, state + 64, state + 96, gamma, acc, 1, add1, add2, dx, dgamma, dbeta,
                              stream);
}

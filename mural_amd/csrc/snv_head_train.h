// The head of a tower inside the composed training step in two launches per direction (included by train_ops.hip; gfx950 / CDNA4).
//
// Reference: MuRaL/model/model_snv.py:386-388, 428-430 -- ... -> ReLU -> max over the columns -> distal_fc = BatchNorm1d(32) -> Dropout ->
// Linear(32, n_class) under model.train() (training.py:424-427).  Per tower these were gmax + bn_stats + bn2d_apply_dropout + linear
// forward and linear (input + weight gradient) + dropout + bn reduce + bn apply + gmax backward: ten launches of a few microseconds each
// on (B, 32) tensors, every one a dependent step of the tower's chain.  Here:
//     forward  1: global max (+ arg-max, ReLU) with the batch sums of the BatchNorm in its epilogue
//              2: BatchNorm (finalised by every workgroup) + dropout + Linear
//     backward 1: d logits x W -> dropout mask, the BatchNorm's backward sums
//              2: BatchNorm-backward apply -> scatter through the arg-max and the ReLU mask
// and the Linear's weight gradient (no consumer inside the step) leaves the chain: the caller launches it last.
#pragma once

#include "mfma_tile.h"

namespace mural {
namespace headtrain {

constexpr int HC = 32;      // channels of a tower

// ---- forward 1: feat[b][c] = relu?(max_l x[b][l][c]), arg; batch sums of feat, feat^2
__global__ __launch_bounds__(256) void hd_gmax_stats_kernel(const float* __restrict__ x, int64_t B, int L, int relu, float* __restrict__ feat,
                                                            int32_t* __restrict__ arg, double* __restrict__ acc) {
  __shared__ float red[2][32][HC + 1];
  const int tid = threadIdx.x, chunk = tid & 7, rsub = tid >> 3;
  f32x4 s1 = splat(0.f), s2 = splat(0.f);
  for (int64_t b = (int64_t)blockIdx.x * 32 + rsub; b < B; b += (int64_t)gridDim.x * 32) {
    f32x4 m = splat(-INFINITY);
    int am[4] = {0, 0, 0, 0};
    for (int l = 0; l < L; ++l) {
      const f32x4 v = ld4(x + ((size_t)(b * L + l)) * HC + 4 * chunk);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (v[q] > m[q]) {
          m[q] = v[q];
          am[q] = l;
        }
    }
    if (relu) m = max4(m, splat(0.f));
    st4(feat + (size_t)b * HC + 4 * chunk, m);
    int32_t* ap = arg + (size_t)b * HC + 4 * chunk;
    ap[0] = am[0]; ap[1] = am[1]; ap[2] = am[2]; ap[3] = am[3];
    s1 += m;
    s2 += m * m;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    red[0][rsub][4 * chunk + q] = s1[q];
    red[1][rsub][4 * chunk + q] = s2[q];
  }
  __syncthreads();
  if (tid < 2 * HC) {
    const int which = tid >> 5, c = tid & 31;
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) t += red[which][r][c];
    atomicAdd(&acc[((size_t)(blockIdx.x % MURAL_BN_SLOTS) * 2 + which) * HC + c], (double)t);
  }
}

// ---- forward 2: fd = dropout(BatchNorm(feat)), logits = fd W^T + bias; lane = channel, a half-wave = a row
__global__ __launch_bounds__(256) void hd_bn_drop_fc_kernel(const float* __restrict__ feat, int64_t B, const double* __restrict__ acc,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                            float* __restrict__ state, float p, uint64_t seed, const uint64_t* __restrict__ seed_dev,
                                                            float* __restrict__ fd, const float* __restrict__ W, const float* __restrict__ bias, int nc,
                                                            float* __restrict__ logits) {
  __shared__ float cst[2][HC];
  const int tid = threadIdx.x;
  if (tid < HC) {
    const int c = tid;
    const double n = (double)B;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < MURAL_BN_SLOTS; ++k) {
      s1 += acc[((size_t)k * 2 + 0) * HC + c];
      s2 += acc[((size_t)k * 2 + 1) * HC + c];
    }
    const double mean = s1 / n;
    double var = s2 / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float sc = (float)(gamma[c] * invstd), sh = (float)(beta[c] - mean * gamma[c] * invstd);
    cst[0][c] = sc;
    cst[1][c] = sh;
    if (blockIdx.x == 0) {
      state[c] = sc;
      state[HC + c] = sh;
      state[2 * HC + c] = (float)mean;
      state[3 * HC + c] = (float)invstd;
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
  const int c = tid & 31;
  const float sc = cst[0][c], sh = cst[1][c];
  float w[SNV_MAXCLASS];
#pragma unroll
  for (int k = 0; k < SNV_MAXCLASS; ++k) w[k] = k < nc ? W[k * HC + c] : 0.f;
  for (int64_t b = (int64_t)blockIdx.x * 8 + (tid >> 5); b < B; b += (int64_t)gridDim.x * 8) {
    const int64_t i = b * HC + c;
    float v = fmaf(sc, feat[i], sh);
    if (p > 0.f) {
      const uint64_t r = mix64(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1));
      const float u = (float)(r >> 40) * (1.f / 16777216.f);
      v = u >= p ? v * keep_scale : 0.f;
    }
    fd[i] = v;
#pragma unroll
    for (int k = 0; k < SNV_MAXCLASS; ++k) {
      if (k >= nc) break;
      float t = w[k] * v;
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) t += __shfl_xor(t, off);
      if (c == 0) logits[b * nc + k] = t + bias[k];
    }
  }
}

// ---- backward 1: dd = dropout mask (d logits W); batch sums of dd, dd * xhat
__global__ __launch_bounds__(256) void hd_bwd1_kernel(const float* __restrict__ dlogits, const float* __restrict__ W, int nc, int64_t B,
                                                      const float* __restrict__ feat, const float* __restrict__ state, float p, uint64_t seed,
                                                      const uint64_t* __restrict__ seed_dev, float* __restrict__ dd, double* __restrict__ acc) {
  __shared__ float red[2][8][HC + 1];
  const int tid = threadIdx.x, c = tid & 31, rsub = tid >> 5;
  if (seed_dev) seed += *seed_dev;
  const float keep_scale = 1.f / (1.f - p);
  const float mu = state[2 * HC + c], is = state[3 * HC + c];
  float w[SNV_MAXCLASS];
#pragma unroll
  for (int k = 0; k < SNV_MAXCLASS; ++k) w[k] = k < nc ? W[k * HC + c] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  for (int64_t b = (int64_t)blockIdx.x * 8 + rsub; b < B; b += (int64_t)gridDim.x * 8) {
    float d = 0.f;
#pragma unroll
    for (int k = 0; k < SNV_MAXCLASS; ++k)
      if (k < nc) d = fmaf(dlogits[b * nc + k], w[k], d);
    const int64_t i = b * HC + c;
    if (p > 0.f) {
      const uint64_t r = mix64(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1));
      const float u = (float)(r >> 40) * (1.f / 16777216.f);
      d = u >= p ? d * keep_scale : 0.f;
    }
    dd[i] = d;
    s1 += d;
    s2 += d * ((feat[i] - mu) * is);
  }
  red[0][rsub][c] = s1;
  red[1][rsub][c] = s2;
  __syncthreads();
  if (tid < 2 * HC) {
    const int which = tid >> 5, cc = tid & 31;
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += red[which][r][cc];
    atomicAdd(&acc[((size_t)(blockIdx.x % MURAL_BN_SLOTS) * 2 + which) * HC + cc], (double)t);
  }
}

// ---- backward 2: d feat = gamma invstd (dd - mean(dd) - xhat mean(dd xhat)); dx[b][l][c] = (l == arg && c3 > 0) ? d feat : 0
__global__ __launch_bounds__(256) void hd_bwd2_kernel(const float* __restrict__ dd, const float* __restrict__ feat, const float* __restrict__ state,
                                                      const float* __restrict__ gamma, const double* __restrict__ acc, int64_t B, int L,
                                                      const int32_t* __restrict__ arg, const float* __restrict__ c3, float* __restrict__ dx,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float cst[5][HC];
  const int tid = threadIdx.x;
  if (tid < HC) {
    const int c = tid;
    const double n = (double)B;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < MURAL_BN_SLOTS; ++k) {
      s1 += acc[((size_t)k * 2 + 0) * HC + c];
      s2 += acc[((size_t)k * 2 + 1) * HC + c];
    }
    cst[0][c] = gamma[c] * state[3 * HC + c];
    cst[1][c] = (float)(s1 / n);
    cst[2][c] = (float)(s2 / n);
    cst[3][c] = state[2 * HC + c];
    cst[4][c] = state[3 * HC + c];
    if (blockIdx.x == 0) {
      dgamma[c] = (float)s2;
      dbeta[c] = (float)s1;
    }
  }
  __syncthreads();
  const int chunk = tid & 7;
  const f32x4 k0 = ld4(&cst[0][4 * chunk]), m1 = ld4(&cst[1][4 * chunk]), m2 = ld4(&cst[2][4 * chunk]), mu = ld4(&cst[3][4 * chunk]),
              is = ld4(&cst[4][4 * chunk]);
  for (int64_t b = (int64_t)blockIdx.x * 32 + (tid >> 3); b < B; b += (int64_t)gridDim.x * 32) {
    const f32x4 d = ld4(dd + (size_t)b * HC + 4 * chunk), f = ld4(feat + (size_t)b * HC + 4 * chunk);
    const int32_t* ap = arg + (size_t)b * HC + 4 * chunk;
    const int a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
    f32x4 g;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float xh = (f[q] - mu[q]) * is[q];
      g[q] = k0[q] * (d[q] - m1[q] - xh * m2[q]);
    }
    for (int l = 0; l < L; ++l) {
      const size_t o = ((size_t)(b * L + l)) * HC + 4 * chunk;
      const f32x4 v = ld4(c3 + o);
      f32x4 out;
      out[0] = (a0 == l && v[0] > 0.f) ? g[0] : 0.f;
      out[1] = (a1 == l && v[1] > 0.f) ? g[1] : 0.f;
      out[2] = (a2 == l && v[2] > 0.f) ? g[2] : 0.f;
      out[3] = (a3 == l && v[3] > 0.f) ? g[3] : 0.f;
      st4(dx + o, out);
    }
  }
}

}  // namespace headtrain

bool head_train_fused_ok(int nc) {
  const char* e = dev_env("MURAL_TRAIN_HEAD_OPS");
  return !(e && atoi(e) != 0) && nc >= 1 && nc <= SNV_MAXCLASS;
}

// forward of a tower's head: c3 [B][L][32] (raw conv output) -> feat, arg, fd, logits; acc zeroed by the caller
int head_train_fwd(const float* c3, int64_t B, int L, float* feat, int32_t* arg, double* acc, const float* gamma, const float* beta, float eps,
                   float momentum, float* running_mean, float* running_var, float* state, float p, uint64_t seed, const uint64_t* seed_dev,
                   float* fd, const float* W, const float* bias, int nc, float* logits, hipStream_t stream) {
  using namespace headtrain;
  if (B == 0) return MURAL_OK;
  const unsigned g1 = (unsigned)std::min<int64_t>((B + 31) / 32, 1024), g2 = (unsigned)std::min<int64_t>((B + 7) / 8, 2048);
  hipLaunchKernelGGL(hd_gmax_stats_kernel, dim3(g1), dim3(256), 0, stream, c3, B, L, 1, feat, arg, acc);
  hipLaunchKernelGGL(hd_bn_drop_fc_kernel, dim3(g2), dim3(256), 0, stream, feat, B, acc, gamma, beta, eps, momentum, running_mean, running_var, state,
                     p, seed, seed_dev, fd, W, bias, nc, logits);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

// backward: d logits -> dx [B][L][32] (gradient of the raw conv output), dgamma / dbeta of the BatchNorm; dd: [B][32] scratch; the Linear's
// weight gradient is head_train_wgrad's (any time later on the stream)
int head_train_bwd(const float* dlogits, const float* W, int nc, int64_t B, int L, const float* feat, const float* state, const float* gamma,
                   float p, uint64_t seed, const uint64_t* seed_dev, float* dd, double* acc, const int32_t* arg, const float* c3, float* dx,
                   float* dgamma, float* dbeta, hipStream_t stream) {
  using namespace headtrain;
  if (B == 0) return MURAL_OK;
  const unsigned g1 = (unsigned)std::min<int64_t>((B + 7) / 8, 2048), g2 = (unsigned)std::min<int64_t>((B + 31) / 32, 2048);
  hipLaunchKernelGGL(hd_bwd1_kernel, dim3(g1), dim3(256), 0, stream, dlogits, W, nc, B, feat, state, p, seed, seed_dev, dd, acc);
  hipLaunchKernelGGL(hd_bwd2_kernel, dim3(g2), dim3(256), 0, stream, dd, feat, state, gamma, acc, B, L, arg, c3, dx, dgamma, dbeta);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

int head_train_wgrad(const float* dlogits, const float* fd, int64_t B, int nc, float* dW, float* db, hipStream_t stream) {
  if (B == 0) return MURAL_OK;
  hipLaunchKernelGGL(linear_wgrad_mfma_kernel, dim3((unsigned)(((nc + 15) / 16) * ((headtrain::HC + 15) / 16))), dim3(1024), 0, stream, dlogits, fd, B,
                     headtrain::HC, nc, dW, db);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

}  // namespace mural

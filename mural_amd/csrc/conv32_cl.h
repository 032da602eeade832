// Pieces shared by the channel-last conv kernels of the composed SNV training step (conv32_cl.hip: workgroup tiles; conv32_wave.hip:
// wave-private units): the BatchNorm finalisation every conv launch runs in its prologue and the filter fragments in the k order of
// the swizzled [column][32 channels] LDS image (mfma_tile.h).  Reference: nn.BatchNorm1d / nn.Conv1d of MuRaL/model/model_snv.py:350-430
// under model.train() (training.py:424).
#pragma once
#include "snv_tower_conv.h"

namespace mural {

constexpr int CL_C = 32;

// the same fold of the batch sums into scale / shift / state as conv32_mfma.hip (256 threads, eight loads each)
struct ClFin {
  const double* acc;
  double n;
  const float* gamma;
  const float* beta;
  float eps, momentum;
  float* running_mean;
  float* running_var;
  float* state;
};

__device__ __forceinline__ void cl_finalize(const ClFin& f, float* aux /* scale | beta | mean */, double* red, int tid) {
  if (tid < 256) {      // (workgroups of 256 or 512 threads)
    const int c = tid & 31, grp = tid >> 5;
    double v[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[2 * q] = f.acc[((size_t)(4 * grp + q) * 2 + 0) * CL_C + c];
      v[2 * q + 1] = f.acc[((size_t)(4 * grp + q) * 2 + 1) * CL_C + c];
    }
    red[(grp * 2 + 0) * CL_C + c] = (v[0] + v[2]) + (v[4] + v[6]);
    red[(grp * 2 + 1) * CL_C + c] = (v[1] + v[3]) + (v[5] + v[7]);
  }
  __syncthreads();
  if (tid < CL_C) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int g = 0; g < MURAL_BN_SLOTS / 4; ++g) {
      s1 += red[(g * 2 + 0) * CL_C + tid];
      s2 += red[(g * 2 + 1) * CL_C + tid];
    }
    const double mean = s1 / f.n;
    double var = s2 / f.n - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)f.eps);
    // the affine map is applied centred, beta + scale * (v - mean): folding the mean into a shift (beta - mean * scale) costs one
    // instruction less per element and 2-3 x the round-off (two terms of the size of the mean cancel), which at batch 4096 shows up
    // as extra ReLU-mask and arg-max flips in the backward
    const float sc = (float)(f.gamma[tid] * invstd);
    aux[tid] = sc;
    aux[CL_C + tid] = f.beta[tid];
    aux[2 * CL_C + tid] = (float)mean;
    if (blockIdx.x == 0) {
      f.state[tid] = sc;
      f.state[CL_C + tid] = f.beta[tid];
      f.state[2 * CL_C + tid] = (float)mean;
      f.state[3 * CL_C + tid] = (float)invstd;
      if (f.running_mean) {
        const double unbiased = f.n > 1.0 ? var * f.n / (f.n - 1.0) : var;
        f.running_mean[tid] = (float)((1.0 - f.momentum) * f.running_mean[tid] + f.momentum * mean);
        f.running_var[tid] = (float)((1.0 - f.momentum) * f.running_var[tid] + f.momentum * unbiased);
      }
    }
  }
  __syncthreads();
}

// filter fragments in the order conv_layer / mfma_tap expect: k-step s = 8 tap + 4 half + q <-> input channel 16 half + 4 kk + q,
// output channel 16 mb + n16; dgrad: the transposed, tap-flipped filter
__device__ __forceinline__ void cl_frags(const float* __restrict__ W, int dgrad, int mb, int n16, int kk, float (&a)[SNV_KSTEPS]) {
#pragma unroll
  for (int s = 0; s < SNV_KSTEPS; ++s) {
    const int t = s / 8, h = (s % 8) / 4, q = s % 4;
    const int cin = 16 * h + 4 * kk + q, cout = 16 * mb + n16;
    a[s] = W[dgrad ? (cin * CL_C + cout) * 3 + (2 - t) : (cout * CL_C + cin) * 3 + t];
  }
}

}  // namespace mural

// Post-head calibration on the device: what MuRaL/scripts/run_predict.py:214-225 does to the model
// output on the host, one pass over the (n, n_class) rows, fused into ONE kernel behind the head:
//   F.softmax(pred_y, dim=1)                                                   run_predict.py:214 (float32)
//   FullDirichletCalibrator.predict_proba: softmax(W . [log(clip(p, tiny, 1 - tiny)); 1])
//                                              dirichlet_python/dirichletcal/calib/fulldirichlet.py:78-80, utils.py:5-7 (clip and
//                                              log in the INPUT's dtype = float32), calib/multinomial.py:60-64, :235-244 (float64)
//   poisson_calibrate                          MuRaL/model/calibration.py:10-23
//   apply_scaling                              MuRaL/scripts/scaling.py:10-28
// One thread per row (n_class <= 16: the 4 x 5 / 8 x 9 weight matrix sits in registers / scalar loads); HBM-bound: 4 k bytes in,
// 8 k bytes out per row.
#include <cfloat>

#include "common.h"

namespace mural {

constexpr int CAL_MAXCLASS = 16;

struct CalArgs {
  const float* in;        // [n][k] model output (log-probabilities / scores) or probabilities
  int64_t n;
  int k;
  int in_is_prob;
  const double* w;        // [k][k+1] or nullptr
  int poisson;
  double scale;           // 0: no scaling
  void* out;
  int out_f64;
};

template <int K>
__global__ __launch_bounds__(256) void calibrate_rows_kernel(const CalArgs a) {
  const int k = K > 0 ? K : a.k;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < a.n; row += (int64_t)gridDim.x * blockDim.x) {
    float pf[K > 0 ? K : CAL_MAXCLASS];
    const float* src = a.in + row * k;
#pragma unroll
    for (int c = 0; c < k; ++c) pf[c] = src[c];
    if (!a.in_is_prob) {                       // float32 softmax, as torch evaluates F.softmax on the float32 output
      float m = pf[0];
#pragma unroll
      for (int c = 1; c < k; ++c) m = fmaxf(m, pf[c]);
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < k; ++c) {
        pf[c] = expf(pf[c] - m);
        sum += pf[c];
      }
#pragma unroll
      for (int c = 0; c < k; ++c) pf[c] = pf[c] / sum;
    }
    double p[K > 0 ? K : CAL_MAXCLASS];
    if (a.w) {
      double s[K > 0 ? K : CAL_MAXCLASS];
#pragma unroll
      for (int c = 0; c < k; ++c) {            // clip_for_log in float32: 1 - tiny rounds to 1
        const float cl = fminf(fmaxf(pf[c], FLT_MIN), 1.0f);
        s[c] = (double)logf(cl);
      }
      double z[K > 0 ? K : CAL_MAXCLASS];
      double zmax = -INFINITY;
#pragma unroll
      for (int o = 0; o < k; ++o) {
        const double* wr = a.w + (size_t)o * (k + 1);
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < k; ++c) acc += s[c] * wr[c];
        acc += wr[k];
        z[o] = acc;
        zmax = fmax(zmax, acc);
      }
      double sum = 0.0;
#pragma unroll
      for (int o = 0; o < k; ++o) {
        p[o] = exp(z[o] - zmax);
        sum += p[o];
      }
#pragma unroll
      for (int o = 0; o < k; ++o) p[o] = p[o] / sum;
    } else {
#pragma unroll
      for (int c = 0; c < k; ++c) p[c] = (double)pf[c];
    }
    if (a.poisson) {
      const double p0 = fmin(fmax(p[0], 1e-10), 1.0);
      const double lam = -log(p0);
      const double den = 1.0 - p0;
#pragma unroll
      for (int c = 1; c < k; ++c) p[c] = lam * p[c] / den;
      p[0] = 1.0 - lam;
    }
    if (a.scale != 0.0) {
      double rest = 0.0;
#pragma unroll
      for (int c = 1; c < k; ++c) {
        p[c] *= a.scale;
        rest += p[c];
      }
      p[0] = 1.0 - rest;
    }
    if (a.out_f64) {
      double* dst = static_cast<double*>(a.out) + row * k;
#pragma unroll
      for (int c = 0; c < k; ++c) dst[c] = p[c];
    } else {
      float* dst = static_cast<float*>(a.out) + row * k;
#pragma unroll
      for (int c = 0; c < k; ++c) dst[c] = (float)p[c];
    }
  }
}

}  // namespace mural

using namespace mural;

extern "C" int mural_calibrate_rows(const float* in, int64_t n, int32_t n_class, int32_t in_is_prob, const double* dirichlet_w,
                                    int32_t poisson, double scale, void* out, int32_t out_f64, void* stream) {
  MURAL_REQUIRE(n >= 0, "negative row count");
  MURAL_REQUIRE(n_class >= 2 && n_class <= CAL_MAXCLASS, "n_class must be in [2,%d], got %d", CAL_MAXCLASS, n_class);
  if (n == 0) return MURAL_OK;
  MURAL_REQUIRE(in && out, "in/out must not be NULL");
  CalArgs a{in, n, n_class, in_is_prob, dirichlet_w, poisson, scale, out, out_f64};
  const int64_t want = (n + 255) / 256;
  const dim3 grid((unsigned)(want < 4096 ? want : 4096)), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (n_class) {
    case 2: hipLaunchKernelGGL(calibrate_rows_kernel<2>, grid, block, 0, s, a); break;
    case 4: hipLaunchKernelGGL(calibrate_rows_kernel<4>, grid, block, 0, s, a); break;
    case 8: hipLaunchKernelGGL(calibrate_rows_kernel<8>, grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL(calibrate_rows_kernel<0>, grid, block, 0, s, a); break;
  }
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

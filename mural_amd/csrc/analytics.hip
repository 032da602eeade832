// Validation-epoch analytics of the training loop as device-side segmented reductions.
// Counterparts in the reference (pandas / per-row Python, the dominant cost of an epoch once the train step is fast):
//   freq_kmer_comp_multi        MuRaL/evaluation/evaluation.py:48-67     group-by flanking k-mer -> observed / predicted rate
//   corr_calc_sub               MuRaL/evaluation/evaluation.py:124-193   group-by genomic window  -> observed / predicted rate
//   Evaluator.evaluate_regional_score                       :544-587     group-by (region of rows, k-mer)
//   ECELoss / ClasswiseECELoss / BrierScore / CE(mean)      :209-290, calibrate_prob :340-358
//   MultinomialRegression fit (loss / gradient / Hessian)   dirichlet_python/dirichletcal/calib/multinomial.py:153-172
// Every kernel reduces rows into a small float64 table; the correlations / Newton step on those tables are host work.
// All of this is HBM-bound streaming (one pass over n x (n_class + a few) values); float64 accumulation throughout.
#include "common.h"

namespace mural {
namespace {

constexpr int AN_THREADS = 256;
constexpr int AN_LDS_DOUBLES = 4096;      // 32 KB privatised table

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// key = base-5 number of the 2d flanking order-1 codes (us_d .. us_1, ds_1 .. ds_d), optionally offset by the row's region
__global__ __launch_bounds__(AN_THREADS) void kmer_keys_kernel(const int64_t* __restrict__ codes, int64_t n, int ncols, int left0,
                                                               int right0, int d, int64_t region_size, int64_t n_regions,
                                                               int32_t groups, int32_t* __restrict__ keys,
                                                               int32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * AN_THREADS + threadIdx.x;
  if (i >= n) return;
  const int64_t* row = codes + i * ncols;
  int32_t key = 0;
  bool bad = false;
  for (int j = 0; j < d; ++j) {
    const int64_t c = row[left0 + j];
    bad |= (c < 0 || c > 4);
    key = key * 5 + (int32_t)c;
  }
  for (int j = 0; j < d; ++j) {
    const int64_t c = row[right0 + j];
    bad |= (c < 0 || c > 4);
    key = key * 5 + (int32_t)c;
  }
  if (bad) {
    atomicOr(status, 1);
    keys[i] = -1;
    return;
  }
  if (region_size > 0) {
    const int64_t reg = i / region_size;
    key = reg < n_regions ? (int32_t)(reg * groups) + key : -1;
  }
  keys[i] = key;
}

__global__ __launch_bounds__(AN_THREADS) void window_keys_kernel(const int32_t* __restrict__ chrom_id,
                                                                 const int64_t* __restrict__ start, int64_t n, int64_t window,
                                                                 const int64_t* __restrict__ chrom_base, int32_t n_chrom,
                                                                 int32_t* __restrict__ keys, int32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * AN_THREADS + threadIdx.x;
  if (i >= n) return;
  const int32_t c = chrom_id[i];
  const int64_t s = start[i];
  if (c < 0 || c >= n_chrom || s < 0) {
    atomicOr(status, 1);
    keys[i] = -1;
    return;
  }
  keys[i] = (int32_t)(chrom_base[c] + s / window);
}

// table[g] = { rows, rows with label == c (c < nc), sum of prob[:, c] (c < nc) }
template <typename P>
__global__ __launch_bounds__(AN_THREADS) void group_obs_pred_kernel(const int32_t* __restrict__ keys,
                                                                    const int32_t* __restrict__ label,
                                                                    const P* __restrict__ prob, int64_t n, int nc,
                                                                    int32_t n_groups, int use_lds, double* __restrict__ table,
                                                                    int32_t* __restrict__ status) {
  __shared__ double lds[AN_LDS_DOUBLES];
  const int stride = 1 + 2 * nc;
  const int cells = n_groups * stride;
  if (use_lds) {
    for (int t = threadIdx.x; t < cells; t += AN_THREADS) lds[t] = 0.0;
    __syncthreads();
  }
  double* dst = use_lds ? lds : table;
  const int64_t step = (int64_t)gridDim.x * AN_THREADS;
  const int64_t n_round = (n + AN_THREADS - 1) / AN_THREADS * AN_THREADS;      // whole waves stay converged for the shuffles
  for (int64_t i = (int64_t)blockIdx.x * AN_THREADS + threadIdx.x; i < n_round; i += step) {
    int32_t key = -1, lab = -1;
    if (i < n) {
      key = keys[i];
      lab = label[i];
      if (key >= n_groups || lab < 0 || lab >= nc) {
        atomicOr(status, 2);
        key = -1;
      }
    }
    const int32_t k0 = __shfl(key, 0, 64);
    if (__all(key == k0)) {
      // sorted inputs (regional windows, row regions): one atomic per wave instead of 64 on one address
      if (k0 < 0) continue;
      const double cnt = 64.0;
      double* g = dst + (int64_t)k0 * stride;
      if ((threadIdx.x & 63) == 0) atomicAdd(&g[0], cnt);
      for (int c = 0; c < nc; ++c) {
        const double o = wave_sum(lab == c ? 1.0 : 0.0);
        const double p = wave_sum((double)prob[i * nc + c]);
        if ((threadIdx.x & 63) == 0) {
          if (o != 0.0) atomicAdd(&g[1 + c], o);
          atomicAdd(&g[1 + nc + c], p);
        }
      }
    } else if (key >= 0) {
      double* g = dst + (int64_t)key * stride;
      atomicAdd(&g[0], 1.0);
      atomicAdd(&g[1 + lab], 1.0);
      for (int c = 0; c < nc; ++c) atomicAdd(&g[1 + nc + c], (double)prob[i * nc + c]);
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int t = threadIdx.x; t < cells; t += AN_THREADS)
      if (lds[t] != 0.0) atomicAdd(&table[t], lds[t]);
  }
}

// out = { nll sum, brier sum, top-label bins [nb][3] = (rows, confidence sum, correct sum),
//         class bins [nc][nb][3] = (rows, confidence sum, rows whose label is the class) }
template <typename P>
__global__ __launch_bounds__(AN_THREADS) void calib_metrics_kernel(const P* __restrict__ prob, const int32_t* __restrict__ label,
                                                                   int64_t n, int nc, int nb, const float* __restrict__ bounds,
                                                                   double* __restrict__ out, int32_t* __restrict__ status) {
  __shared__ double lds[AN_LDS_DOUBLES];
  const int cells = 2 + 3 * nb * (1 + nc);
  for (int t = threadIdx.x; t < cells; t += AN_THREADS) lds[t] = 0.0;
  __syncthreads();
  double nll = 0.0, brier = 0.0;
  const int64_t step = (int64_t)gridDim.x * AN_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * AN_THREADS + threadIdx.x; i < n; i += step) {
    const int lab = label[i];
    if (lab < 0 || lab >= nc) {
      atomicOr(status, 2);
      continue;
    }
    // the reference's pseudo-logits are log(prob) and its scores softmax(log(prob)), evaluated in prob's own precision
    P q[16];
    P m = (P)-INFINITY;
    for (int c = 0; c < nc; ++c) {
      q[c] = (P)log((P)prob[i * nc + c]);
      m = q[c] > m ? q[c] : m;
    }
    P s = (P)0;
    for (int c = 0; c < nc; ++c) {
      q[c] = (P)exp(q[c] - m);
      s += q[c];
    }
    P conf = (P)-1;
    int arg = 0;
    for (int c = 0; c < nc; ++c) {
      q[c] = q[c] / s;
      if (q[c] > conf) {
        conf = q[c];
        arg = c;
      }
      const P dlt = (c == lab ? (P)1 : (P)0) - q[c];
      brier += (double)(dlt * dlt);
    }
    nll -= (double)log(q[lab]);
    for (int c = -1; c < nc; ++c) {
      const P v = c < 0 ? conf : q[c];
      int b = -1;
      for (int t = 0; t < nb; ++t)
        if ((double)v > (double)bounds[t] && (double)v <= (double)bounds[t + 1]) b = t;
      if (b < 0) continue;
      double* cell = lds + 2 + 3 * ((c + 1) * nb + b);
      atomicAdd(&cell[0], 1.0);
      atomicAdd(&cell[1], (double)v);
      const bool hit = c < 0 ? (arg == lab) : (lab == c);
      if (hit) atomicAdd(&cell[2], 1.0);
    }
  }
  atomicAdd(&lds[0], nll);
  atomicAdd(&lds[1], brier);
  __syncthreads();
  for (int t = threadIdx.x; t < cells; t += AN_THREADS)
    if (lds[t] != 0.0) atomicAdd(&out[t], lds[t]);
}

// Multinomial regression on x = [log clip(prob); 1] with weights W [k][k+1]:
//   out = { sum of -log clip(s_y, eps, 1 - eps), d/dW [k][m], d2/dW2 [k*m][k*m] }    (sums over rows; the caller divides by n)
constexpr int FIT_MAX_K = 8;
constexpr int FIT_MAX_M = FIT_MAX_K + 1;
constexpr int FIT_ROWS = 128;
constexpr int FIT_MAX_E = (FIT_MAX_K * FIT_MAX_M * FIT_MAX_K * FIT_MAX_M + AN_THREADS - 1) / AN_THREADS;

template <typename P>
__global__ __launch_bounds__(AN_THREADS) void dirichlet_fit_kernel(const P* __restrict__ prob, const int32_t* __restrict__ label,
                                                                   int64_t n, int k, const double* __restrict__ W, int need_hess,
                                                                   double* __restrict__ out, int32_t* __restrict__ status) {
  __shared__ double sW[FIT_MAX_K * FIT_MAX_M];
  __shared__ double sx[FIT_ROWS][FIT_MAX_M];
  __shared__ double ss[FIT_ROWS][FIT_MAX_K];      // softmax outputs; zeroed for rows without a gradient
  __shared__ double sr[FIT_ROWS][FIT_MAX_K];      // s - onehot(y)
  __shared__ double sloss[AN_THREADS / 64];
  const int m = k + 1, km = k * m;
  for (int t = threadIdx.x; t < km; t += AN_THREADS) sW[t] = W[t];
  const double tiny = sizeof(P) == 4 ? 1.1754943508222875e-38 : 2.2250738585072014e-308;
  const double eps = 2.220446049250313e-16;

  double acc[FIT_MAX_E];
  double gacc = 0.0;         // gradient entry of thread t < km
  double lacc = 0.0;
#pragma unroll
  for (int e = 0; e < FIT_MAX_E; ++e) acc[e] = 0.0;
  const int n_ent = need_hess ? km * km : 0;

  const int64_t tiles = (n + FIT_ROWS - 1) / FIT_ROWS;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    __syncthreads();
    if (threadIdx.x < FIT_ROWS) {
      const int r = threadIdx.x;
      const int64_t i = tile * FIT_ROWS + r;
      bool live = i < n;
      int lab = 0;
      if (live) {
        lab = label[i];
        if (lab < 0 || lab >= k) {
          atomicOr(status, 2);
          live = false;
        }
      }
      if (live) {
        for (int c = 0; c < k; ++c) {
          P p = prob[i * k + c];
          const P lo = (P)tiny, hi = (P)1 - (P)tiny;
          p = p < lo ? lo : (p > hi ? hi : p);
          sx[r][c] = (double)(P)log(p);
        }
        sx[r][k] = 1.0;
        double z[FIT_MAX_K], zmax = -INFINITY;
        for (int j = 0; j < k; ++j) {
          double a = 0.0;
          for (int c = 0; c < m; ++c) a += sW[j * m + c] * sx[r][c];
          z[j] = a;
          zmax = a > zmax ? a : zmax;
        }
        double tot = 0.0;
        for (int j = 0; j < k; ++j) {
          z[j] = exp(z[j] - zmax);
          tot += z[j];
        }
        const double sy = z[lab] / tot;
        const bool clipped = sy < eps || sy > 1.0 - eps;
        lacc -= log(clipped ? (sy < eps ? eps : 1.0 - eps) : sy);
        // d(-log s_y)/dz_j = s_j - [j == y];  d2/dz_j dz_j' = s_j [j == j'] - s_j s_j'   (both vanish where the clip is active)
        for (int j = 0; j < k; ++j) {
          const double s = clipped ? 0.0 : z[j] / tot;
          ss[r][j] = s;
          sr[r][j] = clipped ? 0.0 : s - (j == lab ? 1.0 : 0.0);
        }
      } else {
        for (int c = 0; c < m; ++c) sx[r][c] = 0.0;
        for (int j = 0; j < k; ++j) ss[r][j] = sr[r][j] = 0.0;
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < km) {
      const int j = threadIdx.x / m, c = threadIdx.x % m;
      double a = 0.0;
      for (int r = 0; r < FIT_ROWS; ++r) a += sr[r][j] * sx[r][c];
      gacc += a;
    }
#pragma unroll
    for (int e = 0; e < FIT_MAX_E; ++e) {
      const int idx = threadIdx.x + e * AN_THREADS;
      if (idx < n_ent) {
        const int p = idx / km, q = idx % km;
        const int j = p / m, c = p % m, j2 = q / m, c2 = q % m;
        double a = 0.0;
        for (int r = 0; r < FIT_ROWS; ++r) {
          const double sj = ss[r][j];
          a += (j == j2 ? sj : 0.0) * sx[r][c] * sx[r][c2] - sj * ss[r][j2] * sx[r][c] * sx[r][c2];
        }
        acc[e] += a;
      }
    }
  }
  lacc = wave_sum(lacc);
  if ((threadIdx.x & 63) == 0) sloss[threadIdx.x >> 6] = lacc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double l = 0.0;
    for (int w = 0; w < AN_THREADS / 64; ++w) l += sloss[w];
    atomicAdd(&out[0], l);
  }
  if ((int)threadIdx.x < km) atomicAdd(&out[1 + threadIdx.x], gacc);
#pragma unroll
  for (int e = 0; e < FIT_MAX_E; ++e) {
    const int idx = threadIdx.x + e * AN_THREADS;
    if (idx < n_ent) atomicAdd(&out[1 + km + idx], acc[e]);
  }
}

int grid_for(int64_t n, int per_block, int cap) {
  int64_t g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

}  // namespace
}  // namespace mural

using namespace mural;

extern "C" int mural_eval_kmer_keys(const int64_t* codes, int64_t n, int32_t ncols, int32_t left0, int32_t right0, int32_t d,
                                    int64_t region_size, int64_t n_regions, int32_t* keys, int32_t* status, void* stream) {
  MURAL_REQUIRE(n >= 0 && ncols > 0 && d >= 1 && d <= 6, "kmer_keys: n >= 0, ncols > 0, 1 <= d <= 6 required");
  MURAL_REQUIRE(left0 >= 0 && left0 + d <= ncols && right0 >= 0 && right0 + d <= ncols, "kmer_keys: flank columns outside the table");
  int64_t groups = 1;
  for (int j = 0; j < 2 * d; ++j) groups *= 5;
  MURAL_REQUIRE(region_size >= 0 && n_regions >= 0 && (region_size == 0 || n_regions * groups < (1ll << 31)),
                "kmer_keys: region keys do not fit 31 bits");
  if (n == 0) return MURAL_OK;
  MURAL_REQUIRE(codes && keys && status, "kmer_keys: null pointer");
  hipLaunchKernelGGL(kmer_keys_kernel, dim3((unsigned)((n + AN_THREADS - 1) / AN_THREADS)), dim3(AN_THREADS), 0, (hipStream_t)stream,
                     codes, n, ncols, left0, right0, d, region_size, n_regions, (int32_t)groups, keys, status);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

extern "C" int mural_eval_window_keys(const int32_t* chrom_id, const int64_t* start, int64_t n, int64_t window,
                                      const int64_t* chrom_base, int32_t n_chrom, int32_t* keys, int32_t* status, void* stream) {
  MURAL_REQUIRE(n >= 0 && window > 0 && n_chrom > 0, "window_keys: n >= 0, window > 0, n_chrom > 0 required");
  if (n == 0) return MURAL_OK;
  MURAL_REQUIRE(chrom_id && start && chrom_base && keys && status, "window_keys: null pointer");
  hipLaunchKernelGGL(window_keys_kernel, dim3((unsigned)((n + AN_THREADS - 1) / AN_THREADS)), dim3(AN_THREADS), 0, (hipStream_t)stream,
                     chrom_id, start, n, window, chrom_base, n_chrom, keys, status);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

extern "C" int mural_eval_group_obs_pred(const int32_t* keys, const int32_t* label, const void* prob, int32_t prob_f64, int64_t n,
                                         int32_t n_class, int32_t n_groups, double* table, int32_t* status, void* stream) {
  MURAL_REQUIRE(n >= 0 && n_class >= 1 && n_class <= 16 && n_groups >= 1, "group_obs_pred: bad sizes");
  MURAL_REQUIRE((int64_t)n_groups * (1 + 2 * n_class) < (1ll << 31), "group_obs_pred: table too large");
  if (n == 0) return MURAL_OK;
  MURAL_REQUIRE(keys && label && prob && table && status, "group_obs_pred: null pointer");
  const int use_lds = (int64_t)n_groups * (1 + 2 * n_class) <= AN_LDS_DOUBLES;
  const int grid = grid_for(n, AN_THREADS * 8, 2048);
  if (prob_f64)
    hipLaunchKernelGGL(group_obs_pred_kernel<double>, dim3(grid), dim3(AN_THREADS), 0, (hipStream_t)stream, keys, label,
                       (const double*)prob, n, n_class, n_groups, use_lds, table, status);
  else
    hipLaunchKernelGGL(group_obs_pred_kernel<float>, dim3(grid), dim3(AN_THREADS), 0, (hipStream_t)stream, keys, label,
                       (const float*)prob, n, n_class, n_groups, use_lds, table, status);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

extern "C" int mural_eval_calib_metrics(const void* prob, int32_t prob_f64, const int32_t* label, int64_t n, int32_t n_class,
                                        int32_t n_bins, const float* bounds, double* out, int32_t* status, void* stream) {
  MURAL_REQUIRE(n >= 0 && n_class >= 1 && n_class <= 16 && n_bins >= 1, "calib_metrics: bad sizes");
  MURAL_REQUIRE(2 + 3 * n_bins * (1 + n_class) <= AN_LDS_DOUBLES, "calib_metrics: n_bins * (n_class + 1) too large");
  if (n == 0) return MURAL_OK;
  MURAL_REQUIRE(prob && label && bounds && out && status, "calib_metrics: null pointer");
  const int grid = grid_for(n, AN_THREADS * 8, 1024);
  if (prob_f64)
    hipLaunchKernelGGL(calib_metrics_kernel<double>, dim3(grid), dim3(AN_THREADS), 0, (hipStream_t)stream, (const double*)prob,
                       label, n, n_class, n_bins, bounds, out, status);
  else
    hipLaunchKernelGGL(calib_metrics_kernel<float>, dim3(grid), dim3(AN_THREADS), 0, (hipStream_t)stream, (const float*)prob, label,
                       n, n_class, n_bins, bounds, out, status);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

extern "C" int mural_eval_dirichlet_fit_terms(const void* prob, int32_t prob_f64, const int32_t* label, int64_t n, int32_t n_class,
                                              const double* weights, int32_t need_hessian, double* out, int32_t* status,
                                              void* stream) {
  MURAL_REQUIRE(n >= 0 && n_class >= 2 && n_class <= FIT_MAX_K, "dirichlet_fit_terms: 2 <= n_class <= 8 required");
  if (n == 0) return MURAL_OK;
  MURAL_REQUIRE(prob && label && weights && out && status, "dirichlet_fit_terms: null pointer");
  const int grid = grid_for(n, FIT_ROWS * 4, 512);
  if (prob_f64)
    hipLaunchKernelGGL(dirichlet_fit_kernel<double>, dim3(grid), dim3(AN_THREADS), 0, (hipStream_t)stream, (const double*)prob,
                       label, n, n_class, weights, need_hessian, out, status);
  else
    hipLaunchKernelGGL(dirichlet_fit_kernel<float>, dim3(grid), dim3(AN_THREADS), 0, (hipStream_t)stream, (const float*)prob, label,
                       n, n_class, weights, need_hessian, out, status);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

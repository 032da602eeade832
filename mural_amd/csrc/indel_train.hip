// Training-mode building blocks of the INDEL U-Net (reference MuRaL/model/model_indel.py:6-19, :151-176 run under
// model.train(), MuRaL/training.py:404-450): a general Conv1d (any stride, nearest-neighbour upsampled input) with its
// input- and weight-gradient kernels, and the SiLU / Softplus / ReLU element-wise pair.  BatchNorm (batch statistics),
// Linear, Dropout and the global max come from train_ops.hip.  Tensors are [B][C][L] fp32, weights in torch's
// [Cout][Cin][K] layout.  Direct convolutions on the vector ALU: at 4..48 channels the layers are HBM / latency bound.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "conv1d.h"
#include "mfma_tile.h"

namespace mural {
// conv_wgrad_mfma.hip: the weight gradient as an implicit GEMM on the matrix cores (nonzero return: shape not covered)
int launch_conv_wgrad_mfma(const float* dy, const float* x, float* part, int64_t B, int Cin, int Lin, int Cout, int Lout, int K, int stride,
                           int pad, int up, int max_chunks, int* chunks_out, hipStream_t st);
namespace {

constexpr int IT_THREADS = 256;

// ------------------------------------------------------------------------------------------------ input gradient
// dx[b][ci][j] = sum over the `up` positions l of the upsampled input that read x[j], taps k and output channels co of
//   dy[b][co][(l + pad - k) / stride] * W[co][ci][k]        where (l + pad - k) is a non-negative multiple of stride.
// A thread owns one (b, j) and CG input channels; weights sit in LDS as [co][k][ci] (a wave reads one address: broadcast).
template <int CG>
__global__ __launch_bounds__(IT_THREADS) void conv_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ W,
                                                                float* dx, int64_t rows /* B * Lin */, int Cin,
                                                                int Lin, int Cout, int Lout, int K, int stride, int pad, int up,
                                                                const float* add /* optional, may be dx itself */) {
  extern __shared__ float wl[];                   // [Cout][K][CG]
  const int ci0 = blockIdx.y * CG;
  for (int i = threadIdx.x; i < Cout * K * CG; i += IT_THREADS) {
    const int c = i % CG, r = i / CG, k = r % K, co = r / K;
    wl[i] = (ci0 + c < Cin) ? W[((size_t)co * Cin + ci0 + c) * K + k] : 0.f;
  }
  __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * IT_THREADS + threadIdx.x;
  if (row >= rows) return;
  const int b = (int)(row / Lin), j = (int)(row - (int64_t)b * Lin);
  float acc[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) acc[c] = 0.f;
  const float* dyb = dy + (size_t)b * Cout * Lout;
  for (int u = 0; u < up; ++u) {
    const int l = j * up + u;
    for (int k = 0; k < K; ++k) {
      const int t = l + pad - k;
      if (t < 0 || t % stride != 0) continue;
      const int lo = t / stride;
      if (lo >= Lout) continue;
      // eight gradient loads in flight before the first is used (the channel counts are multiples of 8): a rolled loop waits a
      // global round trip per output channel
      for (int co0 = 0; co0 < Cout; co0 += 8) {
        float g[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) g[q] = co0 + q < Cout ? dyb[(size_t)(co0 + q) * Lout + lo] : 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float* w = wl + ((co0 + q < Cout ? co0 + q : 0) * K + k) * CG;
#pragma unroll
          for (int c = 0; c < CG; ++c) acc[c] = fmaf(g[q], w[c], acc[c]);
        }
      }
    }
  }
  // (+ a gradient arriving at x through another consumer; it may live in dx itself: an element is read and written by one thread.
  // Its CG values are requested together -- a load behind a per-element branch carries a full wait of its own)
  if (add) {
    float e[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) e[c] = ci0 + c < Cin ? add[((size_t)b * Cin + ci0 + c) * Lin + j] : 0.f;
#pragma unroll
    for (int c = 0; c < CG; ++c) acc[c] += e[c];
  }
#pragma unroll
  for (int c = 0; c < CG; ++c)
    if (ci0 + c < Cin) dx[((size_t)b * Cin + ci0 + c) * Lin + j] = acc[c];
}

// The same gradient for the U-Net's strided (stride 4 / 5 / 2, up 1) and upsampled (stride 1, up 2 / 5 / 4) layers with the taps of a
// position gathered first.  stride 1: the `up` positions that read x[j] and the K taps collapse to M = K + up - 1 distinct gradient
// columns lo = j * up + m' + pad - (K - 1), m' < M, each with the SUM of the taps that reach it (built in LDS); up 1: the taps with
// k = (j + pad) mod stride, + stride, ... reach lo = (j + pad - k) / stride.  Either way a position has at most DG_NT (column, weight
// row) pairs (2 at stride 4 / 5, 4 at stride 2, 8 / 10 / 11 at up 2 / 4 / 5): the loads of DG_COB output channels x all pairs are in
// flight together (32 - 48 per thread), Cout / DG_COB dependent rounds instead of the up * K * Cout / 8 of the kernel above
// (210 -> 12 for the 40 -> 48-channel level).
template <int CG, int DG_NT, int DG_COB>
__global__ __launch_bounds__(IT_THREADS) void conv_dgrad_gather_kernel(const float* __restrict__ dy, const float* __restrict__ W,
                                                                       float* dx, int64_t rows /* B * Lin */, int Cin,
                                                                       int Lin, int Cout, int Lout, int K, int stride, int pad, int up,
                                                                       const float* add /* optional, may be dx itself */) {
  extern __shared__ float wl[];                   // [Cout][T][CG]: T = K + up - 1 summed taps (stride 1) or the K taps (up 1)
  const int ci0 = blockIdx.y * CG;
  const bool summed = stride == 1;
  const int T = summed ? K + up - 1 : K;
  for (int i = threadIdx.x; i < Cout * T * CG; i += IT_THREADS) {
    const int c = i % CG, r = i / CG, t = r % T, co = r / T;
    float v = 0.f;
    if (ci0 + c < Cin) {
      const float* w = W + ((size_t)co * Cin + ci0 + c) * K;
      if (summed) {
        const int k0 = K - 1 - t > 0 ? K - 1 - t : 0, k1 = K - 2 - t + up < K - 1 ? K - 2 - t + up : K - 1;
        for (int k = k0; k <= k1; ++k) v += w[k];
      } else {
        v = w[t];
      }
    }
    wl[i] = v;
  }
  __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * IT_THREADS + threadIdx.x;
  if (row >= rows) return;
  const int b = (int)(row / Lin), j = (int)(row - (int64_t)b * Lin);
  int lo[DG_NT], wi[DG_NT];                       // gradient column (-1: none) and weight row of every pair
  if (summed) {
    const int base = j * up + pad - (K - 1);
#pragma unroll
    for (int t = 0; t < DG_NT; ++t) {
      const int l = base + t;
      lo[t] = (t < T && l >= 0 && l < Lout) ? l : -1;
      wi[t] = t < T ? t : 0;
    }
  } else {
    const int r = (j + pad) % stride, top = (j + pad - r) / stride;
#pragma unroll
    for (int t = 0; t < DG_NT; ++t) {
      const int k = r + t * stride, l = top - t;
      lo[t] = (k < K && l >= 0 && l < Lout) ? l : -1;
      wi[t] = k < K ? k : 0;
    }
  }
  float acc[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) acc[c] = 0.f;
  const float* dyb = dy + (size_t)b * Cout * Lout;
  for (int co0 = 0; co0 < Cout; co0 += DG_COB) {
    float g[DG_COB][DG_NT];
#pragma unroll
    for (int q = 0; q < DG_COB; ++q)
#pragma unroll
      for (int t = 0; t < DG_NT; ++t) {
        const bool ok = lo[t] >= 0 && co0 + q < Cout;
        g[q][t] = dyb[ok ? (size_t)(co0 + q) * Lout + lo[t] : 0];
        if (!ok) g[q][t] = 0.f;
      }
#pragma unroll
    for (int q = 0; q < DG_COB; ++q) {
      const int co = co0 + q < Cout ? co0 + q : 0;
#pragma unroll
      for (int t = 0; t < DG_NT; ++t) {
        const float* w = wl + ((size_t)co * T + wi[t]) * CG;
#pragma unroll
        for (int c = 0; c < CG; ++c) acc[c] = fmaf(g[q][t], w[c], acc[c]);
      }
    }
  }
  // (+ a gradient arriving at x through another consumer; it may live in dx itself: an element is read and written by one thread.
  // Its CG values are requested together -- a load behind a per-element branch carries a full wait of its own)
  if (add) {
    float e[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) e[c] = ci0 + c < Cin ? add[((size_t)b * Cin + ci0 + c) * Lin + j] : 0.f;
#pragma unroll
    for (int c = 0; c < CG; ++c) acc[c] += e[c];
  }
#pragma unroll
  for (int c = 0; c < CG; ++c)
    if (ci0 + c < Cin) dx[((size_t)b * Cin + ci0 + c) * Lin + j] = acc[c];
}

// ------------------------------------------------------------------------------------------------ weight gradient
// part[chunk][co][ci * K + k] = sum over the chunk's (b, lo) of dy[b][co][lo] * xu[b][ci][lo * stride - pad + k], xu = x
// upsampled by `up`; part[chunk][co][Cin * K] = sum of dy (bias gradient).  Workgroup = (chunk, group of 4 output channels).
// With E = Cin * K <= 256 entries the 256 threads form 256 / E slices that split the 64 positions of a tile between them
// (summed through LDS at the end); with more entries a thread owns two of them and all 64 positions.
constexpr int WG_CHUNKS = 1024;

// CO output channels per workgroup (4 / 8 / 16), tiles of WG_TL positions (64 / 128 / 256), WG_EP = 4 entries per thread: a tile's
// input span is staged once for all CO channels, and the CO gradients of a position -- 16-byte LDS reads that return 1 KB per wave
// whatever the lanes ask for -- serve four entries each.  (With one entry per thread the launch ran at the LDS return bandwidth,
// 4x below the FMA rate.)  The threads form slices = 256 / ceil(E / 4) groups that split the positions of a tile between them;
// the slices are summed through LDS at the end.
constexpr int WG_EP = 4;

template <int CO, int WG_TL>
__global__ __launch_bounds__(IT_THREADS) void conv_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                float* __restrict__ part, int B, int Cin, int Lin, int Cout,
                                                                int Lout, int K, int stride, int pad, int up, int tiles_per_row,
                                                                int64_t tiles, int chunks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int span = (WG_TL - 1) * stride + K, spanp = span | 1;
  float* ds = lds;                               // [WG_TL][CO]
  float* xs = lds + CO * WG_TL;                  // [Cin][spanp]; reused for the slice reduction at the end
  const int chunk = blockIdx.x, co0 = blockIdx.y * CO;
  const int entries = Cin * K;
  const int eq = (entries + WG_EP - 1) / WG_EP;  // threads per slice
  const int slices = IT_THREADS / eq;
  const int sl = threadIdx.x / eq, q0 = threadIdx.x - sl * eq;
  const bool active = sl < slices;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 acc[WG_EP][CO / 2];                        // channel pairs: one v_pk_fma_f32 per pair, entry and position
  int xo[WG_EP];
#pragma unroll
  for (int j = 0; j < WG_EP; ++j) {
#pragma unroll
    for (int g = 0; g < CO / 2; ++g) acc[j][g] = f32x2{0.f, 0.f};
    const int e = WG_EP * q0 + j;
    const int ec = e < entries ? e : entries - 1;  // (a padding entry recomputes the last one; it is not written)
    xo[j] = (ec / K) * spanp + ec % K;
  }
  constexpr int DV = CO * WG_TL / IT_THREADS;      // gradient values per thread and tile
  float bsum[DV];                                  // bias gradient: this thread's share of the dy values it stages
#pragma unroll
  for (int q = 0; q < DV; ++q) bsum[q] = 0.f;
  const int Lup = Lin * up;
  for (int64_t tile = chunk; tile < tiles; tile += chunks) {
    const int b = (int)(tile / tiles_per_row), lo0 = (int)(tile % tiles_per_row) * WG_TL;
    __syncthreads();
    const float* xb = x + (size_t)b * Cin * Lin;
    const int base = lo0 * stride - pad;
    {
      float dv[DV];                                  // requested first: position-major over the threads (coalesced along lo)
#pragma unroll
      for (int q = 0; q < DV; ++q) {
        const int i = threadIdx.x + q * IT_THREADS;
        const int g = i / WG_TL, t = i - g * WG_TL;
        const int lo = lo0 + t, co = co0 + g;
        const bool ok = lo < Lout && co < Cout;
        dv[q] = dy[ok ? ((size_t)b * Cout + co) * Lout + lo : 0];
        if (!ok) dv[q] = 0.f;
      }
      constexpr int UN = 4;                          // input loads of a thread in flight (a round per load = a global round trip)
      for (int i0 = threadIdx.x; i0 < Cin * span; i0 += IT_THREADS * UN) {
        float v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int ii = i0 + IT_THREADS * u;
          const int ci = ii / span, p = ii - ci * span;
          const int l = base + p;
          const bool ok = ii < Cin * span && l >= 0 && l < Lup;
          v[u] = xb[ok ? (size_t)ci * Lin + l / up : 0];
          if (!ok) v[u] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int ii = i0 + IT_THREADS * u;
          if (ii < Cin * span) {
            const int ci = ii / span, p = ii - ci * span;
            xs[ci * spanp + p] = v[u];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < DV; ++q) {
        const int i = threadIdx.x + q * IT_THREADS;
        const int g = i / WG_TL, t = i - g * WG_TL;
        ds[t * CO + g] = dv[q];
        bsum[q] += dv[q];
      }
    }
    __syncthreads();
    if (active) {
      for (int t = sl; t < WG_TL; t += slices) {
        f32x2 xv[WG_EP];
#pragma unroll
        for (int j = 0; j < WG_EP; ++j) {
          const float v = xs[xo[j] + t * stride];
          xv[j] = f32x2{v, v};
        }
#pragma unroll
        for (int g4 = 0; g4 < CO / 4; ++g4) {
          const f32x4 d = *reinterpret_cast<const f32x4*>(ds + t * CO + 4 * g4);
          const f32x2 d0 = {d.x, d.y}, d1 = {d.z, d.w};
#pragma unroll
          for (int j = 0; j < WG_EP; ++j) {
            acc[j][2 * g4 + 0] = __builtin_elementwise_fma(d0, xv[j], acc[j][2 * g4 + 0]);
            acc[j][2 * g4 + 1] = __builtin_elementwise_fma(d1, xv[j], acc[j][2 * g4 + 1]);
          }
        }
      }
    }
  }
  const int rowlen = entries + 1;
  // slice reduction, four channels per round: xs holds [slice][entry][4]
#pragma unroll
  for (int g4 = 0; g4 < CO / 4; ++g4) {
    __syncthreads();
    if (active) {
#pragma unroll
      for (int j = 0; j < WG_EP; ++j) {
        const int e = WG_EP * q0 + j;
        if (e < entries)
          *reinterpret_cast<f32x4*>(xs + ((size_t)sl * entries + e) * 4) =
              f32x4{acc[j][2 * g4][0], acc[j][2 * g4][1], acc[j][2 * g4 + 1][0], acc[j][2 * g4 + 1][1]};
      }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < entries * 4; o += IT_THREADS) {
      const int e = o >> 2, c = o & 3;
      float sm = 0.f;
      for (int q = 0; q < slices; ++q) sm += xs[((size_t)q * entries + e) * 4 + c];
      const int co = co0 + 4 * g4 + c;
      if (co < Cout) part[((size_t)chunk * Cout + co) * rowlen + e] = sm;
    }
  }
  // thread i staged channel i / WG_TL + q * (IT_THREADS / WG_TL) as its q-th value: gather the per-thread sums per channel
  __syncthreads();
#pragma unroll
  for (int q = 0; q < DV; ++q) ds[q * IT_THREADS + threadIdx.x] = bsum[q];
  __syncthreads();
  if (threadIdx.x < CO) {
    constexpr int R = IT_THREADS / WG_TL;          // channels per round of IT_THREADS values
    const int g = threadIdx.x, q = g / R, t0 = (g % R) * WG_TL;
    float sm = 0.f;
    for (int t = 0; t < WG_TL; ++t) sm += ds[q * IT_THREADS + t0 + t];
    if (co0 + g < Cout) part[((size_t)chunk * Cout + co0 + g) * rowlen + entries] = sm;
  }
}

// one wave per output element: lanes stride over the chunks, float64 sum
__global__ __launch_bounds__(IT_THREADS) void conv_wgrad_reduce_kernel(const float* __restrict__ part, int chunks, int Cout,
                                                                       int entries, float* __restrict__ dW, float* __restrict__ db) {
  const int i = blockIdx.x * (IT_THREADS / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int rowlen = entries + 1;
  if (i >= Cout * rowlen) return;
  double s = 0.0;
  for (int c = lane; c < chunks; c += 64) s += (double)part[(size_t)c * Cout * rowlen + i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane != 0) return;
  const int co = i / rowlen, r = i - co * rowlen;
  if (r < entries) dW[(size_t)co * entries + r] = (float)s;
  else if (db) db[co] = (float)s;
}

// The same for up to WD_MAXJOBS layers in ONE launch (blockIdx.y = layer): the composed training step (indel_train_step.hip) gives
// every layer its own partial-row region and reduces them all at the end of its backward instead of once per layer.
constexpr int WD_MAXJOBS = 48;
struct WgradJob { const float* part; float* dW; float* db; int chunks, Cout, entries, pad; };
struct WgradJobs { WgradJob j[WD_MAXJOBS]; };
// Threads run along the elements of a partial row (coalesced 256-byte reads per wave), the four waves of a workgroup take every
// fourth chunk with four independent double accumulators each and meet in LDS in a fixed order (bitwise reproducible).  (One wave
// per element with its lanes striding over the chunks read one 4-byte word per cache line: 256 us for the 35 layers of the step.)
__global__ __launch_bounds__(IT_THREADS) void conv_wgrad_reduce_multi_kernel(const WgradJobs jobs) {
  static_assert(IT_THREADS == 256, "four waves per workgroup");
  const WgradJob jb = jobs.j[blockIdx.y];
  const int rowlen = jb.entries + 1;
  const int total = jb.Cout * rowlen;
  __shared__ double sm[4][64];
  const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
  for (int i0 = blockIdx.x * 64; i0 < total; i0 += gridDim.x * 64) {
    const int i = i0 + e;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (i < total) {
      const float* p = jb.part + i;
      int c = sl;
      for (; c + 12 < jb.chunks; c += 16) {
        const float v0 = p[(size_t)c * total], v1 = p[(size_t)(c + 4) * total], v2 = p[(size_t)(c + 8) * total],
                    v3 = p[(size_t)(c + 12) * total];
        s0 += (double)v0;
        s1 += (double)v1;
        s2 += (double)v2;
        s3 += (double)v3;
      }
      for (; c < jb.chunks; c += 4) s0 += (double)p[(size_t)c * total];
    }
    sm[sl][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && i < total) {
      const double s = (sm[0][e] + sm[1][e]) + (sm[2][e] + sm[3][e]);
      const int co = i / rowlen, r = i - co * rowlen;
      if (r < jb.entries) jb.dW[(size_t)co * jb.entries + r] = (float)s;
      else if (jb.db) jb.db[co] = (float)s;
    }
    __syncthreads();
  }
}

struct WgradDefer {
  bool on = false;
  int n = 0;
  WgradJobs jobs;
};
static thread_local WgradDefer g_wgrad_defer;

}  // namespace
// begin collecting (the caller guarantees distinct `part` regions per layer until the flush); flush launches the one reduction
void wgrad_defer_begin() {
  g_wgrad_defer.on = true;
  g_wgrad_defer.n = 0;
}
int wgrad_defer_flush(hipStream_t st) {
  WgradDefer& d = g_wgrad_defer;
  d.on = false;
  if (d.n == 0) return MURAL_OK;
  int most = 0;
  for (int k = 0; k < d.n; ++k) most = std::max(most, d.jobs.j[k].Cout * (d.jobs.j[k].entries + 1));
  const int gx = std::min((most + 63) / 64, 64);
  hipLaunchKernelGGL(conv_wgrad_reduce_multi_kernel, dim3(gx, d.n), dim3(IT_THREADS), 0, st, d.jobs);
  d.n = 0;
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}
namespace {

// ------------------------------------------------------------------------------------------------ activations
// kind: 1 ReLU, 2 SiLU, 3 Softplus (beta 1, linear above 20, like torch.nn.Softplus)
__device__ __forceinline__ float act_f(float v, int kind) {
  if (kind == 1) return v > 0.f ? v : 0.f;
  if (kind == 2) return v / (1.f + expf(-v));
  return v > 20.f ? v : log1pf(expf(v));
}
__device__ __forceinline__ float act_d(float v, int kind) {
  if (kind == 1) return v > 0.f ? 1.f : 0.f;
  const float s = 1.f / (1.f + expf(-v));
  if (kind == 2) return s * (1.f + v * (1.f - s));
  return v > 20.f ? 1.f : s;
}

__global__ __launch_bounds__(IT_THREADS) void act_fwd_kernel(const float* __restrict__ x, int64_t n, int kind, float* __restrict__ y) {
  const int64_t step = (int64_t)gridDim.x * IT_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * IT_THREADS + threadIdx.x; i < n; i += step) y[i] = act_f(x[i], kind);
}
__global__ __launch_bounds__(IT_THREADS) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, int64_t n,
                                                             int kind, float* __restrict__ dx) {
  const int64_t step = (int64_t)gridDim.x * IT_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * IT_THREADS + threadIdx.x; i < n; i += step) dx[i] = dy[i] * act_d(x[i], kind);
}

// ------------------------------------------------------------------------------------------------ weight layouts of a whole model
// One launch per training step instead of one (forward) + one (input gradient) per layer: element i of the concatenated weights
// finds its job by binary search and is written in the forward layout [Cin][K][Cout] and, where asked for, the input-gradient
// layout [Cout][K flipped][Cin].
__global__ __launch_bounds__(IT_THREADS) void relayout_multi_kernel(const MuralRelayoutJob* __restrict__ jobs, int n, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * IT_THREADS + threadIdx.x;
  if (i >= total) return;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].start <= i) lo = mid;
    else hi = mid - 1;
  }
  const MuralRelayoutJob j = jobs[lo];
  const int e = (int)(i - j.start);
  const int co = e / (j.Cin * j.K), r = e - co * j.Cin * j.K, ci = r / j.K, k = r - ci * j.K;
  const float v = j.W[e];
  j.wt_fwd[((size_t)ci * j.K + k) * j.Cout + co] = v;
  if (j.wt_dgrad) j.wt_dgrad[((size_t)co * j.K + (j.K - 1 - k)) * j.Cin + ci] = v;
}

// ------------------------------------------------------------------------------------------------ conv -> BatchNorm -> act
// The U-Net's unit is Conv1d -> BatchNorm1d (batch statistics) [-> SiLU / ReLU] [+ residuals] (model_indel.py:6-19, :117-123): one
// launch finalises the statistics (every workgroup derives scale / shift of all <= 96 channels from the batch sums, workgroup 0
// also writes the state and the running statistics), applies BatchNorm and the activation and adds the residuals; the backward
// re-derives the activation's slope from the saved conv output, so the BatchNorm output is never stored.
constexpr int CB_SLOTS = MURAL_BN_SLOTS;

// The two batch sums of every channel into LDS (sums[0][c], sums[1][c]), by the whole workgroup: 32 lanes per channel read one
// accumulator copy each and reduce by shuffles, 8 channels per pass.  nslots: the copies the producer's workgroups actually used
// (its grid.y, at most CB_SLOTS; the others are still zero).  Every workgroup of the consumers below starts with this; one thread
// per channel walking 2 x 32 copies made it 6-12 us of latency per workgroup (0.6 ms of a 6 ms step).
constexpr int CB_MAXC = 128;
__device__ __forceinline__ void cb_slot_sums(const double* __restrict__ acc, int C, int nslots, double (*sums)[CB_MAXC]) {
  const int k = threadIdx.x & 31;
  for (int base = 0; base < C; base += IT_THREADS / 32) {
    const int c = base + (threadIdx.x >> 5);
    double s1 = 0.0, s2 = 0.0;
    if (c < C && k < nslots) {
      s1 = acc[((size_t)k * 2 + 0) * C + c];
      s2 = acc[((size_t)k * 2 + 1) * C + c];
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
      s1 += __shfl_xor(s1, off, 64);
      s2 += __shfl_xor(s2, off, 64);
    }
    if (k == 0 && c < C) {
      sums[0][c] = s1;
      sums[1][c] = s2;
    }
  }
  __syncthreads();
}

// z = act(scale * y + shift) [+ res1] [+ res2];  state = scale | shift | mean | invstd ([4][C])
template <int NRES>      // residual tensors read: 0, 1 (res1) or 2 (res1 and res2) -- decided at compile time: a load behind a run-time
                         // `if (res1)` carries a full wait of its own, three serial round trips per 16 bytes instead of one
__global__ __launch_bounds__(IT_THREADS) void bn_post_apply_kernel(const float* __restrict__ y, int64_t total, int C, int L,
                                                                   const double* __restrict__ acc, int nslots, double n,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float eps, float momentum, float* __restrict__ running_mean,
                                                                   float* __restrict__ running_var, float* __restrict__ state, int act,
                                                                   const float* __restrict__ res1, const float* __restrict__ res2,
                                                                   float* __restrict__ z) {
  extern __shared__ float cst[];      // [C][2] scale, shift
  __shared__ double sums[2][CB_MAXC];
  cb_slot_sums(acc, C, nslots, sums);
  for (int c = threadIdx.x; c < C; c += IT_THREADS) {
    const double mean = sums[0][c] / n;
    double var = sums[1][c] / n - mean * mean;
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
  if ((L & 3) == 0 && total < (int64_t(1) << 31)) {
    // four consecutive positions of a row per thread: 16-byte accesses and one pair of 32-bit divisions per four elements (two
    // 64-bit divisions per element made this map ALU-bound)
    const uint32_t nq = (uint32_t)(total >> 2), qstep = gridDim.x * IT_THREADS;
    for (uint32_t q = blockIdx.x * IT_THREADS + threadIdx.x; q < nq; q += qstep) {
      const uint32_t e = q << 2;
      const int c = (int)((e / (uint32_t)L) % (uint32_t)C);
      const float sc = cst[2 * c], sh = cst[2 * c + 1];
      const f32x4 yv = ld4(y + e);
      f32x4 r1 = splat(0.f), r2 = splat(0.f);
      if (NRES >= 1) r1 = ld4(res1 + e);
      if (NRES >= 2) r2 = ld4(res2 + e);
      f32x4 r;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        r[t] = fmaf(sc, yv[t], sh);
        if (act) r[t] = act_f(r[t], act);
      }
      if (NRES >= 1) r += r1;
      if (NRES >= 2) r += r2;
      st4(z + e, r);
    }
    return;
  }
  const int64_t step = (int64_t)gridDim.x * IT_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * IT_THREADS + threadIdx.x; i < total; i += step) {
    const int c = (int)((i / L) % C);
    float v = fmaf(cst[2 * c], y[i], cst[2 * c + 1]);
    if (act) v = act_f(v, act);
    if (NRES >= 1) v += res1[i];
    if (NRES >= 2) v += res2[i];
    z[i] = v;
  }
}

// sums over (B, L) of g = dz * act'(u) and g * xhat per channel, u = scale * y + shift, xhat = (y - mean) * invstd
__global__ __launch_bounds__(IT_THREADS) void bn_post_bwd_reduce_kernel(const float* __restrict__ dz, const float* __restrict__ y, int B,
                                                                        int C, int L, const float* __restrict__ state, int act,
                                                                        double* __restrict__ acc) {
  const int c = blockIdx.x;
  const float sc = state[c], sh = state[C + c], mu = state[2 * C + c], is = state[3 * C + c];
  double a = 0.0, bq = 0.0;
  const int64_t per = (int64_t)B * L;
  if ((L & 3) == 0 && per * C < (int64_t(1) << 31)) {   // four positions per thread, 32-bit index math
    const uint32_t nq = (uint32_t)(per >> 2), qstep = gridDim.y * IT_THREADS;
    for (uint32_t q = blockIdx.y * IT_THREADS + threadIdx.x; q < nq; q += qstep) {
      const uint32_t e = q << 2, b = e / (uint32_t)L, l = e - b * (uint32_t)L;
      const uint32_t o = (b * (uint32_t)C + (uint32_t)c) * (uint32_t)L + l;
      const f32x4 yv = ld4(y + o), gz = ld4(dz + o);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float g = gz[t];
        if (act) g *= act_d(fmaf(sc, yv[t], sh), act);
        a += g;
        bq += (double)g * ((yv[t] - mu) * is);
      }
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.y * IT_THREADS + threadIdx.x; i < per; i += (int64_t)gridDim.y * IT_THREADS) {
      const int64_t b = i / L;
      const int l = (int)(i - b * L);
      const size_t o = (size_t)(b * C + c) * L + l;
      const float v = y[o];
      float g = dz[o];
      if (act) g *= act_d(fmaf(sc, v, sh), act);
      a += g;
      bq += (double)g * ((v - mu) * is);
    }
  }
  __shared__ double sh2[2][IT_THREADS];
  sh2[0][threadIdx.x] = a;
  sh2[1][threadIdx.x] = bq;
  __syncthreads();
  for (int off = IT_THREADS / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      sh2[0][threadIdx.x] += sh2[0][threadIdx.x + off];
      sh2[1][threadIdx.x] += sh2[1][threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {   // accumulator copy by workgroup: same-address atomics serialise in L2
    double* slot = acc + (size_t)(blockIdx.y % CB_SLOTS) * 2 * C;
    atomicAdd(&slot[c], sh2[0][0]);
    atomicAdd(&slot[C + c], sh2[1][0]);
  }
}

// dy = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)); workgroup 0 also writes dgamma = sum(g * xhat), dbeta = sum(g)
__global__ __launch_bounds__(IT_THREADS) void bn_post_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                                       int64_t total, int C, int L, const float* __restrict__ state,
                                                                       const float* __restrict__ gamma, const double* __restrict__ acc,
                                                                       int nslots, double n, int act, float* __restrict__ dy,
                                                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
  extern __shared__ float cst[];      // [C][6]: gamma * invstd, mean(g), mean(g * xhat), mean, scale, shift
  __shared__ double sums[2][CB_MAXC];
  cb_slot_sums(acc, C, nslots, sums);
  for (int c = threadIdx.x; c < C; c += IT_THREADS) {
    const double s1 = sums[0][c], s2 = sums[1][c];
    cst[6 * c + 0] = gamma[c] * state[3 * C + c];
    cst[6 * c + 1] = (float)(s1 / n);
    cst[6 * c + 2] = (float)(s2 / n);
    cst[6 * c + 3] = state[2 * C + c];
    cst[6 * c + 4] = state[c];
    cst[6 * c + 5] = state[C + c];
    if (blockIdx.x == 0) {
      dgamma[c] = (float)s2;
      dbeta[c] = (float)s1;
    }
  }
  __syncthreads();
  if ((L & 3) == 0 && total < (int64_t(1) << 31)) {   // (as in bn_post_apply_kernel)
    const uint32_t nq = (uint32_t)(total >> 2), qstep = gridDim.x * IT_THREADS;
    for (uint32_t q = blockIdx.x * IT_THREADS + threadIdx.x; q < nq; q += qstep) {
      const uint32_t e = q << 2;
      const int c = (int)((e / (uint32_t)L) % (uint32_t)C);
      const float k0 = cst[6 * c + 0], m1 = cst[6 * c + 1], m2 = cst[6 * c + 2], mu = cst[6 * c + 3], sc = cst[6 * c + 4],
                  sh = cst[6 * c + 5], is = state[3 * C + c];
      const f32x4 yv = ld4(y + e), gz = ld4(dz + e);
      f32x4 r;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float g = gz[t];
        if (act) g *= act_d(fmaf(sc, yv[t], sh), act);
        r[t] = k0 * (g - m1 - (yv[t] - mu) * is * m2);
      }
      st4(dy + e, r);
    }
    return;
  }
  const int64_t step = (int64_t)gridDim.x * IT_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * IT_THREADS + threadIdx.x; i < total; i += step) {
    const int c = (int)((i / L) % C);
    const float v = y[i];
    float g = dz[i];
    if (act) g *= act_d(fmaf(cst[6 * c + 4], v, cst[6 * c + 5]), act);
    const float xh = (v - cst[6 * c + 3]) * state[3 * C + c];
    dy[i] = cst[6 * c + 0] * (g - cst[6 * c + 1] - xh * cst[6 * c + 2]);
  }
}

// Short tensors (the deep U-Net levels: B * L <= CB_ONEPASS elements per channel): one workgroup per channel does both passes of a
// direction -- sums, then the map -- so the batch-sum launch, its atomics and the accumulator block drop out.  A thread keeps its
// CB_OP_PER elements in registers between the passes.
// Two sizes: 256 threads x 8 elements (B * L <= 2048: the two deepest levels at a batch of 128) and 1024 threads x 12 (<= 12288: the
// 80-column level) -- beyond that a workgroup per channel is too few workgroups for the tensor.
constexpr int CB_OP_PER = 8, CB_ONEPASS = IT_THREADS * CB_OP_PER;
constexpr int CB_OP2_THREADS = 1024, CB_OP2_PER = 12, CB_ONEPASS2 = CB_OP2_THREADS * CB_OP2_PER;

// block sums of (a, b): shuffles inside a wave, one LDS round across the waves (fixed order: bitwise reproducible)
template <int THREADS>
__device__ __forceinline__ void cb_block_sum2(double& a, double& b, double (*sh)[THREADS / 64]) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_xor(a, off, 64);
    b += __shfl_xor(b, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = a;
    sh[1][threadIdx.x >> 6] = b;
  }
  __syncthreads();
  a = 0.0;
  b = 0.0;
#pragma unroll
  for (int w = 0; w < THREADS / 64; ++w) {
    a += sh[0][w];
    b += sh[1][w];
  }
}

template <int THREADS, int PER>
__global__ __launch_bounds__(THREADS) void bn_post_onepass_fwd_kernel(const float* __restrict__ y, int B, int C, int L,
                                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                         float eps, float momentum, float* __restrict__ running_mean,
                                                                         float* __restrict__ running_var, float* __restrict__ state, int act,
                                                                         const float* __restrict__ res1, const float* __restrict__ res2,
                                                                         float* __restrict__ z) {
  __shared__ double sh[2][THREADS / 64];
  const int c = blockIdx.x, per = B * L;
  float v[PER];
  uint32_t o[PER];
  double s = 0.0, q = 0.0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {            // all of a thread's loads in flight together
    const int i = (int)threadIdx.x + j * THREADS;
    const int b = i / L, l = i - b * L;
    o[j] = (uint32_t)((b * C + c) * L + l);
    v[j] = i < per ? y[o[j]] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    s += v[j];
    q += (double)v[j] * v[j];
  }
  cb_block_sum2<THREADS>(s, q, sh);
  const double n = (double)per, mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const float sc = (float)(gamma[c] * invstd), sft = (float)(beta[c] - mean * gamma[c] * invstd);
  if (threadIdx.x == 0) {
    state[c] = sc;
    state[C + c] = sft;
    state[2 * C + c] = (float)mean;
    state[3 * C + c] = (float)invstd;
    if (running_mean) {
      const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
      running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
  }
  // the residuals of all of a thread's elements are requested together (one branch per tensor: a load behind a per-element branch
  // carries a full wait of its own -- 2 x PER serial round trips in a launch that is one round trip long otherwise)
  float e[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) e[j] = 0.f;
  if (res1) {
#pragma unroll
    for (int j = 0; j < PER; ++j) e[j] = (int)threadIdx.x + j * THREADS < per ? res1[o[j]] : 0.f;
  }
  if (res2) {
    float e2[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) e2[j] = (int)threadIdx.x + j * THREADS < per ? res2[o[j]] : 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) e[j] += e2[j];
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = (int)threadIdx.x + j * THREADS;
    if (i < per) {
      float r = fmaf(sc, v[j], sft);
      if (act) r = act_f(r, act);
      z[o[j]] = r + e[j];
    }
  }
}

template <int THREADS, int PER>
__global__ __launch_bounds__(THREADS) void bn_post_onepass_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ y, int B,
                                                                         int C, int L, const float* __restrict__ state,
                                                                         const float* __restrict__ gamma, int act, float* __restrict__ dy,
                                                                         float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double sh[2][THREADS / 64];
  const int c = blockIdx.x, per = B * L;
  const float sc = state[c], sft = state[C + c], mu = state[2 * C + c], is = state[3 * C + c];
  float g[PER], xh[PER];
  uint32_t o[PER];
  double a = 0.0, bq = 0.0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = (int)threadIdx.x + j * THREADS;
    const int b = i / L, l = i - b * L;
    o[j] = (uint32_t)((b * C + c) * L + l);
    const bool ok = i < per;
    const float v = ok ? y[o[j]] : 0.f;
    g[j] = ok ? dz[o[j]] : 0.f;
    if (act) g[j] *= act_d(fmaf(sc, v, sft), act);
    xh[j] = ok ? (v - mu) * is : 0.f;
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    a += g[j];
    bq += (double)g[j] * xh[j];
  }
  cb_block_sum2<THREADS>(a, bq, sh);
  if (threadIdx.x == 0) {
    dgamma[c] = (float)bq;
    dbeta[c] = (float)a;
  }
  const float k0 = gamma[c] * is, m1 = (float)(a / per), m2 = (float)(bq / per);
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = (int)threadIdx.x + j * THREADS;
    if (i < per) dy[o[j]] = k0 * (g[j] - m1 - xh[j] * m2);
  }
}

int out_length(int Lin, int K, int stride, int pad, int up) { return (Lin * up + 2 * pad - K) / stride + 1; }

}  // namespace
}  // namespace mural

using namespace mural;

extern "C" int mural_op_convg_out_length(int32_t Lin, int32_t K, int32_t stride, int32_t pad, int32_t up) {
  if (K < 1 || stride < 1 || up < 1 || pad < 0 || Lin * up + 2 * pad < K) return -1;
  return out_length(Lin, K, stride, pad, up);
}

// y = Conv1d(upsample_up(x)); wt = scratch of Cout * Cin * K floats (receives the [Cin][K][Cout] layout of W)
extern "C" int mural_op_convg_fwd(const float* x, const float* W, const float* bias, float* wt, float* y, int64_t B, int32_t Cin,
                                  int32_t Lin, int32_t Cout, int32_t K, int32_t stride, int32_t pad, int32_t up, void* stream) {
  MURAL_REQUIRE(B >= 0 && Cin >= 1 && Cout >= 1 && Lin >= 1, "convg_fwd: bad sizes");
  MURAL_REQUIRE(mural_op_convg_out_length(Lin, K, stride, pad, up) >= 1, "convg_fwd: bad geometry");
  if (B == 0) return MURAL_OK;
  MURAL_REQUIRE(x && wt && y, "convg_fwd: null pointer");
  if (W != nullptr) {       // W == NULL: wt already holds the forward layout (mural_op_relayout_multi)
    int rc = mural_op_relayout(W, wt, Cout, Cin, K, 0, stream);
    if (rc != MURAL_OK) return rc;
  }
  Conv1dArgs a;
  std::memset(&a, 0, sizeof(a));
  a.in = x; a.wt = wt; a.bias = bias; a.out = y;
  a.B = (int)B; a.Cin = Cin; a.Lin = Lin; a.Cout = Cout; a.Lout = out_length(Lin, K, stride, pad, up);
  a.K = K; a.stride = stride; a.pad = pad; a.up = up;
  a.act = ACT_NONE;
  return launch_conv1d(a, (hipStream_t)stream);
}

// dx (optional) [B][Cin][Lin], dW [Cout][Cin][K], db (optional) [Cout]; part = scratch of mural_op_convg_bwd_scratch floats
extern "C" size_t mural_op_convg_bwd_scratch(int32_t Cin, int32_t Cout, int32_t K) {
  return (size_t)WG_CHUNKS * Cout * ((size_t)Cin * K + 1);
}

static int convg_bwd_impl(const float* dy, const float* x, const float* W, const float* wt_dgrad, int64_t B, int32_t Cin, int32_t Lin,
                          int32_t Cout, int32_t K, int32_t stride, int32_t pad, int32_t up, float* dx, const float* dx_add, float* dW, float* db,
                          float* part, size_t part_floats, void* stream);

extern "C" int mural_op_convg_bwd(const float* dy, const float* x, const float* W, int64_t B, int32_t Cin, int32_t Lin, int32_t Cout,
                                  int32_t K, int32_t stride, int32_t pad, int32_t up, float* dx, float* dW, float* db, float* part,
                                  size_t part_floats, void* stream) {
  return convg_bwd_impl(dy, x, W, nullptr, B, Cin, Lin, Cout, K, stride, pad, up, dx, nullptr, dW, db, part, part_floats, stream);
}

// wt_dgrad (optional): the input-gradient layout of W prepared by mural_op_relayout_multi; dx_add (optional, may alias dx): a second
// gradient of x (a residual / skip connection's) added while dx is written
static int convg_bwd_impl(const float* dy, const float* x, const float* W, const float* wt_dgrad, int64_t B, int32_t Cin, int32_t Lin,
                          int32_t Cout, int32_t K, int32_t stride, int32_t pad, int32_t up, float* dx, const float* dx_add, float* dW, float* db,
                          float* part, size_t part_floats, void* stream) {
  MURAL_REQUIRE(B >= 1 && Cin >= 1 && Cout >= 1 && Lin >= 1, "convg_bwd: bad sizes");
  const int Lout = mural_op_convg_out_length(Lin, K, stride, pad, up);
  MURAL_REQUIRE(Lout >= 1, "convg_bwd: bad geometry");
  MURAL_REQUIRE(dy && x && W && dW && part, "convg_bwd: null pointer");
  MURAL_REQUIRE(Cin * K <= 4 * IT_THREADS, "convg_bwd: Cin * K = %d exceeds %d", Cin * K, 4 * IT_THREADS);
  hipStream_t st = (hipStream_t)stream;
  if (dx && stride == 1 && up == 1 && Cin % 4 == 0) {
    // a stride-1 conv's input gradient is the conv of dy with the transposed, tap-flipped weights: the forward engines (MFMA on short
    // rows / deep reductions, packed FMAs otherwise) take it; the deep levels' rows of 8..80 positions left the direct kernel below
    // with a handful of workgroups (30-80 us per launch).  The flipped weights borrow the weight gradient's scratch, which is
    // written after this launch on the same stream.
    if (!wt_dgrad) {
      MURAL_REQUIRE(part_floats >= (size_t)Cout * Cin * K, "convg_bwd: scratch too small");
      if (int rc = mural_op_relayout(W, part, Cout, Cin, K, 1, stream)) return rc;
      wt_dgrad = part;
    }
    Conv1dArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = dy; a.wt = wt_dgrad; a.bias = nullptr; a.out = dx;
    a.B = (int)B; a.Cin = Cout; a.Lin = Lout; a.Cout = Cin; a.Lout = Lin;
    a.K = K; a.stride = 1; a.pad = K - 1 - pad; a.up = 1;
    a.act = ACT_NONE;
    a.res1 = dx_add;
    if (int rc = launch_conv1d(a, st)) return rc;
  } else if (dx && Cin > 4 && (stride == 1 || up == 1) && (stride == 1 ? K + up - 1 : (K + stride - 1) / stride) <= 12 &&
             (size_t)Cout * (stride == 1 ? K + up - 1 : K) * 8 * sizeof(float) <= 64 * 1024) {
    const int64_t rows = B * Lin;
    const int nt = stride == 1 ? K + up - 1 : (K + stride - 1) / stride;      // (column, weight row) pairs per position
    const dim3 grid((unsigned)((rows + IT_THREADS - 1) / IT_THREADS), (Cin + 7) / 8);
    const size_t lds = (size_t)Cout * (stride == 1 ? K + up - 1 : K) * 8 * sizeof(float);
#define MURAL_DGRAD(NT_, COB_)                                                                                                        \
  hipLaunchKernelGGL((conv_dgrad_gather_kernel<8, NT_, COB_>), grid, dim3(IT_THREADS), lds, st, dy, W, dx, rows, Cin, Lin, Cout, Lout, K, \
                     stride, pad, up, dx_add)
    if (nt <= 2) MURAL_DGRAD(2, 16);
    else if (nt <= 4) MURAL_DGRAD(4, 8);
    else if (nt <= 8) MURAL_DGRAD(8, 4);
    else MURAL_DGRAD(12, 4);
#undef MURAL_DGRAD
    MURAL_HIP_CHECK(hipGetLastError());
  } else if (dx) {
    const int64_t rows = B * Lin;
    const bool small = Cin <= 4;
    const int cg = small ? 4 : 8;
    const dim3 grid((unsigned)((rows + IT_THREADS - 1) / IT_THREADS), (Cin + cg - 1) / cg);
    const size_t lds = (size_t)Cout * K * cg * sizeof(float);
    if (small)
      hipLaunchKernelGGL(conv_dgrad_kernel<4>, grid, dim3(IT_THREADS), lds, st, dy, W, dx, rows, Cin, Lin, Cout, Lout, K, stride, pad, up, dx_add);
    else
      hipLaunchKernelGGL(conv_dgrad_kernel<8>, grid, dim3(IT_THREADS), lds, st, dy, W, dx, rows, Cin, Lin, Cout, Lout, K, stride, pad, up, dx_add);
    MURAL_HIP_CHECK(hipGetLastError());
  }
  const int entries = Cin * K;
  static const bool wgrad_mfma = !(dev_env("MURAL_WGRAD_MFMA") && atoi(dev_env("MURAL_WGRAD_MFMA")) == 0);      // A/B switch of the tools
  if (wgrad_mfma) {
    const int cap = (int)std::min<size_t>(WG_CHUNKS, part_floats / ((size_t)Cout * (entries + 1)));
    int chunks = 0;
    if (cap >= 1 && launch_conv_wgrad_mfma(dy, x, part, B, Cin, Lin, Cout, Lout, K, stride, pad, up, cap, &chunks, st) == MURAL_OK) {
      if (g_wgrad_defer.on && g_wgrad_defer.n < WD_MAXJOBS) {
        g_wgrad_defer.jobs.j[g_wgrad_defer.n++] = WgradJob{part, dW, db, chunks, Cout, entries, 0};
        return MURAL_OK;
      }
      hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((Cout * (entries + 1) + 3) / 4), dim3(IT_THREADS), 0, st, part, chunks, Cout,
                         entries, dW, db);
      MURAL_HIP_CHECK(hipGetLastError());
      return MURAL_OK;
    }
  }
  const int co = Cout % 16 == 0 ? 16 : (Cout % 8 == 0 ? 8 : 4);
  // tile length: fewest visits per row, a visit priced as one latency + its share of FMAs; the tile must leave 5 workgroups per CU
  int tl = 64;
  size_t lds = 0;
  {
    double best = 0.0;
    for (int cand = 256; cand >= 64; cand >>= 1) {
      size_t fl = (size_t)Cin * (((cand - 1) * stride + K) | 1);
      if (fl < (size_t)IT_THREADS * WG_EP * 4) fl = (size_t)IT_THREADS * WG_EP * 4;      // slice reduction scratch: [slices][entries][4]
      const size_t bytes = (fl + (size_t)co * cand) * sizeof(float);
      if (cand > 64 && bytes > 32 * 1024) continue;
      const double cost = (double)((Lout + cand - 1) / cand) * (1.0 + cand / 256.0);
      if (lds == 0 || cost < best) { best = cost; tl = cand; lds = bytes; }
    }
  }
  MURAL_REQUIRE(lds <= 64 * 1024, "convg_bwd: input tile of %zu bytes exceeds 64 KB of LDS", lds);
  const int tiles_per_row = (Lout + tl - 1) / tl;
  const int64_t tiles = B * tiles_per_row;
  const int chunks = (int)(tiles < WG_CHUNKS ? tiles : WG_CHUNKS);
  MURAL_REQUIRE(part_floats >= (size_t)chunks * Cout * (entries + 1), "convg_bwd: scratch too small");
  const dim3 wgrid(chunks, (Cout + co - 1) / co);
#define MURAL_WGRAD(CO_, TL_)                                                                                                        \
  hipLaunchKernelGGL((conv_wgrad_kernel<CO_, TL_>), wgrid, dim3(IT_THREADS), lds, st, dy, x, part, (int)B, Cin, Lin, Cout, Lout, K, \
                     stride, pad, up, tiles_per_row, tiles, chunks)
#define MURAL_WGRAD_TL(CO_)                \
  if (tl == 256) MURAL_WGRAD(CO_, 256);    \
  else if (tl == 128) MURAL_WGRAD(CO_, 128); \
  else MURAL_WGRAD(CO_, 64)
  if (co == 16) { MURAL_WGRAD_TL(16); }
  else if (co == 8) { MURAL_WGRAD_TL(8); }
  else { MURAL_WGRAD_TL(4); }
#undef MURAL_WGRAD_TL
#undef MURAL_WGRAD
  MURAL_HIP_CHECK(hipGetLastError());
  if (g_wgrad_defer.on && g_wgrad_defer.n < WD_MAXJOBS) {      // reduced with every other layer's rows by wgrad_defer_flush
    g_wgrad_defer.jobs.j[g_wgrad_defer.n++] = WgradJob{part, dW, db, chunks, Cout, entries, 0};
    return MURAL_OK;
  }
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((Cout * (entries + 1) + 3) / 4), dim3(IT_THREADS), 0, st, part, chunks, Cout,
                     entries, dW, db);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

extern "C" int mural_op_act_fwd(const float* x, int64_t n, int32_t kind, float* y, void* stream) {
  MURAL_REQUIRE(kind >= 1 && kind <= 3, "act_fwd: kind must be 1 (ReLU), 2 (SiLU) or 3 (Softplus)");
  if (n <= 0) return MURAL_OK;
  int64_t g = (n + IT_THREADS * 4 - 1) / (IT_THREADS * 4);
  hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(IT_THREADS), 0, (hipStream_t)stream, x, n, kind, y);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

extern "C" int mural_op_act_bwd(const float* dy, const float* x, int64_t n, int32_t kind, float* dx, void* stream) {
  MURAL_REQUIRE(kind >= 1 && kind <= 3, "act_bwd: kind must be 1 (ReLU), 2 (SiLU) or 3 (Softplus)");
  if (n <= 0) return MURAL_OK;
  int64_t g = (n + IT_THREADS * 4 - 1) / (IT_THREADS * 4);
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(IT_THREADS), 0, (hipStream_t)stream, dy, x, n, kind, dx);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

// z = act(BatchNorm(Conv1d(upsample_up(x)))) [+ res1] [+ res2] with batch statistics, running statistics updated in place (momentum
// like nn.BatchNorm1d).  y0 receives the conv output (saved for the backward), state [4][Cout] = scale | shift | mean | invstd,
// acc = zeroed accumulator block [MURAL_BN_SLOTS][2][Cout] of doubles, wt = scratch of Cout * Cin * K floats.
extern "C" int mural_op_convg_bn_fwd(const float* x, const float* W, const float* bias, float* wt, float* y0, int64_t B, int32_t Cin,
                                     int32_t Lin, int32_t Cout, int32_t K, int32_t stride, int32_t pad, int32_t up, const float* gamma,
                                     const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                     double* acc, float* state, int32_t act, const float* res1, const float* res2, float* z,
                                     void* stream) {
  MURAL_REQUIRE(act >= 0 && act <= 3, "convg_bn_fwd: act must be 0 (none), 1 (ReLU), 2 (SiLU) or 3 (Softplus)");
  MURAL_REQUIRE(gamma && beta && acc && state && z && y0, "convg_bn_fwd: null pointer");
  MURAL_REQUIRE(Cout <= CB_MAXC, "convg_bn_fwd: at most %d output channels (got %d)", CB_MAXC, Cout);
  if (int rc = mural_op_convg_fwd(x, W, bias, wt, y0, B, Cin, Lin, Cout, K, stride, pad, up, stream)) return rc;
  if (B == 0) return MURAL_OK;
  const int Lout = out_length(Lin, K, stride, pad, up);
  if (B * Lout <= CB_ONEPASS2 && B * Cout * Lout < (int64_t(1) << 31)) {       // short tensor: sums and map in one launch, one workgroup per channel
    if (B * Lout <= CB_ONEPASS)
      hipLaunchKernelGGL((bn_post_onepass_fwd_kernel<IT_THREADS, CB_OP_PER>), dim3(Cout), dim3(IT_THREADS), 0, (hipStream_t)stream, y0, (int)B,
                         Cout, Lout, gamma, beta, eps, momentum, running_mean, running_var, state, act, res1, res2, z);
    else
      hipLaunchKernelGGL((bn_post_onepass_fwd_kernel<CB_OP2_THREADS, CB_OP2_PER>), dim3(Cout), dim3(CB_OP2_THREADS), 0, (hipStream_t)stream, y0,
                         (int)B, Cout, Lout, gamma, beta, eps, momentum, running_mean, running_var, state, act, res1, res2, z);
    MURAL_HIP_CHECK(hipGetLastError());
    return MURAL_OK;
  }
  if (int rc = mural_op_bn_stats(y0, B, Cout, Lout, 0, acc, stream)) return rc;
  const int64_t total = B * Cout * Lout;
  int64_t gy = (B * Lout + 256 * 8 - 1) / (256 * 8);        // grid.y of mural_op_bn_stats: workgroup y adds into copy y % CB_SLOTS
  const int nslots = (int)(gy < 1 ? 1 : (gy > CB_SLOTS ? CB_SLOTS : gy));
  int64_t g = (total + IT_THREADS * 4 - 1) / (IT_THREADS * 4);
  // (a lone second residual travels as the first)
  const float *ra = res1 ? res1 : res2, *rb = res1 ? res2 : nullptr;
#define MURAL_BNPA(N_)                                                                                                                     \
  hipLaunchKernelGGL(bn_post_apply_kernel<N_>, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(IT_THREADS), (size_t)Cout * 2 * sizeof(float), \
                     (hipStream_t)stream, y0, total, Cout, Lout, acc, nslots, (double)(B * Lout), gamma, beta, eps, momentum, running_mean, \
                     running_var, state, act, ra, rb, z)
  if (rb) MURAL_BNPA(2);
  else if (ra) MURAL_BNPA(1);
  else MURAL_BNPA(0);
#undef MURAL_BNPA
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

// backward of the same unit from dz (the gradient of z; the residuals' gradients are dz itself): dgamma, dbeta, then the conv's dx
// (optional), dW, db (optional).  acc = zeroed accumulator block, dy0 = scratch [B][Cout][Lout], part as for mural_op_convg_bwd.
namespace mural {
// mural_op_convg_bn_bwd with dx_add (optional, may alias dx): dx = the conv's input gradient + dx_add -- the composed step
// (indel_train_step.hip) hands the gradient that reaches x through a residual or skip connection here instead of a separate add pass
int convg_bn_bwd_add(const float* dz, const float* x, const float* W, const float* y0, const float* state, const float* gamma, int64_t B,
                     int32_t Cin, int32_t Lin, int32_t Cout, int32_t K, int32_t stride, int32_t pad, int32_t up, int32_t act, double* acc,
                     float* dy0, float* dx, const float* dx_add, float* dW, float* db, float* dgamma, float* dbeta, float* part,
                     size_t part_floats, const float* wt_dgrad, void* stream);
}
extern "C" int mural_op_convg_bn_bwd(const float* dz, const float* x, const float* W, const float* y0, const float* state,
                                     const float* gamma, int64_t B, int32_t Cin, int32_t Lin, int32_t Cout, int32_t K, int32_t stride,
                                     int32_t pad, int32_t up, int32_t act, double* acc, float* dy0, float* dx, float* dW, float* db,
                                     float* dgamma, float* dbeta, float* part, size_t part_floats, const float* wt_dgrad, void* stream) {
  return convg_bn_bwd_add(dz, x, W, y0, state, gamma, B, Cin, Lin, Cout, K, stride, pad, up, act, acc, dy0, dx, nullptr, dW, db, dgamma, dbeta,
                          part, part_floats, wt_dgrad, stream);
}
int mural::convg_bn_bwd_add(const float* dz, const float* x, const float* W, const float* y0, const float* state, const float* gamma, int64_t B,
                            int32_t Cin, int32_t Lin, int32_t Cout, int32_t K, int32_t stride, int32_t pad, int32_t up, int32_t act,
                            double* acc, float* dy0, float* dx, const float* dx_add, float* dW, float* db, float* dgamma, float* dbeta,
                            float* part, size_t part_floats, const float* wt_dgrad, void* stream) {
  MURAL_REQUIRE(act >= 0 && act <= 3, "convg_bn_bwd: act must be 0 (none), 1 (ReLU), 2 (SiLU) or 3 (Softplus)");
  MURAL_REQUIRE(B >= 1 && dz && y0 && state && gamma && acc && dy0 && dgamma && dbeta, "convg_bn_bwd: null pointer / empty batch");
  MURAL_REQUIRE(Cout <= CB_MAXC, "convg_bn_bwd: at most %d output channels (got %d)", CB_MAXC, Cout);
  const int Lout = mural_op_convg_out_length(Lin, K, stride, pad, up);
  MURAL_REQUIRE(Lout >= 1, "convg_bn_bwd: bad geometry");
  hipStream_t st = (hipStream_t)stream;
  const int64_t per = B * Lout, total = per * Cout;
  if (per <= CB_ONEPASS) {            // short tensor: sums and map in one launch, one workgroup per channel
    hipLaunchKernelGGL((bn_post_onepass_bwd_kernel<IT_THREADS, CB_OP_PER>), dim3(Cout), dim3(IT_THREADS), 0, st, dz, y0, (int)B, Cout, Lout,
                       state, gamma, act, dy0, dgamma, dbeta);
  } else if (per <= CB_ONEPASS2 && total < (int64_t(1) << 31)) {
    hipLaunchKernelGGL((bn_post_onepass_bwd_kernel<CB_OP2_THREADS, CB_OP2_PER>), dim3(Cout), dim3(CB_OP2_THREADS), 0, st, dz, y0, (int)B, Cout,
                       Lout, state, gamma, act, dy0, dgamma, dbeta);
  } else {
    int gy = (int)((per + IT_THREADS * 8 - 1) / (IT_THREADS * 8));
    gy = gy < 1 ? 1 : (gy > 256 ? 256 : gy);
    hipLaunchKernelGGL(bn_post_bwd_reduce_kernel, dim3(Cout, gy), dim3(IT_THREADS), 0, st, dz, y0, (int)B, Cout, Lout, state, act, acc);
    int64_t g = (total + IT_THREADS * 4 - 1) / (IT_THREADS * 4);
    hipLaunchKernelGGL(bn_post_bwd_apply_kernel, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(IT_THREADS), (size_t)Cout * 6 * sizeof(float),
                       st, dz, y0, total, Cout, Lout, state, gamma, acc, gy > CB_SLOTS ? CB_SLOTS : gy, (double)per, act, dy0, dgamma,
                       dbeta);
  }
  MURAL_HIP_CHECK(hipGetLastError());
  MURAL_REQUIRE(dx || !dx_add, "convg_bn_bwd: dx_add without dx");
  return convg_bwd_impl(dy0, x, W, wt_dgrad, B, Cin, Lin, Cout, K, stride, pad, up, dx, dx_add, dW, db, part, part_floats, stream);
}

// every conv weight of a model into its forward (and, where wt_dgrad is set, input-gradient) layout in one launch; jobs: device
// array sorted by `start` (the running sum of Cout * Cin * K), total = the sum over all jobs
extern "C" int mural_op_relayout_multi(const MuralRelayoutJob* jobs, int32_t n_jobs, int64_t total, void* stream) {
  MURAL_REQUIRE(jobs && n_jobs >= 1 && total >= 1, "relayout_multi: empty job list");
  hipLaunchKernelGGL(relayout_multi_kernel, dim3((unsigned)((total + IT_THREADS - 1) / IT_THREADS)), dim3(IT_THREADS), 0,
                     (hipStream_t)stream, jobs, n_jobs, total);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

// Local branch of the SNV models: shared k-mer embedding -> Linear/ReLU/BN x2 -> Linear (eval mode).
// Reference semantics: MuRaL/model/model_snv.py:451-468, :492 (Network2) and :74-93 (Network0).
// 0.6 % of the model's FLOPs.  Two kernels: snv_local_mlp_mfma (below; fp32 MFMA with all weights resident in LDS, used
// whenever they fit -- every shipped configuration) and snv_local_mlp, the plain fp32 VALU version for wider layers
// (activations of a 32-position tile in LDS, BN-folded transposed weights streamed from L2).
#include <algorithm>
#include <cstdlib>

#include "mfma_tile.h"
#include "snv.h"

namespace mural {

constexpr int LOC_TP = 32;        // positions per workgroup tile
constexpr int LOC_PG = 8;         // positions per thread task
constexpr int LOC_THREADS = 256;

__device__ __forceinline__ void dense_layer(const float* __restrict__ X, int xs, const float* __restrict__ wt,
                                            const float* __restrict__ bias, int K4, int H, float* __restrict__ Y, int ys,
                                            bool relu) {
  // Y[p][h] = act(bias[h] + sum_k X[p][k] * wt[k][h]) for the LOC_TP positions of the tile
  const int tasks = H * (LOC_TP / LOC_PG);
  for (int task = threadIdx.x; task < tasks; task += LOC_THREADS) {
    const int pg = task / H;
    const int h = task - pg * H;
    float acc[LOC_PG];
    const float b = bias[h];
#pragma unroll
    for (int i = 0; i < LOC_PG; ++i) acc[i] = b;
    for (int k4 = 0; k4 < K4; ++k4) {
      const float w0 = wt[(4 * k4 + 0) * H + h], w1 = wt[(4 * k4 + 1) * H + h];
      const float w2 = wt[(4 * k4 + 2) * H + h], w3 = wt[(4 * k4 + 3) * H + h];
#pragma unroll
      for (int i = 0; i < LOC_PG; ++i) {
        const float4 x = *reinterpret_cast<const float4*>(X + (pg * LOC_PG + i) * xs + 4 * k4);
        acc[i] = fmaf(x.x, w0, acc[i]);
        acc[i] = fmaf(x.y, w1, acc[i]);
        acc[i] = fmaf(x.z, w2, acc[i]);
        acc[i] = fmaf(x.w, w3, acc[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < LOC_PG; ++i) Y[(pg * LOC_PG + i) * ys + h] = relu ? fmaxf(acc[i], 0.f) : acc[i];
  }
}

__global__ __launch_bounds__(LOC_THREADS) void snv_local_mlp(LocalDev L, const int64_t* __restrict__ cat, int64_t n,
                                                             float* __restrict__ out, int xs, int h1s, int h2s) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  float* X = lsm;                 // [LOC_TP][xs]
  float* H1 = X + LOC_TP * xs;    // [LOC_TP][h1s]
  float* H2 = H1 + LOC_TP * h1s;  // [LOC_TP][h2s]
  float* O = H2 + LOC_TP * h2s;   // [LOC_TP][SNV_MAXCLASS]
  const int64_t n_tiles = (n + LOC_TP - 1) / LOC_TP;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * LOC_TP;
    // gather embeddings (zero-padded to xs columns)
    for (int i = threadIdx.x; i < LOC_TP * xs; i += LOC_THREADS) {
      const int p = i / xs, k = i - p * xs;
      float v = 0.f;
      const int64_t row = row0 + p;
      if (k < L.in1 && row < n) {
        const int col = k / 5, d = k - 5 * col;
        int64_t id = cat[row * L.cols + col];
        id = id < 0 ? 0 : (id >= L.emb_rows ? L.emb_rows - 1 : id);
        v = L.emb[id * 5 + d];
      }
      X[i] = v;
    }
    // zero the K padding of the hidden buffers once per tile (cheap, keeps the k4 loops branch-free)
    for (int i = threadIdx.x; i < LOC_TP * (h1s - L.h1); i += LOC_THREADS) {
      const int p = i / (h1s - L.h1), k = L.h1 + i % (h1s - L.h1);
      H1[p * h1s + k] = 0.f;
    }
    for (int i = threadIdx.x; i < LOC_TP * (h2s - L.h2); i += LOC_THREADS) {
      const int p = i / (h2s - L.h2), k = L.h2 + i % (h2s - L.h2);
      H2[p * h2s + k] = 0.f;
    }
    __syncthreads();
    dense_layer(X, xs, L.w1t, L.b1, xs / 4, L.h1, H1, h1s, true);
    __syncthreads();
    dense_layer(H1, h1s, L.w2t, L.b2, h1s / 4, L.h2, H2, h2s, true);
    __syncthreads();
    dense_layer(H2, h2s, L.w3t, L.b3, h2s / 4, L.n_class, O, SNV_MAXCLASS, false);
    __syncthreads();
    for (int i = threadIdx.x; i < LOC_TP * L.n_class; i += LOC_THREADS) {
      const int p = i / L.n_class, k = i - p * L.n_class;
      if (row0 + p < n) out[(row0 + p) * L.n_class + k] = O[p * SNV_MAXCLASS + k];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same MLP on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation).  One persistent workgroup per CU keeps
// all three weight matrices in LDS in MFMA A-fragment order (29.4 k floats for 95 -> 150 -> 75 -> 4) and walks tiles of 32
// positions: D[feature 16][position 16] += W[feature][k] * act[k][position].  A "unit" is one block of 16 output features
// against both 16-position blocks of the tile (two independent accumulator chains sharing the A operand); the units of a
// layer are dealt to the four waves so that the three layers together balance.  Operands are fetched with ds_read_b128:
// lane (m|n = lane % 16, kk = lane / 16) reads k = 16 j + 4 kk .. + 3 and feeds component t to the MFMA of k-step 4 j + t
// (the order of the k-steps of a dot product is free), and activation rows are pitched = 16 (mod 32) floats, which spreads
// the 64 sixteen-byte accesses of an instruction evenly over the banks.
constexpr int LM_TP = 32;

__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct LocalMfmaDims { int K1p, K2p, K3p, n1b, n2b, s1, sx, dbg; };

// one unit: features [16 nb, 16 nb + 16) x both position blocks; DUAL: the units nb and nb2 together (four independent
// accumulator chains: with a single wave per SIMD two chains leave the MFMA pipe half idle).  RELU output into Y (pitch ys)
// or, for the last layer, the logits.
template <bool LAST, bool DUAL>
__device__ __forceinline__ void mlp_unit(const float* __restrict__ A, int J, int nb, int nb2, const float* __restrict__ X, int xs,
                                         const float* __restrict__ bias, float* __restrict__ Y, int ys, int lane,
                                         float* __restrict__ out, int64_t row0, int64_t n, int nc) {
  const int n16 = lane & 15, kk = lane >> 4;
  const int f0 = 16 * nb + 4 * kk, f2 = 16 * nb2 + 4 * kk;
  f32x4 acc0 = ld4(bias + f0), acc1 = acc0;      // LDS copy of the bias, zero-padded to the block grid
  f32x4 acc2 = DUAL ? ld4(bias + f2) : splat(0.f), acc3 = acc2;
  const float* ap = A + ((size_t)nb * J * 64 + lane) * 4;
  const float* ap2 = A + ((size_t)nb2 * J * 64 + lane) * 4;
  const float* x0 = X + n16 * xs + 4 * kk;
  const float* x1 = x0 + 16 * xs;
  f32x4 a = ld4(ap), a2 = DUAL ? ld4(ap2) : splat(0.f), p0 = ld4(x0), p1 = ld4(x1);
  for (int j = 0; j < J; ++j) {          // operands of step j + 1 are in flight under the MFMAs of step j
    const int jn = j + 1 < J ? j + 1 : j;
    const f32x4 an = ld4(ap + jn * 256), p0n = ld4(x0 + 16 * jn), p1n = ld4(x1 + 16 * jn);
    f32x4 a2n = splat(0.f);
    if (DUAL) a2n = ld4(ap2 + jn * 256);
    __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the MFMAs (the scheduler would sink them to their use)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], p0[t], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], p1[t], acc1, 0, 0, 0);
      if (DUAL) {
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[t], p0[t], acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[t], p1[t], acc3, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    a = an;
    a2 = a2n;
    p0 = p0n;
    p1 = p1n;
  }
  if (!LAST) {
    st4(Y + n16 * ys + f0, max4(acc0, splat(0.f)));
    st4(Y + (16 + n16) * ys + f0, max4(acc1, splat(0.f)));
    if (DUAL) {
      st4(Y + n16 * ys + f2, max4(acc2, splat(0.f)));
      st4(Y + (16 + n16) * ys + f2, max4(acc3, splat(0.f)));
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (f0 + r < nc) {
        if (row0 + n16 < n) out[(row0 + n16) * nc + f0 + r] = acc0[r];
        if (row0 + 16 + n16 < n) out[(row0 + 16 + n16) * nc + f0 + r] = acc1[r];
      }
    }
  }
}

// the units u0, u0 + 4, ... < count of one layer for this wave, two at a time
__device__ __forceinline__ void mlp_layer(const float* __restrict__ A, int J, int u0, int count, const float* __restrict__ X, int xs,
                                          const float* __restrict__ bias, float* __restrict__ Y, int ys, int lane) {
  int u = u0;
  for (; u + 4 < count; u += 8) mlp_unit<false, true>(A, J, u, u + 4, X, xs, bias, Y, ys, lane, nullptr, 0, 0, 0);
  if (u < count) mlp_unit<false, false>(A, J, u, u, X, xs, bias, Y, ys, lane, nullptr, 0, 0, 0);
}

__global__ __launch_bounds__(LOC_THREADS) void snv_local_mlp_mfma(LocalDev L, const int64_t* __restrict__ cat, int64_t n,
                                                                  float* __restrict__ out, LocalMfmaDims d) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  const int J1 = d.K1p / 16, J2 = d.K2p / 16, J3 = d.K3p / 16;
  float* A1 = lsm;
  float* A2 = A1 + (size_t)d.n1b * J1 * 256;
  float* A3 = A2 + (size_t)d.n2b * J2 * 256;
  float* H1 = A3 + (size_t)J3 * 256;       // [LM_TP][s1]
  float* XH = H1 + LM_TP * d.s1;           // [LM_TP][sx]: embeddings, later the second hidden layer
  float* EM = XH + LM_TP * d.sx;           // the embedding table [emb_rows][5]
  {   // A1 | A2 | A3: eight loads in flight per thread (a one-tile call is this copy's latency: 40 dependent round trips otherwise)
    constexpr int UN = 8;
    const int n4 = L.frag_floats / 4;
    for (int i0 = threadIdx.x; i0 < n4; i0 += UN * LOC_THREADS) {
      f32x4 v[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = i0 + u * LOC_THREADS;
        v[u] = ld4(L.frag + 4 * (i < n4 ? i : i0));
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int i = i0 + u * LOC_THREADS;
        if (i < n4) st4(A1 + 4 * i, v[u]);
      }
    }
  }
  for (int i = threadIdx.x; i < L.emb_rows * 5; i += LOC_THREADS) EM[i] = L.emb[i];
  float* BI1 = EM + ((L.emb_rows * 5 + 3) & ~3);      // biases, zero-padded to whole 16-feature blocks
  float* BI2 = BI1 + d.K2p;
  float* BI3 = BI2 + d.K3p;
  for (int i = threadIdx.x; i < d.K2p; i += LOC_THREADS) BI1[i] = i < L.h1 ? L.b1[i] : 0.f;
  for (int i = threadIdx.x; i < d.K3p; i += LOC_THREADS) BI2[i] = i < L.h2 ? L.b2[i] : 0.f;
  if (threadIdx.x < 16) BI3[threadIdx.x] = (int)threadIdx.x < L.n_class ? L.b3[threadIdx.x] : 0.f;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n_tiles = (n + LM_TP - 1) / LM_TP;
  // k-mer ids of a tile: (position, column) tasks, LM_IDS per thread, fetched one tile ahead so that their global-memory
  // latency hides under the previous tile's layers
  constexpr int LM_IDS = 4;
  const int id_tasks = LM_TP * L.cols;
  int ids[LM_IDS];
  auto fetch_ids = [&](int64_t tile_) {
#pragma unroll
    for (int q = 0; q < LM_IDS; ++q) {
      const int task = threadIdx.x + q * LOC_THREADS;
      int v = -1;
      if (task < id_tasks && tile_ < n_tiles) {
        const int p = task / L.cols, col = task - p * L.cols;
        const int64_t row = tile_ * LM_TP + p;
        if (row < n) {
          const int64_t id = cat[row * L.cols + col];
          v = (int)(id < 0 ? 0 : (id >= L.emb_rows ? L.emb_rows - 1 : id));
        }
      }
      ids[q] = v;
    }
  };
  fetch_ids(blockIdx.x);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * LM_TP;
    __syncthreads();                       // fragments built / previous tile's last layer done with XH
#pragma unroll
    for (int q = 0; q < LM_IDS; ++q) {
      const int task = threadIdx.x + q * LOC_THREADS;
      if (task < id_tasks) {
        const int p = task / L.cols, col = task - p * L.cols;
        float* dst = XH + p * d.sx + 5 * col;
        const float* e = EM + 5 * (ids[q] < 0 ? 0 : ids[q]);
        const float live = ids[q] < 0 ? 0.f : 1.f;          // rows past the end of the batch read as zeros
        float v[5];
#pragma unroll
        for (int dd = 0; dd < 5; ++dd) v[dd] = e[dd];
#pragma unroll
        for (int dd = 0; dd < 5; ++dd) dst[dd] = v[dd] * live;
      }
    }
    for (int i = threadIdx.x; i < LM_TP * (d.K1p - L.in1); i += LOC_THREADS) {      // zero the K padding of the input rows
      const int p = i / (d.K1p - L.in1), k = L.in1 + i % (d.K1p - L.in1);
      XH[p * d.sx + k] = 0.f;
    }
    fetch_ids(tile + gridDim.x);
    lds_only_barrier();                    // (a __syncthreads() would also wait for the ids just requested)
    if (!(d.dbg & 2)) mlp_layer(A1, J1, wave, d.n1b, XH, d.sx, BI1, H1, d.s1, lane);
    lds_only_barrier();
    if (!(d.dbg & 4)) mlp_layer(A2, J2, 3 - wave, d.n2b, H1, d.s1, BI2, XH, d.sx, lane);
    lds_only_barrier();
    if (wave == 2 && !(d.dbg & 8)) mlp_unit<true, false>(A3, J3, 0, 0, XH, d.sx, BI3, nullptr, 0, lane, out, row0, n, L.n_class);
  }
}

int launch_snv_local(const LocalDev& L, const int64_t* cat, int64_t n, float* out, hipStream_t stream) {
  if (n == 0) return MURAL_OK;
  // MFMA path when the three fragment images and one tile of activations fit one CU's LDS (every shipped configuration)
  {
    LocalMfmaDims d;
    d.K1p = (L.in1 + 15) & ~15;
    d.n1b = (L.h1 + 15) / 16;
    d.K2p = 16 * d.n1b;
    d.n2b = (L.h2 + 15) / 16;
    d.K3p = 16 * d.n2b;
    auto pitch = [](int k) { return ((k + 31) & ~31) + 16; };     // = 16 (mod 32)
    d.s1 = pitch(d.K2p);
    d.sx = std::max(pitch(d.K1p), pitch(d.K3p));
    d.dbg = getenv("MURAL_DEBUG_MLP") ? atoi(getenv("MURAL_DEBUG_MLP")) : 0;
    const size_t floats = (size_t)256 * (d.n1b * (d.K1p / 16) + d.n2b * (d.K2p / 16) + d.K3p / 16) + (size_t)LM_TP * (d.s1 + d.sx) +
                          (size_t)((L.emb_rows * 5 + 3) & ~3) + d.K2p + d.K3p + 16;
    // (the choice must not depend on the batch size: results are bitwise independent of how a site list is chunked)
    if (L.n_class <= 16 && LM_TP * L.cols <= 4 * LOC_THREADS && floats * sizeof(float) <= 160 * 1024 && !getenv("MURAL_DEBUG_LOCAL_VALU")) {
      static DynLdsOnce big_lds;
      if (int rc = big_lds.ensure(&snv_local_mlp_mfma)) return rc;
      const int64_t n_tiles = (n + LM_TP - 1) / LM_TP;
      const int grid = (int)(n_tiles < 256 ? n_tiles : 256);
      hipLaunchKernelGGL(snv_local_mlp_mfma, dim3(grid), dim3(LOC_THREADS), floats * sizeof(float), stream, L, cat, n, out, d);
      MURAL_HIP_CHECK(hipGetLastError());
      return MURAL_OK;
    }
  }
  const int xs = (L.in1 + 3) & ~3, h1s = (L.h1 + 3) & ~3, h2s = (L.h2 + 3) & ~3;
  const size_t lds = (size_t)LOC_TP * (xs + h1s + h2s + SNV_MAXCLASS) * sizeof(float);
  MURAL_REQUIRE(lds <= 64 * 1024, "local branch too wide for the LDS tile (%zu bytes)", lds);
  const int64_t n_tiles = (n + LOC_TP - 1) / LOC_TP;
  const int grid = (int)(n_tiles < 4096 ? n_tiles : 4096);
  hipLaunchKernelGGL(snv_local_mlp, dim3(grid), dim3(LOC_THREADS), lds, stream, L, cat, n, out, xs, h1s, h2s);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

}  // namespace mural

// Local branch of the SNV models: shared k-mer embedding -> Linear/ReLU/BN x2 -> Linear (eval mode).
// Reference semantics: MuRaL/model/model_snv.py:451-468, :492 (Network2) and :74-93 (Network0).
// 0.6 % of the model's FLOPs.  Two kernels: snv_local_mlp_mfma (below; fp32 MFMA with all weights resident in LDS, used
// whenever they fit -- every shipped configuration) and snv_local_mlp, the plain fp32 VALU version for wider layers
// (activations of a 32-position tile in LDS, BN-folded transposed weights streamed from L2).
#include <algorithm>
#include <cstdlib>

#include "snv_local_mfma.h"

namespace mural {

constexpr int LOC_TP = 32;        // positions per workgroup tile
constexpr int LOC_PG = 8;         // positions per thread task

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
__global__ __launch_bounds__(LOC_THREADS) void snv_local_mlp_mfma(LocalDev L, const int64_t* __restrict__ cat, int64_t n,
                                                                  float* __restrict__ out, LocalMfmaDims d) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  local_mlp_mfma_body(L, cat, n, out, d, lsm, (int)blockIdx.x, (int)gridDim.x);
}

__global__ __launch_bounds__(LOC_THREADS, 2) void snv_local_mlp_reg(LocalDev L, const int64_t* __restrict__ cat, int64_t n,
                                                                    float* __restrict__ out, LocalMfmaDims d) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  local_mlp_reg_body(L, cat, n, out, d, lsm, (int)blockIdx.x, (int)gridDim.x);
}

bool local_mfma_plan(const LocalDev& L, LocalMfmaDims* dp, size_t* lds_bytes) {
  LocalMfmaDims d;
  d.K1p = (L.in1 + 15) & ~15;
  d.n1b = (L.h1 + 15) / 16;
  d.K2p = 16 * d.n1b;
  d.n2b = (L.h2 + 15) / 16;
  d.K3p = 16 * d.n2b;
  auto pitch = [](int k) { return ((k + 31) & ~31) + 16; };     // = 16 (mod 32)
  d.s1 = pitch(d.K2p);
  d.sx = std::max(pitch(d.K1p), pitch(d.K3p));
  d.dbg = dev_env("MURAL_DEBUG_MLP") ? atoi(dev_env("MURAL_DEBUG_MLP")) : 0;
  const size_t floats = (size_t)256 * (d.n1b * (d.K1p / 16) + d.n2b * (d.K2p / 16) + d.K3p / 16) + (size_t)LM_TP * (d.s1 + d.sx) +
                        (size_t)((L.emb_rows * 5 + 3) & ~3) + d.K2p + d.K3p + 16;
  *dp = d;
  *lds_bytes = floats * sizeof(float);
  // (the choice must not depend on the batch size: results are bitwise independent of how a site list is chunked)
  return L.n_class <= 16 && LM_TP * L.cols <= 4 * LOC_THREADS && floats * sizeof(float) <= 160 * 1024 && !dev_env("MURAL_DEBUG_LOCAL_VALU");
}

int launch_snv_local(const LocalDev& L, const int64_t* cat, int64_t n, float* out, hipStream_t stream) {
  if (n == 0) return MURAL_OK;
  // MFMA path when the three fragment images and one tile of activations fit one CU's LDS (every shipped configuration)
  {
    LocalMfmaDims d;
    size_t lds = 0;
    if (local_mfma_plan(L, &d, &lds)) {
      // the shipped dimensions: fragments in registers, two workgroups per CU (snv_local_mfma.h; MURAL_LOCAL_REG=0: fragments in LDS)
      const bool reg_off = dev_env("MURAL_LOCAL_REG") && atoi(dev_env("MURAL_LOCAL_REG")) == 0;
      if (!reg_off && d.K1p == 16 * LR_J1 && d.n1b == LR_N1B && d.K2p == 16 * LR_J2 && d.n2b == LR_N2B && d.K3p == 16 * LR_J3 && !d.dbg) {
        const size_t lds_reg = ((size_t)LM_TP * (d.s1 + d.sx) + (size_t)((L.emb_rows * 5 + 3) & ~3) + d.K2p + d.K3p + 16) * sizeof(float);
        const int64_t n_tiles = (n + LM_TP - 1) / LM_TP;
        const int grid = (int)(n_tiles < 512 ? n_tiles : 512);
        hipLaunchKernelGGL(snv_local_mlp_reg, dim3(grid), dim3(LOC_THREADS), lds_reg, stream, L, cat, n, out, d);
        MURAL_HIP_CHECK(hipGetLastError());
        return MURAL_OK;
      }
      static DynLdsOnce big_lds;
      if (int rc = big_lds.ensure(&snv_local_mlp_mfma)) return rc;
      const int64_t n_tiles = (n + LM_TP - 1) / LM_TP;
      const int grid = (int)(n_tiles < 256 ? n_tiles : 256);
      hipLaunchKernelGGL(snv_local_mlp_mfma, dim3(grid), dim3(LOC_THREADS), lds, stream, L, cat, n, out, d);
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

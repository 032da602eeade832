// Local branch of the SNV models on fp32 MFMA: device code shared by snv_local_mlp_mfma (snv_local.hip) and the small-batch
// first-stage launch (snv_stage1.hip).
#pragma once
#include "mfma_tile.h"
#include "snv.h"

namespace mural {

constexpr int LOC_THREADS = 256;
constexpr int LM_TP = 32;

__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

// The kernel body, for the first LOC_THREADS threads of a workgroup (any others must have left the kernel: the barriers below
// count live waves only): workgroup `block` of `nblocks`.  snv_local_mlp_mfma (snv_local.hip) is this; the small-batch first-stage
// launch runs it in one extra workgroup beside its site workgroups (snv_stage1.hip), so that a 16-site call does not wait for
// a launch of its own.
__device__ __forceinline__ void local_mlp_mfma_body(const LocalDev& L, const int64_t* __restrict__ cat, int64_t n,
                                                    float* __restrict__ out, const LocalMfmaDims& d, float* lsm, int block,
                                                    int nblocks) {
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
  fetch_ids(block);
  for (int64_t tile = block; tile < n_tiles; tile += nblocks) {
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
    fetch_ids(tile + nblocks);
    lds_only_barrier();                    // (a __syncthreads() would also wait for the ids just requested)
    if (!(d.dbg & 2)) mlp_layer(A1, J1, wave, d.n1b, XH, d.sx, BI1, H1, d.s1, lane);
    lds_only_barrier();
    if (!(d.dbg & 4)) mlp_layer(A2, J2, 3 - wave, d.n2b, H1, d.s1, BI2, XH, d.sx, lane);
    lds_only_barrier();
    if (wave == 2 && !(d.dbg & 8)) mlp_unit<true, false>(A3, J3, 0, 0, XH, d.sx, BI3, nullptr, 0, lane, out, row0, n, L.n_class);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same MLP with the A fragments of a wave's units in REGISTERS (the shipped 95 -> 150 -> 75 -> n_class branch: J = 6 / 10 / 5
// sixteen-wide k chunks, 10 / 5 / 1 feature blocks).  With the three fragment images in LDS (118 KB) a CU holds ONE four-wave
// workgroup -- one wave per SIMD, every MFMA chain waiting for its own ds_read_b128 -- and the launch took 122 us per 100 k sites for
// 33 us of MFMA time.  Here LDS carries the activations alone (38 KB), two workgroups share a CU and a step's A operand is already
// there.  Same units per wave, same k order per accumulator as local_mlp_mfma_body: the results are bitwise the same.
constexpr int LR_J1 = 6, LR_N1B = 10, LR_J2 = 10, LR_N2B = 5, LR_J3 = 5;

template <int J, bool LAST, bool DUAL>
__device__ __forceinline__ void mlp_unit_reg(const f32x4 (&a)[J], const f32x4 (&a2)[J], int nb, int nb2, const float* __restrict__ X, int xs,
                                             const float* __restrict__ bias, float* __restrict__ Y, int ys, int lane,
                                             float* __restrict__ out, int64_t row0, int64_t n, int nc) {
  const int n16 = lane & 15, kk = lane >> 4;
  const int f0 = 16 * nb + 4 * kk, f2 = 16 * nb2 + 4 * kk;
  f32x4 acc0 = ld4(bias + f0), acc1 = acc0;
  f32x4 acc2 = DUAL ? ld4(bias + f2) : splat(0.f), acc3 = acc2;
  const float* x0 = X + n16 * xs + 4 * kk;
  const float* x1 = x0 + 16 * xs;
  f32x4 p0 = ld4(x0), p1 = ld4(x1);
#pragma unroll
  for (int j = 0; j < J; ++j) {          // the activations of step j + 1 are in flight under the MFMAs of step j
    const int jn = j + 1 < J ? j + 1 : j;
    const f32x4 p0n = ld4(x0 + 16 * jn), p1n = ld4(x1 + 16 * jn);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][t], p0[t], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][t], p1[t], acc1, 0, 0, 0);
      if (DUAL) {
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[j][t], p0[t], acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[j][t], p1[t], acc3, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
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

__device__ __forceinline__ void local_mlp_reg_body(const LocalDev& L, const int64_t* __restrict__ cat, int64_t n,
                                                   float* __restrict__ out, const LocalMfmaDims& d, float* lsm, int block, int nblocks) {
  float* H1 = lsm;                         // [LM_TP][s1]
  float* XH = H1 + LM_TP * d.s1;           // [LM_TP][sx]: embeddings, later the second hidden layer
  float* EM = XH + LM_TP * d.sx;           // the embedding table [emb_rows][5]
  for (int i = threadIdx.x; i < L.emb_rows * 5; i += LOC_THREADS) EM[i] = L.emb[i];
  float* BI1 = EM + ((L.emb_rows * 5 + 3) & ~3);      // biases, zero-padded to whole 16-feature blocks
  float* BI2 = BI1 + d.K2p;
  float* BI3 = BI2 + d.K3p;
  for (int i = threadIdx.x; i < d.K2p; i += LOC_THREADS) BI1[i] = i < L.h1 ? L.b1[i] : 0.f;
  for (int i = threadIdx.x; i < d.K3p; i += LOC_THREADS) BI2[i] = i < L.h2 ? L.b2[i] : 0.f;
  if (threadIdx.x < 16) BI3[threadIdx.x] = (int)threadIdx.x < L.n_class ? L.b3[threadIdx.x] : 0.f;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // this wave's units (as mlp_layer deals them): layer 1 units wave, wave + 4, wave + 8 (< 10); layer 2 units 3 - wave, 7 - wave (< 5);
  // layer 3 on wave 2.  Fragment image of L.frag: A1 [n1b][J1][64 lanes][4] | A2 [n2b][J2][64][4] | A3 [J3][64][4]
  const f32x4* F1 = reinterpret_cast<const f32x4*>(L.frag);
  const f32x4* F2 = F1 + LR_N1B * LR_J1 * 64;
  const f32x4* F3 = F2 + LR_N2B * LR_J2 * 64;
  f32x4 a1[3][LR_J1], a2[2][LR_J2], a3[LR_J3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int u = wave + 4 * q;
#pragma unroll
    for (int j = 0; j < LR_J1; ++j) a1[q][j] = u < LR_N1B ? F1[(u * LR_J1 + j) * 64 + lane] : splat(0.f);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int u = 3 - wave + 4 * q;
#pragma unroll
    for (int j = 0; j < LR_J2; ++j) a2[q][j] = u < LR_N2B ? F2[(u * LR_J2 + j) * 64 + lane] : splat(0.f);
  }
#pragma unroll
  for (int j = 0; j < LR_J3; ++j) a3[j] = wave == 2 ? F3[j * 64 + lane] : splat(0.f);
  const int64_t n_tiles = (n + LM_TP - 1) / LM_TP;
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
  fetch_ids(block);
  for (int64_t tile = block; tile < n_tiles; tile += nblocks) {
    const int64_t row0 = tile * LM_TP;
    __syncthreads();                       // tables built / previous tile's last layer done with XH
#pragma unroll
    for (int q = 0; q < LM_IDS; ++q) {
      const int task = threadIdx.x + q * LOC_THREADS;
      if (task < id_tasks) {
        const int p = task / L.cols, col = task - p * L.cols;
        float* dst = XH + p * d.sx + 5 * col;
        const float* e = EM + 5 * (ids[q] < 0 ? 0 : ids[q]);
        const float live = ids[q] < 0 ? 0.f : 1.f;
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
    fetch_ids(tile + nblocks);
    lds_only_barrier();
    // layer 1: units (wave, wave + 4) together, then wave + 8 alone (waves 0 and 1)
    mlp_unit_reg<LR_J1, false, true>(a1[0], a1[1], wave, wave + 4, XH, d.sx, BI1, H1, d.s1, lane, nullptr, 0, 0, 0);
    if (wave + 8 < LR_N1B) mlp_unit_reg<LR_J1, false, false>(a1[2], a1[2], wave + 8, wave + 8, XH, d.sx, BI1, H1, d.s1, lane, nullptr, 0, 0, 0);
    lds_only_barrier();
    // layer 2: unit 3 - wave (and 7 - wave = 4 on wave 3, as a pair)
    if (7 - wave < LR_N2B) mlp_unit_reg<LR_J2, false, true>(a2[0], a2[1], 3 - wave, 7 - wave, H1, d.s1, BI2, XH, d.sx, lane, nullptr, 0, 0, 0);
    else mlp_unit_reg<LR_J2, false, false>(a2[0], a2[0], 3 - wave, 3 - wave, H1, d.s1, BI2, XH, d.sx, lane, nullptr, 0, 0, 0);
    lds_only_barrier();
    if (wave == 2) mlp_unit_reg<LR_J3, true, false>(a3, a3, 0, 0, XH, d.sx, BI3, nullptr, 0, lane, out, row0, n, L.n_class);
  }
}

// host: LDS layout of the kernel for this model; false when the weights do not fit one CU's LDS (snv_local.hip)
bool local_mfma_plan(const LocalDev& L, LocalMfmaDims* d, size_t* lds_bytes);

}  // namespace mural

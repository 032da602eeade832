// Wave-private form of the channel-last 32->32 k=3 conv kernels of the composed SNV training step (gfx950 / CDNA4).
//
// Reference semantics: nn.BatchNorm1d -> nn.Conv1d of MuRaL/model/model_snv.py:350-430 and the ResBlock of :794-812 under model.train()
// (training.py:424), and their gradients under loss.backward() (training.py:427) -- the same math as conv32_cl.hip, which stays the
// kernel of rows too long for a wave's image and the A/B reference (MURAL_TRAIN_CONV_CL=1).  What changes is who owns what, the recipe
// of the prediction kernel (snv_tower_wave.hip):
//
//   * a WAVE owns whole batch rows: a unit is P rows of L columns on a flattened column axis with zero separators (<= 9 blocks of
//     16 columns), all 32 output channels.  There is no workgroup barrier in the unit loop; the phases of conv32_cl.hip (stage,
//     MFMA, stream-out: strictly additive there) overlap inside the wave instead.
//   * ONE wave per SIMD (4 per CU) with 512 registers and two LDS images each:
//       forward : the images double-buffer the units.  While unit u runs its 48 MFMAs per block, the rows of unit u+1 (requested at
//                 the start of unit u) arrive in registers, take ReLU + BatchNorm a few 16-byte pieces per block and are written to
//                 the other image; outputs go to memory straight from the accumulators (lane = column, 16 bytes = 4 channels) with
//                 the residual operands loaded two blocks ahead in the same layout; the batch sums of act(y) for the next BatchNorm
//                 ride in the epilogue bursts.
//       backward: the images hold dy and xhat = (act(x) - mean) * invstd.  The whole next unit (dy, x) travels in registers under
//                 the two MFMA phases: weight gradient (dW~ += dy (x) xhat, K = columns, 8-byte operand reads) and input gradient
//                 (the forward conv with the transposed, tap-flipped filter on the dy image; dz leaves from the accumulators).
//                 Keeping xhat instead of BN(act(x)) in the image makes sum(dz * xhat) a read of that image in accumulator layout
//                 (the workgroup-tile kernel re-reads x from memory for it) and costs one fix-up per partial row:
//                     dW[co][ci][t] = gamma[ci] * dW~[co][ci][t] + beta[ci] * S_t[co],
//                 S_t = sum of dy over the columns whose tap-t input is not zero padding (S_1 = bias gradient, S_0 / S_2 leave out
//                 the first / last column of every row), because the zero padding is applied behind the BatchNorm.
//   * units are tickets of an atomic counter (zero at launch), requested a unit ahead.
//   * out-of-range work never branches: every global access goes through a per-unit buffer descriptor whose num_records is the
//     bytes of the rows that exist; lanes of separator / padding columns carry an offset no descriptor covers (loads return 0,
//     stores are dropped).
#include <cstdlib>
#include <cstring>

#include "conv32_cl.h"

namespace mural {
namespace {

constexpr int CW_NBMAX = 9;                 // 16-column blocks per unit at most
constexpr uint32_t CW_BLK = 2048u;          // bytes between blocks of an image (16 columns x 128 B; the swizzle key has period 16)
constexpr uint32_t CW_OOB = 0x80000000u;    // lane offset outside every unit descriptor
constexpr int CW_DUMP = 128;                // floats behind each image of a wave: 32 16-byte dump slots (where the lanes of staging pieces
                                            // behind the unit store; slot + image stride is the second image's slot)
constexpr int CW_AUX = 256;                 // floats in front of the waves' regions (BatchNorm constants, final sums)
constexpr int CW_CUS = 256;

struct CwGeom {
  int L, Sc, P, nb;
  int nchunk;              // 16-byte pieces of a whole unit: P * L * 8
  FastDiv dL, dSc;
};

bool cw_geom(int64_t B, int L, CwGeom* g) {
  std::memset(g, 0, sizeof(*g));
  if (L < 1) return false;
  const int Sc = L + 1;
  const int pmax = (16 * CW_NBMAX - 1) / Sc;
  if (pmax < 1) return false;
  int64_t p = B / (4 * CW_CUS);             // one unit per wave when the rows are short
  if (p > pmax) p = pmax;
  if (p < 1) p = 1;
  g->L = L;
  g->Sc = Sc;
  g->P = (int)p;
  g->nb = (1 + g->P * Sc + 15) / 16;
  g->nchunk = g->P * L * 8;
  g->dL = FastDiv::make((uint32_t)L);
  g->dSc = FastDiv::make((uint32_t)Sc);
  return true;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t cw_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void buf_st4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 0);
}
__device__ __forceinline__ f32x4 pk_fma(const f32x4& a, const f32x4& b, const f32x4& c) {
  const f32x2 lo = __builtin_elementwise_fma(f32x2{a.x, a.y}, f32x2{b.x, b.y}, f32x2{c.x, c.y});
  const f32x2 hi = __builtin_elementwise_fma(f32x2{a.z, a.w}, f32x2{b.z, b.w}, f32x2{c.z, c.w});
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}

// byte offset inside a unit of this lane's 16 bytes of block b in accumulator layout (column n16 of the block, channels 4 kk .. + 3
// of M-block 0; M-block 1 sits 64 bytes further); CW_OOB for separator / padding columns
__device__ __forceinline__ uint32_t cw_acc_offset(const CwGeom& g, int b, int n16, int kk) {
  const uint32_t c = 16u * (uint32_t)b + (uint32_t)n16;
  const uint32_t u = c - 1u;
  const uint32_t p = g.dSc.div(u);
  const uint32_t j = u - p * (uint32_t)g.Sc;
  const bool ok = c >= 1u && p < (uint32_t)g.P && j < (uint32_t)g.L;
  return ok ? (((p * (uint32_t)g.L + j) << 7) + 16u * (uint32_t)kk) : CW_OOB;
}

// staging slot u of a lane: piece lane + 64 u of the unit in memory order (byte offset 16 lane + 1024 u: pieces behind the unit fall
// outside its descriptor) -> byte offset in the image
__device__ __forceinline__ void cw_stage_slot(const CwGeom& g, int u, int lane, uint32_t dump, uint32_t* soff) {
  const uint32_t task = (uint32_t)lane + 64u * (uint32_t)u;
  const uint32_t col = task >> 3;
  const uint32_t p = g.dL.div(col);
  const uint32_t l = col - p * (uint32_t)g.L;
  const bool ok = task < (uint32_t)g.nchunk;
  *soff = ok ? 4u * (uint32_t)lds_off(2 + (int)(p * (uint32_t)g.Sc + l), (int)(task & 7u)) : dump;
}

#define CW_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)

// one conv tap (8 k-steps) of a block for both M-blocks: the two accumulator chains alternate pair by pair, `burst` (vector / LDS /
// memory instructions of the pipeline around the MFMAs) sits behind the first pair; a scheduling barrier closes every pair (left alone
// the backend serialises each chain: 8 dependent MFMAs in a row cost 40 instead of 32 cycles each)
template <int T, class F>
__device__ __forceinline__ void cw_tap(const float (&a0)[SNV_KSTEPS], const float (&a1)[SNV_KSTEPS], const f32x4 (&bv)[2], f32x4& acc0,
                                       f32x4& acc1, F&& burst) {
#define CW_PAIR(I)                                                 \
  acc0 = CW_MFMA(a0[8 * T + (I)], bv[(I) >> 2][(I) & 3], acc0);    \
  acc1 = CW_MFMA(a1[8 * T + (I)], bv[(I) >> 2][(I) & 3], acc1);    \
  if constexpr ((I) == 0) burst();                                 \
  __builtin_amdgcn_sched_barrier(0);
  CW_PAIR(0)
  CW_PAIR(1)
  CW_PAIR(2)
  CW_PAIR(3)
  CW_PAIR(4)
  CW_PAIR(5)
  CW_PAIR(6)
  CW_PAIR(7)
#undef CW_PAIR
}

// NB blocks of a 32->32 k=3 conv on a read-only image: slot(b) runs behind the first MFMA pair of block b, epi(b, m, acc) -- the
// epilogue of M-block m of block b -- behind the first pairs of taps 1 / 2 of block b + 1 (the last block's epilogues follow the loop)
template <int NB, class Slot, class Epi>
__device__ __forceinline__ void cw_conv_blocks(const char* img, const uint32_t (&rd)[6], const float (&a0)[SNV_KSTEPS],
                                               const float (&a1)[SNV_KSTEPS], const f32x4 (&pb)[2], Slot&& slot, Epi&& epi) {
  f32x4 X[2], Y[2], Z[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    X[h] = lds_ld4(img, rd[h]);
    Y[h] = lds_ld4(img, rd[2 + h]);
    Z[h] = lds_ld4(img, rd[4 + h]);
  }
  f32x4 pa0 = splat(0.f), pa1 = splat(0.f);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    f32x4 acc0 = pb[0], acc1 = pb[1];
    __builtin_amdgcn_sched_barrier(0);
    cw_tap<0>(a0, a1, X, acc0, acc1, [&]() __attribute__((always_inline)) { slot(b); });
    if (b + 1 < NB) {
#pragma unroll
      for (int h = 0; h < 2; ++h) X[h] = lds_ld4(img, rd[h] + CW_BLK * (b + 1));
    }
    __builtin_amdgcn_sched_barrier(0);
    cw_tap<1>(a0, a1, Y, acc0, acc1, [&]() __attribute__((always_inline)) {
      if (b > 0) epi(b > 0 ? b - 1 : 0, 0, pa0);
    });
    if (b + 1 < NB) {
#pragma unroll
      for (int h = 0; h < 2; ++h) Y[h] = lds_ld4(img, rd[2 + h] + CW_BLK * (b + 1));
    }
    __builtin_amdgcn_sched_barrier(0);
    cw_tap<2>(a0, a1, Z, acc0, acc1, [&]() __attribute__((always_inline)) {
      if (b > 0) epi(b > 0 ? b - 1 : 0, 1, pa1);
    });
    if (b + 1 < NB) {
#pragma unroll
      for (int h = 0; h < 2; ++h) Z[h] = lds_ld4(img, rd[4 + h] + CW_BLK * (b + 1));
    }
    pa0 = acc0;
    pa1 = acc1;
    __builtin_amdgcn_sched_barrier(0);
  }
  epi(NB - 1, 0, pa0);
  epi(NB - 1, 1, pa1);
}

// sums held per lane in accumulator layout (column n16, channels 16 m + 4 kk .. + 3) -> the workgroup's accumulator slot: lanes of a
// kk group meet through shuffles, the four waves through `red` ([4 waves][NV][32] floats), NV * 32 double atomics per workgroup
template <int NV>
__device__ __forceinline__ void cw_acc_slot_add(const f32x4 (&v)[NV][2], double* slot, float* red, int tid) {
  const int lane = tid & 63, wave = tid >> 6, n16 = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float s = v[i][m][q];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) s += __shfl_xor(s, off);
        if (n16 == 0) red[(wave * NV + i) * CL_C + 16 * m + 4 * kk + q] = s;
      }
  __syncthreads();
  if (tid < NV * CL_C) {
    const int i = tid >> 5, c = tid & 31;
    const float t = (red[(0 * NV + i) * CL_C + c] + red[(1 * NV + i) * CL_C + c]) + (red[(2 * NV + i) * CL_C + c] + red[(3 * NV + i) * CL_C + c]);
    atomicAdd(&slot[i * CL_C + c], (double)t);
  }
}

// ------------------------------------------------------------------------------------------------------------ forward
struct CwFwdArgs {
  CwGeom g;
  const float* x;
  float* y;
  const float* W;
  const float* bias;
  const float* res1;
  const float* res2;
  ClFin fin;
  int pre_relu;
  double* stat_out;       // batch sums of relu(y), relu(y)^2 for the next BatchNorm; nullptr: none
  int64_t B;
  int64_t n_units;
  int* counter;           // unit tickets (zero at launch); nullptr: fixed stride
  int dbg;                // timing experiments (MURAL_DEBUG_CW): 1 no next-unit loads, 2 no stores, 4 no residual loads, 8 no conv
};

template <int NB>
__global__ __launch_bounds__(SNV_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv32w_fwd_kernel(const CwFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int IMG_FLOATS = (16 * NB + 2) * CL_C;
  constexpr uint32_t IMG_BYTES = (IMG_FLOATS + CW_DUMP) * 4u;   // stride between the two images: image | dump slots
  constexpr int WAVE_FLOATS = 2 * (IMG_FLOATS + CW_DUMP);
  constexpr int NLD = 2 * NB;                                   // staging slots per lane: 64 NLD >= pieces of the widest unit
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kk = lane >> 4, chunk = lane & 7;
  const CwGeom& g = a.g;
  float* aux = smem;                                            // scale | beta | mean
  float* wbase = smem + CW_AUX + wave * WAVE_FLOATS;
  char* wb = reinterpret_cast<char*>(wbase);
  cl_finalize(a.fin, aux, reinterpret_cast<double*>(smem + CW_AUX), tid);      // (ends with a workgroup barrier)
  const f32x4 s4 = ld4(aux + 4 * chunk), t4 = ld4(aux + CL_C + 4 * chunk), m4 = ld4(aux + 2 * CL_C + 4 * chunk);
  for (int i = lane; i < WAVE_FLOATS / 4; i += 64) st4(wbase + 4 * i, splat(0.f));      // gap columns stay zero for the launch
  float a0[SNV_KSTEPS], a1[SNV_KSTEPS];
  cl_frags(a.W, 0, 0, n16, kk, a0);
  cl_frags(a.W, 0, 1, n16, kk, a1);
  f32x4 pb[2];
  pb[0] = a.bias ? ld4(a.bias + 4 * kk) : splat(0.f);
  pb[1] = a.bias ? ld4(a.bias + 16 + 4 * kk) : splat(0.f);
  uint32_t rd[6];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) rd[2 * t + h] = 4u * (uint32_t)lds_off(n16 + t, 4 * h + kk);
  uint32_t vo[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) vo[b] = cw_acc_offset(g, b, n16, kk);
  const uint32_t dump = IMG_FLOATS * 4u + 16u * (uint32_t)(lane & 31);
  uint32_t so[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) cw_stage_slot(g, u, lane, dump, &so[u]);
  const uint32_t lane16 = 16u * (uint32_t)lane;
  const float lo_pre = a.pre_relu ? 0.f : -INFINITY;
  const uint32_t row_bytes = (uint32_t)g.L * 128u;
  const size_t unit_stride = (size_t)g.P * g.L * CL_C;          // floats

  const bool dyn = a.counter != nullptr;
  const int64_t unit_step = (int64_t)gridDim.x * SNV_WAVES;
  int ticket = 0;
  int64_t unit = (int64_t)blockIdx.x * SNV_WAVES + wave;
  if (dyn) {
    if (lane == 0) ticket = atomicAdd(a.counter, 1);
    unit = __builtin_amdgcn_readfirstlane(ticket);
    if (lane == 0) ticket = atomicAdd(a.counter, 1);
  }
  auto unit_bytes = [&](int64_t u) -> uint32_t {               // bytes of the rows of unit u that exist
    if (u >= a.n_units) return 0u;
    const int64_t rows = a.B - u * g.P;
    return (uint32_t)(rows < g.P ? rows : g.P) * row_bytes;
  };
  auto stage = [&](const f32x4& raw, uint32_t off) __attribute__((always_inline)) {
    const f32x4 v = max4(raw, splat(lo_pre));
    lds_st4(wb, off, pk_fma(s4, v - m4, t4));
  };
  f32x4 xin[NLD];
  {
    const __amdgpu_buffer_rsrc_t xd = cw_rsrc(a.x + (size_t)(unit < a.n_units ? unit : 0) * unit_stride, unit_bytes(unit));
#pragma unroll
    for (int u = 0; u < NLD; ++u) xin[u] = buf_ld4(xd, lane16 + 1024u * u);
#pragma unroll
    for (int u = 0; u < NLD; ++u) stage(xin[u], so[u]);
  }
  uint32_t cur = 0u;                                            // byte offset of the current unit's image in the wave's region
  f32x4 sum[2][2];                                              // [act(y) | act(y)^2][M-block]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int m = 0; m < 2; ++m) sum[i][m] = splat(0.f);
  constexpr int BS = NB > 2 ? 2 : NB - 1;                       // first block that consumes staged pieces of the next unit

  for (int64_t next = 0; unit < a.n_units; unit = next) {
    if (dyn) {
      next = __builtin_amdgcn_readfirstlane(ticket);
      if (lane == 0) ticket = atomicAdd(a.counter, 1);
    } else {
      next = unit + unit_step;
    }
    const uint32_t ub = unit_bytes(unit);
    const size_t ubase = (size_t)unit * unit_stride;
    const __amdgpu_buffer_rsrc_t yd = cw_rsrc(a.y + ubase, (a.dbg & 2) ? 0u : ub);
    const __amdgpu_buffer_rsrc_t r1d = cw_rsrc((a.res1 ? a.res1 : a.x) + ubase, (a.res1 && !(a.dbg & 4)) ? ub : 0u);
    const __amdgpu_buffer_rsrc_t r2d = cw_rsrc((a.res2 ? a.res2 : a.x) + ubase, (a.res2 && !(a.dbg & 4)) ? ub : 0u);
    const __amdgpu_buffer_rsrc_t nxd =
        cw_rsrc(a.x + (size_t)(next < a.n_units ? next : 0) * unit_stride, (a.dbg & 1) ? 0u : unit_bytes(next));
    f32x4 R1[NB][2], R2[NB][2];
#pragma unroll
    for (int b = 0; b < 2 && b < NB; ++b)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        R1[b][m] = buf_ld4(r1d, vo[b] + 64u * m);
        R2[b][m] = buf_ld4(r2d, vo[b] + 64u * m);
      }
#pragma unroll
    for (int u = 0; u < NLD; ++u) xin[u] = buf_ld4(nxd, lane16 + 1024u * u);
    uint32_t rdc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) rdc[i] = rd[i] + cur;
    const uint32_t nxt = IMG_BYTES - cur;
    __builtin_amdgcn_sched_barrier(0);
    if (a.dbg & 8) {
#pragma unroll
      for (int u = 0; u < NLD; ++u) stage(xin[u], so[u] + nxt);
    } else
    cw_conv_blocks<NB>(
        wb, rdc, a0, a1, pb,
        [&](int b) __attribute__((always_inline)) {
          if (b + 2 < NB) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              R1[b + 2 < NB ? b + 2 : 0][m] = buf_ld4(r1d, vo[b + 2 < NB ? b + 2 : 0] + 64u * m);
              R2[b + 2 < NB ? b + 2 : 0][m] = buf_ld4(r2d, vo[b + 2 < NB ? b + 2 : 0] + 64u * m);
            }
          }
          if (b >= BS) {                                        // this block's share of the next unit's pieces -> the other image
#pragma unroll
            for (int u = 0; u < NLD; ++u)
              if (BS + (u * (NB - BS)) / NLD == b) stage(xin[u], so[u] + nxt);
          }
        },
        [&](int b, int m, const f32x4& acc) __attribute__((always_inline)) {
          const f32x4 y = (acc + R1[b][m]) + R2[b][m];
          buf_st4(yd, vo[b] + 64u * m, y);
          const f32x4 w = max4(y, splat(0.f));
          const f32x4 mk = splat(vo[b] < ub ? 1.f : 0.f);
          sum[0][m] = pk_fma(w, mk, sum[0][m]);
          sum[1][m] = pk_fma(w * w, mk, sum[1][m]);
        });
    cur = nxt;
  }
  if (a.stat_out) {
    __syncthreads();                                            // every wave is past its last unit: the images are dead
    cw_acc_slot_add<2>(sum, a.stat_out + (size_t)(blockIdx.x % MURAL_BN_SLOTS) * 2 * CL_C, smem, tid);
  }
}

// ------------------------------------------------------------------------------------------------------------ backward
struct CwBwdArgs {
  CwGeom g;
  const float* dy;
  const float* x;
  const float* W;
  const float* state;     // scale | beta | mean | invstd of the BatchNorm in front of the conv
  const float* gamma;
  int pre_relu;
  float* part;            // [grid][32*32*3 + 32]
  float* dz;
  double* stat_out;       // sum(dz), sum(dz * xhat)
  int64_t B;
  int64_t n_units;
  int* counter;
  int dbg;                // timing experiments (MURAL_DEBUG_CW): 16 no next-unit loads, 32 no weight gradient, 64 no input gradient, 128 no stores
};

template <int NB>
__global__ __launch_bounds__(SNV_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv32w_bwd_kernel(const CwBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int IMG_FLOATS = (16 * NB + 2) * CL_C;
  constexpr uint32_t IMG_BYTES = (IMG_FLOATS + CW_DUMP) * 4u;   // stride between the two images: image | dump slots
  constexpr int WAVE_FLOATS = 2 * (IMG_FLOATS + CW_DUMP);
  constexpr int NLD = 2 * NB;
  constexpr int NW = CL_C * CL_C * 3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kk = lane >> 4, chunk = lane & 7;
  const CwGeom& g = a.g;
  float* wbase = smem + CW_AUX + wave * WAVE_FLOATS;            // dy image | xhat image | dump
  char* wb = reinterpret_cast<char*>(wbase);
  for (int i = lane; i < WAVE_FLOATS / 4; i += 64) st4(wbase + 4 * i, splat(0.f));
  float a0[SNV_KSTEPS], a1[SNV_KSTEPS];
  cl_frags(a.W, 1, 0, n16, kk, a0);
  cl_frags(a.W, 1, 1, n16, kk, a1);
  const f32x4 mean4 = ld4(a.state + 2 * CL_C + 4 * chunk), inv4 = ld4(a.state + 3 * CL_C + 4 * chunk);
  uint32_t rd[6], wr[2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) rd[2 * t + h] = 4u * (uint32_t)lds_off(n16 + t, 4 * h + kk);
#pragma unroll
  for (int m = 0; m < 2; ++m) wr[m] = IMG_BYTES + 4u * (uint32_t)lds_off(n16 + 1, 4 * m + kk);
  // weight gradient: k-step s = 4 q + j covers the logical columns 16 q + 4 j + kk; a lane reads channels 2 n16, 2 n16 + 1 (8 bytes)
  // of dy at image column pc + 1 and of xhat at image columns pc + tap
  uint32_t wa[4][3];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int tp = 0; tp < 3; ++tp) wa[j][tp] = 4u * (uint32_t)(lds_off(4 * j + kk + tp, n16 >> 1) + 2 * (n16 & 1));
  uint32_t vo[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) vo[b] = cw_acc_offset(g, b, n16, kk);
  const uint32_t dump = IMG_FLOATS * 4u + 16u * (uint32_t)(lane & 31);
  uint32_t so[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) cw_stage_slot(g, u, lane, dump, &so[u]);
  const uint32_t lane16 = 16u * (uint32_t)lane;
  const float lo_pre = a.pre_relu ? 0.f : -INFINITY;
  const uint32_t row_bytes = (uint32_t)g.L * 128u;
  const size_t unit_stride = (size_t)g.P * g.L * CL_C;

  const bool dyn = a.counter != nullptr;
  const int64_t unit_step = (int64_t)gridDim.x * SNV_WAVES;
  int ticket = 0;
  int64_t unit = (int64_t)blockIdx.x * SNV_WAVES + wave;
  if (dyn) {
    if (lane == 0) ticket = atomicAdd(a.counter, 1);
    unit = __builtin_amdgcn_readfirstlane(ticket);
    if (lane == 0) ticket = atomicAdd(a.counter, 1);
  }
  auto unit_bytes = [&](int64_t u) -> uint32_t {
    if (u >= a.n_units) return 0u;
    const int64_t rows = a.B - u * g.P;
    return (uint32_t)(rows < g.P ? rows : g.P) * row_bytes;
  };
  f32x4 gin[NLD], xin[NLD];
  {
    const size_t o = (size_t)(unit < a.n_units ? unit : 0) * unit_stride;
    const uint32_t ub = unit_bytes(unit);
    const __amdgpu_buffer_rsrc_t gd = cw_rsrc(a.dy + o, ub), xd = cw_rsrc(a.x + o, ub);
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      gin[u] = buf_ld4(gd, lane16 + 1024u * u);
      xin[u] = buf_ld4(xd, lane16 + 1024u * u);
    }
  }
  f32x4 wacc[2][3][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int tp = 0; tp < 3; ++tp)
#pragma unroll
      for (int h = 0; h < 2; ++h) wacc[m][tp][h] = splat(0.f);
  f32x4 sum[2][2];                                              // [dz | dz * xhat][M-block]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int m = 0; m < 2; ++m) sum[i][m] = splat(0.f);
  f32x4 bsum = splat(0.f), esum = splat(0.f);                   // staging layout: channels 4 chunk .. + 3; esum: first (lane bit 3 = 0) / last column
  const f32x4 pb[2] = {splat(0.f), splat(0.f)};
  const int edge_last = (lane >> 3) & 1;

  for (int64_t next = 0; unit < a.n_units; unit = next) {
    if (dyn) {
      next = __builtin_amdgcn_readfirstlane(ticket);
      if (lane == 0) ticket = atomicAdd(a.counter, 1);
    } else {
      next = unit + unit_step;
    }
    // ---- this unit's rows: registers -> images (dy as it is, x as xhat); the bias gradient rides along
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      bsum += gin[u];
      lds_st4(wb, so[u], gin[u]);
      const f32x4 v = max4(xin[u], splat(lo_pre));
      lds_st4(wb, so[u] + IMG_BYTES, (v - mean4) * inv4);
    }
    // ---- the next unit's rows travel under the MFMA phases
    {
      const size_t o = (size_t)(next < a.n_units ? next : 0) * unit_stride;
      const uint32_t nb_ = (a.dbg & 16) ? 0u : unit_bytes(next);
      const __amdgpu_buffer_rsrc_t gd = cw_rsrc(a.dy + o, nb_), xd = cw_rsrc(a.x + o, nb_);
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        gin[u] = buf_ld4(gd, lane16 + 1024u * u);
        xin[u] = buf_ld4(xd, lane16 + 1024u * u);
      }
    }
    const uint32_t ub = unit_bytes(unit);
    const __amdgpu_buffer_rsrc_t zd = cw_rsrc(a.dz + (size_t)unit * unit_stride, (a.dbg & 128) ? 0u : ub);
    // ---- first / last column of every row (S_0 / S_2 of the weight-gradient fix-up)
    for (int t = lane; t < g.P * 16; t += 64) {
      const int row = t >> 4;
      esum += ld4(wbase + lds_off(2 + row * g.Sc + (edge_last ? g.L - 1 : 0), chunk));
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- weight gradient
    if (!(a.dbg & 32)) {
#pragma unroll
      for (int q = 0; q < NB; ++q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x2 gv = *reinterpret_cast<const f32x2*>(wb + wa[j][1] + CW_BLK * q);
          f32x2 bv[3];
#pragma unroll
          for (int tp = 0; tp < 3; ++tp) bv[tp] = *reinterpret_cast<const f32x2*>(wb + IMG_BYTES + wa[j][tp] + CW_BLK * q);
#pragma unroll
          for (int tp = 0; tp < 3; ++tp)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              wacc[0][tp][h] = CW_MFMA(gv[0], bv[tp][h], wacc[0][tp][h]);
              wacc[1][tp][h] = CW_MFMA(gv[1], bv[tp][h], wacc[1][tp][h]);
            }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- input gradient: the forward conv with the transposed, tap-flipped filter on the dy image
    if (!(a.dbg & 64)) {
      cw_conv_blocks<NB>(
          wb, rd, a0, a1, pb, [&](int) __attribute__((always_inline)) {},
          [&](int b, int m, const f32x4& acc) __attribute__((always_inline)) {
            buf_st4(zd, vo[b] + 64u * m, acc);
            const f32x4 xh = lds_ld4(wb, wr[m] + CW_BLK * b);
            const f32x4 mk = splat(vo[b] < ub ? 1.f : 0.f);
            sum[0][m] = pk_fma(acc, mk, sum[0][m]);
            sum[1][m] = pk_fma(acc, xh, sum[1][m]);
          });
    }
  }
  // ---- BatchNorm-backward sums
  __syncthreads();                                              // every wave is past its last unit
  cw_acc_slot_add<2>(sum, a.stat_out + (size_t)(blockIdx.x % MURAL_BN_SLOTS) * 2 * CL_C, smem, tid);
  // ---- partial row of the weight / bias gradient: every wave parks its tiles and edge sums in its own region
  // D[row 4 kk + r][col n16] of tile (m, tap, h) <-> dW~[co = 2 (4 kk + r) + m][ci = 2 n16 + h][tap]
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int tp = 0; tp < 3; ++tp)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) wbase[((2 * (4 * kk + r) + m) * CL_C + 2 * n16 + h) * 3 + tp] = wacc[m][tp][h][r];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v = bsum[q], e = esum[q];
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) v += __shfl_xor(v, off);
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) e += __shfl_xor(e, off);
    if (lane < 8) wbase[NW + 4 * chunk + q] = v;
    if (lane < 16) wbase[NW + CL_C + edge_last * CL_C + 4 * chunk + q] = e;      // S_1 - S_0 (first columns) | S_1 - S_2 (last columns)
  }
  __syncthreads();
  float* fin = smem;                                            // bias gradient | first-column sums | last-column sums of the workgroup
  if (tid < 3 * CL_C) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < SNV_WAVES; ++w) t += smem[CW_AUX + w * WAVE_FLOATS + NW + tid];
    fin[tid] = t;
  }
  __syncthreads();
  float* dst = a.part + (size_t)blockIdx.x * (NW + CL_C);
  for (int i = tid; i < NW + CL_C; i += SNV_THREADS) {
    if (i < NW) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < SNV_WAVES; ++w) t += smem[CW_AUX + w * WAVE_FLOATS + i];
      const int co = i / (3 * CL_C), rem = i - co * 3 * CL_C, ci = rem / 3, tp = rem - 3 * ci;
      const float S = fin[co] - (tp == 0 ? fin[CL_C + co] : tp == 2 ? fin[2 * CL_C + co] : 0.f);
      dst[i] = a.gamma[ci] * t + a.state[CL_C + ci] * S;
    } else {
      dst[i] = fin[i - NW];
    }
  }
}

int cw_debug() {
  static const int v = getenv("MURAL_DEBUG_CW") ? atoi(getenv("MURAL_DEBUG_CW")) : 0;
  return v;
}

template <int NB>
int cw_launch_fwd(const CwFwdArgs& a, int grid, hipStream_t stream) {
  constexpr size_t lds = (size_t)(CW_AUX + SNV_WAVES * 2 * ((16 * NB + 2) * CL_C + CW_DUMP)) * 4;
  static DynLdsOnce big_lds;
  if (int rc = big_lds.ensure(&conv32w_fwd_kernel<NB>)) return rc;
  hipLaunchKernelGGL(conv32w_fwd_kernel<NB>, dim3(grid), dim3(SNV_THREADS), lds, stream, a);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

template <int NBV>
int cw_launch_bwd(const CwBwdArgs& a, int grid, hipStream_t stream) {
  constexpr int NB = NBV < 3 ? 3 : NBV;      // (a wave's region also holds its partial tiles: three blocks at least)
  constexpr size_t lds = (size_t)(CW_AUX + SNV_WAVES * 2 * ((16 * NB + 2) * CL_C + CW_DUMP)) * 4;
  static_assert(2 * ((16 * NB + 2) * CL_C + CW_DUMP) >= CL_C * CL_C * 3 + 3 * CL_C, "a wave's region holds its partial tiles");
  static DynLdsOnce big_lds;
  if (int rc = big_lds.ensure(&conv32w_bwd_kernel<NB>)) return rc;
  hipLaunchKernelGGL(conv32w_bwd_kernel<NB>, dim3(grid), dim3(SNV_THREADS), lds, stream, a);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

#define CW_DISPATCH(FN, NBV, ...)                 \
  switch (NBV) {                                  \
    case 2: return FN<2>(__VA_ARGS__);            \
    case 3: return FN<3>(__VA_ARGS__);            \
    case 4: return FN<4>(__VA_ARGS__);            \
    case 5: return FN<5>(__VA_ARGS__);            \
    case 6: return FN<6>(__VA_ARGS__);            \
    case 7: return FN<7>(__VA_ARGS__);            \
    case 8: return FN<8>(__VA_ARGS__);            \
    default: return FN<9>(__VA_ARGS__);           \
  }

int cw_grid(int64_t n_units) {
  const int64_t wgs = (n_units + SNV_WAVES - 1) / SNV_WAVES;
  return (int)(wgs < CW_CUS ? wgs : CW_CUS);
}

}  // namespace

// ---- host entry points (snv_train.hip) --------------------------------------------------------------------------------------
int cw_conv32_supported(int L) {
  CwGeom g;
  return cw_geom(2, L, &g) ? 1 : 0;
}

int cw_conv32_fwd(const float* x, int64_t B, int L, int pre_relu, const double* acc, const float* gamma, const float* beta, float eps,
                  float momentum, float* running_mean, float* running_var, float* state, const float* W, const float* bias, int post_relu,
                  const float* res1, const float* res2, double* acc_out, int out_relu, float* y, int* counter, hipStream_t stream) {
  if (B == 0 || L == 0) return MURAL_OK;
  CwFwdArgs a;
  std::memset(&a, 0, sizeof(a));
  MURAL_REQUIRE(cw_geom(B, L, &a.g), "conv32 (wave-private): L = %d does not fit a wave's image", L);
  if (!res1) { res1 = res2; res2 = nullptr; }
  MURAL_REQUIRE(!post_relu && (out_relu || !acc_out), "conv32 (wave-private): raw output with the batch sums of relu(y) only");
  a.x = x; a.y = y; a.W = W; a.bias = bias; a.res1 = res1; a.res2 = res2; a.pre_relu = pre_relu;
  a.stat_out = acc_out;
  a.fin = ClFin{acc, (double)B * L, gamma, beta, eps, momentum, running_mean, running_var, state};
  a.B = B;
  a.n_units = (B + a.g.P - 1) / a.g.P;
  a.counter = counter;
  a.dbg = cw_debug();
  CW_DISPATCH(cw_launch_fwd, a.g.nb, a, cw_grid(a.n_units), stream)
}

int cw_conv32_bwd(const float* dy, const float* x, const float* W, int64_t B, int L, const float* state, const float* gamma, int pre_relu,
                  float* dz, double* stat_out, float* part, int* nrow, int* counter, hipStream_t stream) {
  CwBwdArgs a;
  std::memset(&a, 0, sizeof(a));
  MURAL_REQUIRE(cw_geom(B, L, &a.g), "conv32_bwd (wave-private): L = %d does not fit a wave's image", L);
  a.dy = dy; a.x = x; a.W = W; a.state = state; a.gamma = gamma; a.pre_relu = pre_relu; a.part = part; a.dz = dz; a.stat_out = stat_out;
  a.B = B;
  a.n_units = (B + a.g.P - 1) / a.g.P;
  a.counter = counter;
  a.dbg = cw_debug();
  const int grid = cw_grid(a.n_units);
  *nrow = grid;
  CW_DISPATCH(cw_launch_bwd, a.g.nb, a, grid, stream)
}

}  // namespace mural

// ---- validation hooks (tests/test_gpu_train.py, tools/gpu_debug_conv32_cl.py): the wave-private conv kernels on their own -----
extern "C" int mural_debug_cw_conv32_fwd(const float* x, int64_t B, int32_t L, int32_t pre_relu, const double* acc, const float* gamma,
                                         const float* beta, float* running_mean, float* running_var, float* state, const float* W,
                                         const float* bias, int32_t post_relu, const float* res1, const float* res2, double* acc_out,
                                         int32_t out_relu, float* y, int32_t* counter, void* stream) {
  return mural::cw_conv32_fwd(x, B, L, pre_relu, acc, gamma, beta, 1e-5f, 0.1f, running_mean, running_var, state, W, bias, post_relu, res1, res2,
                              acc_out, out_relu, y, counter, (hipStream_t)stream);
}

extern "C" int mural_debug_cw_conv32_bwd(const float* dy, const float* x, const float* W, int64_t B, int32_t L, const float* state,
                                         const float* gamma, int32_t pre_relu, float* dz, double* stat_out, float* part, int32_t* nrow,
                                         int32_t* counter, void* stream) {
  int n = 0;
  const int rc = mural::cw_conv32_bwd(dy, x, W, B, L, state, gamma, pre_relu, dz, stat_out, part, &n, counter, (hipStream_t)stream);
  if (nrow) *nrow = n;
  return rc;
}

// Wave-private form of the channel-last 32->32 k=3 conv kernels of the composed SNV training step (gfx950 / CDNA4).
//
// Reference semantics: nn.BatchNorm1d -> nn.Conv1d of MuRaL/model/model_snv.py:350-430 and the ResBlock of :794-812 under model.train()
// (training.py:424), and their gradients under loss.backward() (training.py:427) -- the same math as conv32_cl.hip, which stays the
// kernel of rows too long for a wave's image, of the one layer with a ReLU behind it (conv3) and the A/B reference
// (MURAL_TRAIN_CONV_CL=1).  What changes is who owns what, the recipe of the prediction kernel (snv_tower_wave.hip):
//
//   * a WAVE owns whole batch rows: a unit is P rows of L columns on a flattened column axis with zero separators (<= 9 blocks of
//     16 columns), all 32 output channels, ONE LDS image [column][32 channels] (<= 18.7 KB).  Eight waves per CU = two per SIMD: there
//     is no workgroup barrier in the unit loop, the waves drift apart and one wave's memory phase (request its rows, wait, ReLU +
//     BatchNorm, write the image) runs under its SIMD partner's MFMAs.  (First form of this file: one wave per SIMD with two images
//     and the next unit travelling in registers under the MFMAs -- 55 / 86 us per 4096 x 134 layer against 60 / 101 us of the
//     workgroup-tile kernels: with a lone wave per SIMD every vector instruction and every MFMA -> VALU switch costs MFMA issue.)
//   * outputs leave straight from the accumulators (lane = column, 16 bytes = 4 channels; the two M-blocks complete a 128-byte row),
//     residual operands arrive two blocks ahead in the same layout; nothing is written back to the image.
//   * forward: the batch sums of relu(y) for the next BatchNorm ride in the epilogue bursts.
//   * backward: ONE image (dy) serves both gradients.
//       input gradient : the forward conv with the transposed, tap-flipped filter; its epilogue is a store and nothing else.
//       weight gradient: dWr[co][ci][t] += dy[col - t + 1][co] * r[col][ci], K = columns; the dy operand of the three taps comes from
//                        the image (8-byte reads of channels 2 n16, 2 n16 + 1), r = act(x) straight from memory in operand layout
//                        (8 bytes per lane, twelve k-steps ahead), NOT normalised: columns without data read as zero and stay zero.
//       The BatchNorm algebra moves to the partial row of the workgroup (all of it is linear in dWr and in three vectors of dy sums):
//           S_t[co]      = sum of dy over the columns whose tap-t partner is not zero padding = S - (first columns | 0 | last columns)
//           dWx          = invstd[ci] * (dWr - mean[ci] * S_t[co])                  (= sum dy (x) xhat)
//           dW           = gamma[ci] * dWx + beta[ci] * S_t[co]                     (the zero padding sits behind the BatchNorm)
//           sum dz       = sum_{co,t} W[co][ci][t] * S_t[co]
//           sum dz*xhat  = sum_{co,t} W[co][ci][t] * dWx[co][ci][t]                 (exact identities: dz = conv^T(dy))
//       so the BatchNorm-backward sums cost no pass over dz and no second image.
//   * a workgroup is four waves with <= 80 KB of LDS, so two of them -- of one launch or of two launches on different streams (the
//     two towers) -- share a CU.  Its waves walk the workgroup's contiguous share of the units with a fixed stride: the assignment
//     decides the order of the float sums, a fixed one keeps the step bitwise reproducible (one global ticket counter for 4096 units
//     of 10 us also serialises in L2: measured 98 instead of 55 us per launch).
//   * out-of-range work never branches: every global access goes through a per-unit buffer descriptor whose num_records is the
//     bytes of the rows that exist; lanes of separator / padding columns carry an offset no descriptor covers (loads return 0,
//     stores are dropped).
#include <cstdlib>
#include <cstring>

#include "conv32_cl.h"
#include "conv32_jobs.h"

namespace mural {
unsigned long long* g_cw_stamps = nullptr;      // diagnostic (mural_debug_cw_set_stamps)
namespace {

constexpr int CW_NBMAX = 9;                 // 16-column blocks per unit at most
constexpr int CW_THREADS = 256;         // four waves; two workgroups share a CU (<= 80 KB of LDS each), possibly of two launches
constexpr int CW_WAVES = CW_THREADS / 64;
constexpr uint32_t CW_BLK = 2048u;          // bytes between blocks of an image (16 columns x 128 B; the swizzle key has period 16)
constexpr uint32_t CW_OOB = 0x80000000u;    // lane offset outside every unit descriptor
constexpr int CW_DUMP = 128;                // floats behind a wave's image: 32 16-byte dump slots (staging pieces behind the unit)
constexpr int CW_AUX = 512;                 // floats in front of the waves' regions: BatchNorm constants | offset table
constexpr int CW_AUX_TAB = 128;             // uint32[4 * CW_NBMAX][4]: memory offset of column 4 s + kk of a unit (weight gradient)
constexpr int CW_CUS = 256;
constexpr int CW_LA = 12;                   // k-steps the weight gradient's x operand is requested ahead

struct CwGeom {
  int L, Sc, P, nb;
  int nchunk;              // 16-byte pieces of a whole unit: P * L * 8
  FastDiv dL, dSc;
};

// Workgroup slots a launch asks for.  A CU holds two of these workgroups whatever launch they belong to, and the step's two towers
// run on two streams: a launch that takes all 512 slots only time-shares the chip with its neighbour, and every workgroup pays a
// prologue (fragments, BatchNorm finalisation) and an epilogue (batch sums / the partial row) for one unit per wave.  Rows short
// enough to share a unit (P >= 2) ask for ONE slot per CU -- four rows per wave in the short stages, two units per wave in the mid
// tower's first stage -- and leave the other to the other tower; rows that need a unit each keep both (two units per wave at batch
// 4096).  Same-box A/B of the whole step, batch 4096: 498 -> 528 steps/s (tools/r6_train_slots.sh; MURAL_CW_FULL_GRID=1 is the old
// rule).  The choice changes the order of the float sums, not their reproducibility.
int cw_slots(int L) {
  static const bool full = dev_env("MURAL_CW_FULL_GRID") && atoi(dev_env("MURAL_CW_FULL_GRID")) != 0;
  const int pmax = (16 * CW_NBMAX - 1) / (L + 1);
  return (full || pmax < 2) ? 2 * CW_CUS : CW_CUS;
}

bool cw_geom(int64_t B, int L, CwGeom* g) {
  std::memset(g, 0, sizeof(*g));
  if (L < 1) return false;
  const int Sc = L + 1;
  const int pmax = (16 * CW_NBMAX - 1) / Sc;
  if (pmax < 1) return false;
  int64_t p = B / (CW_WAVES * cw_slots(L));  // one unit per wave when the rows are short
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
// (soff: a scalar offset inside the lane's own 128-byte row -- whether a lane is in range is decided by its lane offset alone)
__device__ __forceinline__ void buf_st4s(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, uint32_t soff, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, soff, 0);
}
__device__ __forceinline__ f32x4 buf_ld4s(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, uint32_t soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, soff, 0));
}
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;
__device__ __forceinline__ f32x2 buf_ld2(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
}
__device__ __forceinline__ f32x4 pk_fma(const f32x4& a, const f32x4& b, const f32x4& c) {
  const f32x2 lo = __builtin_elementwise_fma(f32x2{a.x, a.y}, f32x2{b.x, b.y}, f32x2{c.x, c.y});
  const f32x2 hi = __builtin_elementwise_fma(f32x2{a.z, a.w}, f32x2{b.z, b.w}, f32x2{c.z, c.w});
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}

// Staging-slot addresses without a register per slot.  Slot u of a lane is the 16-byte piece at byte 16 lane + 1024 u of the unit; the
// twelve-bit immediate of a buffer instruction reaches 4095, so a lane keeps ONE offset per group of four slots (16 lane + 4096 k,
// made from an opaque copy of the lane offset inside the unit loop: a hoisted table of 18 per-slot offsets was what spilled, and
// their reloads -- one s_waitcnt vmcnt(0) each -- serialised the unit's loads) and the slot inside the group rides in the immediate.
template <int NLD>
struct CwSlots {
  uint32_t g[(NLD + 3) / 4];
  __device__ __forceinline__ void make(uint32_t lane16) {
    uint32_t l = lane16;
    asm volatile("" : "+v"(l));
#pragma unroll
    for (int k = 0; k < (NLD + 3) / 4; ++k) g[k] = l + 4096u * k;
  }
  __device__ __forceinline__ uint32_t at(int u) const { return g[u >> 2] + 1024u * (uint32_t)(u & 3); }
};
// a loop-carried sum must be complete here: left alone the IR sinks the adds of a unit's staging to the loop latch and keeps every
// staged value (72 registers at nine blocks) alive through both MFMA phases
__device__ __forceinline__ void cw_pin(f32x4& v) { asm volatile("" : "+v"(v)); }

// byte offset inside a unit of the 128-byte row of logical column c; CW_OOB for separator / padding columns
__device__ __forceinline__ uint32_t cw_col_offset(const CwGeom& g, uint32_t c) {
  const uint32_t u = c - 1u;
  const uint32_t p = g.dSc.div(u);
  const uint32_t j = u - p * (uint32_t)g.Sc;
  const bool ok = c >= 1u && p < (uint32_t)g.P && j < (uint32_t)g.L;
  return ok ? ((p * (uint32_t)g.L + j) << 7) : CW_OOB;
}

// staging slot u of a lane: piece lane + 64 u of the unit in memory order (byte offset 16 lane + 1024 u: pieces behind the unit fall
// outside its descriptor) -> byte offset in the image
__device__ __forceinline__ uint32_t cw_stage_slot(const CwGeom& g, int u, int lane, uint32_t dump) {
  const uint32_t task = (uint32_t)lane + 64u * (uint32_t)u;
  const uint32_t col = task >> 3;
  const uint32_t p = g.dL.div(col);
  const uint32_t l = col - p * (uint32_t)g.L;
  return task < (uint32_t)g.nchunk ? 4u * (uint32_t)lds_off(2 + (int)(p * (uint32_t)g.Sc + l), (int)(task & 7u)) : dump;
}

// filter fragments (conv32_cl.h: cl_frags) from a copy of W [32][32][3] in LDS (rows of 96 floats at a pitch of 97: the 16 lanes of a
// read sit 96 floats apart, i.e. in one bank): the strided gather from memory touches 64 cache lines per load and instruction
constexpr int CW_WPITCH = 97;
constexpr int CW_WFLOATS = CL_C * CW_WPITCH;
__device__ __forceinline__ void cw_copy_w(const float* __restrict__ W, float* Wl, int tid) {
  for (int i = tid; i < CL_C * CL_C * 3 / 4; i += CW_THREADS) {
    const f32x4 v = ld4(W + 4 * i);
    const int row = (4 * i) / 96, col = (4 * i) - 96 * row;
    float* d = Wl + row * CW_WPITCH + col;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}
__device__ __forceinline__ void cw_frags_lds(const float* Wl, int dgrad, int mb, int n16, int kk, float (&a)[SNV_KSTEPS]) {
#pragma unroll
  for (int s = 0; s < SNV_KSTEPS; ++s) {
    const int t = s / 8, h = (s % 8) / 4, q = s % 4;
    const int cin = 16 * h + 4 * kk + q, cout = 16 * mb + n16;
    a[s] = Wl[dgrad ? cin * CW_WPITCH + cout * 3 + (2 - t) : cout * CW_WPITCH + cin * 3 + t];
  }
}

// The same fragments, written once per step for all conv layers of the step by cw_wfrag_kernel: [dir: forward | input gradient]
// [mb][g][lane][4] floats = k-steps 4 g .. 4 g + 3 of a lane side by side, so that a wave reads a layer's fragments with twelve
// coalesced 16-byte loads per lane and the prologue of a conv launch has no LDS copy of W and no barrier for it
constexpr int CW_WFRAG = 2 * SNV_KSTEPS * 64;      // floats per (layer, direction)
struct CwFragJobs {
  const float* W[24];
  float* out;             // [n][2][CW_WFRAG]
  int n;
};
__global__ __launch_bounds__(256) void cw_wfrag_kernel(const CwFragJobs jobs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= jobs.n * 2 * CW_WFRAG) return;
  const int q = i & 3, lane = (i >> 2) & 63, g = (i >> 8) % 6, mb = (i >> 8) / 6 % 2, dir = (i / CW_WFRAG) & 1, layer = i / (2 * CW_WFRAG);
  const int s = 4 * g + q, t = s / 8, h = (s % 8) / 4;
  const int n16 = lane & 15, kk = lane >> 4;
  const int cin = 16 * h + 4 * kk + q, cout = 16 * mb + n16;
  jobs.out[i] = jobs.W[layer][dir ? (cin * CL_C + cout) * 3 + (2 - t) : (cout * CL_C + cin) * 3 + t];
}
__device__ __forceinline__ void cw_frags_global(const float* __restrict__ frag, int lane, float (&a0)[SNV_KSTEPS], float (&a1)[SNV_KSTEPS]) {
#pragma unroll
  for (int g = 0; g < 6; ++g) {
    const f32x4 u = ld4(frag + (size_t)g * 256 + 4 * lane), v = ld4(frag + (size_t)(6 + g) * 256 + 4 * lane);
    a0[4 * g] = u.x; a0[4 * g + 1] = u.y; a0[4 * g + 2] = u.z; a0[4 * g + 3] = u.w;
    a1[4 * g] = v.x; a1[4 * g + 1] = v.y; a1[4 * g + 2] = v.z; a1[4 * g + 3] = v.w;
  }
}

#define CW_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)

// one conv tap (8 k-steps) of a block for both M-blocks: the two accumulator chains alternate pair by pair, `burst` (vector / LDS /
// memory instructions around the MFMAs) sits behind the first pair; a scheduling barrier closes every pair (left alone the backend
// serialises each chain: 8 dependent MFMAs in a row cost 40 instead of 32 cycles each)
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

// the workgroup's contiguous share of the units; its waves walk it with a fixed stride (which wave computes a unit decides the order
// of the float sums in its partial row: a fixed assignment keeps a step bitwise reproducible)
struct CwUnits {
  int64_t lo, hi;
};
__device__ __forceinline__ void cw_units_init(CwUnits& w, int64_t n_units, int grid) {
  const int64_t per = (n_units + grid - 1) / grid;
  w.lo = (int64_t)blockIdx.x * per;
  w.hi = w.lo + per < n_units ? w.lo + per : n_units;
}

// 32 slots x 2 sums of one channel set, summed by ONE wave without LDS: lane = (which sum, channel) loads its 32 slots (independent
// loads, one round trip), lane c then takes the second sum from lane c + 32
__device__ __forceinline__ void cw_wave_slot_sums(const double* __restrict__ acc, int lane, double* s1, double* s2) {
  const int c = lane & 31, which = lane >> 5;
  double v[MURAL_BN_SLOTS];
#pragma unroll
  for (int k = 0; k < MURAL_BN_SLOTS; ++k) v[k] = acc[((size_t)k * 2 + which) * CL_C + c];
  double t[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < MURAL_BN_SLOTS; ++k) t[k & 3] += v[k];
  const double s = (t[0] + t[1]) + (t[2] + t[3]);
  const double o = __shfl(s, lane ^ 32);
  *s1 = which ? o : s;
  *s2 = which ? s : o;
}

// the BatchNorm finalisation of conv32_cl.h (cl_finalize) by the first wave of a workgroup alone -- no scratch, no barrier of its
// own: the caller's one prologue barrier publishes aux (scale | beta | mean)
__device__ __forceinline__ void cw_finalize_wave0(const ClFin& f, float* aux, int lane) {
  double s1, s2;
  cw_wave_slot_sums(f.acc, lane, &s1, &s2);
  if (lane < CL_C) {
    const double mean = s1 / f.n;
    double var = s2 / f.n - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)f.eps);
    const float sc = (float)(f.gamma[lane] * invstd);
    aux[lane] = sc;
    aux[CL_C + lane] = f.beta[lane];
    aux[2 * CL_C + lane] = (float)mean;
    if (blockIdx.x == 0) {
      f.state[lane] = sc;
      f.state[CL_C + lane] = f.beta[lane];
      f.state[2 * CL_C + lane] = (float)mean;
      f.state[3 * CL_C + lane] = (float)invstd;
      if (f.running_mean) {
        const double unbiased = f.n > 1.0 ? var * f.n / (f.n - 1.0) : var;
        f.running_mean[lane] = (float)((1.0 - f.momentum) * f.running_mean[lane] + f.momentum * mean);
        f.running_var[lane] = (float)((1.0 - f.momentum) * f.running_var[lane] + f.momentum * unbiased);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------ forward
struct CwFwdArgs {
  CwGeom g;
  const float* x;
  float* y;
  const float* W;
  const float* wfrag;     // the forward fragments of W (cw_wfrag_kernel) or nullptr: gathered from a copy of W in LDS
  const float* bias;
  const float* res1;
  const float* res2;
  ClFin fin;
  int pre_relu;
  double* stat_out;       // batch sums of relu(y), relu(y)^2 for the next BatchNorm; nullptr: none
  int64_t B;
  int64_t n_units;
  int grid;               // workgroups of this job (blockIdx.x beyond it: nothing to do)
  int dbg;                // timing experiments (MURAL_DEBUG_CW): 1 no loads of x, 2 no stores, 4 no residual loads, 8 no conv
  unsigned long long* stamps;      // diagnostic (mural_debug_cw_set_stamps): [workgroup][4] wall-clock ticks (100 MHz) at entry, behind the
                                   // prologue, behind the unit loop, at exit
};
struct CwFwdArgs2 { CwFwdArgs j[TOWER_JOBS]; };

template <int NB>
__global__ __launch_bounds__(CW_THREADS, 2) void conv32w_fwd_kernel(const CwFwdArgs2 aa) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const CwFwdArgs& a = aa.j[blockIdx.y];
  if ((int)blockIdx.x >= a.grid) return;
  if (a.stamps && threadIdx.x == 0) a.stamps[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memrealtime();
  constexpr int IMG_FLOATS = (16 * NB + 2) * CL_C;
  constexpr int WAVE_FLOATS = IMG_FLOATS + CW_DUMP;
  constexpr int NLD = 2 * NB;                                   // staging slots per lane: 64 NLD >= pieces of the widest unit
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kk = lane >> 4, chunk = lane & 7;
  const CwGeom& g = a.g;
  float* aux = smem;                                            // scale | beta | mean
  float* wbase = smem + CW_AUX + wave * WAVE_FLOATS;
  char* wb = reinterpret_cast<char*>(wbase);
  CwUnits units;
  cw_units_init(units, a.n_units, a.grid);
  const uint32_t lane16 = 16u * (uint32_t)lane;
  const uint32_t row_bytes = (uint32_t)g.L * 128u;
  const size_t unit_stride = (size_t)g.P * g.L * CL_C;          // floats
  auto unit_bytes = [&](int64_t u) -> uint32_t {               // bytes of the rows of unit u that exist
    const int64_t rows = a.B - u * g.P;
    return u < units.hi ? (uint32_t)(rows < g.P ? rows : g.P) * row_bytes : 0u;
  };
  int64_t unit = units.lo + wave;
  // prologue with ONE barrier: fragment loads in flight; the first wave finalises the BatchNorm (32 slots per sum, one round trip)
  // while every wave clears its own image and writes its share of the staging-offset table
  float a0[SNV_KSTEPS], a1[SNV_KSTEPS];
  if (a.wfrag) {
    cw_frags_global(a.wfrag, lane, a0, a1);
  } else {      // (validation hooks without the per-step fragments: gathered from a copy of W in LDS)
    cw_copy_w(a.W, smem + CW_AUX + 4096, tid);
    __syncthreads();
    cw_frags_lds(smem + CW_AUX + 4096, 0, 0, n16, kk, a0);
    cw_frags_lds(smem + CW_AUX + 4096, 0, 1, n16, kk, a1);
    __syncthreads();
  }
  if (wave == 0) cw_finalize_wave0(a.fin, aux, lane);
  for (int i = lane; i < WAVE_FLOATS / 4; i += 64) st4(wbase + 4 * i, splat(0.f));      // gap columns stay zero for the launch
  const uint32_t dump = IMG_FLOATS * 4u + 16u * (uint32_t)(lane & 31);
  // image offsets of the staging slots: a 16-bit table [slot][lane] behind the waves' regions (18 registers per lane otherwise)
  const uint16_t* sotab = reinterpret_cast<const uint16_t*>(smem + CW_AUX + CW_WAVES * WAVE_FLOATS) + lane;
  for (int u = wave; u < NLD; u += CW_WAVES) const_cast<uint16_t*>(sotab)[64 * u] = (uint16_t)cw_stage_slot(g, u, lane, dump);
  __syncthreads();
  const f32x4 s4 = ld4(aux + 4 * chunk), t4 = ld4(aux + CL_C + 4 * chunk), m4 = ld4(aux + 2 * CL_C + 4 * chunk);
  if (a.stamps && tid == 0) a.stamps[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
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
  for (int b = 0; b < NB; ++b) {
    const uint32_t o = cw_col_offset(g, 16u * b + (uint32_t)n16);
    vo[b] = o == CW_OOB ? CW_OOB : o + 16u * (uint32_t)kk;
  }
  const float lo_pre = a.pre_relu ? 0.f : -INFINITY;
  f32x4 sum[2][2];                                              // [relu(y) | relu(y)^2][M-block]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int m = 0; m < 2; ++m) sum[i][m] = splat(0.f);

  while (unit < units.hi) {
    const uint32_t ub = unit_bytes(unit);
    const size_t ubase = (size_t)unit * unit_stride;
    const __amdgpu_buffer_rsrc_t xd = cw_rsrc(a.x + ubase, (a.dbg & 1) ? 0u : ub);
    const __amdgpu_buffer_rsrc_t yd = cw_rsrc(a.y + ubase, (a.dbg & 2) ? 0u : ub);
    const __amdgpu_buffer_rsrc_t r1d = cw_rsrc((a.res1 ? a.res1 : a.x) + ubase, (a.res1 && !(a.dbg & 4)) ? ub : 0u);
    const __amdgpu_buffer_rsrc_t r2d = cw_rsrc((a.res2 ? a.res2 : a.x) + ubase, (a.res2 && !(a.dbg & 4)) ? ub : 0u);
    // ---- this unit's rows: memory -> ReLU, BatchNorm -> image (the wave's previous unit is past its last operand read)
    {      // (requesting the first unit's rows in front of the prologue keeps 72 registers live across it: spills in the block loop)
      f32x4 xin[NLD];
      CwSlots<NLD> so;
      so.make(lane16);
#pragma unroll
      for (int u = 0; u < NLD; ++u) xin[u] = buf_ld4(xd, so.at(u));
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        const f32x4 v = max4(xin[u], splat(lo_pre));
        lds_st4(wb, sotab[64 * u], pk_fma(s4, v - m4, t4));
      }
    }
    f32x4 R1[NB][2], R2[NB][2];
#pragma unroll
    for (int b = 0; b < 2 && b < NB; ++b)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        R1[b][m] = buf_ld4s(r1d, vo[b], 64u * m);
        R2[b][m] = buf_ld4s(r2d, vo[b], 64u * m);
      }
    __builtin_amdgcn_sched_barrier(0);
    if (!(a.dbg & 8))
      cw_conv_blocks<NB>(
          wb, rd, a0, a1, pb,
          [&](int b) __attribute__((always_inline)) {
            if (b + 2 < NB) {
#pragma unroll
              for (int m = 0; m < 2; ++m) {
                R1[b + 2 < NB ? b + 2 : 0][m] = buf_ld4s(r1d, vo[b + 2 < NB ? b + 2 : 0], 64u * m);
                R2[b + 2 < NB ? b + 2 : 0][m] = buf_ld4s(r2d, vo[b + 2 < NB ? b + 2 : 0], 64u * m);
              }
            }
          },
          [&](int b, int m, const f32x4& acc) __attribute__((always_inline)) {
            const f32x4 y = (acc + R1[b][m]) + R2[b][m];
            buf_st4s(yd, vo[b], 64u * m, y);
            const f32x4 w = max4(y, splat(0.f));
            const f32x4 mk = splat(vo[b] < ub ? 1.f : 0.f);
            sum[0][m] = pk_fma(w, mk, sum[0][m]);
            sum[1][m] = pk_fma(w * w, mk, sum[1][m]);
          });
    unit += CW_WAVES;
  }
  if (a.stamps && tid == 0) a.stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
  if (a.stat_out) {
    // sums held per lane in accumulator layout (column n16, channels 16 m + 4 kk .. + 3): the lanes of a kk group meet through
    // shuffles, the four waves through LDS, 64 double atomics per workgroup (one set per WAVE -- no barrier -- was tried: four times
    // the atomics on the same 2048 addresses cost the short-row launches 4 us each)
    __syncthreads();                                            // every wave is past its last unit: the images are dead
    float* red = smem + CW_AUX;                                 // [4 waves][2][32]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v = sum[i][m][q];
#pragma unroll
          for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off);
          if (n16 == 0) red[(wave * 2 + i) * CL_C + 16 * m + 4 * kk + q] = v;
        }
    __syncthreads();
    if (tid < 2 * CL_C) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < CW_WAVES; ++w) t += red[w * 2 * CL_C + tid];
      atomicAdd(&a.stat_out[(size_t)(blockIdx.x % MURAL_BN_SLOTS) * 2 * CL_C + tid], (double)t);
    }
  }
  if (a.stamps && tid == 0) a.stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
}

// ------------------------------------------------------------------------------------------------------------ backward
struct CwBwdArgs {
  CwGeom g;
  const float* dy;
  const float* x;
  const float* W;
  const float* wfrag;     // the input-gradient fragments of W (cw_wfrag_kernel) or nullptr
  const float* state;     // scale | beta | mean | invstd of the BatchNorm in front of the conv
  const float* gamma;
  int pre_relu;
  float* part;            // [grid][32*32*3 + 32]
  float* dz;
  double* stat_out;       // sum(dz), sum(dz * xhat)
  // FOLD: dy is not read but made while it is staged -- the BatchNorm-backward apply of the layer BEHIND this one (whose input
  // gradient fdz and saved input fx sit in memory, its sums facc complete): dy = a'(fx) * gamma * invstd * (fdz - mean(fdz) - xhat *
  // mean(fdz * xhat)) [+ add1]; workgroup 0 writes that BatchNorm's dgamma / dbeta; dy_out != nullptr: dy is also written out
  const float* fdz;
  const float* fx;
  const float* fstate;    // that BatchNorm's state: scale | beta | mean | invstd
  const float* fgamma;
  const double* facc;
  double fn;
  int frelu;
  float* fdgamma;
  float* fdbeta;
  const float* add1;
  float* dy_out;
  int dy_out_sum;         // dy_out = dy + add1 (ConvBwdFold::dy_out_plus_add1)
  const float* add2;      // ADD2 instances: a second residual gradient
  int64_t B;
  int64_t n_units;
  int grid;
  int dbg;                // timing experiments (MURAL_DEBUG_CW): 16 no loads, 32 no weight gradient, 64 no input gradient, 128 no stores
};
struct CwBwdArgs2 { CwBwdArgs j[TOWER_JOBS]; };
constexpr int CW_AUX_FOLD = 288;            // float[5][32]: gamma * invstd | mean(dz) | mean(dz * xhat) | mean | invstd of the folded BatchNorm

template <int NB, bool FOLD, bool ADD2 = false>
__global__ __launch_bounds__(CW_THREADS, 2) void conv32w_bwd_kernel(const CwBwdArgs2 aa) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const CwBwdArgs& a = aa.j[blockIdx.y];
  if ((int)blockIdx.x >= a.grid) return;
  constexpr int IMG_FLOATS = (16 * NB + 2) * CL_C;
  constexpr int WAVE_FLOATS = IMG_FLOATS + CW_DUMP;
  constexpr int NLD = 2 * NB;
  constexpr int NKS = 4 * NB;                                   // k-steps of the weight gradient: 4 columns each
  constexpr int NW = CL_C * CL_C * 3;
  constexpr int ROW = NW + 3 * CL_C;                            // a wave's partial tiles | sum dy | first columns | last columns
  static_assert(CW_WAVES * WAVE_FLOATS >= 2 * ROW && CW_WAVES * WAVE_FLOATS >= 4096 + CW_WFLOATS, "the waves' regions hold two partial rows / W");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kk = lane >> 4, chunk = lane & 7;
  const CwGeom& g = a.g;
  float* aux = smem;
  float* wbase = smem + CW_AUX + wave * WAVE_FLOATS;            // dy image | dump
  char* wb = reinterpret_cast<char*>(wbase);
  CwUnits units;
  cw_units_init(units, a.n_units, a.grid);
  const uint32_t lane16 = 16u * (uint32_t)lane;
  const uint32_t row_bytes = (uint32_t)g.L * 128u;
  const size_t unit_stride = (size_t)g.P * g.L * CL_C;
  auto unit_bytes = [&](int64_t u) -> uint32_t {               // bytes of the rows of unit u that exist
    const int64_t rows = a.B - u * g.P;
    return (u < units.hi && !(a.dbg & 16)) ? (uint32_t)(rows < g.P ? rows : g.P) * row_bytes : 0u;
  };
  int64_t unit = units.lo + wave;
  if constexpr (FOLD) {      // constants of the folded BatchNorm-backward from its sums (32 slots), its dgamma / dbeta
    float* fc = aux + CW_AUX_FOLD;
    if (wave == 0) {
      double s1, s2;
      cw_wave_slot_sums(a.facc, lane, &s1, &s2);
      if (lane < CL_C) {
        fc[CL_C + lane] = (float)(s1 / a.fn);
        fc[2 * CL_C + lane] = (float)(s2 / a.fn);
        if (blockIdx.x == 0) {
          a.fdbeta[lane] = (float)s1;
          a.fdgamma[lane] = (float)s2;
        }
      }
    } else if (wave == 1 && lane < CL_C) {
      fc[lane] = a.fgamma[lane] * a.fstate[3 * CL_C + lane];
      fc[3 * CL_C + lane] = a.fstate[2 * CL_C + lane];
      fc[4 * CL_C + lane] = a.fstate[3 * CL_C + lane];
    }
  }
  uint32_t* tab = reinterpret_cast<uint32_t*>(aux + CW_AUX_TAB);      // memory offset of the row of column 4 s + kk, s < NKS
  if (tid < 4 * NKS) tab[tid] = cw_col_offset(g, (uint32_t)tid);
  float a0[SNV_KSTEPS], a1[SNV_KSTEPS];
  if (a.wfrag) {
    cw_frags_global(a.wfrag, lane, a0, a1);
  } else {
    cw_copy_w(a.W, smem + CW_AUX, tid);
    __syncthreads();
    cw_frags_lds(smem + CW_AUX, 1, 0, n16, kk, a0);
    cw_frags_lds(smem + CW_AUX, 1, 1, n16, kk, a1);
  }
  const uint32_t dump = IMG_FLOATS * 4u + 16u * (uint32_t)(lane & 31);
  // image offsets of the staging slots: a 16-bit table [slot][lane] behind the waves' regions (18 registers per lane otherwise)
  const uint16_t* sotab = reinterpret_cast<const uint16_t*>(smem + CW_AUX + CW_WAVES * WAVE_FLOATS) + lane;
  for (int u = wave; u < NLD; u += CW_WAVES) const_cast<uint16_t*>(sotab)[64 * u] = (uint16_t)cw_stage_slot(g, u, lane, dump);
  __syncthreads();
  for (int i = lane; i < WAVE_FLOATS / 4; i += 64) st4(wbase + 4 * i, splat(0.f));
  uint32_t rd[6];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) rd[2 * t + h] = 4u * (uint32_t)lds_off(n16 + t, 4 * h + kk);
  // weight gradient: k-step s = 4 q + j covers the logical columns pc = 16 q + 4 j + kk; a lane reads channels 2 n16, 2 n16 + 1
  // (8 bytes) of dy at logical column pc - t + 1 = image column pc - t + 2 for tap t
  uint32_t wa[4][3];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int tp = 0; tp < 3; ++tp) wa[j][tp] = 4u * (uint32_t)(lds_off(4 * j + kk + 2 - tp, n16 >> 1) + 2 * (n16 & 1));
  uint32_t vo[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const uint32_t o = cw_col_offset(g, 16u * b + (uint32_t)n16);
    vo[b] = o == CW_OOB ? CW_OOB : o + 16u * (uint32_t)kk;
  }
  const uint32_t tab_lane = 4u * (uint32_t)(CW_AUX_TAB + kk);   // byte address of this lane's first table entry
  const uint32_t n16x8 = 8u * (uint32_t)n16;
  const float lo_pre = a.pre_relu ? 0.f : -INFINITY;
  f32x4 wacc[2][3][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int tp = 0; tp < 3; ++tp)
#pragma unroll
      for (int h = 0; h < 2; ++h) wacc[m][tp][h] = splat(0.f);
  f32x4 bsum = splat(0.f), esum = splat(0.f);                   // staging layout: channels 4 chunk .. + 3; esum: first (lane bit 3 = 0) / last column
  const f32x4 pb[2] = {splat(0.f), splat(0.f)};
  const int edge_last = (lane >> 3) & 1;
  const char* smc = reinterpret_cast<const char*>(smem);

  while (unit < units.hi) {
    const uint32_t ub = unit_bytes(unit);
    const size_t ubase = (size_t)unit * unit_stride;
    const __amdgpu_buffer_rsrc_t gd = cw_rsrc((FOLD ? a.x : a.dy) + ubase, ub), xd = cw_rsrc(a.x + ubase, ub);
    const __amdgpu_buffer_rsrc_t zd = cw_rsrc(a.dz + ubase, (a.dbg & 128) ? 0u : ub);
    // ---- this unit's dy rows: memory -> image; the bias gradient rides along
    CwSlots<NLD> so;
    so.make(lane16);
    if constexpr (FOLD) {
      // dy = BatchNorm-backward apply of the layer behind (its input passes a ReLU: the host requires it), six staging slots a round
      // (18 loads in flight), branch-free: a piece that does not exist reads x = 0, which the ReLU mask turns into dy = 0 like every
      // x <= 0 -- no range test, and max(x, 0) = x wherever the mask lets a value through
      const float* fc = aux + CW_AUX_FOLD;
      const f32x4 k0 = ld4(fc + 4 * chunk), m1 = ld4(fc + CL_C + 4 * chunk), m2 = ld4(fc + 2 * CL_C + 4 * chunk),
                  mu = ld4(fc + 3 * CL_C + 4 * chunk), is = ld4(fc + 4 * CL_C + 4 * chunk);
      const __amdgpu_buffer_rsrc_t zd2 = cw_rsrc(a.fdz + ubase, ub), xd2 = cw_rsrc(a.fx + ubase, ub);
      const __amdgpu_buffer_rsrc_t ad = cw_rsrc((a.add1 ? a.add1 : a.fdz) + ubase, a.add1 ? ub : 0u);
      const __amdgpu_buffer_rsrc_t od = cw_rsrc((a.dy_out ? a.dy_out : a.dz) + ubase, a.dy_out ? ub : 0u);
      const __amdgpu_buffer_rsrc_t ad2d = cw_rsrc((ADD2 ? a.add2 : a.fdz) + ubase, ADD2 ? ub : 0u);
      constexpr int RS = ADD2 ? 4 : 6;      // (four tensors per slot with a second residual: the same 16-18 loads in flight)
#pragma unroll
      for (int u0 = 0; u0 < NLD; u0 += RS) {
        f32x4 dzn[RS], xn[RS], ad1[RS], ad2[ADD2 ? RS : 1];
#pragma unroll
        for (int q = 0; q < RS; ++q)
          if (u0 + q < NLD) {
            const uint32_t go = so.at(u0 + q);
            dzn[q] = buf_ld4(zd2, go);
            xn[q] = buf_ld4(xd2, go);
            ad1[q] = buf_ld4(ad, go);
            if constexpr (ADD2) ad2[q] = buf_ld4(ad2d, go);
          }
#pragma unroll
        for (int q = 0; q < RS; ++q)
          if (u0 + q < NLD) {
            const f32x4 xh = (xn[q] - mu) * is;
            const f32x4 gq = k0 * ((dzn[q] - m1) - xh * m2);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = xn[q][e] > 0.f ? gq[e] : 0.f;
            o += ad1[q];
            if constexpr (ADD2) o += ad2[q];      // (gq + add1) + add2: bn_bwd_apply_cl_kernel's order
            bsum += o;
            lds_st4(wb, sotab[64 * (u0 + q)], o);
            buf_st4(od, so.at(u0 + q), a.dy_out_sum ? o + ad1[q] : o);
          }
        cw_pin(bsum);
      }
    } else {
      // (the first unit's rows used to travel under the prologue in 72 registers that then stayed allocated around the loop: spills)
      f32x4 gin[NLD];
#pragma unroll
      for (int u = 0; u < NLD; ++u) gin[u] = buf_ld4(gd, so.at(u));
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        bsum += gin[u];
        lds_st4(wb, sotab[64 * u], gin[u]);
      }
      cw_pin(bsum);
    }
    // ---- act(x) of the first k-steps of the weight gradient, in operand layout (channels 2 n16, 2 n16 + 1 of column 4 s + kk)
    f32x2 xop[NKS];
#pragma unroll
    for (int s = 0; s < CW_LA && s < NKS; ++s)
      xop[s] = buf_ld2(xd, *reinterpret_cast<const uint32_t*>(smc + tab_lane + 16u * s) + n16x8);
    // ---- first / last column of every row (S_0 / S_2 of the fix-up)
    for (int t = lane; t < g.P * 16; t += 64) {
      const int row = t >> 4;
      esum += ld4(wbase + lds_off(2 + row * g.Sc + (edge_last ? g.L - 1 : 0), chunk));
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- input gradient: the forward conv with the transposed, tap-flipped filter on the dy image
    if (!(a.dbg & 64))
      cw_conv_blocks<NB>(
          wb, rd, a0, a1, pb, [&](int) __attribute__((always_inline)) {},
          [&](int b, int m, const f32x4& acc) __attribute__((always_inline)) { buf_st4s(zd, vo[b], 64u * m, acc); });
    __builtin_amdgcn_sched_barrier(0);
    // ---- weight gradient
    if (!(a.dbg & 32)) {
#pragma unroll
      for (int s = 0; s < NKS; ++s) {
        const int q = s >> 2, j = s & 3;
        if (s + CW_LA < NKS)
          xop[s + CW_LA < NKS ? s + CW_LA : 0] =
              buf_ld2(xd, *reinterpret_cast<const uint32_t*>(smc + tab_lane + 16u * (s + CW_LA)) + n16x8);
        f32x2 gv[3];
#pragma unroll
        for (int tp = 0; tp < 3; ++tp) gv[tp] = *reinterpret_cast<const f32x2*>(wb + wa[j][tp] + CW_BLK * q);
        const f32x2 r = {fmaxf(xop[s].x, lo_pre), fmaxf(xop[s].y, lo_pre)};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tp = 0; tp < 3; ++tp)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            wacc[0][tp][h] = CW_MFMA(gv[tp][0], r[h], wacc[0][tp][h]);
            wacc[1][tp][h] = CW_MFMA(gv[tp][1], r[h], wacc[1][tp][h]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    unit += CW_WAVES;
  }
  // ---- partial row of the workgroup.  The waves park their tiles and sums in two rows of LDS (two rounds of two waves), then the
  // BatchNorm algebra of the header runs on the workgroup's sums.
  // D[row 4 kk + r][col n16] of tile (m, tap, h) <-> dWr[co = 2 (4 kk + r) + m][ci = 2 n16 + h][tap]
  __syncthreads();                                              // every wave is past its last unit: the images are dead
  float* mine = smem + CW_AUX + (wave & 1) * ROW;
  for (int round = 0; round < 2; ++round) {
    if ((wave >> 1) == round) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int tp = 0; tp < 3; ++tp)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float* p = mine + ((2 * (4 * kk + r) + m) * CL_C + 2 * n16 + h) * 3 + tp;
              *p = round ? *p + wacc[m][tp][h][r] : wacc[m][tp][h][r];
            }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v = bsum[q], e = esum[q];
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) v += __shfl_xor(v, off);
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) e += __shfl_xor(e, off);
        if (lane < 8) {
          float* p = mine + NW + 4 * chunk + q;
          *p = round ? *p + v : v;
        }
        if (lane < 16) {
          float* p = mine + NW + CL_C + edge_last * CL_C + 4 * chunk + q;      // first columns | last columns
          *p = round ? *p + e : e;
        }
      }
    }
    __syncthreads();
  }
  float* fin = aux;                                             // sum dy | first columns | last columns of the workgroup
  if (tid < 3 * CL_C) fin[tid] = smem[CW_AUX + NW + tid] + smem[CW_AUX + ROW + NW + tid];
  __syncthreads();
  // thread = (input channel ci, four output channels): 12 entries (co, tap) each
  {
    const int ci = tid & 31, grp = tid >> 5;                    // 8 groups x 4 output channels
    const float mu = a.state[2 * CL_C + ci], inv = a.state[3 * CL_C + ci], gm = a.gamma[ci], bt = a.state[CL_C + ci];
    float* dst = a.part + (size_t)blockIdx.x * (NW + CL_C);
    float wv[4][3];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
      for (int tp = 0; tp < 3; ++tp) wv[c4][tp] = a.W[((4 * grp + c4) * CL_C + ci) * 3 + tp];
    float sdz = 0.f, sdzx = 0.f;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const int co = 4 * grp + c4;
      const float S1 = fin[co], Sf = fin[CL_C + co], Sl = fin[2 * CL_C + co];
#pragma unroll
      for (int tp = 0; tp < 3; ++tp) {
        const int i = (co * CL_C + ci) * 3 + tp;
        const float dwr = smem[CW_AUX + i] + smem[CW_AUX + ROW + i];
        const float St = S1 - (tp == 0 ? Sf : tp == 2 ? Sl : 0.f);
        const float dwx = inv * (dwr - mu * St);
        dst[i] = gm * dwx + bt * St;
        sdz += wv[c4][tp] * St;
        sdzx += wv[c4][tp] * dwx;
      }
    }
    if (tid < CL_C) dst[NW + tid] = fin[tid];
    __syncthreads();                                            // the partial rows were read: their LDS takes the channel sums
    float* red = smem + CW_AUX;                                 // [8 groups][2][32]
    red[(grp * 2 + 0) * CL_C + ci] = sdz;
    red[(grp * 2 + 1) * CL_C + ci] = sdzx;
    __syncthreads();
    if (tid < 2 * CL_C) {
      float t = 0.f;
#pragma unroll
      for (int gI = 0; gI < 8; ++gI) t += red[gI * 2 * CL_C + tid];
      atomicAdd(&a.stat_out[(size_t)(blockIdx.x % MURAL_BN_SLOTS) * 2 * CL_C + tid], (double)t);
    }
  }
}

int cw_debug() {
  static const int v = dev_env("MURAL_DEBUG_CW") ? atoi(dev_env("MURAL_DEBUG_CW")) : 0;
  return v;
}

template <int NB>
constexpr size_t cw_lds_bytes() {
  return (size_t)(CW_AUX + CW_WAVES * ((16 * NB + 2) * CL_C + CW_DUMP) + NB * 64) * 4;      // aux | images | 16-bit staging offsets
}

template <int NBV>
int cw_launch_fwd(const CwFwdArgs2& a, int gx, int gy, hipStream_t stream) {
  constexpr int NB = NBV < 4 ? 4 : NBV;      // (the waves' regions also hold W and the scratch of the prologue)
  static DynLdsOnce big_lds;
  if (int rc = big_lds.ensure(&conv32w_fwd_kernel<NB>)) return rc;
  hipLaunchKernelGGL(conv32w_fwd_kernel<NB>, dim3(gx, gy), dim3(CW_THREADS), cw_lds_bytes<NB>(), stream, a);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

template <int NBV>
int cw_launch_bwd(const CwBwdArgs2& a, int gx, int gy, hipStream_t stream) {
  constexpr int NB = NBV < 4 ? 4 : NBV;      // (the waves' regions also hold two partial rows / W)
  static DynLdsOnce big_lds;
  if (int rc = big_lds.ensure(&conv32w_bwd_kernel<NB, false>, &conv32w_bwd_kernel<NB, true>, &conv32w_bwd_kernel<NB, true, true>)) return rc;
  if (a.j[0].fdz && a.j[0].add2)
    hipLaunchKernelGGL((conv32w_bwd_kernel<NB, true, true>), dim3(gx, gy), dim3(CW_THREADS), cw_lds_bytes<NB>(), stream, a);
  else if (a.j[0].fdz) hipLaunchKernelGGL((conv32w_bwd_kernel<NB, true>), dim3(gx, gy), dim3(CW_THREADS), cw_lds_bytes<NB>(), stream, a);
  else hipLaunchKernelGGL((conv32w_bwd_kernel<NB, false>), dim3(gx, gy), dim3(CW_THREADS), cw_lds_bytes<NB>(), stream, a);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

#define CW_DISPATCH(FN, NBV, ...)                 \
  switch (NBV) {                                  \
    case 5: return FN<5>(__VA_ARGS__);            \
    case 6: return FN<6>(__VA_ARGS__);            \
    case 7: return FN<7>(__VA_ARGS__);            \
    case 8: return FN<8>(__VA_ARGS__);            \
    case 9: return FN<9>(__VA_ARGS__);            \
    default: return FN<4>(__VA_ARGS__);           \
  }

int cw_grid(int64_t n_units, int L) {
  const int64_t wgs = (n_units + CW_WAVES - 1) / CW_WAVES;
  const int slots = cw_slots(L);
  return (int)(wgs < slots ? wgs : slots);
}

}  // namespace

// ---- host entry points (snv_train.hip) --------------------------------------------------------------------------------------
int cw_conv32_supported(int L) {
  CwGeom g;
  return cw_geom(2, L, &g) ? 1 : 0;
}

// up to TOWER_JOBS layers of the same role in one launch (blockIdx.y = job); the kernel instance is the one of the widest unit
int cw_conv32_fwd_jobs(const ConvFwdJob* jobs, int n, float eps, float momentum, hipStream_t stream) {
  MURAL_REQUIRE(n >= 1 && n <= TOWER_JOBS, "conv32 (wave-private): %d jobs", n);
  CwFwdArgs2 aa;
  std::memset(&aa, 0, sizeof(aa));
  int nb = 0, gx = 0, ny = 0;
  for (int i = 0; i < n; ++i) {
    const ConvFwdJob& j = jobs[i];
    if (j.B == 0 || j.L == 0) continue;
    CwFwdArgs& a = aa.j[ny++];
    MURAL_REQUIRE(cw_geom(j.B, j.L, &a.g), "conv32 (wave-private): L = %d does not fit a wave's image", j.L);
    MURAL_REQUIRE(!j.post_relu && (j.out_relu || !j.acc_out), "conv32 (wave-private): raw output with the batch sums of relu(y) only");
    const float* res1 = j.res1 ? j.res1 : j.res2;
    const float* res2 = j.res1 ? j.res2 : nullptr;
    a.x = j.x; a.y = j.y; a.W = j.W; a.wfrag = j.wfrag; a.bias = j.bias; a.res1 = res1; a.res2 = res2; a.pre_relu = j.pre_relu;
    a.stat_out = j.acc_out;
    a.fin = ClFin{j.acc, (double)j.B * j.L, j.gamma, j.beta, eps, momentum, j.running_mean, j.running_var, j.state};
    a.B = j.B;
    a.n_units = (j.B + a.g.P - 1) / a.g.P;
    a.grid = cw_grid(a.n_units, j.L);
    a.dbg = cw_debug();
    a.stamps = g_cw_stamps;
    nb = a.g.nb > nb ? a.g.nb : nb;
    gx = a.grid > gx ? a.grid : gx;
  }
  if (ny == 0) return MURAL_OK;
  CW_DISPATCH(cw_launch_fwd, nb, aa, gx, ny, stream)
}

int cw_conv32_bwd_jobs(ConvBwdJob* jobs, int n, hipStream_t stream) {
  MURAL_REQUIRE(n >= 1 && n <= TOWER_JOBS, "conv32_bwd (wave-private): %d jobs", n);
  CwBwdArgs2 aa;
  std::memset(&aa, 0, sizeof(aa));
  int nb = 0, gx = 0;
  for (int i = 0; i < n; ++i) {
    ConvBwdJob& j = jobs[i];
    CwBwdArgs& a = aa.j[i];
    MURAL_REQUIRE(cw_geom(j.B, j.L, &a.g), "conv32_bwd (wave-private): L = %d does not fit a wave's image", j.L);
    a.dy = j.dy; a.x = j.x; a.W = j.W; a.wfrag = j.wfrag; a.state = j.state; a.gamma = j.gamma; a.pre_relu = j.pre_relu; a.part = j.part; a.dz = j.dz;
    a.stat_out = j.stat_out;
    MURAL_REQUIRE((j.fold.dz != nullptr) == (jobs[0].fold.dz != nullptr), "conv32_bwd (wave-private): the jobs of a launch fold or do not fold alike");
    MURAL_REQUIRE(j.fold.dz || j.dy, "conv32_bwd (wave-private): dy is NULL");
    if (j.fold.dz) {
      MURAL_REQUIRE(j.fold.relu, "conv32_bwd (wave-private): the folded BatchNorm-backward apply is built for a ReLU in front of that BatchNorm");
      a.fdz = j.fold.dz; a.fx = j.fold.x; a.fstate = j.fold.state; a.fgamma = j.fold.gamma; a.facc = j.fold.acc; a.fn = (double)j.B * j.L;
      a.frelu = j.fold.relu; a.fdgamma = j.fold.dgamma; a.fdbeta = j.fold.dbeta; a.add1 = j.fold.add1; a.dy_out = j.fold.dy_out;
      a.dy_out_sum = j.fold.dy_out_plus_add1 && j.fold.add1 && j.fold.dy_out;
      a.add2 = j.fold.add2;
      MURAL_REQUIRE(!j.fold.add2 || j.fold.add1, "conv32_bwd (wave-private): a second residual gradient without a first");
      MURAL_REQUIRE((j.fold.add2 != nullptr) == (jobs[0].fold.add2 != nullptr), "conv32_bwd (wave-private): the jobs of a launch fold alike");
    }
    a.B = j.B;
    a.n_units = (j.B + a.g.P - 1) / a.g.P;
    a.grid = cw_grid(a.n_units, j.L);
    a.dbg = cw_debug();
    j.nrow = a.grid;
    nb = a.g.nb > nb ? a.g.nb : nb;
    gx = a.grid > gx ? a.grid : gx;
  }
  CW_DISPATCH(cw_launch_bwd, nb, aa, gx, n, stream)
}

// fragments of n <= 24 conv weights [32][32][3] for a whole step: out[n][2][6144 / 2] (forward | input gradient), one launch
size_t cw_wfrag_floats() { return 2 * (size_t)CW_WFRAG; }
int cw_wfrag_build(const float* const* W, int n, float* out, hipStream_t stream) {
  MURAL_REQUIRE(n >= 0 && n <= 24, "conv32 fragments: %d layers", n);
  if (n == 0) return MURAL_OK;
  CwFragJobs jobs;
  std::memset(&jobs, 0, sizeof(jobs));
  for (int i = 0; i < n; ++i) jobs.W[i] = W[i];
  jobs.out = out;
  jobs.n = n;
  hipLaunchKernelGGL(cw_wfrag_kernel, dim3((n * 2 * CW_WFRAG + 255) / 256), dim3(256), 0, stream, jobs);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

int cw_conv32_fwd(const float* x, int64_t B, int L, int pre_relu, const double* acc, const float* gamma, const float* beta, float eps,
                  float momentum, float* running_mean, float* running_var, float* state, const float* W, const float* wfrag, const float* bias,
                  int post_relu, const float* res1, const float* res2, double* acc_out, int out_relu, float* y, hipStream_t stream) {
  const ConvFwdJob j{x, B, L, pre_relu, acc, gamma, beta, running_mean, running_var, state, W, wfrag, bias, post_relu, res1, res2, acc_out, out_relu, y};
  return cw_conv32_fwd_jobs(&j, 1, eps, momentum, stream);
}

int cw_conv32_bwd(const float* dy, const float* x, const float* W, const float* wfrag, int64_t B, int L, const float* state, const float* gamma,
                  int pre_relu, float* dz, double* stat_out, float* part, int* nrow, hipStream_t stream) {
  ConvBwdJob j{dy, x, W, wfrag, B, L, state, gamma, pre_relu, dz, stat_out, part, 0, ConvBwdFold{}};
  const int rc = cw_conv32_bwd_jobs(&j, 1, stream);
  *nrow = j.nrow;
  return rc;
}

}  // namespace mural

// Conv1d (stride 1, or the U-Net's strided encoder convs) with 3 / 5 / 7 taps, few channels and LONG rows (the first two levels of the INDEL U-Net: 4..32
// channels on rows of 8000 / 2000 columns; reference MuRaL/model/model_indel.py:6-19, :29-38 under model.train(), and the input
// gradients of the same layers) on v_mfma_f32_16x16x4_f32 with both operands taken straight from global memory -- no LDS tile, no
// staging loop, no workgroup barrier.  Same contract as conv1d.hip / conv1d_mfma.hip for the cases it takes (Conv1dArgs: bias,
// activation, two residuals; no pre-op, no upsampling), launch_conv1d routes to it.
//
// Why: on these layers the tiled kernels are neither HBM- nor ALU-bound but latency-bound -- a workgroup stages a 2-8 KB tile with
// per-element index arithmetic, waits, computes a microsecond of arithmetic, stores, and the chip holds too few of them to cover the
// three phases (35-55 us per launch against 10-20 us of HBM time).
//
// The taps as a Toeplitz product.  A wave owns 16 consecutive output columns p0 .. p0+15 of one row.  The four k-rows of an MFMA step
// are (input channel 2c + h, tap group g) for h, g in {0, 1}: lane (n = lane & 15, kq = 2h + g) loads ONE 16-byte quad
//     Q = x[ci = 2c + h][p0 + n - pad + 4g .. + 3]              (4-byte aligned; neighbouring lanes overlap: served by the L1)
// whose element j is the B operand of step j: column n of the output reads tap 4g + j of channel ci there.  The A operand of step j is
// W[co][ci][4g + j] (zero for taps >= K: 5 taps use 5 of the 8 slots, 7 taps 7), resident in registers for the whole launch.
// One buffer_load_dwordx4 per lane and channel pair feeds 4 x (Cout / 16) MFMAs; the wave-uniform part of every address is a scalar
// offset, the lane part is loop-invariant.  Segments that touch a row end take per-element loads with the offsets checked (a refused
// element aims past the descriptor and reads 0 -- the zero padding).  Loads run one group of U segments ahead of the MFMAs.
// Output: lane (n, kq) holds rows 4 kq .. + 3 of column n: 16 lanes store 64 consecutive bytes of one channel.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "conv1d.h"
#include "mfma_tile.h"

namespace mural {
namespace {

constexpr uint32_t CD_OOB = 0x80000000u;

struct ConvDArgs {
  Conv1dArgs a;
  int segs_row;            // 16-column segments per row
  int e0, gr, ss;          // first interior segment of a row, interior groups per row, edge-loop segments per row
  int ngroups, nslow;      // interior groups / edge-loop segments of the launch
  uint32_t x_bytes, y_bytes;
  DivWide dGr, dSs;
  int xcd;      // xcd_wave_index (mfma_tile.h)
};

__device__ __forceinline__ float cd_act(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_SILU: return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
    case ACT_SOFTPLUS: {
      const float e = __expf(v);
      return v > 20.f ? v : (v < -15.f ? e : __logf(1.f + e));
    }
    default: return v;
  }
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cd_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 cd_ld4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float cd_ld1(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void cd_st1(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, voff, soff, 0);
}

// MB: 16-channel output blocks; CP: input channel pairs; U: segments per group (NBUF register buffers of U * CP quads each);
// RES: residual tensors are read (their loads are not even issued otherwise)
template <int MB, int CP, int U, bool RES, int NBUF>
__global__ __launch_bounds__(256, 2) void conv1d_direct_kernel(const ConvDArgs g, const float* __restrict__ wt, const float* __restrict__ bias) {
  const Conv1dArgs& a = g.a;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, kq = lane >> 4, half = kq >> 1, tg = kq & 1;
  const __amdgpu_buffer_rsrc_t rx = cd_rsrc(a.in, g.x_bytes), ry = cd_rsrc(a.out, g.y_bytes);
  const __amdgpu_buffer_rsrc_t r1 = cd_rsrc(a.res1 ? a.res1 : a.out, a.res1 ? g.y_bytes : 0u);      // (0 bytes: every load refused, 0)
  const __amdgpu_buffer_rsrc_t r2 = cd_rsrc(a.res2 ? a.res2 : a.out, a.res2 ? g.y_bytes : 0u);

  // A fragments: lane (row co = 16 m + n, k-row kq = (channel half, tap group)), step j = tap 4 tg + j
  float wr[MB][CP][4];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int c = 0; c < CP; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = 16 * m + n, tap = 4 * tg + j;
        wr[m][c][j] = (co < a.Cout && tap < a.K) ? wt[((size_t)(2 * c + half) * a.K + tap) * a.Cout + co] : 0.f;
      }
  // output rows of this lane: co = 16 m + 4 kq + r
  float bz[MB][4];
  uint32_t vo[MB][4];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = 16 * m + 4 * kq + r;
      bz[m][r] = (bias && co < a.Cout) ? bias[co] : 0.f;
      vo[m][r] = co < a.Cout ? (uint32_t)(co * a.Lout + n) * 4u : CD_OOB;
    }
  const int S = a.stride;      // (strided convs: column n reads from S n -- the quads of neighbouring lanes overlap less, nothing else changes)
  const uint32_t vx = (uint32_t)(half * a.Lin + S * n + 4 * tg) * 4u;      // lane part of the interior quad address

  // one segment from its CP quads: MFMAs (two accumulation chains per block -- even / odd channel pairs -- so that a chain's next
  // MFMA never waits for its own result), then bias / activation / residuals / store.  sy: scalar part of the output address;
  // tail: this lane's column lies past the end of the row
  auto finish = [&](const f32x4 (&q)[CP], uint32_t sy, bool tail) {
    float e1[MB][4], e2[MB][4];
    if (RES) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t o = tail ? CD_OOB : vo[m][r];
          e1[m][r] = cd_ld1(r1, o, sy);
          e2[m][r] = cd_ld1(r2, o, sy);
        }
    }
    f32x4 acc[MB], acb[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      acc[m] = f32x4{bz[m][0], bz[m][1], bz[m][2], bz[m][3]};
      acb[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < CP; c += 2)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[m][c][j], q[c][j], acc[m], 0, 0, 0);
          acb[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[m][c + 1][j], q[c + 1][j], acb[m], 0, 0, 0);
        }
    float v[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[m][r] = acc[m][r] + acb[m][r];
    if (a.act != ACT_NONE) {      // (one wave-uniform branch per segment, nothing under it touches memory)
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[m][r] = cd_act(v[m][r], a.act);
    }
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) cd_st1(ry, tail ? CD_OOB : vo[m][r], sy, RES ? v[m][r] + e1[m][r] + e2[m][r] : v[m][r]);
  };

  const int nwaves = gridDim.x * 4;
  const int wid = xcd_wave_index(w, g.xcd);
  // ---------------------------------------------------------------------------------------------- interior groups
  // (no branch around a load, a next group is always fetched -- the last one re-fetches itself -- and scheduling barriers keep the
  // prefetch in front of the MFMAs: see conv_wgrad_mfma.hip)
  {
    struct Buf { f32x4 q[U][CP]; };
    auto load = [&](Buf& t, int grp) {
      const uint32_t b = g.dGr.div((uint32_t)grp);
      const uint32_t p0 = (uint32_t)(g.e0 + ((uint32_t)grp - b * (uint32_t)g.gr) * U) * 16u;
      const uint32_t sx = (b * (uint32_t)(a.Cin * a.Lin) + (uint32_t)S * p0 - (uint32_t)a.pad) * 4u;
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < CP; ++c) t.q[u][c] = cd_ld4(rx, vx, sx + 64u * (uint32_t)(S * u) + (uint32_t)(2 * c * a.Lin) * 4u);
    };
    auto compute = [&](const Buf& t, int grp) {
      const uint32_t b = g.dGr.div((uint32_t)grp);
      const uint32_t p0 = (uint32_t)(g.e0 + ((uint32_t)grp - b * (uint32_t)g.gr) * U) * 16u;
      const uint32_t sy = (b * (uint32_t)(a.Cout * a.Lout) + p0) * 4u;
#pragma unroll
      for (int u = 0; u < U; ++u) finish(t.q[u], sy + 64u * u, false);
    };
    int cur = wid;
    if (cur < g.ngroups) {
      if constexpr (NBUF == 2) {
        Buf t0, t1;
        load(t0, cur);
        for (;;) {
          int nx = cur + nwaves;
          load(t1, nx < g.ngroups ? nx : cur);
          __builtin_amdgcn_sched_barrier(0);
          compute(t0, cur);
          __builtin_amdgcn_sched_barrier(0);
          cur = nx;
          if (cur >= g.ngroups) break;
          nx = cur + nwaves;
          load(t0, nx < g.ngroups ? nx : cur);
          __builtin_amdgcn_sched_barrier(0);
          compute(t1, cur);
          __builtin_amdgcn_sched_barrier(0);
          cur = nx;
          if (cur >= g.ngroups) break;
        }
      } else {
        // three buffers: the loads run TWO groups ahead (a group of the deep-reduction configurations is < 1 us of MFMAs, less than a
        // loaded round trip to HBM; with two waves per SIMD one group of distance left the matrix pipe waiting)
        Buf t0, t1, t2;
        auto ahead = [&](int c) { const int x = c + 2 * nwaves; return x < g.ngroups ? x : c; };
        load(t0, cur);
        load(t1, cur + nwaves < g.ngroups ? cur + nwaves : cur);
        for (;;) {
          load(t2, ahead(cur));
          __builtin_amdgcn_sched_barrier(0);
          compute(t0, cur);
          __builtin_amdgcn_sched_barrier(0);
          cur += nwaves;
          if (cur >= g.ngroups) break;
          load(t0, ahead(cur));
          __builtin_amdgcn_sched_barrier(0);
          compute(t1, cur);
          __builtin_amdgcn_sched_barrier(0);
          cur += nwaves;
          if (cur >= g.ngroups) break;
          load(t1, ahead(cur));
          __builtin_amdgcn_sched_barrier(0);
          compute(t2, cur);
          __builtin_amdgcn_sched_barrier(0);
          cur += nwaves;
          if (cur >= g.ngroups) break;
        }
      }
    }
  }
  // ---------------------------------------------------------------------------------------------- edge segments
  for (int t = wid; t < g.nslow; t += nwaves) {
    const uint32_t b = g.dSs.div((uint32_t)t);
    const int k = t - (int)b * g.ss;
    const int si = k < g.e0 ? k : g.e0 + g.gr * U + (k - g.e0);      // the e0 leading segments, then those behind the interior groups
    const int p0 = 16 * si;
    const uint32_t sx = b * (uint32_t)(a.Cin * a.Lin) * 4u;
    f32x4 q[CP];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int pos = S * (p0 + n) - a.pad + 4 * tg + e;
      const bool ok = (pos >= 0) & (pos < a.Lin);
      uint32_t off = (uint32_t)(half * a.Lin + pos) * 4u;
      asm volatile("" : "+v"(off));      // (a select, not a branch around the loads)
      off = ok ? off : CD_OOB;
#pragma unroll
      for (int c = 0; c < CP; ++c) q[c][e] = cd_ld1(rx, off, sx + (uint32_t)(2 * c * a.Lin) * 4u);
    }
    finish(q, (b * (uint32_t)(a.Cout * a.Lout) + (uint32_t)p0) * 4u, p0 + n >= a.Lout);
  }
}

// workgroups of `fn` that fit the device at a time (2048 if the runtime will not say)
int resident_workgroups(const void* fn, int threads) {
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, 0) == hipSuccess && per_cu > 0)
    return per_cu * prop.multiProcessorCount;
  return 2048;
}

bool direct_plan(const Conv1dArgs& a, int* mb, int* cp) {
  if (a.stride < 1 || a.stride > 8 || a.up != 1 || a.phases > 1 || a.pre_s || a.pre_t || a.pre_relu) return false;
  if (a.K != 3 && a.K != 5 && a.K != 7) return false;
  if (a.Cin != 4 && a.Cin != 8 && a.Cin != 16 && a.Cin != 32) return false;
  if (a.Cout < 1 || a.Cout > 32 || (a.Cout > 16 && a.Cin > 16)) return false;
  if (a.Cout <= 8 && a.Cin >= 16) return false;      // half of every MFMA's rows idle under a deep reduction: the vector-ALU kernel is faster (58 vs 50 us)
  if (a.pad < 0 || a.pad > a.K - 1 || a.Lin + 2 * a.pad < a.K || a.Lout != (a.Lin + 2 * a.pad - a.K) / a.stride + 1) return false;
  const uint64_t xb = (uint64_t)a.B * a.Cin * a.Lin * 4, yb = (uint64_t)a.B * a.Cout * a.Lout * 4;
  if (xb >= (1ull << 31) || yb >= (1ull << 31)) return false;
  *mb = (a.Cout + 15) / 16;
  *cp = a.Cin / 2;
  return true;
}


// ------------------------------------------------------------------------------------------------------------------------------
// The polyphase form of an upsampling conv (Conv1dArgs::phases = up, K = 3 source columns per phase, rows = Cout x up) the same
// barrier-free way, for the decoder's up-convs on the long levels.  With at most four taps ONE quad per channel holds every tap:
// the four k-rows of an MFMA step are the channels 4 c + kq of a channel QUAD, lane (n, kq) loads
//     Q = x[ci = 4 c + kq][p0 + n - pad .. + 3]
// and element j < KT of it is the B operand of step (c, j) -- KT steps per channel quad, no empty tap slots.  A = W[row][ci][j]
// (wt is [Cin][KT][rows], row = co * up + phase), resident: MB x CQ x KT registers.  A workgroup's waves take segments of 16 SOURCE
// columns; blockIdx.y takes slices of MB row blocks (80 rows = one slice of 5 blocks; 120 rows = two of 4).  Output: row -> (co, phase),
// column n -> output position up (p0 + n) + phase: lane (n, kq) stores its four rows as four dwords.
struct ConvPArgs {
  Conv1dArgs a;
  int rows, mb_total;      // GEMM rows (Cout x phases), 16-row blocks
  int segs_row, e0, gr, ss, ngroups, nslow;
  int wide;                // 80 rows = 16 channels x 5 phases in one slice, Lout % 4 == 0: interior segments leave as 16-byte stores
  uint32_t x_bytes, y_bytes;
  DivWide dGr, dSs;
  FastDiv dPh;
  int xcd;      // xcd_wave_index (mfma_tile.h)
};

template <int MB, int CQ, int KT, int U>
__global__ __launch_bounds__(256, 2) void conv1d_direct_poly_kernel(const ConvPArgs g, const float* __restrict__ wt, const float* __restrict__ bias) {
  const Conv1dArgs& a = g.a;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.y * MB;
  const int ph = a.phases;
  const __amdgpu_buffer_rsrc_t rx = cd_rsrc(a.in, g.x_bytes), ry = cd_rsrc(a.out, g.y_bytes);
  float wr[MB][CQ][KT];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int c = 0; c < CQ; ++c)
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        const int row = 16 * (m0 + m) + n;
        wr[m][c][j] = row < g.rows ? wt[((size_t)(4 * c + kq) * KT + j) * g.rows + row] : 0.f;
      }
  float bz[MB][4];
  uint32_t vo[MB][4];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * (m0 + m) + 4 * kq + r;
      const int co = (int)g.dPh.divnb((uint32_t)row), sub = row - co * ph;
      const bool ok = row < g.rows;
      bz[m][r] = (bias && ok) ? bias[co] : 0.f;
      vo[m][r] = ok ? (uint32_t)(co * a.Lout + ph * n + sub) * 4u : CD_OOB;
    }
  const uint32_t vx = (uint32_t)(kq * a.Lin + n) * 4u;
  // Wide stores (MB == 5, g.wide): a segment's 16 channels x 80 consecutive positions go through a wave-private LDS image
  // [channel][84] and leave as five 16-byte stores per lane.  As dwords a lane's four rows land at a stride of `phases` positions: 20
  // scattered dword stores per lane, and the launch took 128 us of which 42 were those stores (86 us with them switched off).
  constexpr bool WIDE = MB == 5;
  constexpr int OBP = 84;
  __shared__ __attribute__((aligned(16))) float obuf[WIDE ? 4 * 16 * OBP : 4];
  uint32_t lw[WIDE ? MB : 1][4], lr[WIDE ? 5 : 1], gw[WIDE ? 5 : 1];
  if constexpr (WIDE) {
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * m + 4 * kq + r;
        const int co = row / 5, sub = row - 5 * co;
        lw[m][r] = (uint32_t)(w * 16 * OBP + co * OBP + 5 * n + sub) * 4u;
      }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int Q = lane + 64 * u, co = Q / 20, t = Q - 20 * co;
      lr[u] = (uint32_t)(w * 16 * OBP + co * OBP + 4 * t) * 4u;
      gw[u] = (uint32_t)(co * a.Lout + 4 * t) * 4u;
    }
  }

  auto finish = [&](const f32x4 (&q)[CQ], uint32_t sy, bool tail, bool interior = false) {
    f32x4 acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = f32x4{bz[m][0], bz[m][1], bz[m][2], bz[m][3]};
#pragma unroll
    for (int c = 0; c < CQ; ++c)
#pragma unroll
      for (int j = 0; j < KT; ++j)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[m][c][j], q[c][j], acc[m], 0, 0, 0);
    float v[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[m][r] = acc[m][r];
    if (a.act != ACT_NONE) {
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[m][r] = cd_act(v[m][r], a.act);
    }
    if constexpr (WIDE) {
      if (interior && g.wide) {      // (wave-uniform)
        char* ob = reinterpret_cast<char*>(obuf);
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) *reinterpret_cast<float*>(ob + lw[m][r]) = v[m][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // this wave's own image: LDS executes a wave's accesses in order
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const f32x4 o4 = *reinterpret_cast<const f32x4*>(ob + lr[u]);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) uint32_t, o4), ry, gw[u], sy, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        return;
      }
    }
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) cd_st1(ry, tail ? CD_OOB : vo[m][r], sy, v[m][r]);
  };

  const int nwaves = gridDim.x * 4;
  const int wid = xcd_wave_index(w, g.xcd);
  {
    struct Buf { f32x4 q[U][CQ]; };
    auto load = [&](Buf& t, int grp) {
      const uint32_t b = g.dGr.div((uint32_t)grp);
      const uint32_t p0 = (uint32_t)(g.e0 + ((uint32_t)grp - b * (uint32_t)g.gr) * U) * 16u;
      const uint32_t sx = (b * (uint32_t)(a.Cin * a.Lin) + p0 - (uint32_t)a.pad) * 4u;
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < CQ; ++c) t.q[u][c] = cd_ld4(rx, vx, sx + 64u * u + (uint32_t)(4 * c * a.Lin) * 4u);
    };
    auto compute = [&](const Buf& t, int grp) {
      const uint32_t b = g.dGr.div((uint32_t)grp);
      const uint32_t p0 = (uint32_t)(g.e0 + ((uint32_t)grp - b * (uint32_t)g.gr) * U) * 16u;
      const uint32_t sy = (b * (uint32_t)(a.Cout * a.Lout) + (uint32_t)ph * p0) * 4u;
#pragma unroll
      for (int u = 0; u < U; ++u) finish(t.q[u], sy + 64u * (uint32_t)(ph * u), false, true);
    };
    int cur = wid;
    if (cur < g.ngroups) {
      Buf t0, t1;
      load(t0, cur);
      for (;;) {
        int nx = cur + nwaves;
        load(t1, nx < g.ngroups ? nx : cur);
        __builtin_amdgcn_sched_barrier(0);
        compute(t0, cur);
        __builtin_amdgcn_sched_barrier(0);
        cur = nx;
        if (cur >= g.ngroups) break;
        nx = cur + nwaves;
        load(t0, nx < g.ngroups ? nx : cur);
        __builtin_amdgcn_sched_barrier(0);
        compute(t1, cur);
        __builtin_amdgcn_sched_barrier(0);
        cur = nx;
        if (cur >= g.ngroups) break;
      }
    }
  }
  for (int t = wid; t < g.nslow; t += nwaves) {
    const uint32_t b = g.dSs.div((uint32_t)t);
    const int k = t - (int)b * g.ss;
    const int si = k < g.e0 ? k : g.e0 + g.gr * U + (k - g.e0);
    const int p0 = 16 * si;
    const uint32_t sx = b * (uint32_t)(a.Cin * a.Lin) * 4u;
    f32x4 q[CQ];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int pos = p0 + n - a.pad + e;
      const bool ok = (pos >= 0) & (pos < a.Lin);
      uint32_t off = (uint32_t)(kq * a.Lin + pos) * 4u;
      asm volatile("" : "+v"(off));
      off = ok ? off : CD_OOB;
#pragma unroll
      for (int c = 0; c < CQ; ++c) q[c][e] = cd_ld1(rx, off, sx + (uint32_t)(4 * c * a.Lin) * 4u);
    }
    finish(q, (b * (uint32_t)(a.Cout * a.Lout) + (uint32_t)(ph * p0)) * 4u, p0 + n >= a.Lin);
  }
}

bool poly_plan(const Conv1dArgs& a, int* mb, int* cq) {
  if (a.phases < 2 || a.phases > 8 || a.stride != 1 || a.up != 1 || a.pre_s || a.pre_t || a.pre_relu || a.res1 || a.res2) return false;
  if (a.K != 3 || a.pad < 0 || a.pad > 2 || a.Lout != a.Lin * a.phases || a.Lin < 64) return false;
  const int rows = a.Cout * a.phases;
  if (a.Cin == 24 && rows <= 80) { *mb = 5; *cq = 6; }
  else if (a.Cin == 32 && rows <= 128) { *mb = 4; *cq = 8; }
  else return false;
  const uint64_t xb = (uint64_t)a.B * a.Cin * a.Lin * 4, yb = (uint64_t)a.B * a.Cout * a.Lout * 4;
  return xb < (1ull << 31) && yb < (1ull << 31);
}

}  // namespace

bool conv1d_direct_supported(const Conv1dArgs& a) {
  int mb, cp;
  return direct_plan(a, &mb, &cp);
}

bool conv1d_direct_poly_supported(const Conv1dArgs& a) {
  int mb, cq;
  return poly_plan(a, &mb, &cq);
}

int launch_conv1d_direct_poly(const Conv1dArgs& a, hipStream_t stream) {
  if (a.B == 0 || a.Lout == 0) return MURAL_OK;
  int mb, cq;
  MURAL_REQUIRE(poly_plan(a, &mb, &cq), "conv1d (direct MFMA, polyphase): unsupported geometry");
  ConvPArgs g;
  std::memset(&g, 0, sizeof(g));
  g.a = a;
  g.rows = a.Cout * a.phases;
  g.mb_total = (g.rows + 15) / 16;
  g.segs_row = (a.Lin + 15) / 16;
  g.x_bytes = (uint32_t)((uint64_t)a.B * a.Cin * a.Lin * 4);
  g.y_bytes = (uint32_t)((uint64_t)a.B * a.Cout * a.Lout * 4);
  g.dPh = FastDiv::make((uint32_t)a.phases);
  g.wide = (mb == 5 && g.rows == 80 && a.phases == 5 && (a.Lout & 3) == 0 && a.act == ACT_NONE && !dev_env("MURAL_DEBUG_POLY_NARROW")) ? 1 : 0;
  const int U = 1;
  // interior segments: the first quad starts inside the row (p0 >= pad), the last one ends inside it (p0 + 15 - pad + 3 < Lin)
  const int e0 = (a.pad + 15) / 16;
  const int last_in = a.Lin - 19 + a.pad >= 0 ? (a.Lin - 19 + a.pad) / 16 : -1;
  const int ir = std::max(0, last_in - e0 + 1);
  g.e0 = e0;
  g.gr = ir / U;
  if (g.gr == 0) g.e0 = 0;
  g.ss = g.segs_row - g.gr * U;
  g.ngroups = a.B * g.gr;
  g.xcd = xcd_swizzle_enabled();
  g.nslow = a.B * g.ss;
  g.dGr = DivWide::make((uint32_t)std::max(1, g.gr), (uint64_t)g.ngroups + 64);
  g.dSs = DivWide::make((uint32_t)std::max(1, g.ss), (uint64_t)g.nslow + 64);
  const int units = g.ngroups + g.nslow;
  const int slices = (g.mb_total + mb - 1) / mb;
  static int cap5 = 0, cap4 = 0;
  int& cap = mb == 5 ? cap5 : cap4;
  if (cap == 0)
    cap = resident_workgroups(mb == 5 ? reinterpret_cast<const void*>(conv1d_direct_poly_kernel<5, 6, 3, 1>)
                                      : reinterpret_cast<const void*>(conv1d_direct_poly_kernel<4, 8, 3, 1>), 256);
  const int wgs = std::max(1, std::min(std::max(1, cap / slices), (units + 3) / 4));
  if (mb == 5) hipLaunchKernelGGL((conv1d_direct_poly_kernel<5, 6, 3, 1>), dim3(wgs, slices), dim3(256), 0, stream, g, a.wt, a.bias);
  else hipLaunchKernelGGL((conv1d_direct_poly_kernel<4, 8, 3, 1>), dim3(wgs, slices), dim3(256), 0, stream, g, a.wt, a.bias);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

int launch_conv1d_direct(const Conv1dArgs& a, hipStream_t stream) {
  if (a.B == 0 || a.Lout == 0) return MURAL_OK;
  int mb, cp;
  MURAL_REQUIRE(direct_plan(a, &mb, &cp), "conv1d (direct MFMA): unsupported geometry");
  ConvDArgs g;
  std::memset(&g, 0, sizeof(g));
  g.a = a;
  g.segs_row = (a.Lout + 15) / 16;
  g.x_bytes = (uint32_t)((uint64_t)a.B * a.Cin * a.Lin * 4);
  g.y_bytes = (uint32_t)((uint64_t)a.B * a.Cout * a.Lout * 4);
  const bool res = a.res1 != nullptr || a.res2 != nullptr;
  int wgs = 1;
  // interior segments of a row: S p0 >= pad, the last quad (S (p0 + 15) + 7 - pad) inside the row, all 16 columns inside the output row
  auto plan = [&](int U) {
    const int S = a.stride;
    const int e0 = (a.pad + 16 * S - 1) / (16 * S);
    const int room = a.Lin - 8 + a.pad - 15 * S;      // S p0 <= room
    const int last_in = room >= 0 ? (room / S) / 16 : -1;
    const int last_out = a.Lout >= 16 ? (a.Lout - 16) / 16 : -1;
    const int ir = std::max(0, std::min(last_in, last_out) - e0 + 1);
    g.e0 = e0;
    g.gr = ir / U;
    if (g.gr == 0) g.e0 = 0;
    g.ss = g.segs_row - g.gr * U;
    g.ngroups = a.B * g.gr;
    g.xcd = xcd_swizzle_enabled();
    g.nslow = a.B * g.ss;
    g.dGr = DivWide::make((uint32_t)std::max(1, g.gr), (uint64_t)g.ngroups + 64);
    g.dSs = DivWide::make((uint32_t)std::max(1, g.ss), (uint64_t)g.nslow + 64);
    const int units = g.ngroups + g.nslow;
    wgs = std::max(1, std::min(2048, std::min((units + 3) / 4, std::max(512, (units + 31) / 32))));
  };
#define MURAL_CD(MB_, CP_, U_, NB_)                                                                                  \
  do {                                                                                                              \
    plan(U_);                                                                                                       \
    /* persistent waves: no more workgroups than are resident at a time (a larger grid runs in rounds, the last one part empty) */ \
    static int cap_res = 0, cap_plain = 0;                                                                          \
    int& cap = res ? cap_res : cap_plain;                                                                           \
    if (cap == 0) {                                                                                                 \
      const void* fn = res ? reinterpret_cast<const void*>(conv1d_direct_kernel<MB_, CP_, U_, true, NB_>)           \
                           : reinterpret_cast<const void*>(conv1d_direct_kernel<MB_, CP_, U_, false, NB_>);         \
      cap = resident_workgroups(fn, 256);                                                                           \
    }                                                                                                               \
    if (dev_env("MURAL_DIRECT_GRID_CAP") == nullptr || atoi(dev_env("MURAL_DIRECT_GRID_CAP")) != 0) wgs = std::min(wgs, cap); \
    if (res) hipLaunchKernelGGL((conv1d_direct_kernel<MB_, CP_, U_, true, NB_>), dim3(wgs), dim3(256), 0, stream, g, a.wt, a.bias);  \
    else hipLaunchKernelGGL((conv1d_direct_kernel<MB_, CP_, U_, false, NB_>), dim3(wgs), dim3(256), 0, stream, g, a.wt, a.bias);     \
  } while (0)
  if (mb == 1) {
    if (cp == 2) MURAL_CD(1, 2, 8, 2);
    else if (cp == 4) MURAL_CD(1, 4, 4, 2);
    else if (cp == 8) MURAL_CD(1, 8, 2, 2);
    else MURAL_CD(1, 16, 1, 2);
  } else {
    if (cp == 2) MURAL_CD(2, 2, 8, 2);
    else if (cp == 4) MURAL_CD(2, 4, 4, 2);
    else MURAL_CD(2, 8, 1, 3);
  }
#undef MURAL_CD
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

}  // namespace mural

// Fused ConvBlock of the INDEL U-Net on fp32 MFMA (reference MuRaL/model/model_indel.py:6-19, eval mode, BatchNorms folded):
//   out = x + W1 . SiLU(W5 * x + b5) + b1 [+ res2],   W5: k=5 conv C -> 2C, W1: 1x1 conv 2C -> C,   C = 16 or 24.
// Same contract as convblock_kernel (conv1d.hip) for the blocks without front / tail; the packed-FMA version is bound by
// vector-instruction issue (one v_pk_fma_f32 per 4 FLOP), here one v_mfma_f32_16x16x4_f32 does 2048 FLOP:
//   GEMM 1  D1[hidden 16][position 16] += W5[hidden][(tap, ci)] * x[ci][position + tap - 2],  K = 5C in steps of 4 input channels
//           of one tap; the B operand is one ds_read_b32 of the channel-major LDS tile (pitch = 16 mod 32) shared by the 2C/16
//           hidden blocks.
//   SiLU    on the accumulators.
//   GEMM 2  D2[out 16][position 16] += W1[out][hidden] * h[hidden][position]: lane (position, kk) holds hidden channels
//           16 mb + 4 kk + r in register r of block mb, which IS a B operand if k-step (mb, r) is defined to cover exactly those
//           four channels (the order of the k-steps of a dot product is free) -- no shuffle, no LDS round trip.
// Both weight matrices live in registers as A fragments for the whole workgroup lifetime (C = 24: 114 VGPRs).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "conv1d.h"
#include "mfma_tile.h"

namespace mural {
namespace {

constexpr int CM_TW = 256 + 8;     // tile: 256 positions + 4 columns each side (2 of them halo; 4 keeps float4 loads aligned)
constexpr int CM_PITCH = 272;      // = 16 (mod 32) floats: the 16 positions x 4 channels of an operand read spread over all banks

__device__ __forceinline__ float silu_fast(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

template <int C>
__global__ __launch_bounds__(256, 2) void convblock_mfma_kernel(const ConvBlockArgs a) {
  constexpr int C2 = 2 * C;
  constexpr int MB1 = C2 / 16;           // hidden blocks
  constexpr int CQ = C / 4;              // channel quads per tap
  constexpr int KS1 = 5 * CQ;            // k-steps of GEMM 1
  constexpr int MB2 = (C + 15) / 16;     // output blocks (C = 24: the second one is half empty)
  constexpr int KS2 = 4 * MB1;           // k-steps of GEMM 2
  __shared__ float tile[C * CM_PITCH];
  __shared__ float otile[C * CM_PITCH];   // the block's outputs, streamed out as float4 rows once the tile is done
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kk = lane >> 4;

  // A fragments: lane (m = n16, kk)
  float a1[MB1][KS1], a2[MB2][KS2];
#pragma unroll
  for (int mb = 0; mb < MB1; ++mb)
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
      const int t = s / CQ, ci = 4 * (s % CQ) + kk;
      a1[mb][s] = a.w5[(ci * 5 + t) * C2 + 16 * mb + n16];
    }
#pragma unroll
  for (int mb = 0; mb < MB2; ++mb)
#pragma unroll
    for (int s = 0; s < KS2; ++s) {
      const int h = 16 * (s >> 2) + 4 * kk + (s & 3), c = 16 * mb + n16;
      a2[mb][s] = c < C ? a.w1[h * C + c] : 0.f;
    }
  // biases in accumulator layout: lane (position, kk), register r <-> row 16 mb + 4 kk + r
  f32x4 bias1[MB1], bias2[MB2];
#pragma unroll
  for (int mb = 0; mb < MB1; ++mb) bias1[mb] = ld4(a.b5 + 16 * mb + 4 * kk);
#pragma unroll
  for (int mb = 0; mb < MB2; ++mb) {
    const int c0 = 16 * mb + 4 * kk;
    bias2[mb] = f32x4{c0 + 0 < C ? a.b1[c0 + 0] : 0.f, c0 + 1 < C ? a.b1[c0 + 1] : 0.f, c0 + 2 < C ? a.b1[c0 + 2] : 0.f,
                      c0 + 3 < C ? a.b1[c0 + 3] : 0.f};
  }

  // persistent workgroup: the weight fragments are loaded once and serve every (row, tile) this workgroup walks
  const int tiles_per_row = (a.L + 255) / 256;
  const int64_t total_tiles = (int64_t)a.B * tiles_per_row;

  // the MFMA phase of one tile: tile (LDS) -> otile (LDS)
  auto compute_tile = [&](int l0) {
    // the tile's 16-position blocks go round the waves (block 4 pb + wave): a partial tile -- the second one of a 400-column row holds 9
    // blocks -- costs ceil(blocks / 4) block times, not the 4 of its fullest wave
#pragma unroll 1
    for (int pb = 0; pb < 4; ++pb) {
      const int blk = 4 * pb + wave;
      if (l0 + 16 * blk >= a.L) break;                  // wave-uniform: nothing of this block (or of the wave's later ones) is inside the row
      const int p = 16 * blk + n16;                     // tile-relative position of this lane's column
      const float* xp = tile + kk * CM_PITCH + p + 2;   // x[ci = 4 cq + kk][p + t - 2] sits at xp[4 cq * pitch + t] (tile origin l0 - 4)
      f32x4 acc1[MB1];
#pragma unroll
      for (int mb = 0; mb < MB1; ++mb) acc1[mb] = bias1[mb];
#pragma unroll
      for (int s = 0; s < KS1; ++s) {
        const float bv = xp[4 * (s % CQ) * CM_PITCH + s / CQ];
#pragma unroll
        for (int mb = 0; mb < MB1; ++mb) acc1[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[mb][s], bv, acc1[mb], 0, 0, 0);
      }
      // SiLU of all the hidden values as ONE burst of vector instructions (every MFMA -> vector -> MFMA switch costs issue cycles on
      // this part, tools/microbench), then GEMM 2 on two accumulator chains per output block (even / odd k-steps)
      float hv[KS2];
#pragma unroll
      for (int s = 0; s < KS2; ++s) hv[s] = silu_fast(acc1[s >> 2][s & 3]);
      f32x4 acc2[MB2], acc2b[MB2];      // (one output block: its k-steps alternate between two chains; two blocks ARE two chains)
#pragma unroll
      for (int mb = 0; mb < MB2; ++mb) {
        acc2[mb] = bias2[mb];
        acc2b[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if constexpr (MB2 == 1) {
#pragma unroll
        for (int s = 0; s < KS2; s += 2) {
          acc2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[0][s], hv[s], acc2[0], 0, 0, 0);
          acc2b[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[0][s + 1], hv[s + 1], acc2b[0], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int s = 0; s < KS2; ++s)
#pragma unroll
          for (int mb = 0; mb < MB2; ++mb) acc2[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[mb][s], hv[s], acc2[mb], 0, 0, 0);
      }
#pragma unroll
      for (int mb = 0; mb < MB2; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * mb + 4 * kk + r;
          if (c < C) otile[c * CM_PITCH + p] = tile[c * CM_PITCH + p + 4] + (MB2 == 1 ? acc2[mb][r] + acc2b[mb][r] : acc2[mb][r]);
        }
    }
  };

  if ((a.L & 3) == 0 && (size_t)a.B * C * a.L * 4 < (size_t(1) << 31)) {
    // Rows start 16-byte aligned: the tile (origin l0 - 4) and the skip tensor move as float4 pieces through range-checked
    // descriptors -- a refused piece (outside the row, past the work list) reads zeros.  SOFTWARE PIPELINE: the pieces of the NEXT
    // tile and this tile's skip pieces are requested before this tile's MFMAs and parked in registers; they reach LDS / the output
    // after the MFMA phase.  (One tile at a time -- request, wait, MFMAs, request the skip, wait, store -- left the matrix cores idle
    // for two global round trips per tile, and the workgroups of a CU, all on the same schedule, idle together: 55 % busy.)
    constexpr int Q = CM_TW / 4;
    constexpr int NX = (C * Q + 255) / 256, NR = (C * 64 + 255) / 256;
    const size_t tbytes = (size_t)a.B * C * a.L * 4;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)tbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res2 ? a.res2 : a.x), 0, a.res2 ? (int)tbytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)tbytes, 0x00020000);
    auto request_x = [&](f32x4 (&xr)[NX], int64_t tix) {
      const bool live = tix < total_tiles;
      const int b = live ? (int)(tix / tiles_per_row) : 0;
      const int l0 = live ? (int)(tix - (int64_t)b * tiles_per_row) * 256 : 0;
#pragma unroll
      for (int u = 0; u < NX; ++u) {
        const int i = tid + 256 * u;
        const int ci = i / Q, q = i - ci * Q;
        const int l = l0 - 4 + 4 * q;
        const bool ok = live & (i < C * Q) & (l >= 0) & (l < a.L);
        uint32_t off = (uint32_t)((b * C + ci) * a.L + l) * 4u;
        asm volatile("" : "+v"(off));
        xr[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : 0x80000000u, 0, 0));
      }
    };
    auto park_x = [&](const f32x4 (&xr)[NX]) {
#pragma unroll
      for (int u = 0; u < NX; ++u) {
        const int i = tid + 256 * u;
        if (i < C * Q) {
          const int ci = i / Q, q = i - ci * Q;
          st4(tile + ci * CM_PITCH + 4 * q, xr[u]);
        }
      }
    };
    f32x4 xr[NX];
    int64_t tix = blockIdx.x;
    request_x(xr, tix);
    park_x(xr);
    __syncthreads();
#pragma unroll 1
    for (; tix < total_tiles; tix += gridDim.x) {
      const int b = (int)(tix / tiles_per_row);
      const int l0 = (int)(tix - (int64_t)b * tiles_per_row) * 256;
      request_x(xr, tix + gridDim.x);                 // the next tile (zeros past the end of the work list)
      f32x4 rr[NR];
      uint32_t ooff[NR];
#pragma unroll
      for (int u = 0; u < NR; ++u) {                  // this tile's skip pieces and output offsets
        const int i = tid + 256 * u;
        const int c = i >> 6, q = i & 63;
        const int l = l0 + 4 * q;
        const bool ok = (i < C * 64) & (l < a.L);
        uint32_t off = (uint32_t)((b * C + c) * a.L + l) * 4u;
        asm volatile("" : "+v"(off));
        ooff[u] = ok ? off : 0x80000000u;
        rr[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ooff[u], 0, 0));
      }
      __builtin_amdgcn_sched_barrier(0);              // (keep the requests in front of the MFMAs)
      compute_tile(l0);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();                                // otile complete, tile consumed
#pragma unroll
      for (int u = 0; u < NR; ++u) {
        const int i = tid + 256 * u;
        const int c = i >> 6, q = i & 63;
        if (i < C * 64) {
          const f32x4 v = ld4(otile + c * CM_PITCH + 4 * q) + rr[u];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ro, ooff[u], 0, 0);
        }
      }
      park_x(xr);
      __syncthreads();                                // the next tile is in place, otile is free
    }
    return;
  }

#pragma unroll 1
  for (int64_t tix = blockIdx.x; tix < total_tiles; tix += gridDim.x) {
    const int b = (int)(tix / tiles_per_row);
    const int l0 = (int)(tix - (int64_t)b * tiles_per_row) * 256;
    __syncthreads();                                    // the previous tile is consumed
    const float* src = a.x + (size_t)b * C * a.L;
    for (int i = tid; i < C * CM_TW; i += 256) {
      const int ci = i / CM_TW, j = i - ci * CM_TW;
      const int l = l0 - 4 + j;
      tile[ci * CM_PITCH + j] = (l >= 0 && l < a.L) ? src[(size_t)ci * a.L + l] : 0.f;
    }
    __syncthreads();
    compute_tile(l0);
    __syncthreads();
    for (int i = tid; i < C * 256; i += 256) {
      const int c = i >> 8, j = i & 255;
      const int l = l0 + j;
      if (l < a.L) {
        const size_t o = ((size_t)b * C + c) * a.L + l;
        float v = otile[c * CM_PITCH + j];
        if (a.res2) v += a.res2[o];
        a.out[o] = v;
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------------
// The same block WITHOUT an LDS tile, barriers or a staging loop (the scheme of conv1d_direct.hip): a wave owns 16 consecutive
// positions p0 .. p0 + 15 of one row; the four k-rows of an MFMA step of GEMM 1 are the channels 4 c + kq of a channel quad, lane
// (n, kq) loads ONE 16-byte quad x[4 c + kq][p0 + n - 2 .. + 1] (taps 0 .. 3) and ONE dword x[4 c + kq][p0 + n + 2] (tap 4) per
// quad -- five taps in five steps, no empty tap slots: the 40 (C = 16) MFMAs of the tiled form, with operands straight from global
// memory, the next segment's loads in flight under this one's MFMAs.  Block input and skip tensor in accumulator layout (channels
// 16 mb + 4 kq + r of column n) are four more dwords each; the result leaves as four dword stores (16 lanes = 64 bytes of a row).
// The tiled form above was bound by its per-tile schedule (request, MFMAs, barrier, LDS -> registers -> global, barrier): 295 us per
// 2048 x 2000 x 16 launch against 160 us of MFMA time, whatever the block order, the SiLU grouping, the grid or a start-up stagger.
struct CbdArgs {
  ConvBlockArgs a;
  int segs_row, e0, gr, ss, ngroups, nslow;
  uint32_t bytes;
  DivWide dGr, dSs;
  int xcd;      // xcd_wave_index (mfma_tile.h)
};

template <int C>
__global__ __launch_bounds__(256, 2) void convblock_direct_kernel(const CbdArgs g) {
  const ConvBlockArgs& a = g.a;
  constexpr int C2 = 2 * C, MB1 = C2 / 16, CQ = C / 4, MB2 = (C + 15) / 16, KS2 = 4 * MB1;
  constexpr uint32_t OOB = 0x80000000u;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, kq = lane >> 4;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)g.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res2 ? a.res2 : a.x), 0, a.res2 ? (int)g.bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)g.bytes, 0x00020000);
  // A fragments: GEMM 1 step (c, j): k-row kq = channel 4 c + kq, tap j; GEMM 2 step s: hidden channel 16 (s / 4) + 4 kq + s % 4
  float a1[MB1][CQ][5], a2[MB2][KS2];
#pragma unroll
  for (int mb = 0; mb < MB1; ++mb)
#pragma unroll
    for (int c = 0; c < CQ; ++c)
#pragma unroll
      for (int j = 0; j < 5; ++j) a1[mb][c][j] = a.w5[((4 * c + kq) * 5 + j) * C2 + 16 * mb + n];
#pragma unroll
  for (int mb = 0; mb < MB2; ++mb)
#pragma unroll
    for (int s = 0; s < KS2; ++s) {
      const int h = 16 * (s >> 2) + 4 * kq + (s & 3), c = 16 * mb + n;
      a2[mb][s] = c < C ? a.w1[h * C + c] : 0.f;
    }
  f32x4 bias1[MB1], bias2[MB2];
#pragma unroll
  for (int mb = 0; mb < MB1; ++mb) bias1[mb] = ld4(a.b5 + 16 * mb + 4 * kq);
  uint32_t vo[MB2][4];      // lane part of the offsets of the lane's output elements (channel 16 mb + 4 kq + r, column n)
#pragma unroll
  for (int mb = 0; mb < MB2; ++mb) {
    const int c0 = 16 * mb + 4 * kq;
    bias2[mb] = f32x4{c0 + 0 < C ? a.b1[c0 + 0] : 0.f, c0 + 1 < C ? a.b1[c0 + 1] : 0.f, c0 + 2 < C ? a.b1[c0 + 2] : 0.f,
                      c0 + 3 < C ? a.b1[c0 + 3] : 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) vo[mb][r] = c0 + r < C ? (uint32_t)((c0 + r) * a.L + n) * 4u : OOB;
  }
  const uint32_t vq = (uint32_t)(kq * a.L + n) * 4u;      // lane part of the operand addresses (channel kq of a quad, column n)

  struct Buf {
    f32x4 q[CQ];      // taps 0 .. 3
    float t4[CQ];     // tap 4
    float xr[MB2][4], sk[MB2][4];
  };
  // MFMAs, SiLU, residuals and stores of one segment; so: scalar part of the output addresses; tail: this lane's column is past the row
  auto finish = [&](const Buf& t, uint32_t so, bool tail) {
    f32x4 acc1[MB1];
#pragma unroll
    for (int mb = 0; mb < MB1; ++mb) acc1[mb] = bias1[mb];
#pragma unroll
    for (int c = 0; c < CQ; ++c)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const float bv = j < 4 ? t.q[c][j < 4 ? j : 0] : t.t4[c];
#pragma unroll
        for (int mb = 0; mb < MB1; ++mb) acc1[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[mb][c][j], bv, acc1[mb], 0, 0, 0);
      }
    float hv[KS2];
#pragma unroll
    for (int s = 0; s < KS2; ++s) hv[s] = silu_fast(acc1[s >> 2][s & 3]);
    f32x4 acc2[MB2], acc2b[MB2];
#pragma unroll
    for (int mb = 0; mb < MB2; ++mb) {
      acc2[mb] = bias2[mb];
      acc2b[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (MB2 == 1) {
#pragma unroll
      for (int s = 0; s < KS2; s += 2) {
        acc2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[0][s], hv[s], acc2[0], 0, 0, 0);
        acc2b[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[0][s + 1], hv[s + 1], acc2b[0], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < KS2; ++s)
#pragma unroll
        for (int mb = 0; mb < MB2; ++mb) acc2[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[mb][s], hv[s], acc2[mb], 0, 0, 0);
    }
#pragma unroll
    for (int mb = 0; mb < MB2; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = ((MB2 == 1 ? acc2[mb][r] + acc2b[mb][r] : acc2[mb][r]) + t.xr[mb][r]) + t.sk[mb][r];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), ro, tail ? OOB : vo[mb][r], so, 0);
      }
  };

  const int nwaves = gridDim.x * 4;
  const int wid = xcd_wave_index(w, g.xcd);
  {
    auto load = [&](Buf& t, int grp) {
      const uint32_t b = g.dGr.div((uint32_t)grp);
      const uint32_t p0 = (uint32_t)(g.e0 + ((uint32_t)grp - b * (uint32_t)g.gr)) * 16u;
      const uint32_t sb = (b * (uint32_t)(C * a.L) + p0) * 4u;      // (row, first column) of the segment
#pragma unroll
      for (int c = 0; c < CQ; ++c) {
        t.q[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vq, sb - 8u + (uint32_t)(4 * c * a.L) * 4u, 0));
        t.t4[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vq, sb + 8u + (uint32_t)(4 * c * a.L) * 4u, 0));
      }
#pragma unroll
      for (int mb = 0; mb < MB2; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          t.xr[mb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo[mb][r], sb, 0));
          t.sk[mb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo[mb][r], sb, 0));
        }
    };
    auto compute = [&](const Buf& t, int grp) {
      const uint32_t b = g.dGr.div((uint32_t)grp);
      const uint32_t p0 = (uint32_t)(g.e0 + ((uint32_t)grp - b * (uint32_t)g.gr)) * 16u;
      finish(t, (b * (uint32_t)(C * a.L) + p0) * 4u, false);
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
  // edge segments: every tap through a checked offset (a refused element reads 0: the zero padding)
  for (int t = wid; t < g.nslow; t += nwaves) {
    const uint32_t b = g.dSs.div((uint32_t)t);
    const int k = t - (int)b * g.ss;
    const int si = k < g.e0 ? k : g.e0 + g.gr + (k - g.e0);
    const int p0 = 16 * si;
    const uint32_t sr = b * (uint32_t)(C * a.L) * 4u;
    Buf q;
#pragma unroll
    for (int e = 0; e < 5; ++e) {
      const int pos = p0 + n - 2 + e;
      const bool ok = (pos >= 0) & (pos < a.L);
      uint32_t off = (uint32_t)(kq * a.L + pos) * 4u;
      asm volatile("" : "+v"(off));
      off = ok ? off : OOB;
#pragma unroll
      for (int c = 0; c < CQ; ++c) {
        const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, sr + (uint32_t)(4 * c * a.L) * 4u, 0));
        if (e < 4) q.q[c][e < 4 ? e : 0] = v;
        else q.t4[c] = v;
      }
    }
    const bool tail = p0 + n >= a.L;
#pragma unroll
    for (int mb = 0; mb < MB2; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        uint32_t off = vo[mb][r];
        asm volatile("" : "+v"(off));
        off = tail ? OOB : off;
        q.xr[mb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, off, sr + (uint32_t)p0 * 4u, 0));
        q.sk[mb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, sr + (uint32_t)p0 * 4u, 0));
      }
    finish(q, sr + (uint32_t)p0 * 4u, tail);
  }
}

template <int C>
static int launch_convblock_direct_t(const ConvBlockArgs& a, hipStream_t stream) {
  CbdArgs g;
  std::memset(&g, 0, sizeof(g));
  g.a = a;
  g.bytes = (uint32_t)((uint64_t)a.B * C * a.L * 4);
  g.segs_row = (a.L + 15) / 16;
  // interior segments: taps p0 - 2 .. p0 + 17 inside the row
  const int e0 = 1;
  const int last_in = a.L >= 34 ? (a.L - 18) / 16 : -1;
  g.gr = std::max(0, last_in - e0 + 1);
  g.e0 = g.gr > 0 ? e0 : 0;
  g.ss = g.segs_row - g.gr;
  g.ngroups = a.B * g.gr;
  g.xcd = xcd_swizzle_enabled();
  g.nslow = a.B * g.ss;
  g.dGr = DivWide::make((uint32_t)std::max(1, g.gr), (uint64_t)g.ngroups + 64);
  g.dSs = DivWide::make((uint32_t)std::max(1, g.ss), (uint64_t)g.nslow + 64);
  static int cap = 0;
  if (cap == 0) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(convblock_direct_kernel<C>), 256, 0) == hipSuccess &&
        per_cu > 0)
      cap = per_cu * prop.multiProcessorCount;
    else
      cap = 512;
  }
  const int units = g.ngroups + g.nslow;
  const int wgs = std::max(1, std::min(cap, (units + 3) / 4));
  hipLaunchKernelGGL(convblock_direct_kernel<C>, dim3(wgs), dim3(256), 0, stream, g);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

}  // namespace

bool convblock_mfma_supported(const ConvBlockArgs& a) {
  return (a.C == 16 || a.C == 24) && !a.f_in && !a.symtab && !a.tail_max && !dev_env("MURAL_DEBUG_CONVBLOCK_VALU");
}

int launch_convblock_mfma(const ConvBlockArgs& a, hipStream_t stream) {
  // MURAL_CONVBLOCK_DIRECT=0: the LDS-tiled form; 2: the barrier-free form whatever the size (A/B switch, validation)
  static const int direct_mode = dev_env("MURAL_CONVBLOCK_DIRECT") ? atoi(dev_env("MURAL_CONVBLOCK_DIRECT")) : 1;      // 2: whatever the size
  if (direct_mode != 0 && (uint64_t)a.B * a.C * a.L * 4 < (1ull << 31) && (direct_mode == 2 || (a.L >= 64 && (int64_t)a.B * a.L >= 32768)))
    return a.C == 16 ? launch_convblock_direct_t<16>(a, stream) : launch_convblock_direct_t<24>(a, stream);
  const int64_t tiles = (int64_t)a.B * ((a.L + 255) / 256);
  // persistent workgroups: exactly as many as are resident at a time (a grid of 2048 on 768 resident ones runs in rounds of 768, 768,
  // 512 -- the last round a third empty)
  static int resident[2] = {0, 0};
  int& res = resident[a.C == 16 ? 0 : 1];
  if (res == 0) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    const void* fn = a.C == 16 ? reinterpret_cast<const void*>(convblock_mfma_kernel<16>) : reinterpret_cast<const void*>(convblock_mfma_kernel<24>);
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) == hipSuccess && per_cu > 0)
      res = per_cu * prop.multiProcessorCount;
    else
      res = 2048;
  }
  const dim3 grid((unsigned)(tiles < res ? tiles : res));
  if (a.C == 16) hipLaunchKernelGGL(convblock_mfma_kernel<16>, grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(convblock_mfma_kernel<24>, grid, dim3(256), 0, stream, a);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

}  // namespace mural

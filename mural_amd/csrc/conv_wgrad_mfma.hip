// Weight gradient of the general Conv1d (indel_train.hip: mural_op_convg_bwd) as an implicit GEMM on v_mfma_f32_16x16x4_f32.
//
//   dW[co][ci][k] = sum over (b, lo) of dy[b][co][lo] * xu[b][ci][lo * stride + k - pad]        (xu = x upsampled by `up`, zero outside)
//   db[co]        = sum over (b, lo) of dy[b][co][lo]
//
// is D = A * B with A = dy as [co][position], B = [position][entry] (entry = ci * K + k, plus one column of ones that yields db) and the
// flattened (b, lo) axis as the reduction dimension: rows of 16 channels x columns of 16 entries per MFMA tile, four positions per
// instruction.  Reference: the gradients torch's autograd produces for nn.Conv1d inside UNet_Small (MuRaL/model/model_indel.py:6-19,
// :100-131) in the step of MuRaL/training.py:424-436.
//
// A wave walks segments of 16 consecutive positions.  Lane (i = lane & 15, q = lane >> 4) holds positions 4q .. 4q+3 of the segment as
// one 16-byte register quad per tile row / column block: element e of every quad forms one k-step (the reduction set of step e is the
// four positions {4q + e}), so both operands come straight from global memory with one buffer_load_dwordx4 per block -- for a stride-1
// conv the column block's quad is the input row shifted by the tap, an unaligned (4-byte aligned) 16-byte load that the L1 serves K
// times.  The wave-uniform part of every address travels in the scalar offset, the per-lane part is loop-invariant: no address VALU
// in the interior of a row.  Segments that touch a row end, ragged rows, strided and upsampled convs take the gathered form (four
// 4-byte loads per block, offsets checked per element; a refused element aims past the descriptor and reads 0).
// Loads run one group of U segments ahead of the MFMAs (register double buffer).  The four waves of a workgroup add their tiles through
// LDS in a fixed order: one partial row [Cout][entries + 1] per workgroup, reduced across workgroups by the caller
// (conv_wgrad_reduce_multi_kernel) -- bitwise reproducible.
#include <algorithm>

#include "common.h"
#include "mfma_tile.h"

namespace mural {
namespace {

constexpr uint32_t WGM_OOB = 0x80000000u;      // a voffset at or past num_records: the load returns 0
constexpr int WGM_RT = 4;                      // tiles per round of the cross-wave sum (16 KB of LDS)

// exact n / d for n <= the bound the host built it for: q = (n * M) >> S with 2^S > bound * d
struct DivWide {
  uint32_t M, S;
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return (uint32_t)(((uint64_t)n * M) >> S); }
};

struct WgmArgs {
  const float* dy;
  const float* x;
  float* part;
  int B, Cin, Lin, Cout, Lout, K, stride, pad, up;
  int entries, rowlen, segs, groups, fast_ok;
  uint32_t total, dy_bytes, x_bytes;
  DivWide dLout;
  FastDiv dK, dUp;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wgm_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 wgm_ld4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float wgm_ld1(__amdgpu_buffer_rsrc_t r, uint32_t voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}

template <int MB, int NBW, int U>
struct WgmBuf {
  f32x4 a[U][MB];
  f32x4 b[U][NBW];
};

template <int MB, int NBW, int U>
__global__ __launch_bounds__(256, 2) void conv_wgrad_mfma_kernel(const WgmArgs a) {
  __shared__ __attribute__((aligned(16))) float red[3][WGM_RT][64][4];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i16 = lane & 15, kq = lane >> 4;
  const int nb0 = blockIdx.y * NBW;
  const __amdgpu_buffer_rsrc_t rd = wgm_rsrc(a.dy, a.dy_bytes), rx = wgm_rsrc(a.x, a.x_bytes);

  // loop-invariant lane parts of the interior addresses
  uint32_t voA[MB], voB[NBW];
#pragma unroll
  for (int m = 0; m < MB; ++m) {
    const int co = 16 * m + i16;
    voA[m] = co < a.Cout ? (uint32_t)(co * a.Lout + 4 * kq) * 4u : WGM_OOB;
  }
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const uint32_t en = (uint32_t)((nb0 + nb) * 16 + i16);
    const uint32_t ci = a.dK.div(en), k = en - ci * (uint32_t)a.K;
    voB[nb] = en < (uint32_t)a.entries ? (ci * (uint32_t)a.Lin + 4u * kq + k) * 4u : WGM_OOB;
  }
  const int bias_nb = a.entries >> 4, bias_lane = a.entries & 15;      // the column of ones

  f32x4 acc[MB][NBW];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) acc[m][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load = [&](WgmBuf<MB, NBW, U>& t, int g) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = g * U + u;                                   // wave-uniform
      const uint32_t P0 = 16u * (uint32_t)s;
      const uint32_t b0 = a.dLout.div(P0 < a.total ? P0 : 0u);
      const uint32_t l0 = P0 - b0 * (uint32_t)a.Lout;
      const bool fast = a.fast_ok && s < a.segs && (int)l0 >= a.pad && (int)l0 + 16 + a.K - 1 - a.pad <= a.Lin;
      if (fast) {
        const uint32_t sA = (b0 * (uint32_t)(a.Cout * a.Lout) + l0) * 4u;
        const uint32_t sB = (b0 * (uint32_t)(a.Cin * a.Lin) + l0 - (uint32_t)a.pad) * 4u;
#pragma unroll
        for (int m = 0; m < MB; ++m) t.a[u][m] = wgm_ld4(rd, voA[m], sA);
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) t.b[u][nb] = wgm_ld4(rx, voB[nb], sB);      // (a block past the row: refused, zeros)
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // every offset is computed unconditionally and then replaced by the refused one with a select: a branch around a load
          // would serialise the loads of the segment (a wait per arm)
          const uint32_t P = P0 + 4u * kq + e;
          const bool ok = P < a.total && s < a.segs;
          const uint32_t bb = a.dLout.div(P);
          const uint32_t l = P - bb * (uint32_t)a.Lout;
#pragma unroll
          for (int m = 0; m < MB; ++m) {
            const int co = 16 * m + i16;
            uint32_t off = ((bb * (uint32_t)a.Cout + co) * (uint32_t)a.Lout + l) * 4u;
            asm volatile("" : "+v"(off));
            t.a[u][m][e] = wgm_ld1(rd, (ok && co < a.Cout) ? off : WGM_OOB);
          }
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb) {
            const uint32_t en = (uint32_t)((nb0 + nb) * 16 + i16);
            const uint32_t ci = a.dK.div(en), k = en - ci * (uint32_t)a.K;
            const int tpos = (int)l * a.stride + (int)k - a.pad;
            const bool okb = ok && en < (uint32_t)a.entries && tpos >= 0 && tpos < a.Lin * a.up;
            const uint32_t xi = a.dUp.div((uint32_t)tpos);
            uint32_t off = ((bb * (uint32_t)a.Cin + ci) * (uint32_t)a.Lin + xi) * 4u;
            asm volatile("" : "+v"(off));
            t.b[u][nb][e] = wgm_ld1(rx, okb ? off : WGM_OOB);
          }
        }
      }
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb)
        if (nb0 + nb == bias_nb && i16 == bias_lane) t.b[u][nb] = f32x4{1.f, 1.f, 1.f, 1.f};
    }
  };
  auto compute = [&](const WgmBuf<MB, NBW, U>& t) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb)
            acc[m][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(t.a[u][m][e], t.b[u][nb][e], acc[m][nb], 0, 0, 0);
  };

  const int step = gridDim.x * 4;
  int g = blockIdx.x * 4 + w;
  WgmBuf<MB, NBW, U> t0, t1;
  if (g < a.groups) load(t0, g);
  while (g < a.groups) {
    const int g1 = g + step, g2 = g + 2 * step;
    if (g1 < a.groups) load(t1, g1);
    compute(t0);
    if (g1 >= a.groups) break;
    if (g2 < a.groups) load(t0, g2);
    compute(t1);
    g = g2;
  }

  // the four waves' tiles, WGM_RT at a time: waves 1..3 park theirs, wave 0 adds them in wave order and writes the partial row
  constexpr int NT = MB * NBW;
  float* prow = a.part + (size_t)blockIdx.x * a.Cout * a.rowlen;
#pragma unroll
  for (int r0 = 0; r0 < NT; r0 += WGM_RT) {
    __syncthreads();
    if (w > 0) {
#pragma unroll
      for (int r = 0; r < WGM_RT; ++r)
        if (r0 + r < NT) *reinterpret_cast<f32x4*>(&red[w - 1][r][lane][0]) = acc[(r0 + r) / NBW][(r0 + r) % NBW];
    }
    __syncthreads();
    if (w == 0) {
#pragma unroll
      for (int r = 0; r < WGM_RT; ++r)
        if (r0 + r < NT) {
          const int m = (r0 + r) / NBW, nb = (r0 + r) % NBW;
          f32x4 v = acc[m][nb];
#pragma unroll
          for (int ww = 0; ww < 3; ++ww) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(&red[ww][r][lane][0]);
            v = f32x4{v.x + o.x, v.y + o.y, v.z + o.z, v.w + o.w};
          }
          const int en = (nb0 + nb) * 16 + i16;
          if (en < a.rowlen) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int co = 16 * m + 4 * kq + j;
              if (co < a.Cout) prow[(size_t)co * a.rowlen + en] = v[j];
            }
          }
        }
    }
  }
}

DivWide make_div_wide(uint32_t d, uint64_t bound) {      // exact for n <= bound; bound * d < 2^62
  uint32_t S = 0;
  while ((1ull << S) <= bound * d) ++S;
  DivWide r;
  r.S = S;
  r.M = (uint32_t)(((1ull << S) + d - 1) / d);
  return r;
}

}  // namespace

// part <- `chunks` partial rows [Cout][Cin * K + 1] (at most max_chunks); returns the number written in *chunks_out.
// Returns MURAL_E_UNSUPPORTED-like 1 (without touching the stream) when the shape is outside what the kernel covers: the caller
// falls back to the vector-ALU kernel.
int launch_conv_wgrad_mfma(const float* dy, const float* x, float* part, int64_t B, int Cin, int Lin, int Cout, int Lout, int K, int stride,
                           int pad, int up, int max_chunks, int* chunks_out, hipStream_t st) {
  const uint64_t total = (uint64_t)B * Lout;
  const uint64_t dyb = total * Cout * 4ull, xb = (uint64_t)B * Cin * Lin * 4ull;
  if (Cout > 96 || total + 64 >= (1ull << 28) || dyb >= (1ull << 31) || xb >= (1ull << 31) || (uint64_t)Lin * up * up >= (1ull << 31) ||
      (uint64_t)(Cin * K + 16) * K >= (1ull << 31))
    return 1;
  if (up > 1 && total > 65536) return 1;      // long upsampled rows are all gathers here: the vector-ALU kernel's LDS tile is faster
  WgmArgs a;
  a.dy = dy; a.x = x; a.part = part;
  a.B = (int)B; a.Cin = Cin; a.Lin = Lin; a.Cout = Cout; a.Lout = Lout; a.K = K; a.stride = stride; a.pad = pad; a.up = up;
  a.entries = Cin * K;
  a.rowlen = a.entries + 1;
  a.total = (uint32_t)total;
  a.segs = (int)((total + 15) / 16);
  a.fast_ok = (stride == 1 && up == 1 && Lout % 16 == 0) ? 1 : 0;
  a.dy_bytes = (uint32_t)dyb;
  a.x_bytes = (uint32_t)xb;
  a.dLout = make_div_wide((uint32_t)Lout, total + 64);
  a.dK = FastDiv::make((uint32_t)K);
  a.dUp = FastDiv::make((uint32_t)up);
  const int MB = (Cout + 15) / 16, nblocks = (a.rowlen + 15) / 16;
  int nbw, U;
  if (MB == 1) { nbw = nblocks <= 4 ? 4 : 8; U = nblocks <= 4 ? 4 : 1; }
  else if (MB == 2) { nbw = nblocks <= 4 ? 4 : 8; U = nblocks <= 4 ? 2 : 1; }
  else if (MB == 3) { nbw = 8; U = 1; }
  else if (MB == 4) { nbw = 6; U = 1; }
  else { nbw = 4; U = 1; }
  const int gy = (nblocks + nbw - 1) / nbw;
  a.groups = (a.segs + U - 1) / U;
  // workgroups: long problems get about four groups per wave (two workgroups per CU and column group are resident); short ones --
  // the deep levels' few thousand positions -- one group per wave, because a wave's groups are a serial chain of load round trips
  const int cap = std::max(1, std::min(max_chunks, std::max(64, 1024 / gy)));
  int chunks = std::max((a.groups + 15) / 16, std::min((a.groups + 3) / 4, std::max(1, 512 / gy)));
  chunks = std::max(1, std::min(chunks, cap));
  const dim3 grid(chunks, gy);
#define MURAL_WGM(MB_, NBW_, U_) hipLaunchKernelGGL((conv_wgrad_mfma_kernel<MB_, NBW_, U_>), grid, dim3(256), 0, st, a)
  if (MB == 1) { if (nbw == 4) MURAL_WGM(1, 4, 4); else MURAL_WGM(1, 8, 1); }
  else if (MB == 2) { if (nbw == 4) MURAL_WGM(2, 4, 2); else MURAL_WGM(2, 8, 1); }
  else if (MB == 3) MURAL_WGM(3, 8, 1);
  else if (MB == 4) MURAL_WGM(4, 6, 1);
  else if (MB == 5) MURAL_WGM(5, 4, 1);
  else MURAL_WGM(6, 4, 1);
#undef MURAL_WGM
  MURAL_HIP_CHECK(hipGetLastError());
  *chunks_out = chunks;
  return MURAL_OK;
}

}  // namespace mural

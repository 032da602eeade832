// Weight gradient of the general Conv1d (indel_train.hip: mural_op_convg_bwd) as an implicit GEMM on v_mfma_f32_16x16x4_f32.
//
//   dW[co][ci][k] = sum over (b, lo) of dy[b][co][lo] * xu[b][ci][lo * stride + k - pad]        (xu = x upsampled by `up`, zero outside)
//   db[co]        = sum over (b, lo) of dy[b][co][lo]
//
// is D = A * B with A = dy as [co][position], B = [position][entry] (entry = ci * K + k) and the flattened (b, lo) axis as the
// reduction dimension: rows of 16 channels x columns of 16 entries per MFMA tile, four positions per instruction.  Reference: the
// gradients torch's autograd produces for nn.Conv1d inside UNet_Small (MuRaL/model/model_indel.py:6-19, :100-131) in the step of
// MuRaL/training.py:424-436.
//
// A wave walks segments of 16 consecutive positions.  Lane (i = lane & 15, q = lane >> 4) holds positions 4q .. 4q+3 of the segment as
// one 16-byte register quad per tile row / column block: element e of every quad forms one k-step (the reduction set of step e is the
// four positions {4q + e}), so both operands come straight from global memory with one buffer_load_dwordx4 per block -- for a stride-1
// conv the column block's quad is the input row shifted by the tap, an unaligned (4-byte aligned) 16-byte load that the L1 serves K
// times.  The wave-uniform part of every address travels in the scalar offset, the per-lane part is loop-invariant: no address VALU
// in the interior of a row.  db is the plain sum of the A quads (vector ALU, a few adds per segment).
//
// Two loops.  INTERIOR: groups of U consecutive segments that lie inside a row of a stride-1 conv; loads run one group ahead of the
// MFMAs (register double buffer).  This loop has NO branch around a load and always fetches a next group (the last one re-fetches
// itself): with loads under control flow the compiler's wait-count bookkeeping merges the paths pessimistically and waits for the
// prefetch it has just issued (measured: vmcnt(0) in front of every MFMA block).  EDGE: the segments that touch a row end, the
// leftovers of a row, and every segment of a ragged / strided / upsampled conv take the gathered form (four 4-byte loads per block,
// offsets checked per element with selects; a refused element aims past the descriptor and reads 0), one segment at a time.
// The four waves of a workgroup add their tiles through LDS in a fixed order: one partial row [Cout][entries + 1] per workgroup,
// reduced across workgroups by the caller (conv_wgrad_reduce_multi_kernel) -- bitwise reproducible.
#include <algorithm>

#include "common.h"
#include "mfma_tile.h"

namespace mural {
namespace {

constexpr uint32_t WGM_OOB = 0x80000000u;      // a voffset at or past num_records: the load returns 0
constexpr int WGM_RT = 4;                      // tiles per round of the cross-wave sum (16 KB of LDS)

struct WgmArgs {
  const float* dy;
  const float* x;
  float* part;
  int B, Cin, Lin, Cout, Lout, K, stride, pad, up;
  int entries, rowlen;
  int e0, gr, ss;          // interior: first interior segment of a row, groups per row, edge-loop segments per row
  int ngroups, nslow;      // interior groups / edge-loop segments of the launch
  int segs_row;            // segments per row (interior mode), 0: the edge loop walks the flattened (b, lo) axis
  uint32_t total, dy_bytes, x_bytes;
  DivWide dLout, dGr, dSs;
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

  f32x4 acc[MB][NBW];
  f32x4 bsum[MB];                               // db: this lane's share of sum(dy) of row 16 m + i16 (positions 4 kq + e)
#pragma unroll
  for (int m = 0; m < MB; ++m) {
    bsum[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) acc[m][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // one segment's MFMAs: 4 k-steps x MB x NBW
  auto mma = [&](const f32x4 (&av)[MB], const f32x4 (&bv)[NBW]) {
#pragma unroll
    for (int m = 0; m < MB; ++m) bsum[m] += av[m];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[m][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][e], bv[nb][e], acc[m][nb], 0, 0, 0);
  };

  const int nwaves = gridDim.x * 4;
  const int wid = blockIdx.x * 4 + w;
  // ---------------------------------------------------------------------------------------------- interior groups
  {
    auto load = [&](WgmBuf<MB, NBW, U>& t, int grp) {      // grp < a.ngroups; wave-uniform; no branch
      const uint32_t b = a.dGr.div((uint32_t)grp);
      const uint32_t l0 = (uint32_t)(a.e0 + ((uint32_t)grp - b * (uint32_t)a.gr) * U) * 16u;
      const uint32_t sA = (b * (uint32_t)(a.Cout * a.Lout) + l0) * 4u;
      const uint32_t sB = (b * (uint32_t)(a.Cin * a.Lin) + l0 - (uint32_t)a.pad) * 4u;
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int m = 0; m < MB; ++m) t.a[u][m] = wgm_ld4(rd, voA[m], sA + 64u * u);
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) t.b[u][nb] = wgm_ld4(rx, voB[nb], sB + 64u * u);      // (a block past the row: refused, zeros)
      }
    };
    auto compute = [&](const WgmBuf<MB, NBW, U>& t) {
#pragma unroll
      for (int u = 0; u < U; ++u) mma(t.a[u], t.b[u]);
    };
    int cur = wid;
    if (cur < a.ngroups) {
      WgmBuf<MB, NBW, U> t0, t1;
      load(t0, cur);
      // (scheduling barriers: left alone, the scheduler sinks the next group's loads down to their first use to save registers --
      // and the MFMAs then wait out every round trip)
      for (;;) {
        int nx = cur + nwaves;
        load(t1, nx < a.ngroups ? nx : cur);
        __builtin_amdgcn_sched_barrier(0);
        compute(t0);
        __builtin_amdgcn_sched_barrier(0);
        cur = nx;
        if (cur >= a.ngroups) break;
        nx = cur + nwaves;
        load(t0, nx < a.ngroups ? nx : cur);
        __builtin_amdgcn_sched_barrier(0);
        compute(t1);
        __builtin_amdgcn_sched_barrier(0);
        cur = nx;
        if (cur >= a.ngroups) break;
      }
    }
  }
  // ---------------------------------------------------------------------------------------------- edge segments
  for (int t = wid; t < a.nslow; t += nwaves) {
    uint32_t P0;
    if (a.segs_row) {      // (row, k-th edge segment of the row): the e0 leading ones, then those behind the interior groups
      const uint32_t b = a.dSs.div((uint32_t)t);
      const int k = t - (int)b * a.ss;
      const int si = k < a.e0 ? k : a.e0 + a.gr * U + (k - a.e0);
      P0 = b * (uint32_t)a.Lout + 16u * (uint32_t)si;
    } else {
      P0 = 16u * (uint32_t)t;
    }
    f32x4 av[MB], bv[NBW];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // every offset is computed unconditionally and then replaced by the refused one with a select: a branch around a load
      // serialises the loads of the segment (a wait per arm)
      const uint32_t P = P0 + 4u * kq + e;
      const bool ok = P < a.total;
      const uint32_t bb = a.dLout.div(P);
      const uint32_t l = P - bb * (uint32_t)a.Lout;
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const int co = 16 * m + i16;
        uint32_t off = ((bb * (uint32_t)a.Cout + co) * (uint32_t)a.Lout + l) * 4u;
        asm volatile("" : "+v"(off));
        av[m][e] = wgm_ld1(rd, (ok & (co < a.Cout)) ? off : WGM_OOB);      // (& not &&: no short-circuit branches)
      }
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) {
        const uint32_t en = (uint32_t)((nb0 + nb) * 16 + i16);
        const uint32_t ci = a.dK.div(en), k = en - ci * (uint32_t)a.K;
        const int tpos = (int)l * a.stride + (int)k - a.pad;
        const bool okb = ok & (en < (uint32_t)a.entries) & (tpos >= 0) & (tpos < a.Lin * a.up);
        const uint32_t xi = a.dUp.div((uint32_t)tpos);
        uint32_t off = ((bb * (uint32_t)a.Cin + ci) * (uint32_t)a.Lin + xi) * 4u;
        asm volatile("" : "+v"(off));
        bv[nb][e] = wgm_ld1(rx, okb ? off : WGM_OOB);
      }
    }
    mma(av, bv);
  }

  // ---------------------------------------------------------------------------------------------- the workgroup's partial row
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
          if (en < a.entries) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int co = 16 * m + 4 * kq + j;
              if (co < a.Cout) prow[(size_t)co * a.rowlen + en] = v[j];
            }
          }
        }
    }
  }
  // db (column `entries` of the partial row; the first column group's workgroups only): the four position quarters of a row sit
  // in lanes i16, i16 + 16, + 32, + 48; then the waves in order
  if (blockIdx.y == 0) {
    __syncthreads();
    float* bs = &red[0][0][0][0];      // [4 waves][MB][16]
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      float v = (bsum[m].x + bsum[m].y) + (bsum[m].z + bsum[m].w);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (kq == 0) bs[(w * MB + m) * 16 + i16] = v;
    }
    __syncthreads();
    if (threadIdx.x < MB * 16) {
      const int m = threadIdx.x >> 4, i = threadIdx.x & 15, co = 16 * m + i;
      const float v = ((bs[(0 * MB + m) * 16 + i] + bs[(1 * MB + m) * 16 + i]) + bs[(2 * MB + m) * 16 + i]) + bs[(3 * MB + m) * 16 + i];
      if (co < a.Cout) prow[(size_t)co * a.rowlen + a.entries] = v;
    }
  }
}

}  // namespace

// part <- `chunks` partial rows [Cout][Cin * K + 1] (at most max_chunks); returns the number written in *chunks_out.
// Returns 1 (without touching the stream) when the shape is outside what the kernel covers: the caller falls back to the
// vector-ALU kernel.
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
  a.dy_bytes = (uint32_t)dyb;
  a.x_bytes = (uint32_t)xb;
  a.dLout = DivWide::make((uint32_t)Lout, total + 64);
  a.dK = FastDiv::make((uint32_t)K);
  a.dUp = FastDiv::make((uint32_t)up);
  const int MB = (Cout + 15) / 16, nblocks = (a.entries + 15) / 16;
  // column blocks per wave: no wider than the row of entries needs (a block past the row still costs its refused load and its MFMAs)
  int nbw;
  if (MB == 1) nbw = nblocks <= 1 ? 1 : (nblocks <= 2 ? 2 : (nblocks <= 4 ? 4 : 8));
  else if (MB == 2) nbw = nblocks <= 2 ? 2 : (nblocks <= 4 ? 4 : (nblocks <= 6 ? 6 : 8));
  else if (MB == 3) nbw = 8;
  else if (MB == 4) nbw = 6;
  else nbw = 4;
  const int gy = (nblocks + nbw - 1) / nbw;
  const bool interior = stride == 1 && up == 1 && Lout % 16 == 0;
  const int cap = std::max(1, std::min(max_chunks, std::max(64, 512 / gy)));      // (two workgroups per CU are resident; the caller reduces `chunks` rows)
  int chunks = 1;
  // interior geometry for U segments per group; then workgroups: long problems get about four groups per wave (two workgroups per CU
  // and column group are resident), short ones -- the deep levels' few thousand positions -- one unit of work per wave, because a
  // wave's units are a serial chain of load round trips
  auto plan = [&](int U) {
    const int SR = Lout / 16;
    a.segs_row = 0; a.e0 = 0; a.gr = 0; a.ss = 1; a.ngroups = 0;
    a.nslow = (int)((total + 15) / 16);
    if (interior) {
      const int e0 = (pad + 15) / 16;                                   // leading segments with l0 < pad
      const int last_ok = (Lin - 16 - (K - 1 - pad)) >= 0 ? (Lin - 16 - (K - 1 - pad)) / 16 : -1;      // l0 + 16 + K - 1 - pad <= Lin
      const int ir = std::max(0, std::min(SR - 1, last_ok) - e0 + 1);
      if (ir / U > 0) {
        a.segs_row = SR; a.e0 = e0; a.gr = ir / U; a.ss = SR - a.gr * U;
        a.ngroups = (int)B * a.gr;
        a.nslow = (int)B * a.ss;
      }
    }
    a.dGr = DivWide::make((uint32_t)std::max(1, a.gr), (uint64_t)a.ngroups + 64);
    a.dSs = DivWide::make((uint32_t)std::max(1, a.ss), (uint64_t)a.nslow + 64);
    const int units = a.ngroups + a.nslow;      // (an edge segment is a full round trip of its own: count it like a group)
    const int c = std::max((units + 15) / 16, std::min((units + 3) / 4, std::max(1, 512 / gy)));
    chunks = std::max(1, std::min(c, cap));
  };
#define MURAL_WGM(MB_, NBW_, U_)                                                                         \
  do {                                                                                                  \
    plan(U_);                                                                                           \
    hipLaunchKernelGGL((conv_wgrad_mfma_kernel<MB_, NBW_, U_>), dim3(chunks, gy), dim3(256), 0, st, a); \
  } while (0)
  if (MB == 1) {
    if (nbw == 1) MURAL_WGM(1, 1, 8);
    else if (nbw == 2) MURAL_WGM(1, 2, 4);
    else if (nbw == 4) MURAL_WGM(1, 4, 4);
    else MURAL_WGM(1, 8, 1);
  } else if (MB == 2) {
    if (nbw == 2) MURAL_WGM(2, 2, 2);
    else if (nbw == 4) MURAL_WGM(2, 4, 2);
    else if (nbw == 6) MURAL_WGM(2, 6, 1);
    else MURAL_WGM(2, 8, 1);
  } else if (MB == 3) MURAL_WGM(3, 8, 1);
  else if (MB == 4) MURAL_WGM(4, 6, 1);
  else if (MB == 5) MURAL_WGM(5, 4, 1);
  else MURAL_WGM(6, 4, 1);
#undef MURAL_WGM
  MURAL_HIP_CHECK(hipGetLastError());
  *chunks_out = chunks;
  return MURAL_OK;
}

}  // namespace mural

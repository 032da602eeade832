// Device code shared by the kernels that run 32->32 k=3 conv layers on the swizzled LDS image of a flattened column axis
// (snv_tower.hip: the per-window tower kernel; snv_reuse.hip: the window-edge kernel of the cross-position reuse path):
// layer roles of a ResBlock stage (MuRaL/model/model_snv.py:477-485, :794-812), per-wave addressing, the branch-free epilogue
// and the software-pipelined implicit-GEMM layer on v_mfma_f32_16x16x4_f32.
#pragma once
#include "mfma_tile.h"
#include "snv.h"

namespace mural {

enum { MODE_PLAIN = 0, MODE_RES_FIRST = 1, MODE_RES_LAST = 2, MODE_ENTRY = 3, MODE_FINAL = 4 };

__device__ __forceinline__ int layer_mode(int layer) {
  // 0 P,1 RF,2 P,3 RL,4 E,5 P,6 RF,7 P,8 RL,9 F   packed 3 bits per layer
  const uint32_t tbl = (0u) | (1u << 3) | (0u << 6) | (2u << 9) | (3u << 12) | (0u << 15) | (1u << 18) | (0u << 21) |
                       (2u << 24) | (4u << 27);
  return (int)((tbl >> (3 * layer)) & 7u);
}

// Per-wave addressing of one stage.  A wave owns M-block `mb` (16 output channels) of the 16-column blocks
// b = cgp + 2i; block i of the wave sits 32 columns = 4096 bytes after block 0, which the swizzle leaves intact,
// so every LDS address of the conv loop is one of these VGPRs plus a compile-time immediate.
struct StageAddr {
  uint32_t rd[6];   // byte offset of B-operand chunk (tap t, half h) for block 0: rd[2t+h]
  uint32_t wr;      // byte offset of this lane's output chunk for block 0
  uint32_t vmask;   // bit i: column of block i held by this lane carries data (not separator / padding)
};

__device__ __forceinline__ StageAddr stage_setup(const TowerGeom& g, int st, int P, int n16, int kk, int mb, int cgp) {
  StageAddr a;
  const int c0 = 16 * cgp + n16;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) a.rd[2 * t + h] = 4u * (uint32_t)lds_off(c0 + t, 4 * h + kk);
  a.wr = 4u * (uint32_t)lds_off(c0 + 1, 4 * mb + kk);
  a.vmask = 0;
#pragma unroll
  for (int i = 0; i < SNV_NB2MAX; ++i)
    a.vmask |= col_is_data(c0 + 32 * i, g.dSc[st], g.Sc[st], g.L[st], P) ? (1u << i) : 0u;
  return a;
}

// Branch-free epilogue.  Per-layer scalars select the role of the layer (see layer_mode):
//   out  = valid ? ps * max(acc, lo) + pt : 0      lo = 0 (ReLU) or -inf (raw); raw layers carry ps = 1, pt = 0
//   xres = ku * acc + kx * xres                    (1,1) first conv2 of a group: z = x1 + x0 keeps the outer skip
//                                                  (model_snv.py:477-479); (1,0) entry conv; (0,1) otherwise
struct LayerK { float lo, ku, kx, kr; };

__device__ __forceinline__ LayerK layer_consts(int mode) {
  LayerK k;
  k.lo = (mode == MODE_RES_LAST) ? -INFINITY : 0.f;
  k.ku = (mode == MODE_RES_FIRST || mode == MODE_ENTRY) ? 1.f : 0.f;
  k.kx = (mode == MODE_ENTRY) ? 0.f : 1.f;
  k.kr = (mode == MODE_RES_FIRST || mode == MODE_RES_LAST) ? 1.f : 0.f;   // accumulator starts from bias + kr * xres
  return k;
}

__device__ __forceinline__ void epilogue(const LayerK& k, f32x4 acc, f32x4& xr, bool valid, f32x4 ps, f32x4 pt, char* out,
                                         uint32_t off) {
  f32x4 o;
  o.x = fmaf(ps.x, fmaxf(acc.x, k.lo), pt.x);
  o.y = fmaf(ps.y, fmaxf(acc.y, k.lo), pt.y);
  o.z = fmaf(ps.z, fmaxf(acc.z, k.lo), pt.z);
  o.w = fmaf(ps.w, fmaxf(acc.w, k.lo), pt.w);
  xr.x = fmaf(acc.x, k.ku, xr.x * k.kx);
  xr.y = fmaf(acc.y, k.ku, xr.y * k.kx);
  xr.z = fmaf(acc.z, k.ku, xr.z * k.kx);
  xr.w = fmaf(acc.w, k.ku, xr.w * k.kx);
  lds_st4(out, off, valid ? o : splat(0.f));
}

__device__ __forceinline__ f32x4 acc_init(const LayerK& k, f32x4 pb, f32x4 xr) {
  return f32x4{fmaf(xr.x, k.kr, pb.x), fmaf(xr.y, k.kr, pb.y), fmaf(xr.z, k.kr, pb.z), fmaf(xr.w, k.kr, pb.w)};
}

// One 32->32 k=3 conv layer for this wave: its M-block against the 16-column blocks it owns.  Full pairs of blocks
// run as two independent accumulator chains, software-pipelined: tap-0 operands were read during the previous
// pair, tap-1/2 reads are in flight under the tap-0/1 MFMAs, and the previous pair's epilogue issues between this
// pair's MFMA groups.  An odd last block runs as a single chain.
__device__ __forceinline__ void conv_layer(const char* in, char* out, const StageAddr& sa, int nbw, const LayerK& k,
                                           const float (&a)[SNV_KSTEPS], f32x4 pb, f32x4 ps, f32x4 pt,
                                           f32x4 (&xres)[SNV_NB2MAX]) {
  constexpr int NPF = SNV_NB2MAX / 2;   // full pairs that fit the register file
  const int nfull = nbw >> 1;
  f32x4 X0[2], X1[2];
  f32x4 pa0 = splat(0.f), pa1 = splat(0.f);
  if (nfull > 0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      X0[h] = lds_ld4(in, sa.rd[h]);
      X1[h] = lds_ld4(in, sa.rd[h] + 4096u);
    }
  }
#pragma unroll
  for (int ip = 0; ip < NPF; ++ip) {
    const int i0 = 2 * ip, i1 = 2 * ip + 1;
    if (ip < nfull) {
      f32x4 Y0[2], Y1[2], Z0[2], Z1[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        Y0[h] = lds_ld4(in, sa.rd[2 + h] + 4096u * i0);
        Y1[h] = lds_ld4(in, sa.rd[2 + h] + 4096u * i1);
      }
      f32x4 acc0 = acc_init(k, pb, xres[i0]), acc1 = acc_init(k, pb, xres[i1]);
      __builtin_amdgcn_sched_barrier(0);
      mfma_tap<true, SNV_KSTEPS>(a, 0, X0, X1, acc0, acc1);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        Z0[h] = lds_ld4(in, sa.rd[4 + h] + 4096u * i0);
        Z1[h] = lds_ld4(in, sa.rd[4 + h] + 4096u * i1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (ip > 0) {   // epilogue of the previous pair rides under this pair's tap-1 MFMAs
        epilogue(k, pa0, xres[i0 - 2], (sa.vmask >> (i0 - 2)) & 1u, ps, pt, out, sa.wr + 4096u * (i0 - 2));
        epilogue(k, pa1, xres[i0 - 1], (sa.vmask >> (i0 - 1)) & 1u, ps, pt, out, sa.wr + 4096u * (i0 - 1));
      }
      mfma_tap<true, SNV_KSTEPS>(a, 1, Y0, Y1, acc0, acc1);
      if (ip > 0) {   // spread the epilogue's VALU work into the MFMA issue gaps
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // 3 VALU
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);     // the two ds_write_b128
      }
      __builtin_amdgcn_sched_barrier(0);
      if (ip + 1 < nfull) {   // tap-0 operands of the next pair
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          X0[h] = lds_ld4(in, sa.rd[h] + 4096u * (i0 + 2));
          X1[h] = lds_ld4(in, sa.rd[h] + 4096u * (i0 + 3));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_tap<true, SNV_KSTEPS>(a, 2, Z0, Z1, acc0, acc1);
      pa0 = acc0;
      pa1 = acc1;
    } else if (ip > 0 && ip == nfull) {   // the previous pair was the last full one
      epilogue(k, pa0, xres[i0 - 2], (sa.vmask >> (i0 - 2)) & 1u, ps, pt, out, sa.wr + 4096u * (i0 - 2));
      epilogue(k, pa1, xres[i0 - 1], (sa.vmask >> (i0 - 1)) & 1u, ps, pt, out, sa.wr + 4096u * (i0 - 1));
    }
  }
  if (nfull == NPF) {
    epilogue(k, pa0, xres[2 * NPF - 2], (sa.vmask >> (2 * NPF - 2)) & 1u, ps, pt, out, sa.wr + 4096u * (2 * NPF - 2));
    epilogue(k, pa1, xres[2 * NPF - 1], (sa.vmask >> (2 * NPF - 1)) & 1u, ps, pt, out, sa.wr + 4096u * (2 * NPF - 1));
  }
  if (nbw & 1) {   // odd last block: a single accumulator chain
#pragma unroll
    for (int i0 = 0; i0 < SNV_NB2MAX; i0 += 2) {
      if (nbw - 1 == i0) {
        f32x4 S0[2], S1[2], S2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          S0[h] = lds_ld4(in, sa.rd[0 + h] + 4096u * i0);
          S1[h] = lds_ld4(in, sa.rd[2 + h] + 4096u * i0);
          S2[h] = lds_ld4(in, sa.rd[4 + h] + 4096u * i0);
        }
        f32x4 acc0 = acc_init(k, pb, xres[i0]), acc1 = acc0;
        __builtin_amdgcn_sched_barrier(0);
        mfma_tap<false, SNV_KSTEPS>(a, 0, S0, S0, acc0, acc1);
        mfma_tap<false, SNV_KSTEPS>(a, 1, S1, S1, acc0, acc1);
        mfma_tap<false, SNV_KSTEPS>(a, 2, S2, S2, acc0, acc1);
        epilogue(k, acc0, xres[i0], (sa.vmask >> i0) & 1u, ps, pt, out, sa.wr + 4096u * i0);
      }
    }
  }
}

// Workgroup barrier for LDS-only hand-offs: __syncthreads() also drains vmcnt, i.e. it would wait for the stage-1
// activations requested one tower ahead (request_x0) at the very next barrier and expose the full HBM latency per tower.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


}  // namespace mural

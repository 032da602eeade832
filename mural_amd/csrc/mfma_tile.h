// Shared device helpers of the MFMA conv kernels: the swizzled LDS image [column][32 channels] of a flattened column
// axis (zero separator columns between rows) and the v_mfma_f32_16x16x4_f32 operand conventions.
#pragma once
#include <cstdlib>
#include "common.h"

namespace mural {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// LDS image [column][32 channels]: the eight 16-byte chunks of a column are XOR-permuted by a per-column key chosen
// (exhaustive search over 16-entry tables, tools/lds_swizzle_search.py) so that the ds_read_b128 operand reads of all
// three conv taps are bank-conflict-free for every 16-lane group of the instruction; the key depends on column mod 16,
// so a wave's blocks, 32 columns apart, keep one base address + immediates.
__device__ __forceinline__ int lds_key(int pc) {
  return (int)((0x2e4c11ee4587ull >> (3 * (pc & 15))) & 7ull);
}
__device__ __forceinline__ int lds_off(int pc, int chunk) { return pc * 32 + ((chunk ^ lds_key(pc)) << 2); }

// Wave index of (workgroup, wave w of 4) in launches whose waves walk consecutive 16-column segments.  Workgroups are handed to the
// eight XCDs round-robin (workgroup b runs on XCD b % 8, each with an L2 of its own), and a 16-column segment is 64 bytes of a channel
// row: with workgroup b on segments 4 b .. 4 b + 3 every 128-byte line is fetched by two XCDs (r05 PMC: the 16-channel INDEL block read
// 1.8 x its input from the fabric).  Slot (b % 8) * (gridDim / 8) + b / 8 gives the waves of one XCD ONE contiguous range per sweep.
__device__ __forceinline__ int xcd_wave_index(int w, int on) {
  const int b = blockIdx.x, nb = gridDim.x;
  const int slot = (on && (nb & 7) == 0) ? (b & 7) * (nb >> 3) + (b >> 3) : b;
  return slot * 4 + w;
}
inline int xcd_swizzle_enabled() {      // MURAL_XCD_SWIZZLE=0: workgroup b takes slot b (A/B switch)
  const char* e = dev_env("MURAL_XCD_SWIZZLE");
  return !(e && atoi(e) == 0);
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 splat(float v) { return f32x4{v, v, v, v}; }
__device__ __forceinline__ f32x4 max4(f32x4 a, f32x4 b) {
  return f32x4{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)};
}
__device__ __forceinline__ f32x4 relu_bn(f32x4 v, f32x4 s, f32x4 t) {
  return f32x4{fmaf(s.x, fmaxf(v.x, 0.f), t.x), fmaf(s.y, fmaxf(v.y, 0.f), t.y), fmaf(s.z, fmaxf(v.z, 0.f), t.z),
               fmaf(s.w, fmaxf(v.w, 0.f), t.w)};
}

// logical column c of a flattened stage geometry holds data (not separator / padding)?
__device__ __forceinline__ bool col_is_data(int c, const FastDiv& dSc, int Sc, int Lv, int P) {
  if (c < 1) return false;
  uint32_t u = (uint32_t)(c - 1);
  uint32_t p = dSc.div(u);
  uint32_t j = u - p * (uint32_t)Sc;
  return (p < (uint32_t)P) && (j < (uint32_t)Lv);
}


// 8 k-steps (one conv tap: 2 halves x 4) of v_mfma_f32_16x16x4_f32 for one (DUAL: two) 16-column block(s)
template <bool DUAL, int NK>
__device__ __forceinline__ void mfma_tap(const float (&a)[NK], int t, const f32x4 (&b0)[2], const f32x4 (&b1)[2],
                                         f32x4& acc0, f32x4& acc1) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[8 * t + 4 * h + q], b0[h][q], acc0, 0, 0, 0);
      if (DUAL) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[8 * t + 4 * h + q], b1[h][q], acc1, 0, 0, 0);
    }
  }
}

__device__ __forceinline__ f32x4 lds_ld4(const char* base, uint32_t off) {
  return *reinterpret_cast<const f32x4*>(base + off);
}
__device__ __forceinline__ void lds_st4(char* base, uint32_t off, f32x4 v) { *reinterpret_cast<f32x4*>(base + off) = v; }


using f32x2 = __attribute__((ext_vector_type(2))) float;

// 16 bytes at (wave-uniform base + 32-bit lane offset) through a raw buffer descriptor: four SGPRs + one VGPR per load -- no 64-bit
// lane pointers for the compiler to precompute and spill (the flat form kept nine of them in scratch and drained vmcnt per load)
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_ld4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

}  // namespace mural

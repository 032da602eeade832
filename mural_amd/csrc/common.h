// Internal helpers shared by the translation units of libmural_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/mural_hip.h"

namespace mural {

void set_error(const char* fmt, ...);

#define MURAL_HIP_CHECK(expr)                                                                      \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      ::mural::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return MURAL_E_RUNTIME;                                                                      \
    }                                                                                              \
  } while (0)

#define MURAL_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      ::mural::set_error(__VA_ARGS__);      \
      return MURAL_E_INVALID;               \
    }                                       \
  } while (0)

// symbols of the in-LDS base alphabet
enum : uint8_t { SYM_A = 0, SYM_C = 1, SYM_G = 2, SYM_T = 3, SYM_N = 4, SYM_PAD = 15, SYM_BAD = 255 };
constexpr int N_SYM = 16;

// exact unsigned division by a runtime constant: valid while n * d < 2^32
struct FastDiv {
  uint32_t d, m;
  __host__ __device__ static FastDiv make(uint32_t d) {
    FastDiv f;
    f.d = d;
    f.m = (uint32_t)((0x100000000ull / d) + 1ull);
    return f;
  }
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return d == 1 ? n : __umulhi(n, m); }
};

// ---------------------------------------------------------------------------------------------
// packed genome access
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t genome_sym(const uint32_t* __restrict__ packed2,
                                               const uint32_t* __restrict__ nmask, int64_t length, int64_t g) {
  if (g < 0 || g >= length) return SYM_N;
  uint32_t w = packed2[g >> 4];
  uint32_t m = nmask[g >> 5];
  uint32_t two = (w >> (2u * (uint32_t)(g & 15))) & 3u;
  return ((m >> (uint32_t)(g & 31)) & 1u) ? (uint32_t)SYM_N : two;
}

// complement within the 16-symbol alphabet (A<->T, C<->G, N, R<->Y, M<->K, S, W, B<->V, D<->H, PAD)
__device__ __forceinline__ uint32_t sym_complement(uint32_t s) {
  // table packed 4 bits per symbol: index 0..15 -> 3,2,1,0,4,6,5,10,8,9,7,14,13,12,11,15
  const uint64_t tbl = 0xFBCDE798A5640123ull;
  return (uint32_t)((tbl >> (4u * s)) & 0xFull);
}

}  // namespace mural

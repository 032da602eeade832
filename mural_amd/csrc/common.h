// Internal helpers shared by the translation units of libmural_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>

#include "../../include/mural_hip.h"

namespace mural {

// ---- environment switches --------------------------------------------------------------------------------------------------------
// The PRODUCT library (libmural_hip.so) reads two variables: MURAL_HOST_THREADS (host threads of the ingest passes, csrc/ingest.hip)
// and TMPDIR (inflated copies of gzip inputs).  Every other switch is a development switch -- an A/B of two kernels, a validation
// path, a diagnostic, or a timing experiment that produces WRONG results -- listed in ONE table (encode.hip: dev_switch_table) and
// read through dev_env(), which answers "not set" unless the debug flavour is loaded (libmural_hip_debug.so: csrc/debug_hooks.hip sets
// g_dev_switches).  A stray MURAL_* variable cannot change what the product library computes.
struct DevSwitch { const char* name; const char* what; };
const DevSwitch* dev_switch_table();      // terminated by {nullptr, nullptr}
extern bool g_dev_switches;
const char* dev_env(const char* name);    // getenv(name) in the debug flavour (the name must be in the table), nullptr otherwise

void set_error(const char* fmt, ...);
// Validation of the workspace carvers (tests only): MURAL_DEBUG_WS_GUARD=<bytes> in the environment puts that many unused bytes behind
// every region a carver hands out, and the regions of the calling thread's latest carve are kept for mural_debug_last_ws_layout, so
// that a test can poison a workspace, run a call and see that nothing was written outside the regions.
size_t ws_guard_bytes();
void ws_layout_reset();
void ws_layout_add(size_t off, size_t bytes);

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

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to (kernel, device): a launch site raises it once per device ordinal
// (never inside a stream capture after the first call on that device).  Thread-safe; setting it twice is harmless.
struct DynLdsOnce {
  std::atomic<uint64_t> done{0};
  template <typename... Fn>
  int ensure(Fn... fns) {
    int dev = 0;
    MURAL_HIP_CHECK(hipGetDevice(&dev));
    const uint64_t bit = dev < 64 ? (1ull << dev) : 0ull;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return MURAL_OK;
    const void* list[] = {reinterpret_cast<const void*>(fns)...};
    for (const void* f : list)
      MURAL_HIP_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done.fetch_or(bit, std::memory_order_release);
    return MURAL_OK;
  }
};

// symbols of the in-LDS base alphabet
enum : uint8_t { SYM_A = 0, SYM_C = 1, SYM_G = 2, SYM_T = 3, SYM_N = 4, SYM_PAD = 15, SYM_BAD = 255 };
constexpr int N_SYM = 16;

// exact unsigned division by a runtime constant: valid while n * d < 2^32
struct FastDiv {
  uint32_t d, m;
  __host__ __device__ static constexpr FastDiv make(uint32_t d) {
    FastDiv f{};
    f.d = d;
    f.m = (uint32_t)((0x100000000ull / d) + 1ull);
    return f;
  }
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return d == 1 ? n : __umulhi(n, m); }
  // the same without the branch on d == 1 (there m == 1 and the high product is 0): for code that divides per element in long
  // unrolled sequences, where a wave-uniform branch per division costs more instruction fetch than two scalar-masked ALU ops
  __device__ __forceinline__ uint32_t divnb(uint32_t n) const { return __umulhi(n, m) + (n & (0u - (uint32_t)(d == 1))); }
};

// exact n / d for n <= the bound the host built it for: q = (n * M) >> S with 2^S > bound * d (FastDiv's single multiply stops at
// n * d < 2^32: a flattened (row, column) index of a long batch needs the wider product)
struct DivWide {
  uint32_t M, S;
  static DivWide make(uint32_t d, uint64_t bound) {      // bound * d < 2^62, bound < 2^31
    uint32_t S = 0;
    while ((1ull << S) <= bound * d) ++S;
    DivWide r;
    r.S = S;
    r.M = (uint32_t)(((1ull << S) + d - 1) / d);
    return r;
  }
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return (uint32_t)(((uint64_t)n * M) >> S); }
};

// ---------------------------------------------------------------------------------------------
// A per-device side stream for independent work inside one library call.  Fork: the side stream waits for everything queued on
// the caller's stream; join: the caller's stream waits for the side stream.  Both are legal inside a stream capture (the side
// stream joins the capture and leaves it at the join).
// ---------------------------------------------------------------------------------------------
int side_priority_mode();      // encode.hip (development switch MURAL_SIDE_PRIORITY)
struct SideStream {
  hipStream_t side = nullptr, side2 = nullptr;
  hipEvent_t fork_ev = nullptr, join_ev = nullptr, join2_ev = nullptr;
  // One set of streams / events per device: a caller holds `mu` from its fork to its join (SideStreamHold), so two host threads that
  // drive the same device cannot cross-wire each other's fork / join events.  (The enqueue between fork and join is host work of a
  // few hundred microseconds; the device side stays concurrent.)
  std::mutex mu;
  bool ready = false;
  int init() {          // called with `mu` held
    if (ready) return MURAL_OK;
    int prio_lo = 0, prio_hi = 0;      // (numerically: hi <= lo; hi = served first)
    MURAL_HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    const int pm = side_priority_mode();
    MURAL_HIP_CHECK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, (pm & 1) ? prio_hi : (pm & 4) ? prio_lo : 0));
    MURAL_HIP_CHECK(hipStreamCreateWithPriority(&side2, hipStreamNonBlocking, (pm & 2) ? prio_hi : (pm & 8) ? prio_lo : 0));
    MURAL_HIP_CHECK(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
    MURAL_HIP_CHECK(hipEventCreateWithFlags(&join_ev, hipEventDisableTiming));
    MURAL_HIP_CHECK(hipEventCreateWithFlags(&join2_ev, hipEventDisableTiming));
    ready = true;
    return MURAL_OK;
  }
  int fork(hipStream_t main, bool both = false) {
    MURAL_HIP_CHECK(hipEventRecord(fork_ev, main));
    MURAL_HIP_CHECK(hipStreamWaitEvent(side, fork_ev, 0));
    if (both) MURAL_HIP_CHECK(hipStreamWaitEvent(side2, fork_ev, 0));
    return MURAL_OK;
  }
  int fork2(hipStream_t main) {      // the second side stream alone (an earlier point of the caller's stream than fork())
    MURAL_HIP_CHECK(hipEventRecord(fork_ev, main));
    MURAL_HIP_CHECK(hipStreamWaitEvent(side2, fork_ev, 0));
    return MURAL_OK;
  }
  int join(hipStream_t main, bool both = false) {
    MURAL_HIP_CHECK(hipEventRecord(join_ev, side));
    MURAL_HIP_CHECK(hipStreamWaitEvent(main, join_ev, 0));
    if (both) {
      MURAL_HIP_CHECK(hipEventRecord(join2_ev, side2));
      MURAL_HIP_CHECK(hipStreamWaitEvent(main, join2_ev, 0));
    }
    return MURAL_OK;
  }
};

SideStream* side_stream_slot(int dev);      // encode.hip: ONE table for the whole library (not one per translation unit)

// Exclusive use of the current device's side streams for the lifetime of the object (fork .. join of one library call).
struct SideStreamHold {
  SideStream* ss = nullptr;
  std::unique_lock<std::mutex> lock;
  int acquire() {
    int dev = 0;
    MURAL_HIP_CHECK(hipGetDevice(&dev));
    MURAL_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    SideStream* s = side_stream_slot(dev);
    lock = std::unique_lock<std::mutex>(s->mu);
    if (int rc = s->init()) return rc;
    ss = s;
    return MURAL_OK;
  }
  SideStream* operator->() const { return ss; }
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

// first index of the ascending side table with amb_pos[i] >= key (plain binary search, one thread)
__device__ __forceinline__ int64_t amb_lower_bound(const int64_t* __restrict__ amb_pos, int64_t n, int64_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (amb_pos[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// the same bound found by a whole wave: 64 probes per round, wave-uniform result
__device__ __forceinline__ int64_t amb_lower_bound_wave(const int64_t* __restrict__ amb_pos, int64_t n, int64_t key, int lane) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t step = (hi - lo + 63) >> 6;
    const int64_t idx = lo + (int64_t)lane * step;
    const bool less = idx < hi && amb_pos[idx] < key;
    const int cnt = __popcll(__ballot(less));          // the table ascends: `less` lanes form a prefix
    if (cnt == 0) {
      hi = lo;
    } else {
      const int64_t nhi = lo + (int64_t)cnt * step;
      lo = lo + (int64_t)(cnt - 1) * step + 1;
      hi = nhi < hi ? nhi : hi;
    }
  }
  return lo;
}

// symbol of genome position g with IUPAC codes resolved through the side table (strand-agnostic, forward symbol)
__device__ __forceinline__ uint32_t genome_sym_iupac(const MuralGenome& gn, int64_t g) {
  if (g < 0 || g >= gn.length) return SYM_N;
  const uint32_t w = gn.packed2[g >> 4];
  const uint32_t m = gn.nmask[g >> 5];
  if (((m >> (uint32_t)(g & 31)) & 1u) == 0u) return (w >> (2u * (uint32_t)(g & 15))) & 3u;
  if (gn.n_amb > 0) {
    const int64_t i = amb_lower_bound(gn.amb_pos, gn.n_amb, g);
    if (i < gn.n_amb && gn.amb_pos[i] == g) return gn.amb_sym[i];
  }
  return SYM_N;
}

// complement within the 16-symbol alphabet (A<->T, C<->G, N, R<->Y, M<->K, S, W, B<->V, D<->H, PAD)
__device__ __forceinline__ uint32_t sym_complement(uint32_t s) {
  // table packed 4 bits per symbol: index 0..15 -> 3,2,1,0,4,6,5,10,8,9,7,14,13,12,11,15
  const uint64_t tbl = 0xFBCDE798A5640123ull;
  return (uint32_t)((tbl >> (4u * s)) & 0xFull);
}

}  // namespace mural

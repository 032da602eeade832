// The prediction table of the reference, written at kernel speed.
//
// MuRaL/scripts/run_predict.py:230-238 ends every prediction run with
//     pred_df.sort_values(['chrom', 'start']); pred_df.to_csv(pred_file, sep='\t', float_format='%.4g', index=False)
// i.e. one text row per site: chrom, start, end, strand, mut_type, prob0 .. prob{k-1}; integers as decimals, probabilities through
// Python's '%.4g' (NaN -> empty field, pandas' na_rep).  pandas formats ~0.2 M rows/s; one MI355X predicts 13-30 M rows/s.  This file
// holds ONE row formatter, compiled for both sides:
//   * mural_tsv_format_device: a kernel formats the rows of a shard (in a caller-given row order) into a text buffer in HBM -- one
//     thread per row into an LDS slot, workgroup scan of the row lengths, compaction in LDS, coalesced copy-out; the host only
//     copies the bytes and write()s them;
//   * mural_tsv_format_host: the same formatter on host threads, for shards that live in host memory (gloo ranks, CPU tests).
// '%.4g' must be byte-identical to CPython / glibc, which round the EXACT binary value to 4 significant digits, ties to even.  The
// formatter scales the value by a power of ten in double precision and decides the rounding exactly: for |p| <= 22 the power is an
// exact double and the sign of fma(a, 10^p, -(N + 0.5)) is the sign of the exact difference; outside that range the scaled value
// decides unless it lies within 1e-6 of the half-way point, in which case an exact big-integer comparison of m * 2^q * 10^p with
// N + 0.5 does (a few dozen 32-bit multiplications, taken by about one value in a million of that range).
// Also here: the reference's per-(segment, strand) focal-base check (MuRaL/data/preprocessing.py:479-484) as a streaming kernel over a
// gathered shard.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

namespace {

#define HD __host__ __device__ __forceinline__

// 10^0 .. 10^308 as correctly rounded doubles (10^0 .. 10^22 are exact)
#define P10_ROW(a) 1e##a##0, 1e##a##1, 1e##a##2, 1e##a##3, 1e##a##4, 1e##a##5, 1e##a##6, 1e##a##7, 1e##a##8, 1e##a##9
#define P10_TABLE                                                                                                                  \
  {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, P10_ROW(1), P10_ROW(2), P10_ROW(3), P10_ROW(4), P10_ROW(5), P10_ROW(6),        \
   P10_ROW(7), P10_ROW(8), P10_ROW(9), P10_ROW(10), P10_ROW(11), P10_ROW(12), P10_ROW(13), P10_ROW(14), P10_ROW(15), P10_ROW(16),   \
   P10_ROW(17), P10_ROW(18), P10_ROW(19), P10_ROW(20), P10_ROW(21), P10_ROW(22), P10_ROW(23), P10_ROW(24), P10_ROW(25), P10_ROW(26), \
   P10_ROW(27), P10_ROW(28), P10_ROW(29), 1e300, 1e301, 1e302, 1e303, 1e304, 1e305, 1e306, 1e307, 1e308}
[[maybe_unused]] __device__ const double kPow10Dev[309] = P10_TABLE;
[[maybe_unused]] const double kPow10Host[309] = P10_TABLE;

HD double pow10_tab(int k) {
#ifdef __HIP_DEVICE_COMPILE__
  return kPow10Dev[k];
#else
  return kPow10Host[k];
#endif
}

HD uint64_t f64_bits(double v) {
  uint64_t b;
  memcpy(&b, &v, 8);
  return b;
}

HD int clz64(uint64_t x) {
#ifdef __HIP_DEVICE_COMPILE__
  return __clzll((long long)x);
#else
  return __builtin_clzll(x);
#endif
}

// ---- exact comparison for the far exponent ranges ------------------------------------------------------------------------------
struct Big {
  uint32_t w[32];
  int n;
};

HD void big_set(Big& b, uint64_t v) {
  b.w[0] = (uint32_t)v;
  b.w[1] = (uint32_t)(v >> 32);
  b.n = b.w[1] ? 2 : 1;
}

HD void big_mul_small(Big& b, uint32_t f) {
  uint64_t carry = 0;
  for (int i = 0; i < b.n; ++i) {
    const uint64_t t = (uint64_t)b.w[i] * f + carry;
    b.w[i] = (uint32_t)t;
    carry = t >> 32;
  }
  if (carry && b.n < 32) b.w[b.n++] = (uint32_t)carry;
}

HD void big_mul_pow5(Big& b, int e) {
  while (e >= 13) {
    big_mul_small(b, 1220703125u);   // 5^13
    e -= 13;
  }
  uint32_t f = 1;
  for (int i = 0; i < e; ++i) f *= 5u;
  if (f > 1) big_mul_small(b, f);
}

HD int big_bitlen(const Big& b) { return 32 * (b.n - 1) + (32 - (clz64((uint64_t)b.w[b.n - 1]) - 32)); }

// bit `i` and up (64 of them) of b
HD uint64_t big_bits_from(const Big& b, int i) {
  uint64_t out = 0;
  for (int k = 0; k < 3; ++k) {
    const int wi = (i >> 5) + k;
    if (wi >= b.n) break;
    const uint64_t w = b.w[wi];
    const int sh = 32 * k - (i & 31);
    if (sh >= 64) break;
    out |= sh >= 0 ? (w << sh) : (w >> (-sh));
  }
  return out;
}

HD bool big_low_bits_nonzero(const Big& b, int nbits) {
  for (int i = 0; i < b.n && 32 * i < nbits; ++i) {
    const int rem = nbits - 32 * i;
    const uint32_t mask = rem >= 32 ? 0xffffffffu : ((1u << rem) - 1u);
    if (b.w[i] & mask) return true;
  }
  return false;
}

// sign of (b * 2^sh - small), small > 0
HD int big_cmp_shifted(const Big& b, int sh, uint64_t small) {
  const int lb = big_bitlen(b) + sh, ls = 64 - clz64(small);
  if (lb != ls) return lb > ls ? 1 : -1;
  if (sh >= 0) {                        // the whole product fits into 64 bits
    const uint64_t v = big_bits_from(b, 0) << sh;
    return v > small ? 1 : (v < small ? -1 : 0);
  }
  const uint64_t v = big_bits_from(b, -sh);
  if (v != small) return v > small ? 1 : -1;
  return big_low_bits_nonzero(b, -sh) ? 1 : 0;
}

// sign of (m * 2^q * 10^p - c / 2) for an odd c = 2N + 1
HD int exact_cmp_half(uint64_t m, int q, int p, uint32_t c) {
  Big b;
  if (p >= 0) {                         // (m * 5^p) * 2^(q + p + 1)  vs  c
    big_set(b, m);
    big_mul_pow5(b, p);
    return big_cmp_shifted(b, q + p + 1, c);
  }
  big_set(b, c);                        // m * 2^(q + 1 + p)  vs  c * 5^-p   (p < 0)
  big_mul_pow5(b, -p);
  return -big_cmp_shifted(b, -(q + 1 + p), m);
}

// |v| * 10^p in double precision (one rounding for |p| <= 22, a few ulp beyond)
HD double scale10(double a, int p) {
  if (p >= 0) {
    if (p <= 300) return a * pow10_tab(p);
    return (a * pow10_tab(300)) * pow10_tab(p - 300);
  }
  return a / pow10_tab(-p);
}

// '%.4g' % v, NaN -> nothing (pandas' na_rep='').  Returns the number of characters written (at most 10).
HD int fmt_g4(double v, char* o) {
  const uint64_t bits = f64_bits(v);
  const uint64_t ab = bits & 0x7fffffffffffffffull;
  if (ab > 0x7ff0000000000000ull) return 0;
  int n = 0;
  if (bits >> 63) o[n++] = '-';
  if (ab == 0x7ff0000000000000ull) {
    o[n++] = 'i';
    o[n++] = 'n';
    o[n++] = 'f';
    return n;
  }
  if (ab == 0) {
    o[n++] = '0';
    return n;
  }
  const int ef = (int)(ab >> 52);
  const uint64_t mant = ab & 0xfffffffffffffull;
  const uint64_t m = ef ? (mant | (1ull << 52)) : mant;
  const int q = ef ? ef - 1075 : -1074;
  const double a = fabs(v);
  const int e2 = 63 - clz64(m) + q;                  // floor(log2 a)
  int e10 = (e2 * 78913) >> 18;                      // floor(e2 * log10 2): floor(log10 a) or one below
  double s = scale10(a, 3 - e10);
  if (s < 1000.0) {
    --e10;
    s = scale10(a, 3 - e10);
  } else if (s >= 10000.0) {
    ++e10;
    s = scale10(a, 3 - e10);
  }
  const int p = 3 - e10;
  int N = (int)s;                                    // 999 .. 10000: floor of the scaled value, or a neighbour of it
  const double h = (double)N + 0.5;
  int sign;                                          // sign of (a * 10^p - h), exactly
  if (p >= 0 && p <= 22) {
    const double d = fma(a, pow10_tab(p), -h);
    sign = d > 0.0 ? 1 : (d < 0.0 ? -1 : 0);
  } else if (p < 0 && p >= -22) {
    const double d = fma(-h, pow10_tab(-p), a);
    sign = d > 0.0 ? 1 : (d < 0.0 ? -1 : 0);
  } else if (fabs(s - h) > 1e-6) {
    sign = s > h ? 1 : -1;
  } else {
    sign = exact_cmp_half(m, q, p, (uint32_t)(2 * N + 1));
  }
  if (sign > 0 || (sign == 0 && (N & 1))) ++N;
  if (N >= 10000) {
    N = 1000;
    ++e10;
  }
  int d[4] = {N / 1000, (N / 100) % 10, (N / 10) % 10, N % 10};
  int nd = 4;
  while (nd > 1 && d[nd - 1] == 0) --nd;
  if (e10 < -4 || e10 >= 4) {
    o[n++] = (char)('0' + d[0]);
    if (nd > 1) {
      o[n++] = '.';
      for (int i = 1; i < nd; ++i) o[n++] = (char)('0' + d[i]);
    }
    o[n++] = 'e';
    int x = e10;
    if (x < 0) {
      o[n++] = '-';
      x = -x;
    } else {
      o[n++] = '+';
    }
    if (x >= 100) o[n++] = (char)('0' + x / 100);
    o[n++] = (char)('0' + (x / 10) % 10);
    o[n++] = (char)('0' + x % 10);
  } else if (e10 >= 0) {
    for (int i = 0; i <= e10; ++i) o[n++] = (char)('0' + d[i]);
    if (nd > e10 + 1) {
      o[n++] = '.';
      for (int i = e10 + 1; i < nd; ++i) o[n++] = (char)('0' + d[i]);
    }
  } else {
    o[n++] = '0';
    o[n++] = '.';
    for (int i = 0; i < -e10 - 1; ++i) o[n++] = '0';
    for (int i = 0; i < nd; ++i) o[n++] = (char)('0' + d[i]);
  }
  return n;
}

HD int fmt_i64(int64_t v, char* o) {
  int n = 0;
  uint64_t u = (uint64_t)v;
  if (v < 0) {
    o[n++] = '-';
    u = 0ull - u;
  }
  char tmp[20];
  int k = 0;
  do {
    tmp[k++] = (char)('0' + (int)(u % 10u));
    u /= 10u;
  } while (u);
  while (k) o[n++] = tmp[--k];
  return n;
}

// numpy's float -> int64 cast (truncation; NaN and out-of-range values give INT64_MIN like the x86 conversion numpy uses)
// (tested on the exponent bits: the library is built with -fno-honor-nans, under which a comparison-based NaN guard may be folded
// away and the cast of a NaN is undefined -- INT64_MIN on the host, 0 on the device)
HD int64_t label_to_i64(float f) {
  uint32_t b;
  memcpy(&b, &f, 4);
  if (((b >> 23) & 0xFFu) >= 127u + 63u) return INT64_MIN;      // NaN, infinities and |f| >= 2^63
  return (int64_t)f;
}

// one table row (input row r) into o; returns its length
HD int format_row(const MuralTsvRows& t, const char* names, int64_t r, char* o) {
  int n = 0;
  const char* nm = names + (int64_t)(t.chrom_id ? t.chrom_id[r] : 0) * t.name_stride;
  for (int i = 0; i < t.name_stride && nm[i]; ++i) o[n++] = nm[i];
  o[n++] = '\t';
  n += fmt_i64(t.start[r], o + n);
  o[n++] = '\t';
  n += fmt_i64(t.end[r], o + n);
  o[n++] = '\t';
  if (t.layout == 1) {                 // BED6: chrom start end name score strand (name = '.')
    o[n++] = '.';
    o[n++] = '\t';
    n += fmt_i64(label_to_i64(t.label[r]), o + n);
    o[n++] = '\t';
    o[n++] = t.strand[r] ? '-' : '+';
    o[n++] = '\n';
    return n;
  }
  o[n++] = t.strand[r] ? '-' : '+';
  o[n++] = '\t';
  n += fmt_i64(label_to_i64(t.label[r]), o + n);
  for (int c = 0; c < t.n_class; ++c) {
    o[n++] = '\t';
    const double v = t.prob_f64 ? ((const double*)t.prob)[r * t.prob_stride + c] : (double)((const float*)t.prob)[r * t.prob_stride + c];
    n += fmt_g4(v, o + n);
  }
  o[n++] = '\n';
  return n;
}

int row_bound(const MuralTsvRows& t, const char* names_host) {
  int longest = 0;
  for (int c = 0; c < t.n_chroms; ++c)
    longest = std::max(longest, (int)strnlen(names_host + (int64_t)c * t.name_stride, (size_t)t.name_stride));
  return longest + 1 + 20 + 1 + 20 + 1 + 1 + 1 + 20 + t.n_class * 11 + 1 + 2;
}

// ---- device side --------------------------------------------------------------------------------------------------------------
// Workgroup = ROWS rows.  Thread r formats its row into LDS slot r (stride `bound`), the row lengths are scanned, and
//   WRITE = false: thread 0 stores the workgroup's byte count (first pass: sizes only),
//   WRITE = true : the rows are compacted inside LDS and streamed to out + block_off[blockIdx] as one contiguous run.
template <bool WRITE>
__global__ void __launch_bounds__(256) tsv_format_kernel(MuralTsvRows t, const char* __restrict__ names, int bound, int rows_per_block,
                                                         int64_t* __restrict__ block_bytes, char* __restrict__ out) {
  extern __shared__ char lds[];
  char* slots = lds;
  char* compact = slots + (size_t)rows_per_block * bound;
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t i = row0 + tid;
  int my = 0;
  if (tid < rows_per_block && i < t.n) my = format_row(t, names, t.perm ? t.perm[i] : i, slots + (size_t)tid * bound);
  // exclusive scan of `my` over the workgroup: wave scan by shuffles, wave totals through LDS
  int incl = my;
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if ((tid & 63) >= d) incl += up;
  }
  __shared__ int wave_tot[4];
  if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < (tid >> 6); ++w) base += wave_tot[w];
  const int off = base + incl - my;
  const int total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  if (!WRITE) {
    if (tid == 0) block_bytes[blockIdx.x] = total;
    return;
  }
  const char* src = slots + (size_t)tid * bound;
  for (int k = 0; k < my; ++k) compact[off + k] = src[k];
  __syncthreads();
  char* dst = out + block_bytes[blockIdx.x];
  for (int k = tid; k < total; k += 256) dst[k] = compact[k];
}

// exclusive scan of the per-workgroup byte counts (in place) by ONE workgroup; total -> *n_bytes
__global__ void __launch_bounds__(1024) tsv_scan_kernel(int64_t* __restrict__ block_bytes, int64_t nb, int64_t* __restrict__ n_bytes) {
  __shared__ int64_t part[1024];
  const int tid = threadIdx.x;
  const int64_t per = (nb + 1023) / 1024;
  const int64_t lo = tid * per < nb ? tid * per : nb, hi = lo + per < nb ? lo + per : nb;
  int64_t sum = 0;
  for (int64_t k = lo; k < hi; ++k) sum += block_bytes[k];
  part[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    int64_t run = 0;
    for (int k = 0; k < 1024; ++k) {
      const int64_t v = part[k];
      part[k] = run;
      run += v;
    }
    *n_bytes = run;
  }
  __syncthreads();
  int64_t run = part[tid];
  for (int64_t k = lo; k < hi; ++k) {
    const int64_t v = block_bytes[k];
    block_bytes[k] = run;
    run += v;
  }
}

// rows i-1 and i of the same (segment, strand) group must carry the same strand-complemented focal base
__global__ void focal_group_check_kernel(const float* __restrict__ rows, int64_t row_stride, int64_t col, const int64_t* __restrict__ group,
                                         int64_t n, int32_t* __restrict__ status) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x + 1;
  if (i >= n) return;
  if (group[i] == group[i - 1] && rows[i * row_stride + col] != rows[(i - 1) * row_stride + col]) atomicOr(status, 1);
}
__global__ void focal_group_check_f64_kernel(const double* __restrict__ rows, int64_t row_stride, int64_t col,
                                             const int64_t* __restrict__ group, int64_t n, int32_t* __restrict__ status) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x + 1;
  if (i >= n) return;
  if (group[i] == group[i - 1] && rows[i * row_stride + col] != rows[(i - 1) * row_stride + col]) atomicOr(status, 1);
}

mural::DynLdsOnce g_tsv_lds;

int validate(const MuralTsvRows* t) {
  MURAL_REQUIRE(t, "NULL argument");
  MURAL_REQUIRE(t->n >= 0 && t->n_class >= 0 && t->n_class <= 64, "bad table shape");
  MURAL_REQUIRE(t->n_chroms >= 1 && t->name_stride >= 2 && t->name_stride <= 1024 && t->chrom_names, "bad chromosome name table");
  MURAL_REQUIRE(t->prob_stride >= t->n_class, "prob_stride smaller than n_class");
  MURAL_REQUIRE(t->layout == 0 || t->layout == 1, "layout must be 0 (prediction table) or 1 (BED6)");
  MURAL_REQUIRE(t->n == 0 || (t->start && t->end && t->strand && t->label && (t->prob || t->n_class == 0 || t->layout == 1)), "NULL column");
  return MURAL_OK;
}

}  // namespace

using namespace mural;

extern "C" int64_t mural_tsv_row_bound(const MuralTsvRows* t) {
  if (validate(t)) return -1;
  return row_bound(*t, t->chrom_names);
}

// The chromosome-name table of a device format call arrives as host memory of the caller (a Python ctypes buffer that may be freed
// right after the call returns) and is needed on the device: it is staged through pinned slots owned by the library, so the copy is
// a true asynchronous copy (from pageable memory hipMemcpyAsync either blocks until the stream has drained or reads the buffer after
// the caller has released it).  A slot is reused after the copy that last used it has completed.
namespace {
struct NameStage {
  static constexpr int SLOTS = 8;
  static constexpr size_t SLOT_BYTES = 16 * 1024;
  std::mutex mu;
  char* base = nullptr;
  hipEvent_t ev[SLOTS] = {};
  bool used[SLOTS] = {};
  int next = 0;
  // copies `bytes` of `src` into a slot and returns it (nullptr: table too large for a slot -- the caller copies synchronously)
  int stage(const char* src, size_t bytes, hipStream_t stream, char* dst_dev) {
    std::lock_guard<std::mutex> lock(mu);
    if (!base) {
      MURAL_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&base), SLOTS * SLOT_BYTES, hipHostMallocDefault));
      for (int i = 0; i < SLOTS; ++i) MURAL_HIP_CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    }
    const int i = next;
    next = (next + 1) % SLOTS;
    if (used[i]) MURAL_HIP_CHECK(hipEventSynchronize(ev[i]));
    std::memcpy(base + (size_t)i * SLOT_BYTES, src, bytes);
    MURAL_HIP_CHECK(hipMemcpyAsync(dst_dev, base + (size_t)i * SLOT_BYTES, bytes, hipMemcpyHostToDevice, stream));
    MURAL_HIP_CHECK(hipEventRecord(ev[i], stream));
    used[i] = true;
    return MURAL_OK;
  }
};
NameStage g_name_stage;
}  // namespace

extern "C" size_t mural_tsv_format_workspace_bytes(int64_t n) {
  const int64_t blocks = (n + 31) / 32;      // the smallest workgroup tile is 32 rows
  return (size_t)(blocks + 1) * 8 + 4096;    // + room for the name table copy is added per call below
}

extern "C" int mural_tsv_format_device(const MuralTsvRows* t, char* out, int64_t cap, int64_t* n_bytes, void* ws, size_t ws_bytes,
                                       void* stream_) {
  if (int rc = validate(t)) return rc;
  MURAL_REQUIRE(out && n_bytes && ws, "NULL argument");
  hipStream_t stream = (hipStream_t)stream_;
  const int bound = row_bound(*t, t->chrom_names);
  MURAL_REQUIRE(cap >= t->n * (int64_t)bound, "text buffer too small: %lld rows need up to %lld bytes", (long long)t->n,
                (long long)(t->n * (int64_t)bound));
  int rows = 256;
  while (rows > 32 && (size_t)2 * rows * bound > 64 * 1024) rows >>= 1;
  const size_t lds = (size_t)2 * rows * bound;
  MURAL_REQUIRE(lds <= 160 * 1024, "rows of up to %d bytes do not fit the formatter's LDS tile", bound);
  const int64_t nb = (t->n + rows - 1) / rows;
  const size_t names_bytes = (size_t)t->n_chroms * t->name_stride;
  const size_t need = (size_t)(nb + 1) * 8 + ((names_bytes + 15) & ~(size_t)15);
  if (ws_bytes < need) {
    set_error("tsv workspace too small: %zu < %zu", ws_bytes, need);
    return MURAL_E_WORKSPACE;
  }
  if (t->n == 0) {
    MURAL_HIP_CHECK(hipMemsetAsync(n_bytes, 0, 8, stream));
    return MURAL_OK;
  }
  if (lds > 64 * 1024)
    if (int rc = g_tsv_lds.ensure(tsv_format_kernel<false>, tsv_format_kernel<true>)) return rc;
  int64_t* block_bytes = static_cast<int64_t*>(ws);
  char* names_dev = static_cast<char*>(ws) + (size_t)(nb + 1) * 8;
  if (names_bytes <= NameStage::SLOT_BYTES) {
    if (int rc = g_name_stage.stage(t->chrom_names, names_bytes, stream, names_dev)) return rc;
  } else {      // (a table of more than 64 names of this stride: a synchronous copy, complete when it returns)
    MURAL_HIP_CHECK(hipStreamSynchronize(stream));
    MURAL_HIP_CHECK(hipMemcpy(names_dev, t->chrom_names, names_bytes, hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(tsv_format_kernel<false>, dim3((unsigned)nb), dim3(256), lds, stream, *t, names_dev, bound, rows, block_bytes, out);
  hipLaunchKernelGGL(tsv_scan_kernel, dim3(1), dim3(1024), 0, stream, block_bytes, nb, n_bytes);
  hipLaunchKernelGGL(tsv_format_kernel<true>, dim3((unsigned)nb), dim3(256), lds, stream, *t, names_dev, bound, rows, block_bytes, out);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

extern "C" int mural_tsv_format_host(const MuralTsvRows* t, char* out, int64_t cap, int64_t* n_bytes, int32_t threads) {
  if (int rc = validate(t)) return rc;
  MURAL_REQUIRE(out && n_bytes, "NULL argument");
  const int bound = row_bound(*t, t->chrom_names);
  const int64_t n = t->n;
  int T = threads > 0 ? threads : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
  T = (int)std::max<int64_t>(1, std::min<int64_t>(T, (n + 65535) / 65536));
  std::vector<std::vector<char>> parts((size_t)T);
  auto work = [&](int k) {
    const int64_t lo = n * k / T, hi = n * (k + 1) / T;
    std::vector<char>& buf = parts[(size_t)k];
    buf.resize((size_t)((hi - lo) * bound + 1));
    size_t w = 0;
    for (int64_t i = lo; i < hi; ++i) w += (size_t)format_row(*t, t->chrom_names, t->perm ? t->perm[i] : i, buf.data() + w);
    buf.resize(w);
  };
  if (T == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int k = 0; k < T; ++k) th.emplace_back(work, k);
    for (auto& x : th) x.join();
  }
  int64_t total = 0;
  for (auto& b : parts) total += (int64_t)b.size();
  MURAL_REQUIRE(total <= cap, "text buffer too small: %lld < %lld", (long long)cap, (long long)total);
  int64_t w = 0;
  for (auto& b : parts) {
    std::memcpy(out + w, b.data(), b.size());
    w += (int64_t)b.size();
  }
  *n_bytes = total;
  return MURAL_OK;
}

// '%.4g' of one value (tests; also what a host-side caller would use for a single number)
extern "C" int mural_tsv_format_g4(double v, char* out12) {
  if (!out12) return -1;
  const int n = fmt_g4(v, out12);
  out12[n] = '\0';
  return n;
}

extern "C" int mural_focal_group_check(const void* rows, int32_t rows_f64, int64_t row_stride, int64_t col, const int64_t* group, int64_t n,
                                       int32_t* status, void* stream) {
  MURAL_REQUIRE(n <= 1 || (rows && group && status), "NULL argument");
  if (n <= 1) return MURAL_OK;
  const int64_t blocks = (n - 1 + 255) / 256;
  if (rows_f64)
    hipLaunchKernelGGL(focal_group_check_f64_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const double*)rows, row_stride,
                       col, group, n, status);
  else
    hipLaunchKernelGGL(focal_group_check_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)rows, row_stride, col,
                       group, n, status);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

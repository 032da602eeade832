// Window encoders on the GPU: k-mer index, one-hot, and dense-tensor -> symbol classification.
// Reference semantics: MuRaL/data/preprocessing.py:636-723 (k-mer), :756-816 (one-hot), :559-567 (windows).
#include <cstring>
#include <cstdlib>
#include "common.h"
#include "dense_symbol.h"

namespace mural {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

const char* last_error_cstr() { return g_err.c_str(); }

bool g_dev_switches = false;

const DevSwitch* dev_switch_table() {
  static const DevSwitch table[] = {
    {"MURAL_TOWER_DYNAMIC_UNITS", "A/B, bit-identical: SNV predict units through a global ticket counter instead of the fixed stride"},
    {"MURAL_SNV_DEFER_SHORT", "A/B, bit-identical: =0 runs the short-stage launches per chunk (what a one-chunk workspace does by itself)"},
    {"MURAL_LOCAL_REG", "A/B, bit-identical: =0 keeps the local MLP's weight fragments in LDS (snv_local_mlp_mfma)"},
    {"MURAL_DEBUG_LW_SEPARATE", "A/B, same results: the long-window first-stage jobs (two segment kinds, mid tower) as three launches instead of one"},
    {"MURAL_DEBUG_NO_CALL_P", "A/B, same results: long-window short-stage launches keep the model's tile size for small calls too"},
    {"MURAL_INDEL_CHUNK", "A/B, same results: positions per pass of the INDEL layer program (256 .. 4096; default 2048)"},
    {"MURAL_INDEL_LANES", "A/B, same results: INDEL forward chunks in flight (1..3 streams; default 2)"},
    {"MURAL_SIDE_PRIORITY", "A/B, same results: bit mask of the side streams created with the highest (1, 2) / lowest (4, 8) stream priority; default 0"},
    {"MURAL_CW_FULL_GRID", "A/B, same sums in another order: training conv launches ask for two workgroup slots per CU whatever the row length"},
    {"MURAL_TRAIN_CONV_CL", "A/B: the training step on the workgroup-tile conv kernels (conv32_cl.hip) instead of the wave-private ones"},
    {"MURAL_TRAIN_NO_FOLD", "A/B: BatchNorm-backward applies of a ResBlock stage as passes of their own"},
    {"MURAL_TRAIN_NO_FIRST_FOLD", "A/B: the first stage's last BatchNorm-backward apply as a pass of its own"},
    {"MURAL_TRAIN_NO_MID_FOLD", "A/B: the short stage's last apply as a pass of its own"},
    {"MURAL_TRAIN_NO_POOL_FOLD", "A/B: the applies in front of the pools as passes of their own"},
    {"MURAL_TRAIN_LOCAL_OPS", "A/B: the local branch of the training step as per-op launches"},
    {"MURAL_TRAIN_HEAD_OPS", "A/B: a tower's head of the training step as per-op launches"},
    {"MURAL_TRAIN_FIRST_SEPARATE", "A/B: symbol histograms / first-layer tables per tower"},
    {"MURAL_TRAIN_ORDER", "experiment: forward 2 = the local branch forked with the towers instead of in front of their prepare chain; backward 1 = the local branch enqueued first"},
    {"MURAL_XCD_SWIZZLE", "A/B, bit-identical: =0 disables the XCD-aware workgroup index of the barrier-free INDEL kernels"},
    {"MURAL_CONV1D_MFMA", "A/B: =0 keeps the generic conv on the vector ALU, =2 routes every conv to the MFMA kernel"},
    {"MURAL_CONV1D_DIRECT", "A/B: =0 disables the barrier-free long-row conv"},
    {"MURAL_CONVBLOCK_DIRECT", "A/B: 0 / 2 never / always the barrier-free ConvBlock"},
    {"MURAL_CONVBLOCK8_VALU", "A/B: the 8-channel ConvBlock entirely on the vector ALU"},
    {"MURAL_INDEL_ENC0", "A/B: =0 disables the persistent level-0 encoder launch"},
    {"MURAL_INDEL_ENC0_DOWN", "A/B: =0 keeps the stride-4 down-conv a launch of its own"},
    {"MURAL_INDEL_DEC0", "A/B: =0 disables the persistent level-0 decoder launch"},
    {"MURAL_INDEL_DEEP", "A/B: =0 runs the deep levels' blocks as two launches each"},
    {"MURAL_INDEL_DEEP_FRONT", "A/B: =0 keeps the fourth level's strided conv a launch of its own"},
    {"MURAL_INDEL_DENSE_SYMBOLS", "A/B: =0 keeps the dense INDEL entry on the three-launch route"},
    {"MURAL_INDEL_ENC0_WGS", "experiment: workgroups per CU of the level-0 encoder launch"},
    {"MURAL_INDEL_DEC0_WGS", "experiment: workgroups per CU of the level-0 decoder launch"},
    {"MURAL_DIRECT_GRID_CAP", "experiment: =0 lifts the grid cap of the barrier-free convs"},
    {"MURAL_WGRAD_MFMA", "A/B: =0 keeps the INDEL weight gradients on the LDS-tiled vector kernel"},
    {"MURAL_SNV_CHUNK", "experiment: sites per launch sequence of the SNV predict path"},
    {"MURAL_DEBUG_WS_GUARD", "validation: unused bytes behind every workspace region (tests poison and check them)"},
    {"MURAL_DEBUG_DROP_PART_ROW", "validation (fault injection): drop one partial row of a conv weight gradient"},
    {"MURAL_DEBUG_NO_SMALL_BATCH", "validation: calls of <= 256 sites through the throughput launches"},
    {"MURAL_DEBUG_NO_LONGWIN", "validation: long windows through the per-layer path"},
    {"MURAL_DEBUG_NO_TOWER_SPLIT", "validation: the towers as one launch per chunk"},
    {"MURAL_DEBUG_NO_PAIR_TABLE", "validation: first layer without the pair table"},
    {"MURAL_DEBUG_NO_LOCAL_FUSE", "validation: the local MLP never rides in the first-stage launch"},
    {"MURAL_DEBUG_LOCAL_VALU", "validation: the local MLP on the vector ALU"},
    {"MURAL_DEBUG_CONVBLOCK_VALU", "validation: every ConvBlock on the vector-ALU kernel"},
    {"MURAL_DEBUG_INDEL_NO_GENOME_FRONT", "validation: the packed INDEL entry through an explicit one-hot tensor"},
    {"MURAL_DEBUG_TOWER_RUNTIME_GEOM", "validation, bit-identical: the first-stage tower instance that reads its geometry from the arguments"},
    {"MURAL_DEBUG_TOWER_STATIC_UNITS", "validation: overrides MURAL_TOWER_DYNAMIC_UNITS"},
    {"MURAL_DEBUG_EDGE_TILE", "validation: the reuse path's edge columns on the workgroup-tile kernel"},
    {"MURAL_DEBUG_POLY_NARROW", "validation: the polyphase up-conv's dword stores"},
    {"MURAL_DEBUG_FIRST_SCATTER", "validation: the first layer's backward through LDS atomics"},
    {"MURAL_DEBUG_CONV32_R", "validation: forces the row tile of the tiled conv32 kernels"},
    {"MURAL_DEBUG_S1_ALIAS", "TIMING ONLY, wrong results: every site writes the x0 rows of site (row mod 64)"},
    {"MURAL_DEBUG_FIRST", "TIMING ONLY, wrong results: phases of the training first-layer kernels switched off"},
    {"MURAL_DEBUG_FIRST_CL", "TIMING ONLY: changes the layout of mural_op_first_fwd / _bwd"},
    {"MURAL_DEBUG_CW", "TIMING ONLY, wrong results: phases of the wave-private conv kernels switched off (bit mask)"},
    {"MURAL_DEBUG_CL", "TIMING ONLY, wrong results: phases of the workgroup-tile conv kernels switched off"},
    {"MURAL_DEBUG_BWD32", "TIMING ONLY, wrong results: phases of the tiled conv32 backward switched off"},
    {"MURAL_DEBUG_MLP", "TIMING ONLY, wrong results: phases of the local MLP switched off"},
    {"MURAL_DEBUG_CB_STAMP_ONLY", "diagnostic: phase stamps of one workgroup only"},
    {"MURAL_DEBUG_SPLIT_P", "diagnostic: caps the tile sizes of the split tower launches"},
    {"MURAL_DEBUG_TOWER_WAVE", "diagnostic: mask of the tower launches that take the wave-private kernel"},
    {"MURAL_DEBUG_TOWER_GRID", "diagnostic: resident workgroups of the tower launches"},
    {"MURAL_DEBUG_TOWER_LDS", "diagnostic: inflates the tower launches' LDS request"},
    {"MURAL_DEBUG_TOWER_STAGGER", "diagnostic: staggered start of the tower workgroups"},
    {nullptr, nullptr}};
  return table;
}

const char* dev_env(const char* name) {
  if (!g_dev_switches) return nullptr;
  for (const DevSwitch* s = dev_switch_table(); s->name; ++s)
    if (!std::strcmp(s->name, name)) return std::getenv(name);
  std::fprintf(stderr, "libmural_hip (debug flavour): %s is read but not listed in dev_switch_table\n", name);
  return nullptr;
}

// Stream priorities of the side streams: an experiment, OFF.  Serving the mid tower's chain first (its launches ask for one workgroup
// slot per CU, so its tail alone leaves half the chip empty while the large tower's tail alone fills it) gained 2 % on the training leg
// run alone (573 / 584 / 576 / 575 -> 593 / 593 / 583 steps/s, tools/r6_train_ab.sh) -- and HALVED it (615 -> 311 steps/s) when the
// process had run the file-to-file prediction leg before (tools/r6_train_after.py): with more streams alive in the process the
// prioritised queue starves the others.  Not worth a behaviour that depends on what else the process did.
int side_priority_mode() {
  const char* e = dev_env("MURAL_SIDE_PRIORITY");
  return e ? atoi(e) : 0;
}

thread_local std::vector<size_t> g_ws_layout;      // (read by the debug flavour's mural_debug_last_ws_layout)

size_t ws_guard_bytes() {
  const char* e = dev_env("MURAL_DEBUG_WS_GUARD");
  return e ? (size_t)atol(e) : 0;
}
void ws_layout_reset() { g_ws_layout.clear(); }
void ws_layout_add(size_t off, size_t bytes) {
  if (g_ws_layout.size() < 4096) {
    g_ws_layout.push_back(off);
    g_ws_layout.push_back(bytes);
  }
}

SideStream* side_stream_slot(int dev) {
  static SideStream per_device[64];
  return &per_device[dev];
}

// one thread per (row, column): order-k index over the strand-oriented window
__global__ void encode_kmer_kernel(MuralGenome g, const int64_t* __restrict__ pos, const uint8_t* __restrict__ strand,
                                   int64_t n, int off, int width, int order, int ncol, int64_t* __restrict__ out) {
  const int64_t total = n * ncol;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / ncol;
    const int col = (int)(i - row * ncol);
    const int64_t ws = pos[row] + off;
    const bool neg = strand[row] != 0;
    int64_t val = 0;
    bool bad = false;
    for (int d = 0; d < order; ++d) {
      const int j = col + d;  // index in the strand-oriented window
      uint32_t s = neg ? genome_sym(g.packed2, g.nmask, g.length, ws + (width - 1 - j))
                       : genome_sym(g.packed2, g.nmask, g.length, ws + j);
      if (s > 3u) bad = true;
      if (neg) s = 3u - (s & 3u);
      val = val * 4 + (int64_t)(s & 3u);
    }
    int64_t sentinel = 1;
    for (int d = 0; d < order; ++d) sentinel *= 4;
    out[i] = bad ? sentinel : val;
  }
}

// one thread per (row, column): writes the 4 channel values (coalesced along the column axis per channel)
__global__ void encode_onehot_kernel(MuralGenome g, const int64_t* __restrict__ pos, const uint8_t* __restrict__ strand,
                                     int64_t n, int off, int width, float* __restrict__ out) {
  const int64_t total = n * width;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / width;
    const int j = (int)(i - row * width);
    const int64_t ws = pos[row] + off;
    const bool neg = strand[row] != 0;
    uint32_t s = genome_sym_iupac(g, neg ? ws + (width - 1 - j) : ws + j);
    if (neg) s = sym_complement(s);
    float* o = out + row * 4 * (int64_t)width + j;
    // channel set of the symbol (bit ch = base ch is possible): one-hot, pairs 0.5, triples 1/3, N 0.25 (preprocessing.py:758-772)
    const uint32_t set = (0xF7BDEC963A5F8421ull >> (4u * s)) & 0xFu;
    const int members = __popc(set);
    const float v = members == 1 ? 1.0f : members == 2 ? 0.5f : members == 3 ? (float)(1.0 / 3.0) : 0.25f;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) o[(int64_t)ch * width] = ((set >> ch) & 1u) ? v : 0.0f;
  }
}

// the windows as symbols (one byte per column, strand-oriented like the one-hot form): what the sequence kernels consume -- the
// training step's first layer works from these, so a loader that holds the packed genome need not expand 16 bytes per column first
// (a thread takes 4 consecutive bytes of the flattened (row, column) index: one division and one aligned 4-byte store per group -- a
// byte store per thread made this 17.6 us for the training batch's 8 MB; a group that straddles two rows goes byte by byte)
__global__ __launch_bounds__(256) void encode_symbols_kernel(MuralGenome g, const int64_t* __restrict__ pos, const uint8_t* __restrict__ strand,
                                                             int64_t n, int off, int width, uint8_t* __restrict__ out) {
  const int64_t total = n * width;
  const int64_t groups = (total + 3) >> 2;
  for (int64_t gi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; gi < groups; gi += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = gi << 2;
    const int64_t row = i0 / width;
    const int j = (int)(i0 - row * width);
    if (j + 3 < width && (reinterpret_cast<uintptr_t>(out) & 3u) == 0u) {
      const int64_t ws = pos[row] + off;
      const bool neg = strand[row] != 0;
      uint32_t packed = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uint32_t s = genome_sym_iupac(g, neg ? ws + (width - 1 - (j + e)) : ws + (j + e));
        if (neg) s = sym_complement(s);
        packed |= s << (8 * e);
      }
      *reinterpret_cast<uint32_t*>(out + i0) = packed;
    } else {
      for (int e = 0; e < 4 && i0 + e < total; ++e) {
        const int64_t i = i0 + e;
        const int64_t r = i / width;
        const int jj = (int)(i - r * width);
        const int64_t ws = pos[r] + off;
        const bool neg = strand[r] != 0;
        uint32_t s = genome_sym_iupac(g, neg ? ws + (width - 1 - jj) : ws + jj);
        if (neg) s = sym_complement(s);
        out[i] = (uint8_t)s;
      }
    }
  }
}

// A thread converts 4 consecutive symbols of the flattened (row, column) index: 16 independent channel loads in flight and one
// aligned 4-byte store (the symbol buffer is linear in that index, so a group may straddle two rows).  HBM-bound: 16 B in,
// 1 B out per column.
// bad_code < 0: a column that is no MuRaL symbol becomes N and raises the status word; otherwise it becomes bad_code, silently (the
// INDEL first level reads such a column's floats itself: indel_level0.hip)
__global__ __launch_bounds__(256) void dense_to_symbols_kernel(const float* __restrict__ x, int64_t n, int L,
                                                               uint8_t* __restrict__ sym, int32_t* __restrict__ status, int bad_code) {
  const int64_t total = n * L;
  const int64_t groups = (total + 3) >> 2;
  for (int64_t gidx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; gidx < groups; gidx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = gidx << 2;
    int64_t row = i0 / L;
    int j = (int)(i0 - row * L);
    float v[4][4];
    if (j + 3 < L) {
      // the group lies inside one row: its four columns of a channel are 16 consecutive bytes (4-byte aligned: rows have odd lengths) --
      // four 16-byte loads instead of sixteen dword loads that each use a quarter of what they fetch
      struct __attribute__((packed, aligned(4))) Quad { float v[4]; };
      const float* p = x + row * 4 * (int64_t)L + j;
      Quad q[4];
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) q[ch] = *reinterpret_cast<const Quad*>(p + (int64_t)ch * L);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) v[e][ch] = q[ch].v[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool live = i0 + e < total;
        const float* p = x + row * 4 * (int64_t)L + j;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) v[e][ch] = live ? p[(int64_t)ch * L] : 0.f;
        if (++j == L) {
          j = 0;
          ++row;
        }
      }
    }
    uint32_t packed = 0;
    bool bad = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int s = dense_symbol(v[e][0], v[e][1], v[e][2], v[e][3]);
      if (s < 0) {
        s = bad_code < 0 ? (int)SYM_N : bad_code;
        bad |= bad_code < 0 && i0 + e < total;
      }
      packed |= (uint32_t)s << (8 * e);
    }
    if (bad && status) atomicOr(status, (int)MURAL_E_ENCODING);
    if (i0 + 3 < total) {
      *reinterpret_cast<uint32_t*>(sym + i0) = packed;
    } else {
      for (int e = 0; i0 + e < total; ++e) sym[i0 + e] = (uint8_t)(packed >> (8 * e));
    }
  }
}

int launch_dense_to_symbols(const float* x, int64_t n, int L, uint8_t* sym, int32_t* status, hipStream_t stream, int bad_code) {
  const int64_t total = n * L;
  if (total == 0) return MURAL_OK;
  MURAL_REQUIRE((reinterpret_cast<uintptr_t>(sym) & 3u) == 0, "dense_to_symbols: the symbol buffer must be 4-byte aligned");
  const int block = 256;
  const int64_t groups = (total + 3) / 4;
  const int grid = (int)((groups + block - 1) / block < 16384 ? (groups + block - 1) / block : 16384);
  hipLaunchKernelGGL(dense_to_symbols_kernel, dim3(grid), dim3(block), 0, stream, x, n, L, sym, status, bad_code);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

}  // namespace mural

using namespace mural;

extern "C" const char* mural_last_error(void) { return mural::last_error_cstr(); }
extern "C" int mural_abi_version(void) { return 2; }

static int window_geometry(int radius, int indel, int* off, int* width) {
  MURAL_REQUIRE(radius >= 1, "radius must be >= 1, got %d", radius);
  *off = indel ? -radius + 1 : -radius;
  *width = indel ? 2 * radius : 2 * radius + 1;
  return MURAL_OK;
}

extern "C" int mural_encode_kmer(const MuralGenome* g, const int64_t* pos, const uint8_t* strand, int64_t n,
                                 int32_t radius, int32_t order, int32_t indel, int64_t* out, void* stream) {
  MURAL_REQUIRE(g && g->packed2 && g->nmask, "genome pointers must not be NULL");
  MURAL_REQUIRE(order >= 1 && order <= 12, "local_order must be in [1,12], got %d", order);
  int off, width;
  if (int rc = window_geometry(radius, indel, &off, &width)) return rc;
  const int ncol = width - (order - 1);
  MURAL_REQUIRE(ncol >= 1, "window too short for order %d", order);
  if (n == 0) return MURAL_OK;
  MURAL_REQUIRE(pos && strand && out, "pos/strand/out must not be NULL");
  const int64_t total = n * ncol;
  const int block = 256;
  const int grid = (int)((total + block - 1) / block < 8192 ? (total + block - 1) / block : 8192);
  hipLaunchKernelGGL(encode_kmer_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, *g, pos, strand, n, off, width,
                     order, ncol, out);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

extern "C" int mural_encode_onehot(const MuralGenome* g, const int64_t* pos, const uint8_t* strand, int64_t n,
                                   int32_t radius, int32_t indel, float* out, void* stream) {
  MURAL_REQUIRE(g && g->packed2 && g->nmask, "genome pointers must not be NULL");
  MURAL_REQUIRE(g->n_amb == 0 || (g->amb_pos && g->amb_sym), "genome: n_amb > 0 needs amb_pos and amb_sym");
  int off, width;
  if (int rc = window_geometry(radius, indel, &off, &width)) return rc;
  if (n == 0) return MURAL_OK;
  MURAL_REQUIRE(pos && strand && out, "pos/strand/out must not be NULL");
  const int64_t total = n * width;
  const int block = 256;
  const int grid = (int)((total + block - 1) / block < 16384 ? (total + block - 1) / block : 16384);
  hipLaunchKernelGGL(encode_onehot_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, *g, pos, strand, n, off,
                     width, out);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

extern "C" int mural_encode_symbols(const MuralGenome* g, const int64_t* pos, const uint8_t* strand, int64_t n, int32_t radius,
                                    int32_t indel, uint8_t* out, void* stream) {
  MURAL_REQUIRE(g && g->packed2 && g->nmask, "genome pointers must not be NULL");
  MURAL_REQUIRE(g->n_amb == 0 || (g->amb_pos && g->amb_sym), "genome: n_amb > 0 needs amb_pos and amb_sym");
  int off, width;
  if (int rc = window_geometry(radius, indel, &off, &width)) return rc;
  if (n == 0) return MURAL_OK;
  MURAL_REQUIRE(pos && strand && out, "pos/strand/out must not be NULL");
  const int64_t groups = (n * width + 3) / 4;
  const int block = 256;
  const int grid = (int)((groups + block - 1) / block < 16384 ? (groups + block - 1) / block : 16384);
  hipLaunchKernelGGL(encode_symbols_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, *g, pos, strand, n, off, width, out);
  MURAL_HIP_CHECK(hipGetLastError());
  return MURAL_OK;
}

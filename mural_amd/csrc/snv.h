// Device-side data model of the fused SNV forward (shared by the host folder and the kernels).
#pragma once
#include "common.h"

namespace mural {

constexpr int SNV_C = 32;          // conv channels handled by the MFMA path
constexpr int SNV_K = 3;           // conv taps
constexpr int SNV_NLAYER = 10;     // C->C convs per tower: 4 (RBs1) + conv2 + 4 (RBs2) + conv3
constexpr int SNV_KSTEPS = 24;     // 96 / 4 k-steps of v_mfma_f32_16x16x4_f32
constexpr int SNV_WFRAG = 2 * SNV_KSTEPS * 64;   // floats per layer: [mblock][kstep][lane]
constexpr int SNV_LUT = 125 * SNV_C;             // 3-mer lookup table (A,C,G,T,N)^3 x channels
constexpr int SNV_TAPS = 3 * N_SYM * SNV_C;      // per-tap, per-symbol contributions (generic path)
constexpr int SNV_LUTBLK = SNV_LUT + SNV_TAPS + SNV_C;   // lut | taps | bias0, contiguous in the blob
constexpr int SNV_LUT4 = 256 * SNV_C;             // pair table (A,C,G,T)^4 x channels: max of the two 3-mer rows of adjacent columns
constexpr int SNV_MAXCLASS = 16;
constexpr int SNV_CHUNK = 131072;  // sites per stage-1 / tower launch sequence (bounds the x0 scratch: 3.4 GB at R=1000)
constexpr int SNV_NB2MAX = 9;      // max 16-column blocks a wave owns in one stage (waves split M x column parity)
constexpr int SNV_THREADS = 256;
constexpr int SNV_WAVES = SNV_THREADS / 64;
constexpr int SNV_MID_HALF = 100;  // model_snv.py:473

// layer roles inside a tower (index into wfrag / bias / post_s / post_t)
enum { L_RB1A_C1 = 0, L_RB1A_C2, L_RB1B_C1, L_RB1B_C2, L_CONV2, L_RB2A_C1, L_RB2A_C2, L_RB2B_C1, L_RB2B_C2, L_CONV3 };
// standalone BN affine maps (index into ex_s / ex_t)
enum { EX_RB1_ENTRY = 0, EX_BN_MID = 1, EX_BN_OUT = 2, EX_FC_BN = 3, EX_COUNT = 4 };

struct TowerGeom {
  int L1;          // columns of the tower input (2R+1 for large, 201 for mid)
  int col0;        // first column of the tower input inside the window (crop start)
  int L[3];        // lengths after maxpool1/2/3  (L2, L3, L4)
  int pk[3], ps[3], pp[3];   // pool kernel / stride / pad
  int Sc[3];       // column stride per position in the flattened geometry (= L + 1 separator)
  int NC[3];       // logical columns: 1 + P*Sc
  int nb[3];       // 16-column MFMA blocks covering [0, NC)
  FastDiv dL[3];   // divide by L[i]
  FastDiv dSc[3];  // divide by Sc[i]
};

struct TowerDev {                 // all device pointers into one blob
  const float* lut;               // [125][32]
  const float* lut4;              // [256][32] max(lut[abc], lut[bcd]) over (A,C,G,T)^4 (snv_stage1_kernel)
  const float* taps;              // [3][16][32]
  const float* bias0;             // [32] first-layer conv bias
  const float* wfrag;             // [10][SNV_WFRAG]
  const float* wfrag4;            // [10][SNV_WFRAG] the same, [mblock][kstep / 4][lane][4]: one 16-byte load = four k-steps of a lane
  const float* bias;              // [10][32]
  const float* post_s;            // [10][32] BN scale applied to relu(layer output) for the next conv
  const float* post_t;            // [10][32]
  const float* ex_s;              // [4][32]
  const float* ex_t;              // [4][32]
  const float* fc_w;              // [n_class][32]
  const float* fc_b;              // [n_class]
};

struct LocalDev {
  const float* emb;               // [emb_rows][5]
  const float* w1t;               // [in1][h1]  (transposed lin_layers.0.weight), in1 = 5*cols
  const float* b1;                // [h1]
  const float* w2t;               // [h1][h2]   BN0 folded in
  const float* b2;                // [h2]
  const float* w3t;               // [h2][n_class]  BN1 folded in
  const float* b3;                // [n_class]
  const float* frag;              // the three weight matrices in MFMA A-fragment order (snv_local_mlp_mfma), frag_floats long
  int frag_floats;
  int cols, emb_rows, in1, h1, h2, n_class;
};

struct LocalMfmaDims { int K1p, K2p, K3p, n1b, n2b, s1, sx, dbg; };   // LDS layout of snv_local_mlp_mfma (snv_local_mfma.h)

struct Stage1Tower { int L1, col0, L2, pk, ps, pp; };

struct Stage1Args {               // snv_stage1_kernel: window decode + first conv layer + maxpool1
  Stage1Tower tw[2];              // 0 = large, 1 = mid
  const float* lut[2];            // lut | taps | bias0 blocks of the two towers
  const float* lut4;              // large tower: pair table [256][32] (throughput kernel, 15-wide pools), or nullptr
  int Lwin, cw, wave_bytes, x0_cols, nwords, radius;
  int64_t n;
  const uint8_t* codes;           // [n][Lwin] symbols (dense path)
  MuralGenome genome;             // packed path
  const int64_t* pos;
  const uint8_t* strand;
  float* x0;                      // [n][x0_cols][32] pooled first-layer activations (large columns, then mid)
  const float* dense;             // small-batch dense entry: [n][4][Lwin] one-hot / IUPAC fractions read directly (codes unused), or nullptr
  int32_t* status;                // dense: MURAL_E_ENCODING is or-ed in for a column that is no valid encoding (nullptr: not reported)
  // small-batch launch, Network2: one extra workgroup (blockIdx n) runs the local branch (snv_local_mfma.h) beside the site workgroups
  int loc_on;
  LocalDev loc;
  LocalMfmaDims loc_d;
  const int64_t* loc_cat;         // [n][loc.cols] k-mer ids
  float* loc_out;                 // [n][n_class] logits
  int* zero;                      // small-batch launch: n ints cleared for the tower launch behind it (SnvFwdArgs::tile_count), or nullptr
  int site_mode;                  // 1: the workgroup-per-site kernel at any batch size (long windows: sixteen per-wave windows do not fit LDS)
  int dbg_alias;                  // timing experiment (MURAL_DEBUG_S1_ALIAS): every site writes the x0 rows of site (row % 64) -- WRONG results
};

// training-mode first layer of ONE tower (snv_stage1.hip: first_train_kernel)
constexpr int FIRST_TRAIN_MAXGRID = 256;
// The BatchNorm-backward apply of the layer BEHIND the first layer folded into the first layer's backward (channel-last form only):
// the pooled gradient is not read but made per element, dy = relu'(x) * gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)) + add1
// + add2 (the arithmetic of bn_bwd_apply_cl_kernel, conv32_cl.hip), from that BatchNorm's input gradient dz, its saved input x (= this
// layer's output), its completed sums acc and up to two residual gradients; workgroup 0 writes its dgamma / dbeta.  dz == nullptr: none.
struct FirstFold {
  const float* dz;
  const float* x;
  const float* add1;
  const float* add2;
  const float* state;             // scale | beta | mean | invstd
  const float* gamma;
  const double* acc;              // [MURAL_BN_SLOTS][2][32]: sum(dz), sum(dz * xhat)
  double n;                       // elements per channel
  float* dgamma;
  float* dbeta;
};
struct FirstTrainArgs {
  Stage1Tower tw;
  int Lwin, cw, wave_bytes;
  int64_t B;
  const uint8_t* sym;             // [B][Lwin] window symbols
  const float* lutblk;            // forward: lut | taps | bias of this step's batch statistics
  float* y;                       // forward: [B][32][L2]
  uint8_t* arg;                   // [B][L2][32] window offset of the pooled maximum (written forward, read backward)
  const float* dy;                // backward: [B][32][L2]
  float* dpart;                   // backward: [grid][SNV_LUTBLK] per-workgroup gradient tables
  int cl;                         // 1: y / dy are channel-last [B][L2][32] (the composed training step), 0: [B][32][L2]
  double* stat;                   // forward, cl only: [MURAL_BN_SLOTS][2][32] batch sums of relu(y), relu(y)^2 (nullptr: none)
  FirstFold fold;                 // backward, cl only
  int dbg;                        // timing experiments (MURAL_DEBUG_FIRST): 1 no LDS atomics, 2 no gradient / arg-max loads, 4 no index reads
  unsigned long long* stamps;     // diagnostic (mural_debug_first_set_stamps): [workgroup][8] wall-clock ticks (100 MHz) of wave 0's phases
};
extern unsigned long long* g_first_stamps;
int first_train_grid(int64_t B);
bool first_train_supported(int C, int pk);
int launch_first_train(FirstTrainArgs a, bool bwd, hipStream_t stream);

struct SnvFwdArgs {
  TowerGeom geom[2];              // 0 = large, 1 = mid
  TowerDev tw[2];
  int P;                          // positions per tile
  int nbuf;                       // floats per activation buffer
  int Lwin;                       // window length (2R+1)
  int n_class;
  int has_local;                  // Network2: mix with local softmax
  int64_t n;                      // rows
  const float* x0;                // [n][x0_cols][32] stage-1 output (snv_stage1_kernel)
  int x0_cols;                    // L2 large + L2 mid
  const float* local_logits;      // [n][n_class]
  float* out;                     // [n][n_class]
  float* taps;                    // debug dump (tile 0) or nullptr
  int tap_stride;                 // floats per dumped buffer
  unsigned long long* stamps;     // diagnostic per-phase cycle sums [grid][32] or nullptr
  const int32_t* status;          // dense entry: encoding status word; while it is set the head writes NaN (nullptr: not checked)
  // Tower range of this launch.  {0, 1}: both towers and the head in one launch.  Split mode runs {0, 0} (large tower,
  // its fc logits go to xlogit) and then {1, 1} (mid tower with its own, larger tile; reads xlogit and does the head).
  int tw_first, tw_last;
  float* xlogit;                  // [n][SNV_MAXCLASS] large-tower logits between the two launches of the split mode
  // Tower-parallel launch (phase 0, tw {0, 1}, small batches): grid = 2 workgroups per tile, one per tower; both publish their
  // logits (xlogit / xlogit2) and count up tile_count[tile] (zero before the launch); the second one to arrive runs the head.
  int par;
  float* xlogit2;                 // [n][SNV_MAXCLASS] mid-tower logits
  int* tile_count;                // [tiles]
  // Stage range of this launch.  phase 0: everything.  phase 1: first conv stage only (entry + 4 residual convs + max-pool
  // 2 + BN), the pooled tile goes to s3[tower].  phase 2: the two short stages (6 convs), global max, fc (+ head), reading
  // s3[tower] - with a tile of many more positions, so that the short stages run full-width layers.
  int phase;
  float* s3[2];                   // [n][L3][32] per tower: input of the second conv stage, in the layout of x0
  // Wave-private launch (snv_tower_wave.hip; stage-split launches of the throughput path): P counts the sites of ONE WAVE, nbuf
  // the floats of one wave's image, geom is laid out for P sites per wave.
  int wave;
  // wave-private launch: units are handed out through this counter (zero before the launch) instead of a fixed stride -- the two
  // waves that share a SIMD do not run at the same speed (the older wave slot wins the issue arbitration), so with equal shares
  // the favoured half of the waves is done at 0.72 of the launch and the rest finishes alone; nullptr: fixed stride
  int* unit_counter;
  int stagger;                    // wave-private launch: late start of every second workgroup of a CU, in units of 8128 cycles
  // Long windows (run-time-geometry first-stage instance, one site per wave): the launch's "rows" are SEGMENTS of the large tower's
  // pooled row read in place from x0 -- row v = segment v % seg_n of site v / seg_n, starting at column seg_col0 + (v % seg_n) *
  // seg_step of that site's x0 row.  seg_n == 0: rows are sites (column offset by tower, as always).
  int seg_n, seg_step, seg_col0;
};

// ---- cross-position reuse (snv_reuse.hip, snv_tower_wave.hip): the edge tile of a site
constexpr int RU_EC = 9;                 // pooled columns per window side fed to the edge pyramid (9 -> 5 valid after 4 convs)
constexpr int RU_EV = 5;                 // window-specific columns per side after the four convs
constexpr int RU_L = 2 * RU_EC;          // data columns per site in the edge tile
constexpr int RU_SC = RU_L + 1;          // + zero separator

struct EdgeArgs {
  TowerGeom ge;             // stage 0 = the edge tile: L = 18, Sc = 19
  TowerDev tw;
  int P;
  int nbuf;                 // floats per LDS buffer
  int64_t n;                // sites of this launch
  const int64_t* pos;       // genome positions of the sites
  const uint8_t* strand;    // 0 '+', 1 '-' per site: selects the row set
  int64_t glen;             // chromosome length
  int64_t t0[2], nb;        // rows per strand: oriented coordinate of row 0; row count
  int woff;                 // oriented offset of the tower's first input column from the site (-R large, -100 mid)
  int L1;                   // conv columns of the tower input (2R+1 / 201)
  int L2, L3;               // columns after maxpool1 / maxpool2
  int D;                    // stride of maxpool1 on the base axis (15 / 3)
  int pk2, ps2, pp2;        // maxpool2
  int u_lo, u_hi;           // pooled columns whose window lies inside the shared rows (gathered from S)
  int right_pad;            // the last pooled column contains the right zero-padded conv column (large at R = 1000: yes)
  const float* F[2];        // per strand: shared pooled first-layer rows
  const float* El[2];       // first pooled column of a window starting at row b
  const float* Er[2];       // last pooled column of a window ending at row b (used when right_pad)
  const float* R[2];        // shared first-conv-stage output rows (raw)
  const float* S[2];        // shared maxpool2 + BN rows
  float* s3;                // [n][L3][32] out
};

// wave-private form of the edge kernel (snv_tower_wave.hip): EW_P sites per wave = nine 16-column blocks, like the first-stage launches
constexpr int EW_P = 7;
size_t edge_wave_lds_bytes();
int launch_snv_edge_wave(const EdgeArgs& e, int* unit_counter, hipStream_t stream);

}  // namespace mural

struct MuralSnvModel {
  MuralSnvShape shape;
  mural::SnvFwdArgs args;         // geometry + device pointers (input/output fields filled per call): both towers, one tile size
  // split mode: four launches per chunk - (tower, phase) = (large, 1), (mid, 1), (large, 2), (mid, 2 + head) - each with the
  // largest tile that keeps two workgroups per CU
  mural::SnvFwdArgs args_split[4];
  size_t lds_split[4];
  bool split;
  // small batches (the reference's default predict call is 16 sites): ONE launch, one site per workgroup, both towers and the head
  // -- the latency of a call is one tile through the layers, so the tile is as narrow as it gets
  mural::SnvFwdArgs args_small;
  size_t lds_small;
  mural::LocalDev local;
  float* blob;                    // device allocation holding every folded tensor
  size_t blob_floats;
  size_t lds_bytes;               // dynamic LDS of the tower kernel
  mural::Stage1Args s1;           // stage-1 kernel arguments (input/output fields filled per call)
  size_t s1_lds_bytes;
  bool s1_pair;                   // the stage-1 kernel carries the large tower's pair table (Stage1Args::lut4)
  // Network2, small batches: the local branch rides in the first-stage launch (Stage1Args::loc_on) when its MFMA kernel applies
  bool loc_fused;
  mural::LocalMfmaDims loc_d;
  size_t loc_lds;
  // Long windows (pooled first-stage rows of the large tower longer than a wave's LDS image: distal_radius 2000, 4000, ...): the first
  // conv stage of the large tower runs on SEGMENTS of the pooled row -- virtual rows of lw_LA (lw_nA per site, starts 0, 7 lw_nj,
  // 14 lw_nj, ...) and lw_LB columns (one per site, start lw_SB, ending at the row's end) with 4 halo columns towards the row's
  // interior, gathered from x0 into scratch; their pooled outputs are scattered into s3[0] and the short stages run on whole rows.
  bool longwin;
  bool front_only;                // long windows whose short stages do not fit the kernels either: only the segmented first stage of the large
                                  // tower runs here (mural_snv_forward_front); the caller finishes per layer (model/generic_eval.py)
  bool front_mid;                 // ... and the mid tower's launches + the head fit: mural_snv_forward_finish runs them fused around the caller's large-tower logits
  int lw_nA, lw_LA, lw_LB, lw_SB, lw_nj;
  mural::SnvFwdArgs args_lwA, args_lwB;
  size_t lds_lwA, lds_lwB;
  int64_t chunk;                  // sites per stage-1 / tower launch sequence (SNV_CHUNK, less for long windows)
};

/*
 * mural_hip.h -- C ABI of libmural_hip.so, the MI355X (gfx950) hot path for MuRaL models.
 *
 * The reference (CaiLiLab/MuRaL) has no FFI layer: its seam is the Python nn.Module protocol of the
 * classes returned by MuRaL/model/nn_utils.py:186 (model_choice).  Each entry point below names the
 * reference interface it replaces (file:line relative to the reference tree).  The binding that a
 * maintainer would add on the reference side is a ctypes stub; see INTEGRATION.md.
 *
 * Conventions
 *   - plain C types only; device pointers are raw HIP device addresses the caller owns ("dev"),
 *     host pointers are marked "host".  The library borrows every pointer for the duration of one call.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises.
 *   - return value: 0 = ok, otherwise a MURAL_E_* code; mural_last_error() gives the message of the
 *     last failure on the calling thread.
 *   - fp32 everywhere; integer outputs are bit-exact w.r.t. the reference.
 */
#ifndef MURAL_HIP_H
#define MURAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  MURAL_OK = 0,
  MURAL_E_INVALID = 1,     /* bad argument / unsupported configuration (maps to ValueError)        */
  MURAL_E_RUNTIME = 2,     /* HIP runtime failure (maps to RuntimeError)                            */
  MURAL_E_WORKSPACE = 3,   /* workspace too small                                                   */
  MURAL_E_ENCODING = 4     /* dense distal input holds a column that is not a MuRaL one-hot/IUPAC
                              column (reported asynchronously through the status word, see below)   */
};

const char* mural_last_error(void);
int mural_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Genome in packed form (product input format):
 *   packed2 : 16 bases per uint32, base i in bits [2*(i%16), +2)  (A0 C1 G2 T3; 0 where masked)
 *   nmask   : 32 bases per uint32, bit (i%32) set when the base is not one of ACGT
 * Bases outside [0, length) read as 'N' (the reference imputes chromosome ends with 'N',
 * MuRaL/data/preprocessing.py:682-695, :791-804).
 * IUPAC ambiguity codes other than N (R Y M S W K B D H V; fractional one-hot columns in the reference,
 * preprocessing.py:762-772) are rare and live in a sparse side table: their bit in nmask is set (the k-mer encoder
 * treats them like N, :655-666) and (amb_pos, amb_sym) lists them in ascending position with their symbol
 * MURAL_SYM_R .. MURAL_SYM_V.  The one-hot encoder and the fused forward resolve them through that table.
 * ---------------------------------------------------------------------------------------------- */
enum { MURAL_SYM_A = 0, MURAL_SYM_C, MURAL_SYM_G, MURAL_SYM_T, MURAL_SYM_N, MURAL_SYM_R, MURAL_SYM_Y, MURAL_SYM_M,
       MURAL_SYM_S, MURAL_SYM_W, MURAL_SYM_K, MURAL_SYM_B, MURAL_SYM_D, MURAL_SYM_H, MURAL_SYM_V };
typedef struct {
  const uint32_t* packed2;   /* dev */
  const uint32_t* nmask;     /* dev */
  int64_t length;            /* bases in this chromosome */
  const int64_t* amb_pos;    /* dev, ascending positions of non-N ambiguity codes (NULL when n_amb == 0) */
  const uint8_t* amb_sym;    /* dev, MURAL_SYM_R .. MURAL_SYM_V per entry */
  int64_t n_amb;
} MuralGenome;

/* Replaces seq_digit_encoder (MuRaL/data/preprocessing.py:636-723) for one site per row.
 * out: dev int64 [n][2*radius + (indel?0:1) - (order-1)], values in [0, 4^order].
 * strand: dev uint8 [n], 0 = '+', 1 = '-'.  indel != 0 selects the indel window (:564-566).      */
int mural_encode_kmer(const MuralGenome* g, const int64_t* pos, const uint8_t* strand, int64_t n,
                      int32_t radius, int32_t order, int32_t indel, int64_t* out, void* stream);

/* Replaces seq_ohe_encoder (MuRaL/data/preprocessing.py:756-816): one-hot columns, 0.25 x 4 for N and the
 * fractional columns of the other IUPAC codes (:762-772, complemented on the '-' strand :774-788).
 * out: dev float [n][4][2*radius + (indel?0:1)].                                                   */
int mural_encode_onehot(const MuralGenome* g, const int64_t* pos, const uint8_t* strand, int64_t n,
                        int32_t radius, int32_t indel, float* out, void* stream);

/* The same windows as one SYMBOL per column (MURAL_SYM_*: A C G T N R Y M S W K B D H V = 0..14, strand-oriented and complemented
 * like the one-hot columns; positions outside the record are N): what mural_op_dense_to_symbols recovers from the one-hot tensor,
 * without the detour -- the `symbols` input of mural_snv_train_forward.  out: dev uint8 [n][2*radius + (indel?0:1)].            */
int mural_encode_symbols(const MuralGenome* g, const int64_t* pos, const uint8_t* strand, int64_t n, int32_t radius,
                         int32_t indel, uint8_t* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SNV model family (Network0 / Network1 / Network2, MuRaL/model/model_snv.py:19-525), eval mode.
 * Raw parameters are handed over as HOST pointers in the reference's state_dict naming; the library
 * folds BatchNorm running statistics, builds the first-layer 3-mer lookup tables and the MFMA weight
 * fragments, and keeps device copies inside the opaque handle.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { const float *weight, *bias, *running_mean, *running_var; } MuralBN;   /* host, [C] each */
typedef struct { const float *weight, *bias; } MuralAffine;                             /* host */

typedef struct {            /* ResBlock, model_snv.py:794-812 */
  MuralBN bn1; MuralAffine conv1;   /* conv weight [C][C][3] */
  MuralBN bn2; MuralAffine conv2;
} MuralResBlock;

typedef struct {            /* one conv tower, model_snv.py:350-388 (mid) / :392-430 (large) */
  MuralBN bn_in;  MuralAffine conv_in;     /* conv1.0 / conv1.1   weight [C][4][K] */
  MuralResBlock rbs1[2];                    /* RBs1.{0,1} */
  MuralBN bn_mid; MuralAffine conv_mid;    /* conv2.0 / conv2.1   weight [C][C][K] */
  MuralResBlock rbs2[2];                    /* RBs2.{0,1} */
  MuralBN bn_out; MuralAffine conv_out;    /* conv3.0 / conv3.1 (+ReLU) */
  MuralBN fc_bn;  MuralAffine fc;          /* distal_fc.0 / distal_fc.2  weight [n_class][C] */
} MuralTower;

typedef struct {            /* local branch, model_snv.py:322-339 */
  const float* emb;                         /* emb_layer.weight [emb_rows][5] */
  MuralAffine lin[2];                       /* lin_layers.{0,1}: [h1][5*cols], [h2][h1] */
  MuralBN bn[2];                            /* bn_layers.{0,1} */
  MuralAffine out;                          /* local_fc.0 (Network2) / model.output_layer (Network0) */
} MuralLocal;

typedef struct {
  int32_t model_no;        /* 0 local-only, 1 towers only, 2 both (nn_utils.py:213-216)            */
  int32_t n_class;
  int32_t local_cols;      /* number of k-mer columns = len(emb_dims)                              */
  int32_t emb_rows;        /* 4^local_order + 1                                                    */
  int32_t hidden1, hidden2;
  int32_t channels;        /* CNN_out_channels (this build: 32)                                    */
  int32_t ksize;           /* CNN_kernel_size  (this build: 3)                                     */
  int32_t distal_len;      /* 2*distal_radius + 1                                                  */
  float bn_eps;            /* 1e-5 */
} MuralSnvShape;

typedef struct { MuralLocal local; MuralTower mid, large; } MuralSnvParams;

typedef struct MuralSnvModel MuralSnvModel;

int  mural_snv_model_create(const MuralSnvShape* shape, const MuralSnvParams* host_params, MuralSnvModel** out);
void mural_snv_model_destroy(MuralSnvModel* m);

/* scratch the forward calls need for a batch of n rows (bytes; allocate once, reuse).
 * dense != 0: for mural_snv_forward_dense (adds n*distal_len symbol bytes); 0: for mural_snv_forward_packed */
size_t mural_snv_workspace_bytes(const MuralSnvModel* m, int64_t n, int32_t dense);
/* The smallest workspace the forward calls accept.  mural_snv_workspace_bytes keeps the pooled stage-3 inputs of up to four
 * 131072-site chunks (5.5 KB per site at the default geometry: + 2.2 GB on calls of >= 524288 sites) so that the short stages run as
 * one launch per tower (~1 % faster); with a workspace between the two sizes they run per chunk -- same results bit for bit.       */
size_t mural_snv_workspace_bytes_min(const MuralSnvModel* m, int64_t n, int32_t dense);

/* Replaces Network{0,1,2}.forward((cont_x, cat_x), distal_x) (model_snv.py:104-108, :226-287, :439-525)
 * for already-encoded tensors.  cat_x: dev int64 [n][local_cols] (ignored for model_no 1);
 * distal_x: dev float [n][4][distal_len] contiguous (ignored for model_no 0).
 * out: dev float [n][n_class] -- log-probabilities (Network1/2) or raw logits (Network0).
 * status: dev int32[1], set to MURAL_E_ENCODING by the kernels if a distal column is not a MuRaL
 * encoding; while it is set the call's output is overwritten with NaN, so the failure is visible
 * without a host round trip (the caller reads and clears the word whenever it synchronises); may be NULL. */
int mural_snv_forward_dense(const MuralSnvModel* m, const int64_t* cat_x, const float* distal_x, int64_t n,
                            float* out, void* workspace, size_t workspace_bytes, int32_t* status, void* stream);
/* The same forward for windows handed over as ONE SYMBOL BYTE per column (dev uint8 [n][distal_len], MURAL_SYM_* codes 0..14 as
 * mural_op_dense_to_symbols / mural_host_dense_to_symbols / mural_encode_symbols write them; the bytes are not checked).  Workspace:
 * mural_snv_workspace_bytes(m, n, 0).                                                                                            */
int mural_snv_forward_symbols(const MuralSnvModel* m, const int64_t* cat_x, const uint8_t* symbols, int64_t n, float* out,
                              void* workspace, size_t workspace_bytes, void* stream);

/* Fused encode + forward straight from the packed genome: the path `mural_snv predict` takes
 * (replaces preprocessing.py:636-723 + :756-816 + model_snv.py:439-525 for n sites).
 * local_radius / local_order describe the k-mer window of the local branch.                        */
int mural_snv_forward_packed(const MuralSnvModel* m, const MuralGenome* g, const int64_t* pos,
                             const uint8_t* strand, int64_t n, int32_t local_radius, int32_t local_order,
                             float* out, void* workspace, size_t workspace_bytes, void* stream);

/* Cross-position reuse for dense same-strand site lists (no reference counterpart beyond the window
 * sharing of its encoders, MuRaL/data/preprocessing.py:602-610, :808-814).  Same result as mural_snv_forward_packed (within
 * rounding of identical operation sequences; parity tests: 1e-5 on probabilities) for n sites of one chromosome that lie inside
 * [pos_min, pos_max] (host-known bounds, at most mural_snv_reuse_chunk_span() bases apart) on the strands named by `strands`
 * (bit 0: '+' sites occur, bit 1: '-' sites occur; a site outside the bounds or on an unnamed strand comes back as NaN): the
 * first conv stage of both towers is evaluated once per base of the span and strand as dilated convs over pooled rows, and per
 * site only the pooled columns next to the window edges are recomputed.  Pays off from a site density of a few percent of the
 * bases.  Needs mural_snv_reuse_supported(m) (model_no 1 / 2, fused shape,
 * at least 18 pooled columns per tower: distal_radius >= 135).                                                       */
int     mural_snv_reuse_supported(const MuralSnvModel* m);
int64_t mural_snv_reuse_chunk_span(void);
size_t  mural_snv_reuse_workspace_bytes(const MuralSnvModel* m, int64_t n, int64_t span, int32_t strands);
int mural_snv_forward_packed_reuse(const MuralSnvModel* m, const MuralGenome* g, const int64_t* pos, const uint8_t* strand,
                                   int64_t n, int32_t strands, int64_t pos_min, int64_t pos_max, int32_t local_radius,
                                   int32_t local_order, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* Debug/validation hook used by the parity tests: same as forward_dense for the first tile, and dumps
 * the LDS-resident stage outputs of that tile to `taps` (dev float, see mural_snv_tap_layout).      */
int mural_snv_debug_taps(const MuralSnvModel* m, const int64_t* cat_x, const float* distal_x, int64_t n,
                         float* out, void* workspace, size_t workspace_bytes, float* taps, size_t taps_floats,
                         void* stream);
/* geometry of the fused kernel for this model: fills out[0..15] =
 * {P, NBUF_floats, L2_large, L3_large, L4_large, L2_mid, L3_mid, L4_mid, n_tap_slots, ...}          */
int mural_snv_tap_layout(const MuralSnvModel* m, int32_t* out16);

/* ------------------------------------------------------------------------------------------------
 * INDEL model (UNet_Small, MuRaL/model/model_indel.py:21-176), eval mode.  Parameters are HOST pointers in the
 * reference's state_dict naming; conv weights are [Cout][Cin][K].
 * ---------------------------------------------------------------------------------------------- */
typedef struct { MuralAffine conv; MuralBN bn; } MuralConvBN;            /* Sequential(Conv1d, BatchNorm1d) */
typedef struct { const float* conv5_w; MuralBN bn1; const float* conv1_w; MuralBN bn2; } MuralConvBlock;   /* :6-19 */

typedef struct {
  int32_t n_class;
  int32_t channels;        /* CNN_out_channels (8 in every shipped model)                          */
  int32_t ksize;           /* CNN_kernel_size (7)                                                  */
  int32_t down[6];         /* down_list                                                            */
  int32_t use_reverse;     /* strand-symmetrising input conv present (insertion models)            */
  int32_t length;          /* input length 2*distal_radius                                         */
  float bn_eps;
} MuralIndelShape;

typedef struct {
  MuralConvBN sym;                 /* conv.{0,1} (only when use_reverse)                           */
  MuralConvBN up_l[6];             /* uplblocks.i.{0,1}                                            */
  MuralConvBlock up_b[6];          /* upblocks.i.0.conv.{0,1,3,4}                                  */
  MuralConvBN down_l[5];           /* downlblocks.j.{1,2}                                          */
  MuralConvBlock down_b[5];        /* downblocks.j.0.conv.{0,1,3,4}                                */
  MuralAffine out1; MuralBN out_bn; MuralAffine out2;   /* out_conv.{0,1,3}                        */
  MuralBN fc_bn; MuralAffine fc;   /* out_fc.{0,2}                                                 */
} MuralIndelParams;

typedef struct MuralIndelModel MuralIndelModel;
int  mural_indel_model_create(const MuralIndelShape* shape, const MuralIndelParams* host_params, MuralIndelModel** out);
void mural_indel_model_destroy(MuralIndelModel* m);
size_t mural_indel_workspace_bytes(const MuralIndelModel* m, int64_t n);
/* Replaces UNet_Small.forward(distal_input) (model_indel.py:151-176).  distal_x: dev float [n][4][length];
 * out: dev float [n][n_class] positive Softplus scores (callers apply softmax, run_predict.py:214).  Any float tensor is accepted;
 * columns that are one-hot / IUPAC-fraction columns of the reference's encoder (preprocessing.py:756-816) travel as one symbol byte
 * each through the table-driven first level of the packed entry (same scores bit for bit), the others are evaluated from their floats. */
int mural_indel_forward_dense(const MuralIndelModel* m, const float* distal_x, int64_t n, float* out,
                              void* workspace, size_t workspace_bytes, void* stream);
/* The same for sites of a packed genome: replaces seq_ohe_encoder (MuRaL/data/preprocessing.py:756-816; indel window
 * [pos - R + 1, pos + R], :564-566) + UNet_Small.forward.  The window is decoded inside the first level's kernel and the
 * strand-symmetrising input conv (model_indel.py:29-32, :154-155) is evaluated there per symbol, so the one-hot window never
 * exists in HBM.  pos: dev int64 [n], strand: dev uint8 [n] (1 = '-'); the model's length must be 2 * R.                    */
int mural_indel_forward_packed(const MuralIndelModel* m, const MuralGenome* genome, const int64_t* pos, const uint8_t* strand,
                               int64_t n, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training-mode building blocks (MuRaL/training.py:404-450 over model_snv.py:439-525): forward with
 * batch-statistics BatchNorm and every backward, on [B][C][L] fp32 device tensors.  Composed by the
 * autograd glue in mural_amd/model/train_ops.py.  "zeroed by the caller" marks accumulation targets.
 * ---------------------------------------------------------------------------------------------- */
int mural_op_relayout(const float* W, float* wt, int32_t Cout, int32_t Cin, int32_t K, int32_t dgrad, void* stream);
int mural_op_conv1d(const float* in, const float* wt, const float* bias, float* out, int64_t B, int32_t Cin,
                    int32_t Cout, int32_t L, int32_t K, const float* pre_s, const float* pre_t, int32_t pre_relu,
                    int32_t post_relu, const float* res1, const float* res2, void* stream);
/* Batch sums of the training-mode BatchNorms live in accumulator blocks double[MURAL_BN_SLOTS][2][C], zeroed by the
 * caller: producers add into the copy picked by their workgroup index (same-address atomics serialise), readers sum. */
#define MURAL_BN_SLOTS 32
int mural_op_bn_stats(const float* x, int64_t B, int32_t C, int32_t L, int32_t relu, double* acc, void* stream);
int mural_op_bn_finalize(const double* acc, double n, int32_t C, const float* gamma,
                         const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                         float* scale, float* shift, float* mean, float* invstd, void* stream);
int mural_op_bn_apply(const float* x, int64_t B, int32_t C, int32_t L, int32_t relu, const float* scale,
                      const float* shift, float* y, void* stream);
int mural_op_bn_backward(const float* dz, const float* x, int64_t B, int32_t C, int32_t L, int32_t relu,
                         const float* mean, const float* invstd, const float* gamma, double* acc,
                         int32_t have_sums, const float* add1, const float* add2, float* dx, float* dgamma, float* dbeta,
                         void* stream);
int mural_op_conv_wgrad(const float* dy, const float* x, int64_t B, int32_t C, int32_t L, int32_t K,
                        const float* scale, const float* shift, int32_t pre_relu, float* dW, float* db,
                        float* part, size_t part_floats, void* stream);
/* fp32-MFMA path of the 32->32 k=3 convs on [B][32][L] tensors (L + 2 <= 288): forward / input gradient and weight +
 * bias gradient.  W: PyTorch [32][32][3]; part: mural_op_conv32_wgrad_scratch() floats.  stat_mode 1 / 2: the epilogue
 * also accumulates per-channel sums of the output into the accumulator block stat_out: 1 = sum / sum of
 * squares of act(y) (batch statistics for the next BatchNorm), 2 = sum(y), sum(y * xhat(stat_x)) (BatchNorm backward).  */
int mural_op_conv32_supported(int32_t L);
int mural_op_conv32(const float* x, const float* W, const float* bias, float* y, int64_t B, int32_t L, int32_t dgrad,
                    const float* pre_s, const float* pre_t, int32_t pre_relu, int32_t post_relu, const float* res1,
                    const float* res2, int32_t stat_mode, int32_t stat_relu, const float* stat_x, const float* stat_mean,
                    const float* stat_invstd, double* stat_out, void* stream);
size_t mural_op_conv32_wgrad_scratch(void);
int mural_op_conv32_wgrad(const float* dy, const float* x, int64_t B, int32_t L, const float* pre_s, const float* pre_t,
                          int32_t pre_relu, float* dW, float* db, float* part, size_t part_floats, void* stream);
/* Whole backward of one BN -> conv32 layer in one pass over dy: dW, db, dz = dL/d(conv input) and the BatchNorm-backward
 * sums of dz (accumulator block stat_out, zeroed by the caller; feed it to mural_op_bn_backward with have_sums = 1).      */
int mural_op_conv32_bwd(const float* dy, const float* x, const float* W, int64_t B, int32_t L, const float* pre_s,
                        const float* pre_t, int32_t pre_relu, const float* mean, const float* invstd, float* dW, float* db,
                        float* dz, double* stat_out, float* part, size_t part_floats, void* stream);
/* One call per BN -> conv32 layer and direction (composition of the kernels above).  state: float[4][32] = scale | shift |
 * mean | invstd of this call's batch statistics, written by the forward and read by the backward.                          */
int mural_op_bnconv32_fwd(const float* x, int64_t B, int32_t L, int32_t pre_relu, double* acc, int32_t have_acc,
                          const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                          float* running_var, float* state, const float* W, const float* bias, int32_t post_relu,
                          const float* res1, const float* res2, double* acc_out, int32_t out_relu, float* y, void* stream);
int mural_op_bnconv32_bwd(const float* dy, const float* x, int64_t B, int32_t L, int32_t pre_relu, const float* state,
                          const float* gamma, const float* W, double* acc, float* part, size_t part_floats, float* dz,
                          const float* add1, const float* add2, float* dW, float* db, float* dx, float* dgamma, float* dbeta,
                          void* stream);
int mural_op_maxpool_fwd(const float* x, int64_t rows, int32_t L, int32_t k, int32_t s, int32_t p, float* y,
                         int32_t* arg /* argmax positions for the backward; may be NULL (inference) */, void* stream);
int mural_op_maxpool_bwd_needs_zero(int32_t k, int32_t s);   /* 1: dx must be zeroed by the caller (overlapping windows) */
int mural_op_maxpool_bwd(const float* dy, const int32_t* arg, int64_t rows, int32_t L, int32_t Lout, int32_t k, int32_t s,
                         int32_t p, float* dx, void* stream);
/* First layer of a tower in training mode: maxpool1(Conv1d(BN(one-hot))) from window symbols (model_snv.py:473-475,
 * 496-497 under training.py:424), BN(4) batch statistics from the symbol histogram.  mural_op_first_plan gives the sizes of
 * the caller-allocated buffers: tab (floats), arg (bytes per pooled output) and the backward scratch (floats).       */
int mural_op_first_plan(int32_t C, int32_t pk, int64_t* tab_floats, int64_t* arg_bytes, int64_t* scratch_floats);
int mural_op_first_fwd(const uint8_t* sym, int64_t B, int32_t Lwin, int32_t col0, int32_t L1, int32_t C, int32_t pk,
                       int32_t ps, int32_t pp, const float* gamma, const float* beta, const float* W,
                       const float* bias, float eps, float momentum, float* running_mean, float* running_var,
                       unsigned long long* counts, float* tab, float* y, void* arg, void* stream);
int mural_op_first_bwd(const float* dy, const void* arg, const uint8_t* sym, int64_t B, int32_t Lwin,
                       int32_t col0, int32_t L1, int32_t C, int32_t pk, int32_t ps, int32_t pp, const float* tab,
                       const float* W, float* scratch, float* dW, float* dbias, float* dgamma, float* dbeta, void* stream);
int mural_op_linear_fwd(const float* x, const float* W, const float* b, int64_t B, int32_t I, int32_t O, float* y,
                        void* stream);
/* dx (optional), dW [O][I] and db [O] are fully written */
int mural_op_linear_bwd(const float* dy, const float* x, const float* W, int64_t B, int32_t I, int32_t O, float* dx,
                        float* dW, float* db, void* stream);
int mural_op_embedding_fwd(const int64_t* cat, const float* E, int64_t B, int32_t cols, int32_t rows, float* y,
                           void* stream);
int mural_op_embedding_bwd(const int64_t* cat, const float* dy, int64_t B, int32_t cols, int32_t rows, float* dE,
                           void* stream);
int mural_op_dropout(const float* x, int64_t total, float p, uint64_t seed, const uint64_t* seed_dev, float* y,
                     void* stream);
int mural_op_relu_mask(const float* g, const float* ref, int64_t total, float* y, void* stream);
int mural_op_head_fwd(const float* loc, const float* mid, const float* lar, int64_t B, int32_t nc, float* out,
                      void* stream);
int mural_op_head_bwd(const float* loc, const float* mid, const float* lar, const float* dout, int64_t B, int32_t nc,
                      float* dloc, float* dmid, float* dlar, void* stream);
int mural_op_dense_to_symbols(const float* x, int64_t n, int32_t L, uint8_t* sym, int32_t* status, void* stream);
/* Its HOST twin for loaders that yield host tensors (the reference's predict loop copies every fp32 window to the device batch by
 * batch, MuRaL/model/nn_utils.py:52-56): xs[b] = HOST float [rows[b]][4][L]; sym = HOST uint8 [sum rows][L], 255 for a column that is
 * no MuRaL encoding (*n_bad counts them).  Host threads; no device work.                                                          */
int mural_host_dense_to_symbols(const float* const* xs, const int64_t* rows, int64_t n_batches, int32_t L, uint8_t* sym,
                                int64_t* n_bad);
/* the small fields (y, cat_x) of the same host batches copied side by side into one (pinned) buffer: bytes[b] bytes from srcs[b]     */
int mural_host_concat(const void* const* srcs, const int64_t* bytes, int64_t n, void* dst);

/* ---------------------------------------------------------------------------------------------------------------
 * Post-head calibration of the prediction path in one pass over the (n, n_class) rows (run_predict.py:214-225):
 * F.softmax of the model output (skipped when in_is_prob) -> full-Dirichlet map softmax(W . [log clip(p); 1]) when
 * dirichlet_w != NULL (dev double [n_class][n_class + 1]; dirichletcal/calib/fulldirichlet.py:78-80, multinomial.py:60-64) ->
 * poisson_calibrate (MuRaL/model/calibration.py:10-23) when poisson != 0 -> apply_scaling (MuRaL/scripts/scaling.py:10-28)
 * when scale != 0.  out: dev double or float [n][n_class].  n_class <= 16.
 * ------------------------------------------------------------------------------------------------------------- */
int mural_calibrate_rows(const float* in, int64_t n, int32_t n_class, int32_t in_is_prob, const double* dirichlet_w,
                         int32_t poisson, double scale, void* out, int32_t out_f64, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * The SNV training step in one call per direction (MuRaL/training.py:424-427: preds = model.forward(...) under model.train(),
 * loss.backward()), Network0 / 1 / 2 with CNN_out_channels = 32, CNN_kernel_size = 3.  `params`: DEVICE pointers in the
 * state_dict naming of MuralSnvParams (running_mean / running_var are updated in place like nn.BatchNorm1d with `momentum`;
 * num_batches_tracked is the caller's).  dropout_p[5] / seeds[5]: emb_dropout, the two local_dropout layers, distal_fc_dropout
 * of the mid and of the large tower, with the seed of each mask (seed_dev: optional device-resident counter added to every
 * seed).  Input: cat_x (dev int64 [B][local_cols]) and either distal_x (dev float [B][4][distal_len]; `status` as in
 * mural_snv_forward_dense) or `symbols` (dev uint8 [B][distal_len] from mural_op_dense_to_symbols).  out: dev float [B][n_class]
 * (log-probabilities; raw logits for Network0).  The workspace (mural_snv_train_workspace_bytes) carries every saved tensor
 * from the forward to the backward of the same batch and must not be touched in between.
 * mural_snv_train_backward: `grads` has the layout of `params`; every weight / bias / emb pointer receives the gradient of
 * that tensor for d(loss)/d(out) = dout (fully written, no accumulation); running_* pointers are ignored.
 * ------------------------------------------------------------------------------------------------------------- */
size_t mural_snv_train_workspace_bytes(const MuralSnvShape* shape, int64_t B);
int mural_snv_train_forward(const MuralSnvShape* shape, const MuralSnvParams* params, const int64_t* cat_x, const float* distal_x,
                            const uint8_t* symbols, int64_t B, const float* dropout_p, const uint64_t* seeds,
                            const uint64_t* seed_dev, float momentum, float* out, void* workspace, size_t workspace_bytes,
                            int32_t* status, void* stream);
int mural_snv_train_backward(const MuralSnvShape* shape, const MuralSnvParams* params, const MuralSnvParams* grads,
                             const int64_t* cat_x, const float* dout, int64_t B, const float* dropout_p, const uint64_t* seeds,
                             const uint64_t* seed_dev, void* workspace, size_t workspace_bytes, void* stream);

/* The criterion of the reference's training loops, nn.CrossEntropyLoss(reduction='sum') on the model output x [B][nc] with labels y
 * (MuRaL/training.py:327, :425), in one launch per direction: loss[0] = -sum_i (x[i][y_i] - logsumexp(x[i])) summed in a fixed order
 * (reproducible), prob = softmax(x) kept for the backward; dx = g[0] (prob - onehot(y)).  A label outside [0, nc) makes the loss NaN. */
int mural_op_ce_sum_fwd(const float* x, const int64_t* y, int64_t B, int32_t nc, float* prob, float* loss, void* stream);
int mural_op_ce_sum_bwd(const float* prob, const int64_t* y, const float* g, int64_t B, int32_t nc, float* dx, void* stream);
/* torch.nn.utils.clip_grad_norm_(parameters, max_norm) of training.py:430 over one flat float32 gradient buffer of n elements (16-byte
 * aligned; zero in the padding between the parameters' slots): total[0] = 2-norm, flat *= min(max_norm / (norm + 1e-6), 1).  scratch64:
 * 64 doubles of device scratch.  Two launches, fixed summation order (bitwise reproducible).                                        */
int mural_op_clip_grad_norm(float* flat, int64_t n, float max_norm, double* scratch64, float* total, void* stream);

/* torch.optim.Adam.step() of the reference's training loops (MuRaL/training.py:346-350, :432; amsgrad and maximize off) over flat float32
 * buffers that share ONE slot layout -- parameters, gradients, exp_avg, exp_avg_sq of every parameter at the same offsets, zero in the
 * padding (it stays zero) -- as one elementwise launch; n = elements (a multiple of 4, buffers 16-byte aligned), step = the 1-based
 * count of this update (bias corrections 1 - beta^step are formed on the host in double).                                             */
int mural_op_adam_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                       double beta2, double eps, double weight_decay, int64_t step, void* stream);

/* Long windows (distal_radius from ~15000 up to the first-stage kernel's LDS limit) whose second conv stage fits no fused kernel: the
 * handle is FRONT-ONLY (mural_snv_tap_layout()[10] == 1; the forward entries above refuse it) and this entry runs what still is fused --
 * window decode + first layer + pool (model_snv.py:473-476 of the large tower) and its first ResBlock stage on halo'd segments of the
 * pooled row, 86 % of the model's arithmetic.  s3_out: dev float [n][L3][32] = the large tower's pooled second-stage input (channel-
 * last), L3 = mural_snv_tap_layout()[3]; the caller finishes per layer (mural_amd/model/generic_eval.py).  Serves every long-window
 * model (distal_radius >= 2000).                                                                                                    */
int mural_snv_forward_front(const MuralSnvModel* m, const MuralGenome* genome, const int64_t* pos, const uint8_t* strand, int64_t n,
                            int32_t local_radius, int32_t local_order, float* s3_out, void* workspace, size_t workspace_bytes, void* stream);
/* ... and its second half where the mid tower and the head still run fused (mural_snv_tap_layout()[12] == 1; then at most
 * mural_snv_tap_layout()[11] sites per front / finish pair, both on THE SAME workspace: the front call leaves the local branch's logits
 * (hence its local_radius / local_order) and the mid tower's pooled row there): large_logits dev float [n][n_class] = the large
 * tower's fc output computed by the caller from s3_out (model_snv.py:482-495) -> out dev float [n][n_class] log-probabilities.      */
int mural_snv_forward_finish(const MuralSnvModel* m, const float* large_logits, int64_t n, float* out, void* workspace,
                             size_t workspace_bytes, void* stream);

/* name of the dominant kernel (the fused tower kernel), for bench.py's roofline report */
const char* mural_snv_kernel_name(void);

/* Live timing of the dominant kernel: between begin and end every launch of it is bracketed by HIP events
 * on the launch stream; end() synchronises those events and returns the summed duration and the count.  */
int mural_profile_begin(void);
int mural_profile_end(double* total_ms, int64_t* launches);

/* ---------------------------------------------------------------------------------------------------------------
 * Host-side ingest (no device work): FASTA -> packed genome, BED -> site arrays, bed_reader row order.
 * Replaces SeqIO.to_dict(SeqIO.parse(...)) (MuRaL/data/preprocessing.py:836), bed_reader (:39-106) and the base maps of
 * the per-character encoders (:655-666, :762-772) for the packed-genome path.
 * ------------------------------------------------------------------------------------------------------------- */
/* names: n_cap x name_cap chars (record id = text after '>' up to the first whitespace); lengths / offsets per record
 * (offset = byte offset of the first sequence line).  n_records = records in the file (may exceed n_cap).           */
int mural_fasta_scan(const char* path, int64_t n_cap, int32_t name_cap, char* names, int64_t* lengths,
                     int64_t* offsets, int64_t* n_records);
/* Pack one record into the MuralGenome format; amb_pos / amb_sym receive the positions and MURAL_SYM_* symbols of IUPAC
 * codes other than N (up to amb_cap; n_amb counts all of them); any non-nucleotide character is MURAL_E_INVALID (the
 * reference raises KeyError).                                                                                        */
int mural_fasta_pack(const char* path, int64_t offset, int64_t length, uint32_t* packed2, uint32_t* nmask,
                     int64_t* amb_pos, uint8_t* amb_sym, int64_t amb_cap, int64_t* n_amb);
/* Six-column BED (chrom start end name score strand); chrom_id indexes chrom_names (order of first appearance);
 * strand: 0 '+', 1 '-'; score = class label.  cap = 0 counts rows / chromosomes.                                    */
int mural_bed_read(const char* path, int64_t cap, int32_t* chrom_id, int64_t* start, int64_t* end, float* score,
                   uint8_t* strand, int32_t n_chrom_cap, int32_t name_cap, char* chrom_names, int64_t* n_rows,
                   int32_t* n_chroms);
/* Rank-local streaming ingest for N-rank prediction -- replaces the reference's "one BedTool per process, split big inputs by hand"
 * (MuRaL/scripts/run_predict.py:107, MuRaL/commands/predict.py:134-137).  mural_bed_index_scan lists the PIECES (<= piece_rows
 * consecutive rows of one chromosome: name, byte range, rows, start of the first row) among the lines that start in
 * [byte_lo, byte_hi); byte_lo = byte_hi = 0 only reports file_bytes (the inflated size of a gzip file).  n_pieces may exceed cap.
 * piece_order (may be NULL): four values per piece -- 1 if every row's (start, strand) is >= its predecessor's ('+' < '-'), i.e. the
 * piece already is in the order of the reference's output table (run_predict.py:227) | start of its last row | strand of its first
 * row | strand of its last row (0 '+', 1 '-', -1 no strand field).
 * mural_bed_parse_range parses rows skip_rows .. skip_rows + n_rows - 1 of the rows in [byte_lo, byte_hi) (all on `chrom`).
 * Every ingest entry point reads gzip files too (inflated once per process; pybedtools / gzip.open in the reference).            */
int mural_bed_index_scan(const char* path, int64_t byte_lo, int64_t byte_hi, int64_t piece_rows, int32_t name_cap, int64_t cap,
                         char* names, int64_t* piece_lo, int64_t* piece_hi, int64_t* piece_rows_out, int64_t* piece_first_start,
                         int64_t* piece_order, int64_t* n_pieces, int64_t* file_bytes);
int mural_bed_parse_range(const char* path, int64_t byte_lo, int64_t byte_hi, int64_t skip_rows, int64_t n_rows, const char* chrom,
                          int64_t* start, int64_t* end, float* score, uint8_t* strand);
/* order[k] = input row of output row k in bed_reader order (+ rows, then - rows of every central_bp-wide segment)    */
int mural_bed_segment_order(const int32_t* chrom_id, const int64_t* start, const uint8_t* strand, int64_t n,
                            int64_t central_bp, int64_t* order, int64_t* group, int64_t* n_groups);

/* ---------------------------------------------------------------------------------------------------------------
 * The prediction table (MuRaL/scripts/run_predict.py:230-238: sort_values(['chrom', 'start']) + to_csv(sep='\t',
 * float_format='%.4g', index=False)) formatted at kernel speed.  One text row per site:
 *     chrom \t start \t end \t strand \t mut_type \t prob0 ... \t prob{k-1} \n
 * integers as decimals, mut_type = int64(label) like numpy's astype, probabilities as Python's '%.4g' (byte-identical:
 * exact round-half-even on the binary value; NaN -> empty field = pandas' na_rep).  Rows are emitted in the order
 * perm[0..n) (NULL = input order): the caller sorts, the formatter writes.  The header line is the caller's.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
  const char* chrom_names;   /* HOST: n_chroms x name_stride chars, NUL-terminated (also for the device call)      */
  int32_t n_chroms, name_stride;
  const int32_t* chrom_id;   /* [n] index into chrom_names, NULL = every row is chromosome 0                        */
  const int64_t* start;      /* [n] */
  const int64_t* end;        /* [n] */
  const uint8_t* strand;     /* [n] 0 '+', 1 '-' */
  const float* label;        /* [n] BED score column (class label) */
  const void* prob;          /* [n][prob_stride] float or double, the first n_class entries of a row are written    */
  int32_t prob_f64, n_class;
  int64_t prob_stride;
  const int64_t* perm;       /* [n] output row i = input row perm[i]; NULL = identity                               */
  int64_t n;
  int32_t layout;            /* 0: the prediction table row above; 1: the six-column BED row bed_reader consumes
                                (chrom start end . label strand; prob is not read) -- synthetic inputs are written with it */
  int32_t reserved;
} MuralTsvRows;
/* upper bound of one formatted row in bytes (text buffers hold n * bound), -1 on a bad table                        */
int64_t mural_tsv_row_bound(const MuralTsvRows* t);
size_t mural_tsv_format_workspace_bytes(int64_t n);
/* column pointers, out, n_bytes (int64) and ws are DEVICE addresses; three launches on `stream`, no synchronisation:
 * the text lands in out[0 .. *n_bytes)                                                                             */
int mural_tsv_format_device(const MuralTsvRows* t, char* out, int64_t cap, int64_t* n_bytes, void* ws, size_t ws_bytes,
                            void* stream);
/* the same formatter on `threads` host threads (0 = up to 16) for tables in host memory                            */
int mural_tsv_format_host(const MuralTsvRows* t, char* out, int64_t cap, int64_t* n_bytes, int32_t threads);
/* '%.4g' of one value into out12 (NUL-terminated); returns the length                                              */
int mural_tsv_format_g4(double v, char* out12);
/* The reference's "different bases" check (MuRaL/data/preprocessing.py:479-484) on a gathered shard: rows is a device
 * (n, row_stride) float / double matrix whose column `col` holds the strand-complemented focal base, group the
 * non-decreasing bed_reader group id per row; bit 0 of *status (device int32) is set when two rows of one group differ. */
int mural_focal_group_check(const void* rows, int32_t rows_f64, int64_t row_stride, int64_t col, const int64_t* group,
                            int64_t n, int32_t* status, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Validation-epoch analytics (device-side segmented reductions into float64 tables; the correlation / Newton algebra
 * on the tables is host work).  Replaces the pandas group-bys and per-row loops of MuRaL/evaluation/evaluation.py:
 * freq_kmer_comp_multi (:48-67), corr_calc_sub (:124-193), Evaluator.evaluate_regional_score (:544-587), ECELoss /
 * ClasswiseECELoss / BrierScore / mean CE (:209-290, :340-358) and the loss / gradient / Hessian sums of the full-Dirichlet
 * fit (dirichlet_python/dirichletcal/calib/multinomial.py:153-172).  status: device int32, bit 0 = bad code / chrom /
 * start, bit 1 = label or key out of range (such rows are skipped).  Tables must be zeroed by the caller.
 * ------------------------------------------------------------------------------------------------------------- */
/* keys[i] = base-5 number of codes[i][left0 .. left0+d) ++ codes[i][right0 .. right0+d) (order-1 codes 0..4);
 * with region_size > 0: + (i / region_size) * 5^(2d), rows of regions >= n_regions get key -1                      */
int mural_eval_kmer_keys(const int64_t* codes, int64_t n, int32_t ncols, int32_t left0, int32_t right0, int32_t d,
                         int64_t region_size, int64_t n_regions, int32_t* keys, int32_t* status, void* stream);
/* keys[i] = chrom_base[chrom_id[i]] + start[i] / window                                                           */
int mural_eval_window_keys(const int32_t* chrom_id, const int64_t* start, int64_t n, int64_t window,
                           const int64_t* chrom_base, int32_t n_chrom, int32_t* keys, int32_t* status, void* stream);
/* table [n_groups][1 + 2 n_class] += { rows, rows with label c, sum of prob[:, c] } of the rows with key g >= 0     */
int mural_eval_group_obs_pred(const int32_t* keys, const int32_t* label, const void* prob, int32_t prob_f64, int64_t n,
                              int32_t n_class, int32_t n_groups, double* table, int32_t* status, void* stream);
/* out [2 + 3 n_bins (1 + n_class)] += { NLL sum, Brier sum, top-label bins [n_bins][3], class bins [n_class][n_bins][3] }
 * with bin b = (bounds[b], bounds[b+1]] and cells (rows, score sum, hits)                                           */
int mural_eval_calib_metrics(const void* prob, int32_t prob_f64, const int32_t* label, int64_t n, int32_t n_class,
                             int32_t n_bins, const float* bounds, double* out, int32_t* status, void* stream);
/* out [1 + km + (km)^2] (km = n_class (n_class + 1)) += { loss sum, gradient, Hessian (if need_hessian) } of the
 * multinomial regression on [log clip(prob); 1] at weights [n_class][n_class + 1] (float64, device); n_class <= 8    */
int mural_eval_dirichlet_fit_terms(const void* prob, int32_t prob_f64, const int32_t* label, int64_t n, int32_t n_class,
                                   const double* weights, int32_t need_hessian, double* out, int32_t* status, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Training-mode ops of the INDEL U-Net (MuRaL/model/model_indel.py:6-19, :151-176 under model.train()): a general
 * Conv1d (stride, zero padding, input upsampled by `up` = nn.Upsample(scale_factor) in front of the conv) with its
 * backward, and the element-wise activations.  x [B][Cin][Lin], W [Cout][Cin][K] (torch layout), y [B][Cout][Lout].
 * ------------------------------------------------------------------------------------------------------------- */
int mural_op_convg_out_length(int32_t Lin, int32_t K, int32_t stride, int32_t pad, int32_t up);   /* -1: bad geometry */
/* wt: scratch of Cout*Cin*K floats */
int mural_op_convg_fwd(const float* x, const float* W, const float* bias, float* wt, float* y, int64_t B, int32_t Cin,
                       int32_t Lin, int32_t Cout, int32_t K, int32_t stride, int32_t pad, int32_t up, void* stream);
size_t mural_op_convg_bwd_scratch(int32_t Cin, int32_t Cout, int32_t K);                          /* floats */
/* dx (optional), dW and db (optional) are fully written */
int mural_op_convg_bwd(const float* dy, const float* x, const float* W, int64_t B, int32_t Cin, int32_t Lin, int32_t Cout,
                       int32_t K, int32_t stride, int32_t pad, int32_t up, float* dx, float* dW, float* db, float* part,
                       size_t part_floats, void* stream);
/* The U-Net's unit in one call per direction: z = act(BatchNorm1d(Conv1d(upsample_up(x)))) [+ res1] [+ res2] in training mode
 * (reference MuRaL/model/model_indel.py:6-19, :117-123 under model.train(): nn.Conv1d -> nn.BatchNorm1d [-> nn.SiLU / nn.ReLU] and the
 * residual adds around them).  act: 0 none, 1 ReLU, 2 SiLU, 3 Softplus.  y0 [B][Cout][Lout] receives the conv output and state
 * [4][Cout] = scale | shift | mean | invstd (both saved for the backward); acc: zeroed [MURAL_BN_SLOTS][2][Cout] doubles; running
 * statistics updated in place like nn.BatchNorm1d (momentum, unbiased variance).  The backward takes dz (the residuals' gradients are
 * dz itself), a fresh zeroed acc, a scratch dy0 [B][Cout][Lout] and the part scratch of mural_op_convg_bwd. */
int mural_op_convg_bn_fwd(const float* x, const float* W, const float* bias, float* wt, float* y0, int64_t B, int32_t Cin,
                          int32_t Lin, int32_t Cout, int32_t K, int32_t stride, int32_t pad, int32_t up, const float* gamma,
                          const float* beta, float eps, float momentum, float* running_mean, float* running_var, double* acc,
                          float* state, int32_t act, const float* res1, const float* res2, float* z, void* stream);
int mural_op_convg_bn_bwd(const float* dz, const float* x, const float* W, const float* y0, const float* state, const float* gamma,
                          int64_t B, int32_t Cin, int32_t Lin, int32_t Cout, int32_t K, int32_t stride, int32_t pad, int32_t up,
                          int32_t act, double* acc, float* dy0, float* dx, float* dW, float* db, float* dgamma, float* dbeta,
                          float* part, size_t part_floats, const float* wt_dgrad, void* stream);
/* The weight layouts the conv kernels read -- forward [Cin][K][Cout], input gradient [Cout][K flipped][Cin] (stride-1 layers) -- for
 * every conv of a model in ONE launch per training step.  jobs: DEVICE array sorted by start (running sum of Cout * Cin * K).
 * mural_op_convg_fwd / _bn_fwd take W == NULL when wt already holds the forward layout; mural_op_convg_bn_bwd takes the prepared
 * input-gradient layout as wt_dgrad (NULL: derived from W inside the call). */
typedef struct MuralRelayoutJob {
  const float* W;        /* [Cout][Cin][K] (torch layout) */
  float* wt_fwd;         /* [Cin][K][Cout] */
  float* wt_dgrad;       /* [Cout][K][Cin], taps flipped; NULL: not needed */
  int32_t Cout, Cin, K, reserved;
  int64_t start;
} MuralRelayoutJob;
int mural_op_relayout_multi(const MuralRelayoutJob* jobs, int32_t n_jobs, int64_t total, void* stream);
/* The INDEL training step as one call per direction (reference: UNet_Small.forward under model.train(),
 * MuRaL/model/model_indel.py:151-176, inside the step of MuRaL/training.py:424-436).  params / grads: DEVICE pointers in the
 * state_dict naming (MuralIndelParams); the forward updates the BatchNorms' running statistics in place (the strand-symmetrising
 * layer's twice, in the reference's order) and leaves everything the backward needs in the workspace; the backward writes every
 * parameter gradient (fully, no accumulation).  x: dev float [B][4][length]; out / dout: dev float [B][n_class].  dropout_p, seed,
 * seed_dev: out_fc's Dropout (see mural_op_dropout).  mural_indel_train_workspace_bytes: 0 on a bad shape.                      */
size_t mural_indel_train_workspace_bytes(const MuralIndelShape* shape, int64_t B);
int mural_indel_train_forward(const MuralIndelShape* shape, const MuralIndelParams* params, const float* x, int64_t B,
                              float dropout_p, uint64_t seed, const uint64_t* seed_dev, float momentum, float* out,
                              void* workspace, size_t workspace_bytes, void* stream);
int mural_indel_train_backward(const MuralIndelShape* shape, const MuralIndelParams* params, const MuralIndelParams* grads,
                               const float* x, const float* dout, int64_t B, float dropout_p, uint64_t seed,
                               const uint64_t* seed_dev, void* workspace, size_t workspace_bytes, void* stream);
/* kind: 1 ReLU, 2 SiLU, 3 Softplus (beta 1, threshold 20); backward takes the forward INPUT x */
int mural_op_act_fwd(const float* x, int64_t n, int32_t kind, float* y, void* stream);
int mural_op_act_bwd(const float* dy, const float* x, int64_t n, int32_t kind, float* dx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MURAL_HIP_H */

// Eval-mode forward of the two SNV conv towers after their first layer + softmax-mixing head (gfx950 / CDNA4).
//
// Reference semantics: MuRaL/model/model_snv.py:477-523 (Network2.forward, tower part and head),
// :794-812 (ResBlock).  One workgroup (4 waves) carries a tile of P positions through EVERY remaining layer with
// the activations resident in LDS: it reads the pooled first-layer activations written by snv_stage1_kernel
// (25.7 KB per site at R=1000) and writes only the (n_class) log-probabilities.
//
//   stage 2-4 every 32->32 k=3 conv is an implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32):
//             D[cout 16][col 16] += W[cout][k] * act[k][col], K = 3 taps x 32 channels = 24 k-steps,
//             two M-blocks per 16-column block.  All positions of the tile share one flattened column axis
//             with a zero separator column between positions (that column IS the conv zero padding).
//             LDS image: [column][32 channels] with a conflict-free 16-byte-chunk XOR swizzle, read with ds_read_b128.
//             Weights live in 48 VGPRs per lane per layer (A fragments, fetched from L2 in fragment order).
//             The residual stream stays in registers in MFMA accumulator layout across a whole stage.
//   head      global max, BN, Linear(32->n_class) per tower, then log(clamp((sm(local)+(sm(mid)+sm(large))/2)/2)).
#include <cstdlib>
#include <vector>

#include "snv_tower_conv.h"

namespace mural {

// diagnostic: accumulate wave 0's cycles per phase into args.stamps[block][phase] (only when stamps != nullptr)
#define SNV_STAMP(id)                                                              \
  do {                                                                             \
    if (args.stamps != nullptr && tid == 0) {                                      \
      const unsigned long long _t = __builtin_amdgcn_s_memtime();                  \
      args.stamps[(size_t)blockIdx.x * 32 + (id)] += _t - t_prev;                  \
      t_prev = _t;                                                                 \
    }                                                                              \
  } while (0)

// Stage-1 activations of (tile, tower) straight into the residual registers, in MFMA accumulator layout
// (lane = column n16 of each owned block, 4 channels starting at chv); separator / padding columns read as 0.
__device__ __forceinline__ void request_x0(const SnvFwdArgs& args, f32x4 (&xres)[SNV_NB2MAX], int64_t tile, int tw_i,
                                           int64_t n_tiles, int cgp, int n16, int chv) {
  const TowerGeom& g = args.geom[tw_i];
  const int x0c = tw_i == 0 ? 0 : args.geom[0].L[0];
  const int nbw0 = g.nb[0] > cgp ? (g.nb[0] - cgp + 1) / 2 : 0;
  const float* base = args.x0 + (size_t)x0c * 32 + chv;
#pragma unroll
  for (int i = 0; i < SNV_NB2MAX; ++i) {
    xres[i] = splat(0.f);
    const int c = 16 * (cgp + 2 * i) + n16;
    if (i < nbw0 && c >= 1 && tile < n_tiles) {
      const uint32_t u = (uint32_t)(c - 1);
      const uint32_t p = g.dSc[0].div(u);
      const uint32_t j2 = u - p * (uint32_t)g.Sc[0];
      const int64_t row = tile * args.P + p;
      if (p < (uint32_t)args.P && j2 < (uint32_t)g.L[0] && row < args.n)
        xres[i] = ld4(base + ((size_t)row * args.x0_cols + j2) * 32);
    }
  }
}

// The same for a first-stage launch (one tower): the per-lane part of the address -- two divisions per block -- is computed
// once per launch.  plan[i] = float offset of the lane's row inside the tile (a multiple of 32) | position p in the low
// five bits, ~0u when the lane's column of block i holds no data.
struct X0Plan { uint32_t off[SNV_NB2MAX]; };

__device__ __forceinline__ X0Plan x0_plan(const SnvFwdArgs& args, int tw_i, int cgp, int n16) {
  const TowerGeom& g = args.geom[tw_i];
  const int nbw0 = g.nb[0] > cgp ? (g.nb[0] - cgp + 1) / 2 : 0;
  X0Plan pl;
#pragma unroll
  for (int i = 0; i < SNV_NB2MAX; ++i) {
    pl.off[i] = ~0u;
    const int c = 16 * (cgp + 2 * i) + n16;
    if (i < nbw0 && c >= 1) {
      const uint32_t u = (uint32_t)(c - 1);
      const uint32_t p = g.dSc[0].div(u);
      const uint32_t j2 = u - p * (uint32_t)g.Sc[0];
      if (p < (uint32_t)args.P && j2 < (uint32_t)g.L[0]) pl.off[i] = ((p * (uint32_t)args.x0_cols + j2) << 5) | p;
    }
  }
  return pl;
}

__device__ __forceinline__ void request_x0_planned(const SnvFwdArgs& args, const X0Plan& pl, f32x4 (&xres)[SNV_NB2MAX], int64_t tile,
                                                   int tw_i, int64_t n_tiles, int chv) {
  const int x0c = tw_i == 0 ? 0 : args.geom[0].L[0];
  const int64_t row0 = tile * args.P;
  const float* base = args.x0 + ((size_t)row0 * args.x0_cols + x0c) * 32 + chv;
  const bool whole = tile < n_tiles && row0 + args.P <= args.n;     // every position of the tile exists (all but the last tile)
#pragma unroll
  for (int i = 0; i < SNV_NB2MAX; ++i) {
    xres[i] = splat(0.f);
    const uint32_t o = pl.off[i];
    if (o != ~0u && (whole || (tile < n_tiles && row0 + (int64_t)(o & 31u) < args.n))) xres[i] = ld4(base + (o & ~31u));
  }
}

// PHASE = SnvFwdArgs::phase at compile time: the stage-split launches do not carry each other's code and registers
template <int PHASE>
__global__ __launch_bounds__(SNV_THREADS, 2) void snv_towers_fused(const SnvFwdArgs args) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  unsigned long long t_prev = args.stamps != nullptr ? __builtin_amdgcn_s_memtime() : 0ull;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mb = wave & 1, cgp = wave >> 1;
  const int n16 = lane & 15, kk = lane >> 4;
  const int P = args.P;
  float* bufA = smem;
  float* bufB = smem + args.nbuf;
  float* feat = smem + 2 * args.nbuf;          // [2][P][32] global max per tower
  float* logit = feat + 2 * P * SNV_C;         // [3][P][SNV_MAXCLASS]: large, mid, local
  float* par = logit + 3 * P * SNV_MAXCLASS;   // per tower: ex_s[4][32] | ex_t[4][32] | fc_w[nc][32] | fc_b[nc..]
  const int par_stride = 2 * EX_COUNT * SNV_C + args.n_class * SNV_C + SNV_MAXCLASS;
  // Stage-1 activations of (tile, tower) are requested straight into the residual registers, in MFMA accumulator layout (below).  The
  // tower-parallel small-call launch asks for them HERE, in front of the parameter staging and its barrier: a 16-site call is one tile
  // per workgroup, and its 7 us of start-up were three memory round trips in a row (parameters, barrier, activations).
  f32x4 xres[SNV_NB2MAX];
  if constexpr (PHASE == 3)
    request_x0(args, xres, (int64_t)(blockIdx.x >> 1), (int)(blockIdx.x & 1u), (args.n + P - 1) / P, cgp, n16, 16 * mb + 4 * kk);
  for (int t2 = 0; t2 < 2; ++t2) {             // small per-channel parameters: resident for the whole launch
    float* d = par + t2 * par_stride;
    const TowerDev& tw = args.tw[t2];
    for (int i = tid; i < EX_COUNT * SNV_C; i += SNV_THREADS) {
      d[i] = tw.ex_s[i];
      d[EX_COUNT * SNV_C + i] = tw.ex_t[i];
    }
    for (int i = tid; i < args.n_class * SNV_C; i += SNV_THREADS) d[2 * EX_COUNT * SNV_C + i] = tw.fc_w[i];
    if (tid < args.n_class) d[2 * EX_COUNT * SNV_C + args.n_class * SNV_C + tid] = tw.fc_b[tid];
  }
  __syncthreads();

  const int chv = 16 * mb + 4 * kk;  // first of this lane's 4 output channels
  const int64_t n_tiles = (args.n + P - 1) / P;
  // Tower-parallel launch (small batches): workgroup 2t runs the large tower of tile t, workgroup 2t + 1 its mid tower, and
  // whichever of the two finishes last runs the head -- the latency of a call is ONE tower, not the sum of both.
  // (PHASE 3 = the stages of PHASE 0 in this launch shape.  The other phases must compile to exactly what they were: the
  // first-stage kernel sits at 254 of 256 VGPRs, and naming these values in locals was enough to make it spill 84 of them.)
  constexpr bool TPAR = PHASE == 3;
#define TW_FIRST (TPAR ? (int)(blockIdx.x & 1u) : args.tw_first)
#define TW_LAST (TPAR ? (int)(blockIdx.x & 1u) : args.tw_last)
#define TILE0 (TPAR ? (int64_t)(blockIdx.x >> 1) : (int64_t)blockIdx.x)
#define TILE_STEP (TPAR ? (gridDim.x >> 1) : gridDim.x)

  // ... one tower ahead: the HBM/L2 latency hides under the previous tower's global max / head.
  X0Plan xplan;
  if (PHASE == 1) {
    xplan = x0_plan(args, TW_FIRST, cgp, n16);
    request_x0_planned(args, xplan, xres, TILE0, TW_FIRST, n_tiles, chv);
  } else if (PHASE != 2 && PHASE != 3) {
    request_x0(args, xres, TILE0, TW_FIRST, n_tiles, cgp, n16, chv);
  }
  const bool do_head = args.tw_last == 1 && PHASE != 1;
  // a first-stage launch runs one tower and one stage: its lane addressing / validity mask is tile-invariant
  const StageAddr sa_first = stage_setup(args.geom[TW_FIRST], 0, P, n16, kk, mb, cgp);
  // a short-stage launch runs one tower through stages 1 and 2: both lane maps are tile-invariant as well
  StageAddr sa_s1 = sa_first, sa_s2 = sa_first;
  if (PHASE == 2) {
    sa_s1 = stage_setup(args.geom[TW_FIRST], 1, P, n16, kk, mb, cgp);
    sa_s2 = stage_setup(args.geom[TW_FIRST], 2, P, n16, kk, mb, cgp);
  }

  float a_cur[SNV_KSTEPS];
  for (int64_t tile = TILE0; tile < n_tiles; tile += TILE_STEP) {
    const int64_t row0 = tile * P;
    if (do_head && !TPAR && tid < P * args.n_class) {   // local-branch logits of this tile -> LDS (third logit vector)
      const int p = tid / args.n_class, k = tid - p * args.n_class;
      float v = 0.f;
      if (args.has_local && row0 + p < args.n) v = args.local_logits[(row0 + p) * args.n_class + k];
      logit[(2 * P + p) * SNV_MAXCLASS + k] = v;
      if (args.tw_first == 1) {               // split mode: the large tower's logits come from the previous launch
        float u = 0.f;
        if (row0 + p < args.n) u = args.xlogit[(row0 + p) * SNV_MAXCLASS + k];
        logit[p * SNV_MAXCLASS + k] = u;
      }
    }

    for (int tw_i = TW_FIRST; tw_i <= TW_LAST; ++tw_i) {
      const TowerGeom& g = args.geom[tw_i];
      const TowerDev& tw = args.tw[tw_i];
      const float* tpar = par + tw_i * par_stride;

      // weights of the first conv: issued now, consumed after the entry barrier.  A stage-split launch runs one tower, so
      // from its second tile on they are already there: the last layer of a tile prefetches the first layer's fragments.
      if (PHASE == 0 || PHASE == 3 || tile == TILE0) {
        const float* wf = tw.wfrag + (size_t)(PHASE == 2 ? 4 : 0) * SNV_WFRAG + (size_t)mb * SNV_KSTEPS * 64 + lane;
#pragma unroll
        for (int s = 0; s < SNV_KSTEPS; ++s) a_cur[s] = wf[s * 64];
      }

      // -------------------------------------------------------------- entry: BN(ReLU(x0)) -> bufA, x0 stays in xres
      StageAddr sa = (PHASE == 1 || tw_i == TW_FIRST) ? sa_first : stage_setup(g, 0, P, n16, kk, mb, cgp);
      if (PHASE != 2) {
        const f32x4 es = ld4(tpar + EX_RB1_ENTRY * 32 + chv), et = ld4(tpar + (EX_COUNT + EX_RB1_ENTRY) * 32 + chv);
        char* A = reinterpret_cast<char*>(bufA);
        const int nbw0 = g.nb[0] > cgp ? (g.nb[0] - cgp + 1) / 2 : 0;
#pragma unroll
        for (int i = 0; i < SNV_NB2MAX; ++i)
          if (i < nbw0) lds_st4(A, sa.wr + 4096u * i, ((sa.vmask >> i) & 1u) ? relu_bn(xres[i], es, et) : splat(0.f));
      }
      __syncthreads();
      SNV_STAMP(3 + 12 * tw_i);   // stage-1 activations landed + BN/ReLU image written

      // -------------------------------------------------------------- the ten 32->32 convs
      for (int layer = PHASE == 2 ? 4 : 0; layer < SNV_NLAYER; ++layer) {
        const int st = layer < 4 ? 0 : (layer < 9 ? 1 : 2);
        if (layer == 4 && PHASE == 2) {
          // second conv stage of a stage-split launch: its input was pooled by the phase-1 launch (s3, layout of x0)
#pragma unroll
          for (int i = 0; i < SNV_NB2MAX; ++i) xres[i] = splat(0.f);
          const int Lout = g.L[1], ScO = g.Sc[1];
          const int total = P * Lout * 8;
          const float* src = args.s3[tw_i] + (size_t)row0 * Lout * 32;
          for (int task = tid; task < total; task += SNV_THREADS) {
            const uint32_t pj = (uint32_t)task >> 3;
            const uint32_t p = g.dL[1].div(pj);
            const int jo = (int)(pj - p * (uint32_t)Lout);
            const f32x4 v = (row0 + p < args.n) ? ld4(src + (size_t)task * 4) : splat(0.f);
            st4(bufB + lds_off(1 + (int)p * ScO + jo + 1, task & 7), v);
          }
          const int nz = 1 + P + (16 * g.nb[1] - g.NC[1]);
          for (int task = tid; task < nz * 8; task += SNV_THREADS) {
            const int k = task >> 3;
            const int c = (k <= P) ? k * ScO : g.NC[1] + (k - P - 1);
            st4(bufB + lds_off(c + 1, task & 7), splat(0.f));
          }
          sa = sa_s1;
          __syncthreads();
        } else if (layer == 4 || layer == 9) {
          // a first-stage launch ends with this pooling: the residual registers are dead, so the next tile's stage-1
          // activations are requested now and their HBM latency hides under the pooling instead of stalling the next entry
          if (PHASE == 1) request_x0_planned(args, xplan, xres, tile + TILE_STEP, tw_i, n_tiles, chv);
          // max-pool (raw y in bufA) + BN (no ReLU) -> bufB in the next stage's geometry
          const int si = st - 1;  // input stage
          const int Lin = g.L[si], Lout = g.L[st], ScI = g.Sc[si], ScO = g.Sc[st];
          const int pk = g.pk[st], ps = g.ps[st], pp = g.pp[st];
          const int total = P * Lout * 8;
          const int cg = tid & 7;
          const int ex = layer == 4 ? EX_BN_MID : EX_BN_OUT;
          const f32x4 pool_s = ld4(tpar + ex * 32 + 4 * cg), pool_t = ld4(tpar + (EX_COUNT + ex) * 32 + 4 * cg);
          for (int task = tid; task < total; task += SNV_THREADS) {
            const uint32_t pj = (uint32_t)task >> 3;
            const uint32_t p = g.dL[st].div(pj);
            const int jo = (int)(pj - p * (uint32_t)Lout);
            const int jlo = jo * ps - pp;
            // window columns clamped into [lo, hi] = the in-range part of the window: a clamped read repeats a column that
            // belongs to the window, so the maximum is unchanged and no -inf masking is needed (every window of the model
            // overlaps its row: padding < kernel)
            const int lo = jlo < 0 ? 0 : jlo;
            const int hi = (jlo + pk - 1) < (Lin - 1) ? (jlo + pk - 1) : (Lin - 1);
            f32x4 v[7];
#pragma unroll
            for (int w = 0; w < 7; ++w) {   // the model's pools are 7- and 3-wide: all reads in flight together
              int j = jlo + w;
              j = j < lo ? lo : (j > hi ? hi : j);
              v[w] = ld4(bufA + lds_off(1 + (int)p * ScI + j + 1, cg));
            }
            f32x4 m = max4(max4(max4(v[0], v[1]), max4(v[2], v[3])), max4(max4(v[4], v[5]), v[6]));
            for (int w = 7; w < pk; ++w) {
              const int j = jlo + w;
              if (j < 0 || j >= Lin) continue;
              m = max4(m, ld4(bufA + lds_off(1 + (int)p * ScI + j + 1, cg)));
            }
            m = f32x4{fmaf(pool_s.x, m.x, pool_t.x), fmaf(pool_s.y, m.y, pool_t.y), fmaf(pool_s.z, m.z, pool_t.z),
                      fmaf(pool_s.w, m.w, pool_t.w)};
            if (PHASE == 1) {      // hand the pooled tile to the phase-2 launch: s3[row][jo][32], 128 bytes per 8 lanes
              if (row0 + p < args.n) st4(args.s3[tw_i] + ((size_t)(row0 + p) * Lout + jo) * 32 + 4 * cg, m);
            } else {
              st4(bufB + lds_off(1 + (int)p * ScO + jo + 1, cg), m);
            }
          }
          if (PHASE == 1) break;   // the first conv stage is done
          {
            const int nz = 1 + P + (16 * g.nb[st] - g.NC[st]);
            for (int task = tid; task < nz * 8; task += SNV_THREADS) {
              const int k = task >> 3;
              const int c = (k <= P) ? k * ScO : g.NC[st] + (k - P - 1);
              st4(bufB + lds_off(c + 1, task & 7), splat(0.f));
            }
          }
          sa = PHASE == 2 ? sa_s2 : stage_setup(g, st, P, n16, kk, mb, cgp);
          __syncthreads();
          SNV_STAMP((layer == 4 ? 5 : 7) + 12 * tw_i);   // max-pool 2 / 3
          if (args.taps != nullptr && tile == 0) {
            float* dst = args.taps + (size_t)(tw_i * 6 + (layer == 4 ? 2 : 4)) * args.tap_stride;
            for (int i = tid; i < args.nbuf; i += SNV_THREADS) dst[i] = bufB[i];
          }
        }

        const LayerK lk = layer_consts(layer_mode(layer));
        const bool in_is_a = ((0xA5u >> layer) & 1u) != 0;
        const char* in = reinterpret_cast<const char*>(in_is_a ? bufA : bufB);
        char* out = reinterpret_cast<char*>(in_is_a ? bufB : bufA);
        const int nb = g.nb[st];
        const int nbw = nb > cgp ? (nb - cgp + 1) / 2 : 0;   // blocks cgp, cgp+2, ... < nb

        // affine maps of THIS layer (needed only in the epilogues, a full MFMA group later) are requested BEFORE the
        // prefetch so that their counted vmcnt wait does not drain the prefetch
        const f32x4 pb = ld4(tw.bias + layer * 32 + chv), ps = ld4(tw.post_s + layer * 32 + chv),
                    pt = ld4(tw.post_t + layer * 32 + chv);
        __builtin_amdgcn_sched_barrier(0);
        // prefetch the next layer's A fragments (consumed after this layer's barrier)
        float a_nxt[SNV_KSTEPS];
        const int ln = PHASE == 1 ? (layer < 3 ? layer + 1 : 0) : (layer < SNV_NLAYER - 1 ? layer + 1 : (PHASE == 2 ? 4 : layer));
        {
          const float* wfn = tw.wfrag + (size_t)ln * SNV_WFRAG + (size_t)mb * SNV_KSTEPS * 64 + lane;
#pragma unroll
          for (int s = 0; s < SNV_KSTEPS; ++s) a_nxt[s] = wfn[s * 64];
        }
        conv_layer(in, out, sa, nbw, lk, a_cur, pb, ps, pt, xres);
        SNV_STAMP(9 + 12 * tw_i);   // conv phases of wave 0 (all layers)
        __syncthreads();
        SNV_STAMP((layer < 4 ? 4 : (layer < 9 ? 6 : 8)) + 12 * tw_i);   // barrier wait after the convs of stage 2 / 3 / 4
#pragma unroll
        for (int s = 0; s < SNV_KSTEPS; ++s) a_cur[s] = a_nxt[s];
        if (args.taps != nullptr && tile == 0 && (layer == 3 || layer == 8 || layer == 9)) {
          float* dst = args.taps + (size_t)(tw_i * 6 + (layer == 3 ? 1 : (layer == 8 ? 3 : 5))) * args.tap_stride;
          const float* o = reinterpret_cast<const float*>(out);
          for (int i = tid; i < args.nbuf; i += SNV_THREADS) dst[i] = o[i];
        }
      }

      if (PHASE == 1) {   // stage-split launch: nothing after the pooling; bufA is free once every wave has pooled
        lds_barrier();
        continue;
      }
      // -------------------------------------------------------------- global max per (position, channel)
      // (the residual registers are dead after the last conv: request the next tower's stage-1 activations now)
      if (PHASE != 2)
        request_x0(args, xres, tw_i < TW_LAST ? tile : tile + TILE_STEP, tw_i < TW_LAST ? tw_i + 1 : TW_FIRST, n_tiles,
                   cgp, n16, chv);
      {
        const int L4 = g.L[2], Sc4 = g.Sc[2];
        float* ft = feat + tw_i * P * SNV_C;
        for (int t = tid; t < P * SNV_C; t += SNV_THREADS) {
          const int p = t >> 5, ch = t & 31;
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {   // all reads in flight together (L4 is 7 / 8 at R = 1000); short rows repeat their last column
            const int pc = 1 + p * Sc4 + (j < L4 ? j : L4 - 1) + 1;
            v[j] = bufA[lds_off(pc, ch >> 2) + (ch & 3)];
          }
          float m = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
          for (int j = 8; j < L4; ++j) m = fmaxf(m, bufA[lds_off(1 + p * Sc4 + j + 1, ch >> 2) + (ch & 3)]);
          ft[t] = m;
        }
        lds_barrier();   // bufA may be overwritten by the next tower's entry step; feat visible to the fc
        SNV_STAMP(10 + 12 * tw_i);   // global max
      }
    }  // towers

    if (PHASE == 1) continue;
    // ------------------------------------------------------------------ BN+Linear per tower (BN folded on the host)
    for (int t = tid; t < (TW_LAST - TW_FIRST + 1) * P * args.n_class; t += SNV_THREADS) {
      const int k = t % args.n_class;
      const int tp = TW_FIRST * P + t / args.n_class;  // tower * P + p
      const int tw_i = tp / P;
      const float* w = par + tw_i * par_stride + 2 * EX_COUNT * SNV_C + k * SNV_C;
      const float* f = feat + tp * SNV_C;
      float acc = par[tw_i * par_stride + 2 * EX_COUNT * SNV_C + args.n_class * SNV_C + k];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x4 wv = ld4(w + 4 * q), fv = ld4(f + 4 * q);
        acc = fmaf(wv.x, fv.x, acc);
        acc = fmaf(wv.y, fv.y, acc);
        acc = fmaf(wv.z, fv.z, acc);
        acc = fmaf(wv.w, fv.w, acc);
      }
      logit[tp * SNV_MAXCLASS + k] = acc;
    }
    lds_barrier();
    SNV_STAMP(25);   // fc

    if (TPAR) {
      // publish this tower's logits; the second workgroup of the tile to get here gathers both and runs the head
      float* mine = TW_FIRST == 0 ? args.xlogit : args.xlogit2;
      if (tid < P * args.n_class) {
        const int p = tid / args.n_class, k = tid - p * args.n_class;
        if (row0 + p < args.n) mine[(row0 + p) * SNV_MAXCLASS + k] = logit[(TW_FIRST * P + p) * SNV_MAXCLASS + k];
      }
      // (one release by the thread that counts, behind the barrier that collects every wave's stores, and one acquire by the same thread:
      // the workgroup's other threads read the pair's logits with device-scope atomic loads.  A fence by all 256 threads on either side --
      // each an L2 write-back / invalidate of the XCD, the pair's workgroups never share one -- cost the last arriver 8.5 us.)
      __syncthreads();
      float* flag = feat + (1 - TW_FIRST) * P * SNV_C;   // the other tower's slot of feat is unused by this workgroup
      if (tid == 0) {
        __threadfence();
        const int old = atomicAdd(&args.tile_count[tile], 1);
        __threadfence();
        flag[0] = __int_as_float(old);
      }
      __syncthreads();
      const bool last = __float_as_int(flag[0]) != 0;
      __syncthreads();   // flag is read by everyone before a later tile's global max could overwrite the slot
      if (!last) continue;
      if (tid < P * args.n_class) {
        const int p = tid / args.n_class, k = tid - p * args.n_class;
        const bool in = row0 + p < args.n;
        const size_t gi = (size_t)(row0 + p) * SNV_MAXCLASS + k;
        logit[p * SNV_MAXCLASS + k] = in ? __hip_atomic_load(args.xlogit + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        logit[(P + p) * SNV_MAXCLASS + k] = in ? __hip_atomic_load(args.xlogit2 + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        logit[(2 * P + p) * SNV_MAXCLASS + k] = (in && args.has_local) ? args.local_logits[(row0 + p) * args.n_class + k] : 0.f;
      }
      lds_barrier();
    } else if (!do_head) {   // split mode, large tower: hand the logits to the launch that runs the mid tower and the head
      if (tid < P * args.n_class) {
        const int p = tid / args.n_class, k = tid - p * args.n_class;
        if (row0 + p < args.n) args.xlogit[(row0 + p) * SNV_MAXCLASS + k] = logit[p * SNV_MAXCLASS + k];
      }
      lds_barrier();
      continue;
    }
    // ------------------------------------------------------------------ head (model_snv.py:515-523 / :284)
    if (tid < P * args.n_class) {
      const int nc = args.n_class;
      const int p = tid / nc, k = tid - p * nc;
      float pr[3];
#pragma unroll 1
      for (int v = 0; v < 3; ++v) {   // softmax of the large / mid / local logits, this thread's class
        float lg[SNV_MAXCLASS];
#pragma unroll
        for (int q = 0; q < SNV_MAXCLASS; ++q)   // every logit read is in flight before the first use
          lg[q] = logit[(v * P + p) * SNV_MAXCLASS + (q < nc ? q : 0)];
        float mx = -INFINITY, own = 0.f;
#pragma unroll
        for (int q = 0; q < SNV_MAXCLASS; ++q)
          if (q < nc) mx = fmaxf(mx, lg[q]);
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < SNV_MAXCLASS; ++q)
          if (q < nc) {
            const float e = __expf(lg[q] - mx);
            sum += e;
            own = (q == k) ? e : own;
          }
        pr[v] = own / sum;
      }
      float prob = (pr[1] + pr[0]) / 2.f;
      if (args.has_local) prob = (pr[2] + prob) / 2.f;
      float res = __logf(fmaxf(prob, 1e-9f));
      if (args.status != nullptr && *args.status != 0) res = __uint_as_float(0x7FC00000u);   // flagged encoding error: loud output
      if (row0 + p < args.n) args.out[(row0 + p) * nc + k] = res;
    }
    if (args.taps != nullptr && tile == 0) {
      float* dst = args.taps + (size_t)12 * args.tap_stride;
      for (int i = tid; i < 2 * P * SNV_C + 3 * P * SNV_MAXCLASS; i += SNV_THREADS) dst[i] = feat[i];
    }
    lds_barrier();
    SNV_STAMP(26);   // head
  }
}

#undef TW_FIRST
#undef TW_LAST
#undef TILE0
#undef TILE_STEP

// ---------------------------------------------------------------------------------------------
// host launch (+ optional per-launch HIP-event timing of this kernel for bench.py's roofline line)
// ---------------------------------------------------------------------------------------------
namespace {
struct KernelProfile {
  bool on = false;
  std::vector<hipEvent_t> ev;   // start/stop pairs
  size_t used = 0;
} g_prof;
}  // namespace

int profile_begin() {
  g_prof.on = true;
  g_prof.used = 0;
  return MURAL_OK;
}

int profile_end(double* total_ms, int64_t* launches) {
  double sum = 0.0;
  for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
    MURAL_HIP_CHECK(hipEventSynchronize(g_prof.ev[i + 1]));
    float ms = 0.f;
    MURAL_HIP_CHECK(hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]));
    sum += ms;
  }
  if (total_ms) *total_ms = sum;
  if (launches) *launches = (int64_t)(g_prof.used / 2);
  g_prof.on = false;
  g_prof.used = 0;
  return MURAL_OK;
}

int launch_snv_tower_wave(const SnvFwdArgs& a, size_t lds_bytes, hipStream_t stream);   // snv_tower_wave.hip

int launch_snv_towers(const MuralSnvModel* m, const SnvFwdArgs& a, size_t lds_bytes, hipStream_t stream) {
  const int64_t n_tiles = (a.n + a.P - 1) / a.P;
  if (n_tiles == 0) return MURAL_OK;
  if (a.wave) {      // wave-private form of a stage-split launch: same per-launch timing hooks
    hipEvent_t w0 = nullptr, w1 = nullptr;
    if (g_prof.on) {
      while (g_prof.ev.size() < g_prof.used + 2) {
        hipEvent_t e;
        MURAL_HIP_CHECK(hipEventCreate(&e));
        g_prof.ev.push_back(e);
      }
      w0 = g_prof.ev[g_prof.used];
      w1 = g_prof.ev[g_prof.used + 1];
      g_prof.used += 2;
      MURAL_HIP_CHECK(hipEventRecord(w0, stream));
    }
    if (int rc = launch_snv_tower_wave(a, lds_bytes, stream)) return rc;
    if (w1) MURAL_HIP_CHECK(hipEventRecord(w1, stream));
    return MURAL_OK;
  }
  int grid = (int)(n_tiles < 2048 ? n_tiles : 2048);
  if (a.par) grid = (int)(2 * n_tiles);   // small batches only (run_towers): one workgroup per (tile, tower)
  if (const char* e = dev_env("MURAL_DEBUG_TOWER_GRID")) {   // diagnostic: e.g. 256 = one workgroup per CU (tools/phase_stamps.py)
    const int v = atoi(e);
    if (v >= 1 && v < grid) grid = v;
  }
  if (const char* e = dev_env("MURAL_DEBUG_TOWER_LDS")) {    // diagnostic: inflate the first-stage launches' LDS request (occupancy study)
    const size_t v = (size_t)atol(e);
    if (a.phase == 1 && v > lds_bytes && v <= 160 * 1024) lds_bytes = v;
  }
  static DynLdsOnce big_lds;
  if (int rc = big_lds.ensure(&snv_towers_fused<0>, &snv_towers_fused<1>, &snv_towers_fused<2>, &snv_towers_fused<3>)) return rc;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (g_prof.on) {
    while (g_prof.ev.size() < g_prof.used + 2) {
      hipEvent_t e;
      MURAL_HIP_CHECK(hipEventCreate(&e));
      g_prof.ev.push_back(e);
    }
    e0 = g_prof.ev[g_prof.used];
    e1 = g_prof.ev[g_prof.used + 1];
    g_prof.used += 2;
    MURAL_HIP_CHECK(hipEventRecord(e0, stream));
  }
  if (a.par) hipLaunchKernelGGL(snv_towers_fused<3>, dim3(grid), dim3(SNV_THREADS), lds_bytes, stream, a);
  else if (a.phase == 1) hipLaunchKernelGGL(snv_towers_fused<1>, dim3(grid), dim3(SNV_THREADS), lds_bytes, stream, a);
  else if (a.phase == 2) hipLaunchKernelGGL(snv_towers_fused<2>, dim3(grid), dim3(SNV_THREADS), lds_bytes, stream, a);
  else hipLaunchKernelGGL(snv_towers_fused<0>, dim3(grid), dim3(SNV_THREADS), lds_bytes, stream, a);
  MURAL_HIP_CHECK(hipGetLastError());
  if (e1) MURAL_HIP_CHECK(hipEventRecord(e1, stream));
  return MURAL_OK;
}

}  // namespace mural

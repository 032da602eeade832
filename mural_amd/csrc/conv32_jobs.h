// Job descriptions of the channel-last kernels of the composed SNV training step (snv_train.hip -> conv32_wave.hip / conv32_cl.hip).
// The two towers of Network1 / Network2 (MuRaL/model/model_snv.py:350-430) have the same layer sequence, so the step walks them in
// lockstep: ONE launch per layer index carries up to two jobs (blockIdx.y = job; the large tower first, so that its workgroups are
// dispatched first and the mid tower's fill the tail) -- half the launches, one launch gap and one tail per layer pair.
#pragma once
#include <cstdint>

namespace mural {

constexpr int TOWER_JOBS = 2;

struct ConvFwdJob {       // y = conv32(BN(act(x))) + bias [+ res1 + res2]; BatchNorm finalised from acc in the prologue
  const float* x;
  int64_t B;
  int L;
  int pre_relu;
  const double* acc;
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  float* state;
  const float* W;
  const float* wfrag;     // forward fragments of W (conv32_wave.hip: cw_wfrag_build) or nullptr
  const float* bias;
  int post_relu;
  const float* res1;
  const float* res2;
  double* acc_out;
  int out_relu;
  float* y;
};

// dy of a conv backward made while it is staged (conv32_wave.hip, FOLD): the BatchNorm-backward apply of the layer BEHIND the conv --
// whose input gradient `dz` and saved input `x` sit in memory, its sums `acc` complete -- plus an optional residual gradient;
// dgamma / dbeta of that BatchNorm are written by the launch; dy_out != nullptr: dy is also written out.  dz == nullptr: no fold.
struct ConvBwdFold {
  const float* dz = nullptr;
  const float* x = nullptr;
  const float* state = nullptr;
  const float* gamma = nullptr;
  const double* acc = nullptr;
  int relu = 0;
  float* dgamma = nullptr;
  float* dbeta = nullptr;
  const float* add1 = nullptr;
  float* dy_out = nullptr;
  const float* add2 = nullptr;      // a second residual gradient (the stage's outer skip): dy = apply + add1 + add2
  int dy_out_plus_add1 = 0;         // dy_out receives dy + add1: the ONE tensor the stage's input gradient then adds (not dy and add1 apart)
};

struct ConvBwdJob {       // dz, BatchNorm-backward sums, partial rows of dW / db
  const float* dy;
  const float* x;
  const float* W;
  const float* wfrag;     // input-gradient fragments of W or nullptr
  int64_t B;
  int L;
  const float* state;
  const float* gamma;
  int pre_relu;
  float* dz;
  double* stat_out;
  float* part;
  int nrow;               // out: partial rows written
  ConvBwdFold fold;
};

struct BnApplyJob {       // dx = a'(x) * gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)) [+ add1 + add2]; dgamma / dbeta
  const float* dz;
  const float* x;
  int64_t rows;
  int relu;
  const float* state;
  const float* gamma;
  const double* acc;
  const float* add1;
  const float* add2;
  float* dx;
  float* dgamma;
  float* dbeta;
};

struct PoolFwdJob { const float* x; int64_t B; int L, k, s, p; float* y; int32_t* arg; double* acc; };
// fold: dy is not read but made per element from the BatchNorm-backward apply of the conv BEHIND the pool (its input gradient dz, its
// saved input x = the pooled tensor, its completed sums): dy = a'(x) * gamma * invstd * (dz - mean(dz) - xhat * mean(dz * xhat)); the
// launch's workgroup 0 writes that BatchNorm's dgamma / dbeta.  fdz == nullptr: no fold.
struct PoolBwdJob {
  const float* dy; const int32_t* arg; int64_t B; int L, Lout, k, s, p; float* dx;
  const float* fdz = nullptr; const float* fx = nullptr; const float* fstate = nullptr; const float* fgamma = nullptr;
  const double* facc = nullptr; int frelu = 0; float* fdgamma = nullptr; float* fdbeta = nullptr;
};
struct GmaxFwdJob { const float* x; int64_t B; int L; int relu; float* feat; int32_t* arg; };
struct GmaxBwdJob { const float* dfeat; const int32_t* arg; const float* c3; int64_t B; int L; float* dx; };

}  // namespace mural

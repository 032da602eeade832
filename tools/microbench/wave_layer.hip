// Microbenchmark (GPU box): the wave-private conv layer of csrc/snv_tower_wave.hip in isolation -- every wave runs
// conv_layer_wave<9> `iters` times on its own LDS image.  Prints TFLOP/s of the executed MFMAs with one and with two
// four-wave workgroups per CU, i.e. what the layer loop itself can reach without entry / pooling / launch tails.
#include "../../mural_amd/csrc/snv_tower_wave.hip"
#include <cstdio>
#include <vector>
namespace mural { void set_error(const char*, ...) {} }
using namespace mural;

template <int NB, int MODE>
__global__ __launch_bounds__(256, 2) void klayer(const float* wfrag, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n16 = lane & 15, kk = lane >> 4;
  float* img = smem + wave * ((16 * NB + 2) * 32);
  for (int i = lane; i < (16 * NB + 2) * 32; i += 64) img[i] = 0.001f * (i & 15);
  WaveAddr sa;
  for (int t = 0; t < 3; ++t)
    for (int h = 0; h < 2; ++h) sa.rd[2 * t + h] = 4u * (uint32_t)lds_off(n16 + t, 4 * h + kk);
  sa.wr[0] = 4u * (uint32_t)lds_off(n16 + 1, kk);
  sa.wr[1] = 4u * (uint32_t)lds_off(n16 + 1, 4 + kk);
  sa.vmask = 0x1ffu ^ (n16 == 3 ? 4u : 0u);
  float a0[24], a1[24];
  for (int s = 0; s < 24; ++s) {
    a0[s] = wfrag[s * 64 + lane];
    a1[s] = wfrag[(24 + s) * 64 + lane];
  }
  f32x4 xr0[TW_NBW], xr1[TW_NBW];
  for (int b = 0; b < TW_NBW; ++b) {
    xr0[b] = splat(0.01f * b);
    xr1[b] = splat(0.02f * b);
  }
  const f32x4 pb[2] = {splat(0.01f), splat(0.02f)}, ps[2] = {splat(0.5f), splat(0.25f)}, pt[2] = {splat(0.f), splat(0.01f)};
  for (int it = 0; it < iters; ++it) {
    const LayerK lk = layer_consts(layer_mode(it & 3));
    FragSrc fs;
    fs.wf = uniform_rsrc(wfrag);
    fs.lane16 = 16u * lane;
    fs.layer_bytes = MODE == 0 ? 0u : ~0u;
    XReq xq;
    xq.on = false;
    conv_layer_wave<NB, false>(reinterpret_cast<char*>(img), sa, lk, a0, a1, fs, pb, ps, pt, xr0, xr1, xq);
  }
  float s = 0.f;
  for (int b = 0; b < NB; ++b) s += xr0[b].x + xr1[b].y;
  out[blockIdx.x * 256 + threadIdx.x] = s + img[lane];
}

int main() {
  float *w, *out;
  hipMalloc(&w, 48 * 64 * 4);
  std::vector<float> hw(48 * 64);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.01f * ((int)(i % 7) - 3);
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&out, 4096 * 256 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&klayer<9, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&klayer<9, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int iters = 400;
  for (int mode = 0; mode < 2; ++mode)
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
      const size_t lds = per_cu == 1 ? 100 * 1024 : 76 * 1024;
      const int grid = 256 * per_cu;
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL((klayer<9, 0>), dim3(grid), dim3(256), lds, 0, w, out, iters);
        else hipLaunchKernelGGL((klayer<9, 1>), dim3(grid), dim3(256), lds, 0, w, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)grid * 4 * iters * 9 * 48 * 2048.0;
      printf("mode %d (%s)  %d workgroup(s)/CU: %.3f ms  %.1f TFLOP/s  (%.3f of 157.3)\n", mode, mode == 0 ? "with fragment prefetch" : "no prefetch",
             per_cu, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
    }
  return 0;
}

// Classification of one column of a dense (n, 4, L) one-hot / IUPAC-fraction window into its symbol (reference encoding:
// MuRaL/data/preprocessing.py:756-816): shared by the dense -> symbol pass (encode.hip) and the small-batch first-stage kernel, which
// reads the dense window itself (snv_stage1.hip).
#pragma once
#include "common.h"

namespace mural {

__device__ __forceinline__ int frac_digit(float v) {
  if (v == 0.0f) return 0;
  if (v == 1.0f) return 1;
  if (v == 0.5f) return 2;
  if (v == 0.25f) return 3;
  if (v == (float)(1.0 / 3.0)) return 4;
  return -1;
}

// dense (n,4,L) one-hot / IUPAC-fraction tensor -> 1 symbol per column (what the fused kernel consumes)
__device__ __forceinline__ int dense_symbol(float v0, float v1, float v2, float v3) {
  const int d0 = frac_digit(v0), d1 = frac_digit(v1), d2 = frac_digit(v2), d3 = frac_digit(v3);
  int s = -1;
  if ((d0 | d1 | d2 | d3) >= 0) {
    switch (d0 + 5 * d1 + 25 * d2 + 125 * d3) {
      case 1: s = 0; break;      // A
      case 5: s = 1; break;      // C
      case 25: s = 2; break;     // G
      case 125: s = 3; break;    // T
      case 468: s = 4; break;    // N
      case 52: s = 5; break;     // R
      case 260: s = 6; break;    // Y
      case 12: s = 7; break;     // M
      case 60: s = 8; break;     // S
      case 252: s = 9; break;    // W
      case 300: s = 10; break;   // K
      case 620: s = 11; break;   // B
      case 604: s = 12; break;   // D
      case 524: s = 13; break;   // H
      case 124: s = 14; break;   // V
      default: break;
    }
  }
  return s;
}

}  // namespace mural

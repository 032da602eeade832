// Host side of model_predict_m's symbol route: one fp32 one-hot window [4][L] -> one symbol byte per column (dense_symbol.h's rule; the
// caller is mural_host_dense_to_symbols, ingest.hip).  Plain C++ built by g++ (not hipcc) so that the column loop can be cloned per
// instruction set (target_clones: AVX2 where the host has it; the loader picks at load time).
//
// Reference: the windows are what seq_ohe_encoder wrote (MuRaL/data/preprocessing.py:756-816): almost every column is one 1.0 over three
// 0.0.  A block of 256 columns whose 1024 values are all +0.0 or 1.0 is classified from bit compares alone -- index = which rows hold
// 1.0, one byte-table read per column: 6.7 GB/s per thread with AVX2, 5.0 with SSE2 -- and any other block (N = four 0.25, IUPAC
// fractions, -0.0, anything else) takes the general digit rule below: 2.8 GB/s.  Same bytes either way.
#include <cstddef>
#include <cstdint>

namespace mural {
namespace {

int64_t classify_general(const float* x, int L, int c0, int m, uint8_t* out, const uint8_t* lut625) {
  const float third = (float)(1.0 / 3.0);
  int64_t bad = 0;
  int32_t key[256];
  for (int c = 0; c < m; ++c) key[c] = 0;
  int mul = 1;
  for (int r = 0; r < 4; ++r) {
    const float* row = x + (size_t)r * L + c0;
    for (int c = 0; c < m; ++c) {            // digit of frac_digit(): 0, 1, .5, .25, 1/3 -> 0..4; anything else poisons the key
      const float v = row[c];
      const int d = (v == 1.0f) * 1 + (v == 0.5f) * 2 + (v == 0.25f) * 3 + (v == third) * 4;
      const int ok = (v == 0.0f) | (d != 0);
      key[c] += ok ? d * mul : 100000;
    }
    mul *= 5;
  }
  for (int c = 0; c < m; ++c) {
    const uint8_t sy = (uint32_t)key[c] < 625u ? lut625[key[c]] : (uint8_t)255;
    out[c0 + c] = sy;
    bad += sy == 255;
  }
  return bad;
}

}  // namespace

// lut625: symbol of key d0 + 5 d1 + 25 d2 + 125 d3 (255 = no MuRaL encoding); lut16: the same for columns of 0 / 1 only, index = bit r set
// when row r holds 1.0.  Returns the number of columns that are no MuRaL encoding (they get 255).
__attribute__((target_clones("avx2", "default"), visibility("hidden"))) int64_t classify_window_host(const float* x, int L, uint8_t* out, const uint8_t* lut625,
                                                                               const uint8_t* lut16) {
  int64_t bad = 0;
  constexpr int BLK = 256;
  constexpr uint32_t ONE = 0x3f800000u;
  const uint32_t* u = reinterpret_cast<const uint32_t*>(x);
  uint8_t idx[BLK];
  for (int c0 = 0; c0 < L; c0 += BLK) {
    const int m = L - c0 < BLK ? L - c0 : BLK;
    const uint32_t *r0 = u + c0, *r1 = u + (size_t)L + c0, *r2 = u + 2 * (size_t)L + c0, *r3 = u + 3 * (size_t)L + c0;
    uint32_t other = 0;
    for (int c = 0; c < m; ++c) {
      const uint32_t a = r0[c], b = r1[c], d = r2[c], e = r3[c];
      const uint32_t a1 = a == ONE, b1 = b == ONE, d1 = d == ONE, e1 = e == ONE;
      other |= (uint32_t)((a != 0) & !a1) | (uint32_t)((b != 0) & !b1) | (uint32_t)((d != 0) & !d1) | (uint32_t)((e != 0) & !e1);
      idx[c] = (uint8_t)(a1 | (b1 << 1) | (d1 << 2) | (e1 << 3));
    }
    if (other) {
      bad += classify_general(x, L, c0, m, out, lut625);
      continue;
    }
    for (int c = 0; c < m; ++c) {
      const uint8_t sy = lut16[idx[c]];
      out[c0 + c] = sy;
      bad += sy == 255;
    }
  }
  return bad;
}

}  // namespace mural

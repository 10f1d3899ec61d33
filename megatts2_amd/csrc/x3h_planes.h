// Host side of the "x3h" operand format (gemm_x3h.hip): a weight matrix as TWO fp16 planes plus one power-of-two scale per
// weight row (= output column of the GEMM).
//
//   s_n   = 2^e_n, e_n chosen so that max_k |w[n, k]| * s_n lies in [2^14, 2^15)  (fp16: largest finite 65504, smallest normal 2^-14)
//   hi    = fp16_rn(w * s_n)                      11 significant bits
//   lo    = fp16_rn((w * s_n - hi) * 2^11)        the next 11 bits (the residual of a round-to-nearest is exact in f32 and at
//                                                 most 2^-11 |w s|; scaled by 2^11 it has the magnitude of w s again)
//   w * s_n = hi + 2^-11 lo + d,  |d| <= 2^-23 |w s_n| for elements within 2^-29 of the row maximum (normal hi); smaller elements
//   keep an ABSOLUTE accuracy of 2^-36 * 2^15 / s_n = 2^-50 of the row maximum - far below the f32 rounding of the row's dot products.
//   inv_n = 2^-e_n is multiplied back in the GEMM epilogue (exact).
// Layout: CHUNK-INTERLEAVED.  Row n of the matrix is stored as ceil(K / 32) blocks of 128 bytes, block c = [hi of k = 32 c .. 32 c + 31
// (64 B) | lo of the same k (64 B)]; K is zero-padded to a multiple of 32.  One block is exactly what a workgroup's ring stage needs of
// row n, and it is ONE 128-byte line: LDS-DMA moves 64-byte row segments (the plane-major layout of x6) at half the rate of 128-byte
// ones (tools/ubench/ldsdma_issue.hip, profiles/r06_ubench_ldsdma_issue.txt: 27 vs 59 B/clk/CU).  Row stride = 4 * Kp bytes - for
// K % 32 == 0 the byte offset of (n, k0) with k0 % 32 == 0 is 4 (n K + k0), the offset of the f32 element itself.
// The conversion is written out (round to nearest even, gradual underflow) so that the planes do not depend on the host compiler's
// _Float16 support.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace mt2 {

inline uint16_t f32_to_f16_rn(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));      // inf / NaN
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                        // >= 65520: rounds to inf
    if (x < 0x33000001u) return (uint16_t)sign;                                                     // <= 2^-25: rounds to zero
    int e = (int)(x >> 23) - 127;
    uint32_t man = (x & 0x7fffffu) | 0x800000u;                                                     // 24-bit significand
    int shift;                                                                                      // bits dropped
    if (e >= -14) shift = 13;
    else { shift = 13 + (-14 - e); e = -15; }                                                       // subnormal: exponent field 0
    const uint32_t keep = man >> shift, rem = man & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    uint32_t r = keep + ((rem > halfway || (rem == halfway && (keep & 1u))) ? 1u : 0u);
    // normal: r has the implicit bit at 0x400 (a carry to 0x800 moves into the exponent by plain addition);
    // subnormal: r is the mantissa field itself (a carry to 0x400 becomes the smallest normal by the same addition)
    const uint32_t h = e >= -14 ? (((uint32_t)(e + 15) << 10) + (r - 0x400u)) : r;
    return (uint16_t)(sign | h);
}
inline float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, ex = (h >> 10) & 31u, man = h & 0x3ffu;
    float v;
    if (ex == 31) { const uint32_t b = sign | 0x7f800000u | (man << 13); std::memcpy(&v, &b, 4); return v; }
    if (ex == 0) v = std::ldexp((float)man, -24);
    else v = std::ldexp((float)(man | 0x400u), (int)ex - 25);
    return sign ? -v : v;
}

inline size_t x3h_padded_k(size_t row_len) { return (row_len + 31) / 32 * 32; }
// w: rows x row_len (row-major).  planes: rows x (2 * x3h_padded_k(row_len)) uint16, chunk-interleaved (above); inv: [rows]
inline void x3h_split_rows(const float* w, size_t rows, size_t row_len, uint16_t* planes, float* inv) {
    const size_t kp = x3h_padded_k(row_len);
    std::memset(planes, 0, rows * 2 * kp * sizeof(uint16_t));
    for (size_t r = 0; r < rows; ++r) {
        float mx = 0.0f;
        const float* row = w + r * row_len;
        for (size_t k = 0; k < row_len; ++k) {
            const float a = std::fabs(row[k]);
            if (a > mx && std::isfinite(a)) mx = a;
        }
        int e = 0;
        if (mx > 0.0f) {
            int q;
            (void)std::frexp(mx, &q);            // mx = m * 2^q, m in [0.5, 1): 2^(q-1) <= mx < 2^q
            e = 15 - q;                          // mx * 2^e in [2^14, 2^15)
            if (e > 100) e = 100;
            if (e < -100) e = -100;
        }
        const float s = std::ldexp(1.0f, e);
        inv[r] = std::ldexp(1.0f, -e);
        for (size_t k = 0; k < row_len; ++k) {
            const float v = row[k] * s;
            const uint16_t h = f32_to_f16_rn(v);
            const float res = (v - f16_to_f32(h)) * 2048.0f;
            uint16_t* blk = planes + r * 2 * kp + (k / 32) * 64;
            blk[k % 32] = h;
            blk[32 + k % 32] = f32_to_f16_rn(res);
        }
    }
}

}  // namespace mt2

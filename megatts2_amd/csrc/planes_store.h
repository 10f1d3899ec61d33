// Device side of the fp16 planes hand-over (GemmP::a_planes): the store every PRODUCER kernel uses - LayerNorm (rowops.hip), the x3h
// loader tile's epilogue (gemm_common.h, GemmP::c_planes) and the attention kernels (attention.hip, AttnP::o_planes).
// Four consecutive columns c .. c + 3 of a row as fp16 planes: x3h_planes.h's block layout without a row scale - per 32 columns a
// 128-byte block [32 hi | 32 lo], hi = fp16_rn(v), lo = fp16_rn((v - hi) * 2^11) - exactly what gemm_x3h.hip's split produces in
// registers, so a consumer that takes the planes computes bit-identical products.  Same bytes and row stride as f32.
// Returns max |v| for the range guard (the caller raises the x3h flag at 65504).
#pragma once
#include <hip/hip_runtime.h>

namespace mt2 {

typedef _Float16 pl_f16x2 __attribute__((ext_vector_type(2)));
typedef float pl_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float store_planes4(float* __restrict__ row, int c, float y0, float y1, float y2, float y3) {
    const pl_f16x2 h01 = __builtin_convertvector((pl_f32x2){y0, y1}, pl_f16x2), h23 = __builtin_convertvector((pl_f32x2){y2, y3}, pl_f16x2);
    const float s = 2048.0f;
    const pl_f16x2 l01 = __builtin_convertvector((pl_f32x2){__builtin_fmaf((float)h01[0], -s, y0 * s), __builtin_fmaf((float)h01[1], -s, y1 * s)}, pl_f16x2);
    const pl_f16x2 l23 = __builtin_convertvector((pl_f32x2){__builtin_fmaf((float)h23[0], -s, y2 * s), __builtin_fmaf((float)h23[1], -s, y3 * s)}, pl_f16x2);
    char* blk = reinterpret_cast<char*>(row) + (c >> 5) * 128 + (c & 31) * 2;
    *reinterpret_cast<uint2*>(blk) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
    *reinterpret_cast<uint2*>(blk + 64) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
    return fmaxf(fmaxf(fabsf(y0), fabsf(y1)), fmaxf(fabsf(y2), fabsf(y3)));
}
__device__ __forceinline__ float store_planes4(float* __restrict__ row, int c, const float4& y) {
    return store_planes4(row, c, y.x, y.y, y.z, y.w);
}

}  // namespace mt2

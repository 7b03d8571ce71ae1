// Layout of split rows and the fp16-pair conversion of one chunk of eight coordinates, shared by the
// stand-alone conversion kernel (score_split.hip) and the prologue kernel that writes pairs directly
// (prologue.hip).  See score_split.hip for the format.
#pragma once
#include "common.h"

namespace hypmi_detail {

__host__ __device__ constexpr int split_half_slots(int ks16, int np = 3) { return (np * ks16 + 15) / 16 * 16; }
// the sum of the two halves is even, so the last slot of a row is always padding
__host__ __device__ constexpr int split_row_slots(int ks16, int np = 3) {
    return (split_half_slots(ks16, np) + np * ks16) | 1;
}

// Power of two that puts a row maximum m (finite, >= 0) in [2^14, 2^15): m = f * 2^ex with f in
// [0.5, 1) is scaled by 2^(15 - ex); capped so that the inverse scale 2^-up stays a normal float
// (rows below 2^-111 keep fewer bits).
__device__ __forceinline__ int pair_scale_exponent(float m) {
    int up = m > 0.f ? 15 - __builtin_amdgcn_frexp_expf(m) : 0;
    return up > 126 ? 126 : up;
}

// largest finite magnitude of eight values (NaN and infinity do not set the scale)
__device__ __forceinline__ float pair_chunk_max(const float (&v)[8]) {
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float a = fabsf(v[e]);
        if (a <= 3.402823466e38f && a > m) m = a;
    }
    return m;
}

// Chunk `chunk` (coordinates 8 chunk .. + 7) of a row scaled by 2^up -> the two slots of its pair.
__device__ __forceinline__ void pair_chunk_store(const float (&v)[8], int up, int chunk, int hs, uint4* out) {
    unsigned b[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = ldexpf(v[e], up);
        const _Float16 x0 = (_Float16)x;  // round to nearest even; NaN and infinity keep their class
        const float r = x - (float)x0;
        const _Float16 x1 = (r == r && fabsf(r) <= 3.402823466e38f) ? (_Float16)r : (_Float16)0.f;
        b[0][e] = __builtin_bit_cast(unsigned short, x0);
        b[1][e] = __builtin_bit_cast(unsigned short, x1);
    }
    const int j = chunk >> 1, h = chunk & 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        uint4 w;
        w.x = b[s][0] | (b[s][1] << 16);
        w.y = b[s][2] | (b[s][3] << 16);
        w.z = b[s][4] | (b[s][5] << 16);
        w.w = b[s][6] | (b[s][7] << 16);
        out[h * hs + 2 * j + s] = w;
    }
}

// padding slots of a pair row and its inverse scale (one lane per row)
__device__ __forceinline__ void pair_row_tail(int up, int ks16, int hs, int row_slots, uint4* out) {
    for (int p = 2 * ks16; p < hs; ++p) out[p] = make_uint4(0, 0, 0, 0);
    for (int p = hs + 2 * ks16; p < row_slots - 1; ++p) out[p] = make_uint4(0, 0, 0, 0);
    out[row_slots - 1] = make_uint4(__float_as_uint(ldexpf(1.f, -up)), 0, 0, 0);
}

}  // namespace hypmi_detail

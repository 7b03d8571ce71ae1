// Shared fp32 MFMA tile main loop:  acc += A[m0.., :K] * B[n0.., :K]^T  for one workgroup tile.
//
// 256 threads = 4 waves arranged 2 x 2; a wave owns TM x TN tiles of 32 x 32 outputs
// (v_mfma_f32_32x32x2_f32), so the workgroup tile is (64 TM) x (64 TN).  Both operands are
// K-contiguous; K is consumed in steps of 32 through LDS (row stride 36 floats keeps the 16-byte
// fragment reads conflict-free) and the next step's global loads are issued before the current
// step's MFMAs.  D layout of a 32 x 32 tile: column = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
#pragma once
#include "common.h"

namespace gemm_tile {

constexpr int BK = 32, LDT = BK + 4;

template <int TM, int TN>
struct Shape {
    static constexpr int BM = 64 * TM, BN = 64 * TN;
    static constexpr int SMEM_FLOATS = (BM + BN) * LDT;
};

__device__ __forceinline__ float4 ld4_guard(const float* p, bool ok) {
    return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// The first `valid` (1..4) components of a 16-byte load, the rest zero: the pad columns behind a row
// of K % 4 != 0 values belong to the caller's buffer and may hold anything (a NaN there would
// otherwise reach the product although its weight is zero).
__device__ __forceinline__ float4 ld4_head(const float* p, bool ok, int valid) {
    float4 v = ld4_guard(p, ok);
    if (valid < 4) {
        v.w = 0.f;
        if (valid < 3) v.z = 0.f;
        if (valid < 2) v.y = 0.f;
    }
    return v;
}

__device__ __forceinline__ float f4_get(const float4& v, int e) {
    return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w;
}

// smem: SMEM_FLOATS floats, 16-byte aligned.  acc is accumulated into (zero it first for a fresh tile).
// k_begin (a multiple of 4) skips a leading part of K that the caller knows to be zero in B.
template <int TM, int TN>
__device__ __forceinline__ void mainloop(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                         int64_t ldb, int64_t m0, int64_t n0, int64_t M, int64_t N, int K,
                                         float* smem, f32x16 (&acc)[TM][TN], int k_begin = 0) {
    constexpr int BM = Shape<TM, TN>::BM, BN = Shape<TM, TN>::BN;
    float* As = smem;
    float* Bs = smem + BM * LDT;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int lrow = tid >> 3;     // 0..31 (+32 per pass)
    const int lk = (tid & 7) * 4;  // float offset inside the K step

    float4 ra[BM / 32], rb[BN / 32];
    auto fetch = [&](int k0) {
        const bool kok = (k0 + lk) < K;
        const int valid = K - (k0 + lk);  // >= 4 except in the last float4 of a row with K % 4 != 0
#pragma unroll
        for (int i = 0; i < BM / 32; ++i) {
            const int64_t gm = m0 + lrow + 32 * i;
            ra[i] = ld4_head(A + gm * lda + k0 + lk, kok && gm < M, valid);
        }
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
            const int64_t gn = n0 + lrow + 32 * i;
            rb[i] = ld4_head(B + gn * ldb + k0 + lk, kok && gn < N, valid);
        }
    };

    fetch(k_begin);
    for (int k0 = k_begin; k0 < K; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < BM / 32; ++i)
            *reinterpret_cast<float4*>(&As[(lrow + 32 * i) * LDT + lk]) = ra[i];
#pragma unroll
        for (int i = 0; i < BN / 32; ++i)
            *reinterpret_cast<float4*>(&Bs[(lrow + 32 * i) * LDT + lk]) = rb[i];
        __syncthreads();
        if (k0 + BK < K) fetch(k0 + BK);
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t)
                a[t] = *reinterpret_cast<const float4*>(&As[(wr * 32 * TM + 32 * t + r) * LDT + 8 * s + 4 * h]);
#pragma unroll
            for (int t = 0; t < TN; ++t)
                b[t] = *reinterpret_cast<const float4*>(&Bs[(wc * 32 * TN + 32 * t + r) * LDT + 8 * s + 4 * h]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4_get(a[i], e), f4_get(b[j], e),
                                                                         acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();  // the caller may start another tile (or reuse smem) right away
}

template <int TM, int TN>
__device__ __forceinline__ void zero(f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
}

// row / column of accumulator element e of tile (i, j) relative to the workgroup tile origin
template <int TM>
__device__ __forceinline__ int row_of(int i, int e) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave >> 1) * 32 * TM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
}
template <int TN>
__device__ __forceinline__ int col_of(int j) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (wave & 1) * 32 * TN + 32 * j + (lane & 31);
}

}  // namespace gemm_tile

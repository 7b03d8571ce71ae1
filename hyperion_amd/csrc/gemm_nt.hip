// Generic fp32 "NT" GEMM on the gfx950 matrix cores:  C[m,n] = sum_k A[m,k] * B[n,k] + bias[n].
//
// Used for the affine steps of the pre-processing chain (A = embeddings, B = transposed weights)
// and as the score kernel for models whose packed rows are too wide for the register-stationary
// kernel in score.hip.  Both operands are K-contiguous, so each lane fetches four consecutive k of
// one row with a single 16-byte LDS read and feeds them to four v_mfma_f32_32x32x2_f32 steps.
//
// Tile: 128 x 128 outputs per 256-thread workgroup, 64 x 64 per wave (2 x 2 MFMA tiles); the main
// loop is shared with the fused prologue kernel (gemm_tile.h).
#include "gemm_tile.h"

namespace {

constexpr int BM = 128, BN = 128;

__global__ __launch_bounds__(256) void gemm_nt_kernel(const float* __restrict__ A, int64_t lda,
                                                      const float* __restrict__ B, int64_t ldb,
                                                      const float* __restrict__ bias,
                                                      float* __restrict__ C, int64_t ldc, int64_t M,
                                                      int64_t N, int K) {
    __shared__ __attribute__((aligned(16))) float smem[gemm_tile::Shape<2, 2>::SMEM_FLOATS];
    const int64_t m0 = (int64_t)blockIdx.y * BM;
    const int64_t n0 = (int64_t)blockIdx.x * BN;
    f32x16 acc[2][2];
    gemm_tile::zero<2, 2>(acc);
    gemm_tile::mainloop<2, 2>(A, lda, B, ldb, m0, n0, M, N, K, smem, acc);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t col = n0 + gemm_tile::col_of<2>(j);
        const float bv = (bias != nullptr && col < N) ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = m0 + gemm_tile::row_of<2>(i, e);
                if (row < M && col < N) C[row * ldc + col] = acc[i][j][e] + bv;
            }
        }
    }
}

}  // namespace

int hypmi_launch_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                         float* C, int64_t ldc, int64_t M, int64_t N, int K, hipStream_t stream) {
    HYPMI_REQUIRE(A && B && C, "gemm_nt: null pointer");
    HYPMI_REQUIRE(M >= 0 && N >= 0 && K > 0, "gemm_nt: bad shape M=%lld N=%lld K=%d", (long long)M,
                  (long long)N, K);
    HYPMI_REQUIRE(aligned16(A) && aligned16(B) && lda % 4 == 0 && ldb % 4 == 0,
                  "gemm_nt: operands must be 16-byte aligned with ld %% 4 == 0");
    HYPMI_REQUIRE(lda >= round_up(K, 4) && ldb >= round_up(K, 4) && ldc >= N,
                  "gemm_nt: leading dimension too small");
    if (M == 0 || N == 0) return HYPMI_OK;
    const int64_t gy = (M + BM - 1) / BM, gx = (N + BN - 1) / BN;
    HYPMI_REQUIRE(gy <= 0x7fffffffLL && gx <= 65535 * 64LL, "gemm_nt: grid too large");
    // blockIdx.y is limited to 65535: split M over several launches if needed
    const int64_t max_y = 65535;
    for (int64_t y0 = 0; y0 < gy; y0 += max_y) {
        const int64_t ny = (gy - y0 < max_y) ? gy - y0 : max_y;
        const int64_t row0 = y0 * BM;
        const int64_t rows = (M - row0 < ny * BM) ? M - row0 : ny * BM;
        dim3 grid((unsigned)gx, (unsigned)ny);
        hipLaunchKernelGGL(gemm_nt_kernel, grid, dim3(256), 0, stream, A + row0 * lda, lda, B, ldb,
                           bias, C + row0 * ldc, ldc, rows, N, K);
    }
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

extern "C" int hypmi_affine(const float* x, int64_t n, int64_t ldx, int d_in, const float* Wt,
                            int64_t ldw, const float* bias, int d_out, float* y, int64_t ldy,
                            void* stream) {
    HYPMI_REQUIRE(d_in > 0 && d_out > 0 && n >= 0, "affine: bad shape");
    HYPMI_REQUIRE(y != nullptr, "affine: null output");
    return hypmi_launch_gemm_nt(x, ldx, Wt, ldw, bias, y, ldy, n, d_out, d_in,
                                static_cast<hipStream_t>(stream));
}

// Shared host-side helpers for the hypmi library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/hypmi.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void hypmi_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

#define HYPMI_REQUIRE(cond, ...)              \
    do {                                      \
        if (!(cond)) {                        \
            hypmi_set_error(__VA_ARGS__);     \
            return HYPMI_EINVAL;              \
        }                                     \
    } while (0)

#define HYPMI_HIP(call)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            hypmi_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),        \
                            __FILE__, __LINE__);                                          \
            return HYPMI_EHIP;                                                            \
        }                                                                                 \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// C[m,n] = sum_k A[m,k] * B[n,k] + bias[n]   (fp32 MFMA, defined in gemm_nt.hip)
int hypmi_launch_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                         float* C, int64_t ldc, int64_t M, int64_t N, int K, hipStream_t stream);

// Width of the packed rows the score kernels accept for a K-dimensional model (score.hip).
int hypmi_kp_for_k(int K);

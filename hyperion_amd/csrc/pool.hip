// Multi-session enrolment helpers: class pooling (sufficient statistics), by-the-book packing and
// block scatter.  All HBM-streaming, one wavefront per output row, coalesced along the row.
#include "common.h"

namespace {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// out[g,:] = sum (or mean) of w[order[r]] x[order[r],:] for r in [offsets[g], offsets[g+1]) (w = 1
// without weights).  Rows are added in list order with a double accumulator, so the result does not
// depend on scheduling.  An empty class gives 0 (sum) or NaN (mean = 0/0), as dividing F by N = 0 does.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void pool_rows_kernel(const TI* __restrict__ x, int64_t ldx, int d,
                                                        const int32_t* __restrict__ order,
                                                        const int64_t* __restrict__ offsets,
                                                        int64_t n_groups, int mean,
                                                        const double* __restrict__ w,
                                                        TO* __restrict__ out, int64_t ldo,
                                                        float* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= n_groups) return;
    const int64_t r0 = offsets[g], r1 = offsets[g + 1];
    const double inv = mean ? 1.0 / (double)(r1 - r0) : 1.0;  // 1/0 = inf, 0 * inf = NaN
    if (counts && lane == 0) counts[g] = (float)(r1 - r0);
    for (int k0 = 0; k0 < d; k0 += 256) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int64_t r = r0; r < r1; ++r) {
            const int64_t src = order[r];
            const TI* row = x + src * ldx;
            const double ws = w ? w[src] : 1.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 64 * u + lane;
                if (k < d) acc[u] += ws * (double)row[k];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 64 * u + lane;
            if (k < d) out[g * ldo + k] = (TO)(acc[u] * inv);
        }
    }
}

// By-the-book packing for one (own count, total count) group:
//   g_k = (s_k + shift_t[k]) * scale_t[k],  h_k = (s_k + shift_n[k]) * scale_n[k],
//   q = (|g|^2 - |h|^2) / 2,  packed row = [g, q, 1] or [g, 1, q + c].
// rows == nullptr means rows 0..n-1 of s; otherwise row i of the output comes from s[rows[i]].
// T = double: statistics and maps in float64 (classes of many segments: |g|^2 - |h|^2 cancels to
// about 1 / count of either term, which fp32 maps cannot carry).
template <typename T>
__global__ __launch_bounds__(256) void book_pack_kernel(const T* __restrict__ s, int64_t lds_, int K,
                                                        const int32_t* __restrict__ rows, int64_t n,
                                                        const T* __restrict__ shift_t,
                                                        const T* __restrict__ scale_t,
                                                        const T* __restrict__ shift_n,
                                                        const T* __restrict__ scale_n, int side,
                                                        double c, float* __restrict__ G, int KP,
                                                        double* __restrict__ G64, int64_t ldg,
                                                        float* __restrict__ bnd) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const T* src = s + (rows ? (int64_t)rows[i] : i) * lds_;
    float* g = G + i * KP;
    double sg = 0.0, sh = 0.0;
    for (int k = lane; k < K; k += 64) {
        const double v = src[k];
        const double gd = (v + (double)shift_t[k]) * (double)scale_t[k];
        const double hv = (v + (double)shift_n[k]) * (double)scale_n[k];
        // q from the unrounded g: |g|^2 and |h|^2 are each about count times larger than their
        // difference, so the rounding of g to fp32 (|g|^2 2^-24) must not enter it
        sg += gd * gd;
        sh += hv * hv;
        g[k] = (float)gd;
        if (G64) G64[i * ldg + k] = gd;
    }
    sg = wave_sum_d(sg);
    sh = wave_sum_d(sh);
    const double q = 0.5 * (sg - sh);
    for (int k = K + lane; k < KP; k += 64) {
        float v = 0.f;
        if (k == K) v = side == HYPMI_SIDE_ENROL ? (float)q : 1.f;
        if (k == K + 1) v = side == HYPMI_SIDE_ENROL ? 1.f : (float)(q + c);
        g[k] = v;
    }
    if (G64 && lane == 0) {  // the same row in float64: what a refinement pass scores with
        const double q64 = q + (side == HYPMI_SIDE_TEST ? c : 0.0);
        G64[i * ldg + K] = q64;
        G64[i * ldg + K + 1] = sqrt(sg);
        bnd[2 * i] = (float)sqrt(sg);
        bnd[2 * i + 1] = (float)fabs(q64);
    }
}

// x[i,:] /= denom[i]  (class sums to class means; a zero count gives NaN or inf as numpy does)
__global__ __launch_bounds__(256) void div_rows_kernel(float* __restrict__ x, int64_t ldx, int64_t n, int d,
                                                       const float* __restrict__ denom) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float dn = denom[i];
    for (int k = lane; k < d; k += 64) x[i * ldx + k] = x[i * ldx + k] / dn;
}

// out[ri[a], ci[b]] = blk[a, b]
__global__ __launch_bounds__(256) void scatter_block_kernel(const float* __restrict__ blk, int64_t ldb,
                                                            const int32_t* __restrict__ ri, int64_t nr,
                                                            const int32_t* __restrict__ ci, int64_t nc,
                                                            float* __restrict__ out, int64_t ldo) {
    const int64_t a = blockIdx.y;
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= nr || b >= nc) return;
    out[(int64_t)ri[a] * ldo + ci[b]] = blk[a * ldb + b];
}

}  // namespace

extern "C" int hypmi_pool_rows(const float* x, int64_t ldx, int d, const int32_t* order,
                               const int64_t* offsets, int64_t n_groups, int mean, float* out,
                               int64_t ldo, float* counts, void* stream) {
    HYPMI_REQUIRE(x && order && offsets && out, "pool_rows: null pointer");
    HYPMI_REQUIRE(d > 0 && ldx >= d && ldo >= d && n_groups >= 0, "pool_rows: bad shape");
    if (n_groups == 0) return HYPMI_OK;
    hipLaunchKernelGGL((pool_rows_kernel<float, float>), dim3((unsigned)((n_groups + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, ldx, d, order, offsets, n_groups, mean,
                       (const double*)nullptr, out, ldo, counts);
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

extern "C" int hypmi_pool_rows_f64(const double* x, int64_t ldx, int d, const int32_t* order,
                                   const int64_t* offsets, int64_t n_groups, const double* w, double* out,
                                   int64_t ldo, void* stream) {
    HYPMI_REQUIRE(x && order && offsets && out, "pool_rows_f64: null pointer");
    HYPMI_REQUIRE(d > 0 && ldx >= d && ldo >= d && n_groups >= 0, "pool_rows_f64: bad shape");
    if (n_groups == 0) return HYPMI_OK;
    hipLaunchKernelGGL((pool_rows_kernel<double, double>), dim3((unsigned)((n_groups + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, ldx, d, order, offsets, n_groups, 0, w, out, ldo,
                       (float*)nullptr);
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

extern "C" int hypmi_div_rows(float* x, int64_t ldx, int64_t n, int d, const float* denom, void* stream) {
    HYPMI_REQUIRE(x && denom && n >= 0 && d > 0 && ldx >= d, "div_rows: bad argument");
    if (n == 0) return HYPMI_OK;
    hipLaunchKernelGGL(div_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, ldx, n, d, denom);
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

extern "C" int hypmi_book_pack(const float* s, int64_t lds_, int K, const int32_t* rows, int64_t n,
                               const float* shift_t, const float* scale_t, const float* shift_n,
                               const float* scale_n, int side, double c, float* G, int kp,
                               void* stream) {
    HYPMI_REQUIRE(s && shift_t && scale_t && shift_n && scale_n && G, "book_pack: null pointer");
    HYPMI_REQUIRE(K > 0 && kp >= K + 2 && kp % 8 == 0 && lds_ >= K && n >= 0, "book_pack: bad shape");
    HYPMI_REQUIRE(side == HYPMI_SIDE_ENROL || side == HYPMI_SIDE_TEST, "book_pack: bad side");
    if (n == 0) return HYPMI_OK;
    hipLaunchKernelGGL(book_pack_kernel<float>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), s, lds_, K, rows, n, shift_t, scale_t, shift_n,
                       scale_n, side, c, G, kp, (double*)nullptr, (int64_t)0, (float*)nullptr);
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

extern "C" int hypmi_book_pack_f64(const double* s, int64_t lds_, int K, const int32_t* rows, int64_t n,
                                   const double* shift_t, const double* scale_t, const double* shift_n,
                                   const double* scale_n, int side, double c, float* G, int kp, void* stream) {
    return hypmi_book_pack_f64_rows(s, lds_, K, rows, n, shift_t, scale_t, shift_n, scale_n, side, c, G, kp,
                                    nullptr, 0, nullptr, stream);
}

extern "C" int hypmi_book_pack_f64_rows(const double* s, int64_t lds_, int K, const int32_t* rows, int64_t n,
                                        const double* shift_t, const double* scale_t, const double* shift_n,
                                        const double* scale_n, int side, double c, float* G, int kp,
                                        double* G64, int64_t ldg, float* bnd, void* stream) {
    HYPMI_REQUIRE(s && shift_t && scale_t && shift_n && scale_n && G, "book_pack_f64: null pointer");
    HYPMI_REQUIRE(K > 0 && kp >= K + 2 && kp % 8 == 0 && lds_ >= K && n >= 0, "book_pack_f64: bad shape");
    HYPMI_REQUIRE(side == HYPMI_SIDE_ENROL || side == HYPMI_SIDE_TEST, "book_pack_f64: bad side");
    HYPMI_REQUIRE(G64 == nullptr || (bnd != nullptr && ldg >= K + 2), "book_pack_f64: float64 rows need bounds and ldg >= K + 2");
    if (n == 0) return HYPMI_OK;
    hipLaunchKernelGGL(book_pack_kernel<double>, dim3((unsigned)((n + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), s, lds_, K, rows, n, shift_t, scale_t, shift_n,
                       scale_n, side, c, G, kp, G64, ldg, bnd);
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

extern "C" int hypmi_scatter_block(const float* blk, int64_t ldb, const int32_t* ri, int64_t nr,
                                   const int32_t* ci, int64_t nc, float* out, int64_t ldo,
                                   void* stream) {
    HYPMI_REQUIRE(blk && ri && ci && out, "scatter_block: null pointer");
    HYPMI_REQUIRE(nr >= 0 && nc >= 0 && ldb >= nc && nr <= 65535LL * 65535LL, "scatter_block: bad shape");
    if (nr == 0 || nc == 0) return HYPMI_OK;
    for (int64_t a0 = 0; a0 < nr; a0 += 65535) {
        const int64_t na = nr - a0 < 65535 ? nr - a0 : 65535;
        hipLaunchKernelGGL(scatter_block_kernel, dim3((unsigned)((nc + 255) / 256), (unsigned)na),
                           dim3(256), 0, static_cast<hipStream_t>(stream), blk + a0 * ldb, ldb, ri + a0,
                           na, ci, nc, out, ldo);
    }
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

// Double-precision side path: exact re-evaluation of the entries that fp32 cannot resolve.
//
// The matrix-core path accumulates K products in fp32, which leaves an absolute error of a few
// 1e-5 on scores of magnitude ~1e2 pieces that cancel to ~0; a relative tolerance on such a score
// cannot be met in fp32.  This file provides
//   * an fp64 embedding prologue (same plan, fp64 copies of the folded weights, f64 matrix cores
//     v_mfma_f64_16x16x4_f64) producing rows [g(0..K-1), q (+ c on the test side), |g|] in double and
//     a compact float2 (|g|, |q|) per row for error bounds;
//   * hypmi_refine_scores: scans a score matrix, flags entries whose fp32 error bound exceeds
//     rel_tol * |s| and replaces them by the fp64 value;
//   * hypmi_refine_topk: re-decides top-k membership of cohort entries that sit within the fp32 error
//     band of the selection boundary (see asnorm.hip for the selection itself).
#include "plan.h"

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int FM = 64, FN = 64, FK = 16, FLD = FK + 2;
constexpr int64_t F64_CHUNK_ROWS = 32768;

// C[m,n] = sum_k A[m,k] B[n,k] + bias[n], all fp64.  64 x 64 tile per workgroup, each wave a 32 x 32
// block as 2 x 2 tiles of v_mfma_f64_16x16x4_f64 (A: lane l holds A[l&15][k = l>>4], B likewise;
// D: column = lane & 15, row = (lane >> 4) + 4 * reg).
__global__ __launch_bounds__(256) void gemm_nt_f64_kernel(const double* __restrict__ A, int64_t lda,
                                                          const double* __restrict__ B, int64_t ldb,
                                                          const double* __restrict__ bias,
                                                          double* __restrict__ C, int64_t ldc, int64_t M,
                                                          int64_t N, int K) {
    __shared__ double As[FM * FLD];
    __shared__ double Bs[FN * FLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.y * FM, n0 = (int64_t)blockIdx.x * FN;
    const int lrow = tid >> 2, lk = (tid & 3) * 4;

    f64x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;

    for (int k0 = 0; k0 < K; k0 += FK) {
        double ra[4], rb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = k0 + lk + u;
            ra[u] = (m0 + lrow < M && kk < K) ? A[(m0 + lrow) * lda + kk] : 0.0;
            rb[u] = (n0 + lrow < N && kk < K) ? B[(n0 + lrow) * ldb + kk] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            As[lrow * FLD + lk + u] = ra[u];
            Bs[lrow * FLD + lk + u] = rb[u];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < FK / 4; ++s) {
            double a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t] = As[(wr * 32 + 16 * t + l15) * FLD + 4 * s + l4];
                b[t] = Bs[(wc * 32 + 16 * t + l15) * FLD + 4 * s + l4];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t col = n0 + wc * 32 + 16 * j + l15;
        const double bv = (bias != nullptr && col < N) ? bias[col] : 0.0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t row = m0 + wr * 32 + 16 * i + l4 + 4 * e;
                if (row < M && col < N) C[row * ldc + col] = acc[i][j][e] + bv;
            }
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void lnorm_rows_f64_kernel(double* __restrict__ y, int64_t n, int64_t ldy,
                                                             int d) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    double* p = y + row * ldy;
    double ss = 0.0;
    for (int k = lane; k < d; k += 64) ss += p[k] * p[k];
    ss = wave_sum(ss);
    const double scale = sqrt((double)d) / (sqrt(ss) + 1e-10);
    for (int k = lane; k < d; k += 64) p[k] *= scale;
}

// Z = [g | h] (2K doubles) -> G64 row [g, q (+c), |g|] and bound (|g|, |q (+c)|) as floats
__global__ __launch_bounds__(256) void pack_f64_kernel(const double* __restrict__ Z, int64_t n, int64_t ldz,
                                                       int K, int side, double c, double* __restrict__ G,
                                                       int64_t ldg, float* __restrict__ bnd) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const double* z = Z + row * ldz;
    double* g = G + row * ldg;
    double sg = 0.0, sh = 0.0;
    for (int k = lane; k < K; k += 64) {
        const double gv = z[k], hv = z[K + k];
        sg += gv * gv;
        sh += hv * hv;
        g[k] = gv;
    }
    sg = wave_sum(sg);
    sh = wave_sum(sh);
    if (lane == 0) {
        const double q = 0.5 * (sg - sh) + (side == HYPMI_SIDE_TEST ? c : 0.0);
        g[K] = q;
        g[K + 1] = sqrt(sg);
        bnd[2 * row] = (float)sqrt(sg);
        bnd[2 * row + 1] = (float)fabs(q);
    }
}

__global__ __launch_bounds__(256) void copy_rows_f64_kernel(const double* __restrict__ in, int64_t ldi,
                                                            int64_t n, int d, double* __restrict__ out,
                                                            int64_t ldo) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    for (int k = lane; k < d; k += 64) out[row * ldo + k] = in[row * ldi + k];
}

int launch_gemm_f64(const double* A, int64_t lda, const double* B, int64_t ldb, const double* bias,
                    double* C, int64_t ldc, int64_t M, int64_t N, int K, hipStream_t stream) {
    if (M == 0 || N == 0) return HYPMI_OK;
    const int64_t gy = (M + FM - 1) / FM, gx = (N + FN - 1) / FN;
    HYPMI_REQUIRE(gy <= 65535 && gx <= 0x7fffffffLL, "gemm_f64: grid too large");
    hipLaunchKernelGGL(gemm_nt_f64_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, stream, A, lda, B,
                       ldb, bias, C, ldc, M, N, K);
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

// ---------------------------------------------------------------------------------------------
// Score refinement.  A block scans RB x CB scores; entries with
//     rel_tol * |s|  <  coeff * (|g1_i| |g2_j| + |q1_i| + |q2_j|)
// (the fp32 error bound of that entry) are queued in LDS and then recomputed in double by 16-lane
// groups: s = <g1_i, g2_j> + q1_i + q2_j.
constexpr int RB = 8, CB = 1024;

__global__ __launch_bounds__(256) void refine_scores_kernel(float* __restrict__ S, int64_t lds_, int64_t n1,
                                                            int64_t n2, const double* __restrict__ G1,
                                                            const double* __restrict__ G2, int64_t ldg,
                                                            int K, const float* __restrict__ b1,
                                                            const float* __restrict__ b2, float rel_tol,
                                                            float coeff,
                                                            unsigned long long* __restrict__ counter) {
    __shared__ unsigned short queue[RB * CB];
    __shared__ unsigned qlen;
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * RB, c0 = (int64_t)blockIdx.x * CB;
    if (tid == 0) qlen = 0;
    __syncthreads();
    const int64_t col = c0 + tid * 4;
    float n2v[4], q2v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const bool ok = col + u < n2;
        n2v[u] = ok ? b2[2 * (col + u)] : 0.f;
        q2v[u] = ok ? b2[2 * (col + u) + 1] : 0.f;
    }
    for (int rr = 0; rr < RB; ++rr) {
        const int64_t row = r0 + rr;
        if (row >= n1) break;
        const float n1v = b1[2 * row], q1v = b1[2 * row + 1];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (col + u < n2) {
                const float s = S[row * lds_ + col + u];
                const float bound = coeff * (n1v * n2v[u] + q1v + q2v[u]);
                if (!(rel_tol * fabsf(s) >= bound)) {  // also catches NaN
                    const unsigned pos = atomicAdd(&qlen, 1u);
                    queue[pos] = (unsigned short)(rr * CB + tid * 4 + u);
                }
            }
        }
    }
    __syncthreads();
    const unsigned total = qlen;
    const int grp = tid >> 4, sub = tid & 15;
    for (unsigned e = grp; e < total; e += 16) {
        const unsigned code = queue[e];
        const int64_t i = r0 + code / CB, j = c0 + code % CB;
        const double* g1 = G1 + i * ldg;
        const double* g2 = G2 + j * ldg;
        double acc = 0.0;
        for (int k = sub; k < K; k += 16) acc = fma(g1[k], g2[k], acc);
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
        if (sub == 0) S[i * lds_ + j] = (float)(acc + g1[K] + g2[K]);
    }
    if (counter && tid == 0 && total) atomicAdd(counter, (unsigned long long)total);
}

// ---------------------------------------------------------------------------------------------
// Refinement of adaptively normalised scores.  A normalised score is a difference of numbers of
// magnitude (|x| + |mean|) / std that cancel to about 0 for most non-target trials, so its fp32 value
// cannot hold a relative tolerance near zero.  A wave looks at 64 consecutive tests of one enrolment row:
// an entry whose fp32 error estimate
//     coeff * [ (|x| + |mz|) iz + (|x| + |mt|) it + 8 + 16 (|z| + |t|) ]
// (x raw score, (mz, iz) / (mt, it) the fp32 cohort statistics of the two parts, z / t the two
// normalised terms: rounding of the inputs and of the means, about four standard deviations of
// accumulation error, and the up to four bits the one-pass variance may lose) exceeds rel_tol * |s| is
// recomputed by the whole wave from the float64 matrices over the same cohort selections, with the
// two-pass mean and deviation of hyperion/score_norm/adapt_s_norm.py:58-63,81-88.
__global__ __launch_bounds__(256) void asnorm_refine_kernel(
    const double* __restrict__ S64, int64_t lds64, const double* __restrict__ E64, int64_t lde,
    const double* __restrict__ T64, int64_t ldt64, int64_t ne, int64_t nt, const uint16_t* __restrict__ t_idx,
    int64_t ldti, const uint16_t* __restrict__ e_idx, int64_t ldei, int k, const float* __restrict__ mz,
    const float* __restrict__ iz, int64_t ldz, const float* __restrict__ mt, const float* __restrict__ it,
    int64_t ldt, float rel_tol, float coeff, double std_floor, float* __restrict__ out, int64_t ldo,
    unsigned long long* __restrict__ counter) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t e = blockIdx.y, t0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
    if (e >= ne || t0 >= nt) return;
    const int64_t t = t0 + lane;
    bool flag = false;
    if (t < nt) {
        const float x = (float)S64[e * lds64 + t];
        const float m_z = mz[e * ldz + t], i_z = iz[e * ldz + t], m_t = mt[t * ldt + e], i_t = it[t * ldt + e];
        const float z = (x - m_z) * i_z, tn = (x - m_t) * i_t;
        const float bound = coeff * ((fabsf(x) + fabsf(m_z)) * i_z + (fabsf(x) + fabsf(m_t)) * i_t + 8.f +
                                     16.f * (fabsf(z) + fabsf(tn)));
        flag = !(rel_tol * fabsf(out[e * ldo + t]) >= bound);  // also catches NaN
    }
    unsigned long long todo = __ballot(flag);
    if (!todo) return;
    const unsigned n_todo = (unsigned)__popcll(todo);
    const double* erow = E64 + e * lde;
    const double inv_k = 1.0 / (double)k;
    while (todo) {
        const int64_t tt = t0 + __builtin_ctzll(todo);
        todo &= todo - 1;
        const double* trow = T64 + tt * ldt64;
        double sz = 0.0, st = 0.0;
        for (int j = lane; j < k; j += 64) {
            const int64_t g = (int64_t)(j >> 3) * 8, u = j & 7;
            sz += erow[t_idx[g * ldti + tt * 8 + u]];
            st += trow[e_idx[g * ldei + e * 8 + u]];
        }
        const double mean_z = wave_sum(sz) * inv_k, mean_t = wave_sum(st) * inv_k;
        double vz = 0.0, vt = 0.0;
        for (int j = lane; j < k; j += 64) {
            const int64_t g = (int64_t)(j >> 3) * 8, u = j & 7;
            const double dz = erow[t_idx[g * ldti + tt * 8 + u]] - mean_z;
            const double dt = trow[e_idx[g * ldei + e * 8 + u]] - mean_t;
            vz = fma(dz, dz, vz);
            vt = fma(dt, dt, vt);
        }
        const double sd_z = fmax(sqrt(wave_sum(vz) * inv_k), std_floor);
        const double sd_t = fmax(sqrt(wave_sum(vt) * inv_k), std_floor);
        if (lane == 0) {
            const double x = S64[e * lds64 + tt];
            out[e * ldo + tt] = (float)(((x - mean_z) / sd_z + (x - mean_t) / sd_t) / sqrt(2.0));
        }
    }
    if (counter && lane == 0) atomicAdd(counter, (unsigned long long)n_todo);
}

}  // namespace

extern "C" int hypmi_gemm_nt_f64(const double* A, int64_t m, int64_t lda, const double* B, int64_t n,
                                 int64_t ldb, int k, const double* bias, double* C, int64_t ldc,
                                 void* stream_) {
    HYPMI_REQUIRE(m >= 0 && n >= 0 && k > 0 && lda >= k && ldb >= k && ldc >= n, "gemm_nt_f64: bad shape");
    if (m == 0 || n == 0) return HYPMI_OK;
    HYPMI_REQUIRE(A && B && C, "gemm_nt_f64: null pointer");
    return launch_gemm_f64(A, lda, B, ldb, bias, C, ldc, m, n, k, static_cast<hipStream_t>(stream_));
}

extern "C" int64_t hypmi_prologue_f64_workspace_bytes(const hypmi_plan* p, int64_t n) {
    if (!p || n <= 0) return 0;
    const int64_t rows = n < F64_CHUNK_ROWS ? n : F64_CHUNK_ROWS;
    return rows * (2 * (int64_t)round_up(p->max_dim, 4) + round_up(2 * p->K, 4)) * (int64_t)sizeof(double);
}

// x: [n x d_in] fp64.  G64: [n x ldg] rows [g, q (+ c), |g|], ldg >= K + 2.  bnd: [n x 2] floats.
extern "C" int hypmi_prologue_f64(const hypmi_plan* p, const double* x, int64_t n, int64_t ldx, int side,
                                  double* G64, int64_t ldg, float* bnd, void* ws, int64_t ws_bytes,
                                  void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HYPMI_REQUIRE(p && p->K > 0, "prologue_f64: plan has no PLDA projection");
    HYPMI_REQUIRE(x && G64 && bnd, "prologue_f64: null pointer");
    HYPMI_REQUIRE(n >= 0 && ldx >= p->d_in && ldg >= p->K + 2, "prologue_f64: bad shape");
    HYPMI_REQUIRE(side == HYPMI_SIDE_ENROL || side == HYPMI_SIDE_TEST, "prologue_f64: bad side %d", side);
    if (n == 0) return HYPMI_OK;
    const int64_t need = hypmi_prologue_f64_workspace_bytes(p, n);
    HYPMI_REQUIRE(ws && ws_bytes >= need, "prologue_f64: workspace too small (%lld < %lld)",
                  (long long)ws_bytes, (long long)need);
    const int64_t chunk = n < F64_CHUNK_ROWS ? n : F64_CHUNK_ROWS;
    const int64_t ld = round_up(p->max_dim, 4), ldz = round_up(2 * p->K, 4);
    double* buf[2] = {static_cast<double*>(ws), static_cast<double*>(ws) + chunk * ld};
    double* z = static_cast<double*>(ws) + 2 * chunk * ld;
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
        const int64_t rows = n - r0 < chunk ? n - r0 : chunk;
        const double* in = x + r0 * ldx;
        int64_t ldin = ldx;
        int which = 0;
        for (int i = 0; i < p->n_seg; ++i) {
            const Segment& s = p->seg[i];
            if (!s.Wt64 && !s.lnorm) continue;
            double* dst;
            if (s.Wt64) {
                dst = buf[which];
                int rc = launch_gemm_f64(in, ldin, s.Wt64, s.ldw, s.bias64, dst, ld, rows, s.d_out, s.d_in,
                                         stream);
                if (rc != HYPMI_OK) return rc;
                which ^= 1;
                ldin = ld;
            } else if (in == x + r0 * ldx) {  // length norm only: work on a copy, never on the input
                dst = buf[which];
                hipLaunchKernelGGL(copy_rows_f64_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                                   stream, in, ldin, rows, s.d_in, dst, ld);
                which ^= 1;
                ldin = ld;
            } else {
                dst = const_cast<double*>(in);
            }
            if (s.lnorm)
                hipLaunchKernelGGL(lnorm_rows_f64_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                                   stream, dst, rows, ldin, s.d_out);
            in = dst;
        }
        int rc = launch_gemm_f64(in, ldin, p->fin.Wt64, p->fin.ldw, p->fin.bias64, z, ldz, rows, 2 * p->K,
                                 p->fin.d_in, stream);
        if (rc != HYPMI_OK) return rc;
        hipLaunchKernelGGL(pack_f64_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, z, rows,
                           ldz, p->K, side, p->c, G64 + r0 * ldg, ldg, bnd + 2 * r0);
        HYPMI_HIP(hipGetLastError());
    }
    return HYPMI_OK;
}

extern "C" int hypmi_refine_scores(float* S, int64_t lds_, int64_t n1, int64_t n2, const double* G1,
                                   const double* G2, int64_t ldg, int K, const float* bnd1,
                                   const float* bnd2, float rel_tol, float err_coeff,
                                   unsigned long long* counter, void* stream_) {
    HYPMI_REQUIRE(S && G1 && G2 && bnd1 && bnd2, "refine_scores: null pointer");
    HYPMI_REQUIRE(n1 >= 0 && n2 >= 0 && lds_ >= n2 && K > 0 && ldg >= K + 2, "refine_scores: bad shape");
    HYPMI_REQUIRE(rel_tol > 0.f && err_coeff > 0.f, "refine_scores: tolerances must be positive");
    if (n1 == 0 || n2 == 0) return HYPMI_OK;
    const int64_t gx = (n2 + CB - 1) / CB;
    for (int64_t y0 = 0; y0 < (n1 + RB - 1) / RB; y0 += 65535) {
        const int64_t gy = ((n1 + RB - 1) / RB - y0 < 65535) ? (n1 + RB - 1) / RB - y0 : 65535;
        const int64_t row0 = y0 * RB;
        hipLaunchKernelGGL(refine_scores_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0,
                           static_cast<hipStream_t>(stream_), S + row0 * lds_, lds_, n1 - row0, n2,
                           G1 + row0 * ldg, G2, ldg, K, bnd1 + 2 * row0, bnd2, rel_tol, err_coeff, counter);
    }
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

extern "C" int hypmi_asnorm_refine(const double* S64, int64_t lds64, const double* E64, int64_t lde,
                                   const double* T64, int64_t ldt64, int64_t ne, int64_t nt, int nc,
                                   const uint16_t* t_idx, int64_t ldti, const uint16_t* e_idx, int64_t ldei, int k,
                                   const float* mz, const float* iz, int64_t ldz, const float* mt, const float* it,
                                   int64_t ldt, float rel_tol, float err_coeff, float std_floor, float* out,
                                   int64_t ldo, unsigned long long* counter, void* stream_) {
    HYPMI_REQUIRE(S64 && E64 && T64 && t_idx && e_idx && mz && iz && mt && it && out, "asnorm_refine: null pointer");
    HYPMI_REQUIRE(ne >= 0 && nt >= 0 && nc > 0 && k > 0 && k <= nc, "asnorm_refine: bad shape");
    HYPMI_REQUIRE(lds64 >= nt && lde >= nc && ldt64 >= nc && ldz >= nt && ldt >= ne && ldo >= nt && ldti >= nt &&
                      ldei >= ne,
                  "asnorm_refine: leading dimension too small");
    HYPMI_REQUIRE(rel_tol > 0.f && err_coeff > 0.f, "asnorm_refine: tolerances must be positive");
    if (ne == 0 || nt == 0) return HYPMI_OK;
    const int64_t gx = (nt + 255) / 256;
    HYPMI_REQUIRE(gx <= 0x7fffffff, "asnorm_refine: too many tests per call");
    for (int64_t e0 = 0; e0 < ne; e0 += 65535) {
        const int64_t rows = ne - e0 < 65535 ? ne - e0 : 65535;
        hipLaunchKernelGGL(asnorm_refine_kernel, dim3((unsigned)gx, (unsigned)rows), dim3(256), 0,
                           static_cast<hipStream_t>(stream_), S64 + e0 * lds64, lds64, E64 + e0 * lde, lde, T64, ldt64,
                           rows, nt, t_idx, ldti, e_idx + e0 * 8, ldei, k, mz + e0 * ldz, iz + e0 * ldz, ldz, mt + e0,
                           it + e0, ldt, rel_tol, err_coeff, (double)std_floor, out + e0 * ldo, ldo, counter);
    }
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

// Second-order statistics for the training side of the back-end (whitening, LDA, PLDA statistics):
//   C = sum_s w_s x_s x_s^T   (d x d, float64),   F = sum_s w_s x_s
// over n float64 rows.  Replaces np.dot(x.T, x) / np.dot(wx.T, x) (hyperion/pdfs/core/normal.py:102-113,
// hyperion/pdfs/plda/plda_base.py:268-272) and, with w = 1 / N_class and the class means, the per-class
// scatter loop of hyperion/transforms/sb_sw.py:36-43.
//
// Gram kernel: f64 matrix cores (v_mfma_f64_16x16x4_f64, the sample index is the contraction).  A
// wave owns a 64 x 64 block of C (4 x 4 tiles) and a range of samples; its operands come straight
// from global memory in operand layout (lane l reads feature f0 + (l & 15) of sample s0 + (l >> 4):
// four 128-byte row segments per load), the next step's eight loads are in flight while the sixteen
// MFMAs of the current one run, and the block is added to C with float64 atomics at the end -- or, in
// ordered mode, written to the wave's own slot of a workspace and summed in slot order by a second
// kernel, which makes the result the same from run to run.  Only
// blocks on or above the diagonal are computed; off-diagonal blocks are added to both halves.
// A wave that ends early returns before anything collective: the kernel has no barrier.
//
// The file also holds the per-class posterior step of SPLDA training (splda_posterior_kernel).
#include "common.h"

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int GB = 64;  // features per block side

__global__ __launch_bounds__(256) void gram_f64_kernel(const double* __restrict__ X, int64_t ldx, int64_t n,
                                                       int d, const double* __restrict__ w,
                                                       int64_t samples_per_wave, double* __restrict__ C,
                                                       int64_t ldc, double* __restrict__ partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    // blockIdx.x runs over the block pairs (bi <= bj) row by row
    const int nblk = (d + GB - 1) / GB;
    int bi = 0, bj = (int)blockIdx.x;
    while (bj >= nblk - bi) {
        bj -= nblk - bi;
        ++bi;
    }
    bj += bi;
    const int f0 = bi * GB, g0 = bj * GB;
    const int64_t s_begin = ((int64_t)blockIdx.y * 4 + wave) * samples_per_wave;
    int64_t s_end = s_begin + samples_per_wave;
    if (s_end > n) s_end = n;
    if (s_begin >= s_end) return;

    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;

    // Loads are unconditional (clamped to an existing feature and sample) and masked afterwards, so
    // that the eight loads of a step are issued back to back and none waits for another.
    bool fa[4], fb[4];  // feature columns of this lane that exist
    int ca[4], cb[4];   // ... clamped into the row
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        fa[t] = f0 + 16 * t + l15 < d;
        fb[t] = g0 + 16 * t + l15 < d;
        ca[t] = fa[t] ? f0 + 16 * t + l15 : d - 1;
        cb[t] = fb[t] ? g0 + 16 * t + l15 : d - 1;
    }
    auto fetch = [&](int64_t s0, double (&a)[4], double (&b)[4]) {
        const int64_t s = s0 + l4;
        const bool live = s < s_end;
        const int64_t sc = live ? s : s_end - 1;
        const double* row = X + sc * ldx;
        double va[4], vb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            va[t] = row[ca[t]];
            vb[t] = row[cb[t]];
        }
        const double ws = w ? w[sc] : 1.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = (live && fa[t]) ? va[t] : 0.0;
            b[t] = (live && fb[t]) ? vb[t] * ws : 0.0;
        }
    };

    double a[4], b[4], na[4], nb[4];
    fetch(s_begin, a, b);
    for (int64_t s0 = s_begin; s0 < s_end; s0 += 4) {
        fetch(s0 + 4, na, nb);  // past the range: zeros
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a[t] = na[t];
            b[t] = nb[t];
        }
    }
    // D layout: column = lane & 15, row = (lane >> 4) + 4 * reg
    if (partial) {
        // ordered mode: the block goes to this wave's own slot (pair, slab, wave); gram_reduce_kernel
        // adds the slots of a pair in slot order, so the sum does not depend on scheduling
        double* slot = partial + (((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 4 + wave) * (GB * GB);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) slot[(16 * i + l4 + 4 * e) * GB + 16 * j + l15] = acc[i][j][e];
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = f0 + 16 * i + l4 + 4 * e, g = g0 + 16 * j + l15;
                if (f < d && g < d) {
                    atomicAdd(&C[(int64_t)f * ldc + g], acc[i][j][e]);
                    if (bi != bj) atomicAdd(&C[(int64_t)g * ldc + f], acc[i][j][e]);
                }
            }
}

// Adds the slots of every block pair in slot order into C (both halves for off-diagonal pairs).
// One thread per element of a 64 x 64 block.
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double* __restrict__ partial, int64_t slots,
                                                          int d, double* __restrict__ C, int64_t ldc) {
    const int nblk = (d + GB - 1) / GB;
    int bi = 0, bj = (int)blockIdx.x;
    while (bj >= nblk - bi) {
        bj -= nblk - bi;
        ++bi;
    }
    bj += bi;
    const int e = blockIdx.y * 256 + threadIdx.x;  // element of the block
    const int f = bi * GB + e / GB, g = bj * GB + e % GB;
    const double* p = partial + (int64_t)blockIdx.x * slots * (GB * GB) + e;
    double sum = 0.0;
    for (int64_t k = 0; k < slots; ++k) sum += p[k * (GB * GB)];
    if (f < d && g < d) {
        C[(int64_t)f * ldc + g] = sum;
        if (bi != bj) C[(int64_t)g * ldc + f] = sum;
    }
}

// F[c] += sum_s w_s X[s, c]: one thread per column and row slab, coalesced across columns.  With
// `partial` the slab sums go to partial[slab, c] and colsum_reduce_kernel adds them in slab order.
__global__ __launch_bounds__(256) void colsum_f64_kernel(const double* __restrict__ X, int64_t ldx, int64_t n,
                                                         int d, const double* __restrict__ w,
                                                         int64_t rows_per_block, double* __restrict__ F,
                                                         double* __restrict__ partial) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    double s = 0.0;
    for (int64_t r = r0; r < r1; ++r) s += X[r * ldx + c] * (w ? w[r] : 1.0);
    if (partial) partial[(int64_t)blockIdx.y * d + c] = s;
    else if (r1 > r0) atomicAdd(&F[c], s);
}

__global__ __launch_bounds__(256) void colsum_reduce_kernel(const double* __restrict__ partial, int64_t slabs,
                                                            int d, double* __restrict__ F) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    double s = 0.0;
    for (int64_t k = 0; k < slabs; ++k) s += partial[k * d + c];
    F[c] = s;
}

// Sum of m doubles in a fixed order: thread t adds elements t, t + 256, ... and the 256 partial sums
// are added pairwise in a fixed tree.  One workgroup.
__global__ __launch_bounds__(256) void ordered_sum_kernel(const double* __restrict__ v, int64_t m,
                                                          double* __restrict__ out) {
    __shared__ double part[256];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < m; i += 256) s += v[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = part[0];
}

// Latent posterior of the two-covariance model in the eigenbasis of V W V^T = Q diag(lam) Q^T, where
// the posterior precision I + N_i V W V^T of every class is diagonal:
//   T[i, j] <- T[i, j] / (1 + N_i lam_j)                      (posterior mean, eigenbasis)
//   logpy_i  = -K/2 log(2 pi) + 1/2 (sum_j log(1 + N_i lam_j) - sum_j T_new[i, j] T_old[i, j])
// A wave walks rows with a grid stride and adds its share of sum_i logpy_i with one atomic.
__global__ __launch_bounds__(256) void splda_posterior_kernel(double* __restrict__ T, int64_t ldt, int64_t m,
                                                              int K, const double* __restrict__ N,
                                                              const double* __restrict__ lam,
                                                              double* __restrict__ logpy,
                                                              double* __restrict__ logpy_sum) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t waves = (int64_t)gridDim.x * 4;
    const double c0 = -0.5 * K * 1.8378770664093454835606594728112;  // log(2 pi)
    double total = 0.0;
    for (int64_t row = wave; row < m; row += waves) {
        const double ni = N[row];
        double* t = T + row * ldt;
        double part = 0.0;
        for (int j = lane; j < K; j += 64) {
            const double p = 1.0 + ni * lam[j];
            const double g = t[j];
            const double y = g / p;
            t[j] = y;
            part += log(p) - y * g;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        const double lp = c0 + 0.5 * part;
        if (lane == 0 && logpy) logpy[row] = lp;
        total += lp;
    }
    if (lane == 0 && logpy_sum && total != 0.0) atomicAdd(logpy_sum, total);
}

}  // namespace

extern "C" int hypmi_splda_posterior(double* T, int64_t m, int64_t ldt, int K, const double* N,
                                     const double* lam, double* logpy, double* logpy_sum, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HYPMI_REQUIRE(m >= 0 && K > 0 && ldt >= K, "splda_posterior: bad shape");
    if (logpy_sum) HYPMI_HIP(hipMemsetAsync(logpy_sum, 0, sizeof(double), stream));
    if (m == 0) return HYPMI_OK;
    HYPMI_REQUIRE(T && N && lam, "splda_posterior: null pointer");
    int64_t blocks = (m + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    // with the per-class values kept, their total is taken from them in a fixed order (same bits from run
    // to run); without, the waves add their shares with atomics
    hipLaunchKernelGGL(splda_posterior_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, T, ldt, m, K, N, lam,
                       logpy, logpy ? nullptr : logpy_sum);
    if (logpy && logpy_sum)
        hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(256), 0, stream, logpy, m, logpy_sum);
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

namespace {
struct GramGrid {
    int npairs;
    int64_t chunks, samples_per_wave, slabs, rows_per_block;
};

GramGrid gram_grid(int64_t n, int d) {
    GramGrid g;
    const int nb = (d + GB - 1) / GB;
    g.npairs = nb * (nb + 1) / 2;  // blocks on and above the diagonal
    // about 2048 workgroups in all; a wave walks at least 64 samples
    g.chunks = (2048 + g.npairs - 1) / g.npairs;
    const int64_t max_chunks = (n + 4 * 64 - 1) / (4 * 64);
    if (g.chunks > max_chunks) g.chunks = max_chunks;
    if (g.chunks > 65535) g.chunks = 65535;
    if (g.chunks < 1) g.chunks = 1;
    g.samples_per_wave = (n + g.chunks * 4 - 1) / (g.chunks * 4);
    g.samples_per_wave = (g.samples_per_wave + 3) / 4 * 4;
    g.slabs = (n + 4095) / 4096;
    if (g.slabs > 65535) g.slabs = 65535;
    if (g.slabs < 1) g.slabs = 1;
    g.rows_per_block = (n + g.slabs - 1) / g.slabs;
    return g;
}

int gram_run(const double* X, int64_t n, int64_t ldx, int d, const double* w, double* C, int64_t ldc, double* F,
             double* ws, hipStream_t stream) {
    HYPMI_HIP(hipMemsetAsync(C, 0, (size_t)d * (size_t)ldc * sizeof(double), stream));
    if (F) HYPMI_HIP(hipMemsetAsync(F, 0, (size_t)d * sizeof(double), stream));
    if (n == 0) return HYPMI_OK;
    HYPMI_REQUIRE(X, "gram_f64: null pointer");
    const GramGrid g = gram_grid(n, d);
    const int64_t slots = g.chunks * 4;
    const size_t gram_doubles = (size_t)g.npairs * slots * GB * GB;
    if (ws) HYPMI_HIP(hipMemsetAsync(ws, 0, gram_doubles * sizeof(double), stream));  // waves without samples
    hipLaunchKernelGGL(gram_f64_kernel, dim3((unsigned)g.npairs, (unsigned)g.chunks), dim3(256), 0, stream, X, ldx, n,
                       d, w, g.samples_per_wave, C, ldc, ws);
    if (ws)
        hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)g.npairs, GB * GB / 256), dim3(256), 0, stream, ws,
                           slots, d, C, ldc);
    if (F) {
        double* fpart = ws ? ws + gram_doubles : nullptr;
        hipLaunchKernelGGL(colsum_f64_kernel, dim3((unsigned)((d + 255) / 256), (unsigned)g.slabs), dim3(256), 0,
                           stream, X, ldx, n, d, w, g.rows_per_block, F, fpart);
        if (ws)
            hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, stream, fpart,
                               g.slabs, d, F);
    }
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}
}  // namespace

extern "C" int hypmi_gram_f64(const double* X, int64_t n, int64_t ldx, int d, const double* w, double* C,
                              int64_t ldc, double* F, void* stream_) {
    HYPMI_REQUIRE(n >= 0 && d > 0 && ldx >= d && ldc >= d, "gram_f64: bad shape");
    HYPMI_REQUIRE(C, "gram_f64: null result");
    return gram_run(X, n, ldx, d, w, C, ldc, F, nullptr, static_cast<hipStream_t>(stream_));
}

extern "C" int64_t hypmi_gram_f64_workspace_bytes(int64_t n, int d) {
    if (n <= 0 || d <= 0) return 0;
    const GramGrid g = gram_grid(n, d);
    return ((int64_t)g.npairs * g.chunks * 4 * GB * GB + g.slabs * (int64_t)d) * (int64_t)sizeof(double);
}

extern "C" int hypmi_gram_f64_ordered(const double* X, int64_t n, int64_t ldx, int d, const double* w, double* C,
                                      int64_t ldc, double* F, void* ws, int64_t ws_bytes, void* stream_) {
    HYPMI_REQUIRE(n >= 0 && d > 0 && ldx >= d && ldc >= d, "gram_f64_ordered: bad shape");
    HYPMI_REQUIRE(C, "gram_f64_ordered: null result");
    const int64_t need = hypmi_gram_f64_workspace_bytes(n, d);
    HYPMI_REQUIRE(need == 0 || (ws && ws_bytes >= need), "gram_f64_ordered: workspace too small (%lld < %lld bytes)",
                  (long long)ws_bytes, (long long)need);
    return gram_run(X, n, ldx, d, w, C, ldc, F, n > 0 ? static_cast<double*>(ws) : nullptr,
                    static_cast<hipStream_t>(stream_));
}

// Plans (device-resident folded models) and the embedding prologue:
//   x -> [affine (+ length norm)]* -> (g, h) = y [At | An] + [bt | bn] -> packed row
//   [g, q, 1] or [g, 1, q + c],  q = (|g|^2 - |h|^2) / 2.
// The last length-normalised segment and the projection run as ONE kernel (prologue_fused_kernel:
// the normalised intermediate never leaves the chip); earlier segments, if any, are a GEMM
// (gemm_nt.hip) plus a row kernel each.  q is accumulated in double by the packing kernel, because it
// feeds every score of its row.
#include <stdarg.h>
#include <string.h>

#include <vector>

#include <mutex>

#include "common.h"
#include "gemm_tile.h"
#include "pair_rows.h"

// ------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

void hypmi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* hypmi_last_error(void) { return g_err; }
extern "C" int hypmi_version(void) { return 1; }

// ------------------------------------------------------------------------------------ plan
#include "plan.h"

namespace {

template <typename T>
int to_device(const std::vector<T>& host, T** out) {
    T* dev = nullptr;
    if (hipMalloc(&dev, host.size() * sizeof(T)) != hipSuccess) {
        hypmi_set_error("plan: device allocation of %zu bytes failed", host.size() * sizeof(T));
        return HYPMI_ENOMEM;
    }
    HYPMI_HIP(hipMemcpy(dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = dev;
    return HYPMI_OK;
}

// A is [d_in x d_out] row-major on the host; the device copies are transposed, [d_out x ld] with
// ld = d_in rounded up to 4 and zero padding, once in fp32 (matrix-core path) and once in fp64
// (refinement path).
int upload_transposed(const double* A, int d_in, int d_out, Segment* seg) {
    const int64_t ld = round_up(d_in, 4);
    std::vector<float> t32((size_t)d_out * ld, 0.f);
    std::vector<double> t64((size_t)d_out * ld, 0.0);
    for (int i = 0; i < d_in; ++i)
        for (int j = 0; j < d_out; ++j) {
            t64[(size_t)j * ld + i] = A[(size_t)i * d_out + j];
            t32[(size_t)j * ld + i] = (float)A[(size_t)i * d_out + j];
        }
    seg->ldw = ld;
    int rc = to_device(t32, &seg->Wt);
    if (rc == HYPMI_OK) rc = to_device(t64, &seg->Wt64);
    return rc;
}

int upload_vec(const double* b, int n, Segment* seg) {
    std::vector<float> t32(n);
    std::vector<double> t64(b, b + n);
    for (int i = 0; i < n; ++i) t32[i] = (float)b[i];
    int rc = to_device(t32, &seg->bias);
    if (rc == HYPMI_OK) rc = to_device(t64, &seg->bias64);
    return rc;
}

}  // namespace

extern "C" void hypmi_plan_destroy(hypmi_plan* p) {
    if (!p) return;
    for (int i = 0; i < HYPMI_MAX_SEG; ++i) {
        p->seg[i].release();
    }
    p->fin.release();
    p->sfin.release();
    if (p->col_map) (void)hipFree(p->col_map);
    p->fused.release();
    delete p;
}

namespace {
int static_scratch(void** out);
}

extern "C" int hypmi_plan_create(const hypmi_plan_desc* d, hypmi_plan** out) {
    HYPMI_REQUIRE(d && out, "plan_create: null argument");
    HYPMI_REQUIRE(d->d_in > 0, "plan_create: d_in must be positive");
    HYPMI_REQUIRE(d->n_seg >= 0 && d->n_seg <= HYPMI_MAX_SEG, "plan_create: n_seg out of range");
    HYPMI_REQUIRE(d->K >= 0, "plan_create: negative K");
    const bool shared = d->K > 0 && d->scale_t != nullptr;
    HYPMI_REQUIRE(d->K == 0 || (d->At && (d->An || shared)), "plan_create: At/An required when K > 0");
    HYPMI_REQUIRE(!shared || (d->shift_t && d->shift_n && d->scale_n),
                  "plan_create: the shared-basis form needs shift_t, scale_t, shift_n and scale_n");
    hypmi_plan* p = new hypmi_plan();
    p->d_in = d->d_in;
    p->n_seg = d->n_seg;
    p->K = d->K;
    p->c = d->c;
    p->max_dim = d->d_in;
    int cur = d->d_in;
    int rc = HYPMI_OK;
    for (int i = 0; i < d->n_seg && rc == HYPMI_OK; ++i) {
        const hypmi_segment& s = d->seg[i];
        Segment& t = p->seg[i];
        t.d_in = cur;
        t.d_out = s.d_out;
        t.lnorm = s.lnorm != 0;
        if (s.d_out <= 0 || (!s.A && s.d_out != cur)) {
            hypmi_set_error("plan_create: segment %d has d_out=%d for d_in=%d", i, s.d_out, cur);
            rc = HYPMI_EINVAL;
            break;
        }
        if (s.A) {
            rc = upload_transposed(s.A, cur, s.d_out, &t);
            if (rc == HYPMI_OK && s.b) rc = upload_vec(s.b, s.d_out, &t);
        } else if (s.b) {
            // bias without a matrix: identity weight keeps one code path
            std::vector<double> I((size_t)cur * cur, 0.0);
            for (int k = 0; k < cur; ++k) I[(size_t)k * cur + k] = 1.0;
            rc = upload_transposed(I.data(), cur, cur, &t);
            if (rc == HYPMI_OK) rc = upload_vec(s.b, s.d_out, &t);
        }
        cur = s.d_out;
        if (cur > p->max_dim) p->max_dim = cur;
    }
    p->d_last = cur;
    if (rc == HYPMI_OK && d->K > 0) {
        const int K = d->K;
        std::vector<double> W((size_t)cur * 2 * K), b(2 * K, 0.0);
        if (shared) {
            // the two-projection form of the same model, for the float64 path
            for (int i = 0; i < cur; ++i)
                for (int j = 0; j < K; ++j) {
                    W[(size_t)i * 2 * K + j] = d->At[(size_t)i * K + j] * d->scale_t[j];
                    W[(size_t)i * 2 * K + K + j] = d->At[(size_t)i * K + j] * d->scale_n[j];
                }
            for (int j = 0; j < K; ++j) {
                const double bj = d->bt ? d->bt[j] : 0.0;
                b[j] = (bj + d->shift_t[j]) * d->scale_t[j];
                b[K + j] = (bj + d->shift_n[j]) * d->scale_n[j];
            }
        } else {
            for (int i = 0; i < cur; ++i)
                for (int j = 0; j < K; ++j) {
                    W[(size_t)i * 2 * K + j] = d->At[(size_t)i * K + j];
                    W[(size_t)i * 2 * K + K + j] = d->An[(size_t)i * K + j];
                }
            for (int j = 0; j < K; ++j) {
                if (d->bt) b[j] = d->bt[j];
                if (d->bn) b[K + j] = d->bn[j];
            }
        }
        p->fin.d_in = cur;
        p->fin.d_out = 2 * K;
        rc = upload_transposed(W.data(), cur, 2 * K, &p->fin);
        if (rc == HYPMI_OK) rc = upload_vec(b.data(), 2 * K, &p->fin);
        p->KP = hypmi_kp_for_k(K);
        if (rc == HYPMI_OK && shared) {
            p->shared = true;
            p->sfin.d_in = cur;
            p->sfin.d_out = K;
            std::vector<double> sb(K, 0.0);
            if (d->bt) sb.assign(d->bt, d->bt + K);
            rc = upload_transposed(d->At, cur, K, &p->sfin);
            if (rc == HYPMI_OK) rc = upload_vec(sb.data(), K, &p->sfin);
            std::vector<float> cm((size_t)4 * K);
            for (int j = 0; j < K; ++j) {
                cm[j] = (float)d->shift_t[j];
                cm[K + j] = (float)d->scale_t[j];
                cm[2 * K + j] = (float)d->shift_n[j];
                cm[3 * K + j] = (float)d->scale_n[j];
            }
            if (rc == HYPMI_OK) rc = to_device(cm, &p->col_map);
            // from here on W / NV describe the projection the fp32 path applies
            W.assign(d->At, d->At + (size_t)cur * K);
        }
        const int NVp = shared ? K : 2 * K;
        // fold the last length-normalised segment into the projection
        if (rc == HYPMI_OK && d->n_seg > 0 && d->seg[d->n_seg - 1].lnorm) {
            const hypmi_segment& ls = d->seg[d->n_seg - 1];
            const int dp = p->seg[d->n_seg - 1].d_in;  // input dim of that segment
            const int D1 = cur, NV = NVp;
            FusedTail& f = p->fused;
            f.d_in = dp;
            f.n_u = ls.A ? D1 : 0;
            f.n_v = NV;
            f.sqrt_dn = (float)sqrt((double)D1);
            f.norm_from_input = ls.A == nullptr;
            f.ldw = round_up(dp, 4);
            std::vector<float> wc((size_t)(f.n_u + NV) * f.ldw, 0.f);
            std::vector<float> bv(NV, 0.f);
            std::vector<double> col(dp);
            // Only the row norm of u = x A1 + b1 is used, and u Q has the same norm for any
            // orthogonal Q.  Householder reflections from the left turn [A1^T | b1^T] into
            // [R | Q^T b1^T] with R upper trapezoidal: column j of A1 Q = R^T is zero above row j, so
            // the kernel starts the contraction of column block n0 at k = n0.
            std::vector<double> bu_rot;
            if (f.n_u) {
                const int nc = dp + (ls.b ? 1 : 0);
                std::vector<double> M((size_t)nc * D1);  // column-major: M[c * D1 + j]
                for (int i = 0; i < dp; ++i)
                    for (int j = 0; j < D1; ++j) M[(size_t)i * D1 + j] = ls.A[(size_t)i * D1 + j];
                if (ls.b)
                    for (int j = 0; j < D1; ++j) M[(size_t)dp * D1 + j] = ls.b[j];
                std::vector<double> hv(D1);
                const int steps = D1 - 1 < dp ? D1 - 1 : dp;
                for (int c = 0; c < steps; ++c) {
                    double* colc = &M[(size_t)c * D1];
                    double nrm = 0.0;
                    for (int j = c; j < D1; ++j) nrm += colc[j] * colc[j];
                    nrm = sqrt(nrm);
                    if (nrm == 0.0) continue;
                    const double alpha = colc[c] > 0 ? -nrm : nrm;
                    double vnorm2 = 0.0;
                    for (int j = c; j < D1; ++j) {
                        hv[j] = colc[j] - (j == c ? alpha : 0.0);
                        vnorm2 += hv[j] * hv[j];
                    }
                    if (vnorm2 == 0.0) continue;
                    for (int cc = c; cc < nc; ++cc) {
                        double* col2 = &M[(size_t)cc * D1];
                        double dot = 0.0;
                        for (int j = c; j < D1; ++j) dot += hv[j] * col2[j];
                        const double t = 2.0 * dot / vnorm2;
                        for (int j = c; j < D1; ++j) col2[j] -= t * hv[j];
                    }
                    for (int j = c + 1; j < D1; ++j) colc[j] = 0.0;  // exact zeros below the diagonal
                }
                for (int j = 0; j < f.n_u; ++j)
                    for (int i = 0; i < dp; ++i) wc[(size_t)j * f.ldw + i] = (float)M[(size_t)i * D1 + j];
                if (ls.b) bu_rot.assign(&M[(size_t)dp * D1], &M[(size_t)dp * D1] + D1);
            }
            for (int j = 0; j < NV; ++j) {
                // column j of A1 A2 (A1 = identity when the segment has no matrix)
                double bacc = 0.0;
                if (ls.A) {
                    for (int i = 0; i < dp; ++i) {
                        double acc = 0.0;
                        for (int t = 0; t < D1; ++t) acc += ls.A[(size_t)i * D1 + t] * W[(size_t)t * NV + j];
                        col[i] = acc;
                    }
                } else {
                    for (int i = 0; i < dp; ++i) col[i] = W[(size_t)i * NV + j];
                }
                if (ls.b)
                    for (int t = 0; t < D1; ++t) bacc += ls.b[t] * W[(size_t)t * NV + j];
                bv[j] = (float)bacc;
                for (int i = 0; i < dp; ++i) wc[(size_t)(f.n_u + j) * f.ldw + i] = (float)col[i];
            }
            rc = to_device(wc, &f.Wcat);
            if (rc == HYPMI_OK && ls.b) {
                // rotated with the matrix when there is one; the bias itself for the identity
                std::vector<float> bu(D1);
                for (int j = 0; j < D1; ++j) bu[j] = (float)(f.n_u ? bu_rot[j] : ls.b[j]);
                rc = to_device(bu, &f.bias_u);
                if (rc == HYPMI_OK) rc = to_device(bv, &f.bias_vin);
            }
            f.enabled = rc == HYPMI_OK;
        }
    }
    if (rc == HYPMI_OK && p->K > 0) {
        // hypmi_prologue_noalloc must not resolve a symbol inside a stream capture: the address of this
        // device's scratch array is looked up here, where the plan's device is current
        void* unused = nullptr;
        rc = static_scratch(&unused);
    }
    if (rc != HYPMI_OK) {
        hypmi_plan_destroy(p);
        return rc;
    }
    *out = p;
    return HYPMI_OK;
}

extern "C" int hypmi_plan_kp(const hypmi_plan* p) { return p ? p->KP : 0; }
extern "C" int hypmi_plan_k(const hypmi_plan* p) { return p ? p->K : 0; }
extern "C" int hypmi_plan_d_in(const hypmi_plan* p) { return p ? p->d_in : 0; }
extern "C" int hypmi_plan_d_out(const hypmi_plan* p) { return p ? p->d_last : 0; }

// ------------------------------------------------------------------------------------ kernels
namespace {

constexpr int64_t CHUNK_ROWS = 65536;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// one wavefront per row: y <- sqrt(d) y / (|y| + 1e-10)
__global__ __launch_bounds__(256) void lnorm_rows_kernel(float* __restrict__ y, int64_t n, int64_t ldy,
                                                         int d) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float* p = y + row * ldy;
    double ss = 0.0;
    for (int k = lane; k < d; k += 64) {
        const double v = p[k];
        ss += v * v;
    }
    ss = wave_sum(ss);
    const float scale = (float)(sqrt((double)d) / (sqrt(ss) + 1e-10));
    for (int k = lane; k < d; k += 64) p[k] *= scale;
}

// one wavefront per row: Z = [g | h] (2K floats) -> packed row of width KP
__global__ __launch_bounds__(256) void pack_gq_kernel(const float* __restrict__ Z, int64_t n, int64_t ldz,
                                                      int K, int KP, int side, double c,
                                                      float* __restrict__ G) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* z = Z + row * ldz;
    float* g = G + row * KP;
    double sg = 0.0, sh = 0.0;
    for (int k = lane; k < K; k += 64) {
        const float gv = z[k];
        const double hv = z[K + k];
        sg += (double)gv * gv;
        sh += hv * hv;
        g[k] = gv;
    }
    const double q = 0.5 * (wave_sum(sg) - wave_sum(sh));
    for (int k = K + lane; k < KP; k += 64) {
        float v = 0.f;
        if (k == K) v = side == HYPMI_SIDE_ENROL ? (float)q : 1.f;
        if (k == K + 1) v = side == HYPMI_SIDE_ENROL ? 1.f : (float)(q + c);
        g[k] = v;
    }
}

// One column chunk (64 TN columns) of the projection walk of prologue_fused_kernel:
// z = sc * (x Wv^T + bias_in) + bias_out for the workgroup's rows.
template <int TM, int TN>
__device__ __forceinline__ void fused_v_chunk(const float* __restrict__ x, int64_t ldx,
                                              const float* __restrict__ Wv, int64_t ldw, int64_t m0, int n0,
                                              int64_t n, int n_v, int d_in, float* smem,
                                              const float (&sc)[TM][16], const float* __restrict__ bias_vin,
                                              const float* __restrict__ bias_vout, float* __restrict__ Z,
                                              int64_t ldz, float* zt = nullptr, int zld = 0) {
    f32x16 acc[TM][TN];
    gemm_tile::zero<TM, TN>(acc);
    gemm_tile::mainloop<TM, TN>(x, ldx, Wv, ldw, m0, n0, n, n_v, d_in, smem, acc);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + gemm_tile::col_of<TN>(j);
        if (col < n_v) {
            const float bi = bias_vin ? bias_vin[col] : 0.f;
            const float bo = bias_vout ? bias_vout[col] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int lr = gemm_tile::row_of<TM>(i, e);
                    const float zv = fmaf(sc[i][e], acc[i][j][e] + bi, bo);
                    if (zt) zt[lr * zld + col] = zv;  // the workgroup's rows stay on the chip
                    else if (m0 + lr < n) Z[(m0 + lr) * ldz + col] = zv;
                }
        }
    }
}

// Fused tail of the prologue: affine + length norm + PLDA projection in one pass over the rows.
// A workgroup owns 64*TM rows.  It first walks the n_u columns of u = x A1 + b1 in 128-column
// chunks, keeping only per-row sums of squares (u never leaves the chip), turns them into the
// per-row scale, then walks the 2K columns of v = x (A1 A2) + b1 A2 and writes
// z = scale * v + b2 = [g | h].
// What the PAIRS variant of prologue_fused_kernel does with its rows instead of writing z: the
// per-column maps and q of the shared-basis form (book_pack_kernel), the packed row [g, q, 1 ...]
// and its conversion to scaled fp16 pairs (pair_rows_kernel), all on the workgroup's LDS copy of z.
struct PairTail {
    const float* col_map;  // shift_t | scale_t | shift_n | scale_n, K floats each
    int K, KP, side, ks16;
    double c;
    uint4* H;        // [n x row_slots] fp16 pairs of side `side`
    uint4* H_other;  // optional: the same rows packed for the other side of a trial
    float* G;        // optional [n x KP] packed fp32 rows of side `side` (nullptr: not wanted)
    float* bnd;      // optional [n x 2]: (|g|, |q (+ c on the test side)|) of side `side`, the terms of the
                     // fp32 error bound of a score against this row
};

template <int TM, bool PAIRS = false>
__global__ __launch_bounds__(256) void prologue_fused_kernel(
    const float* __restrict__ x, int64_t ldx, int64_t n, int d_in, const float* __restrict__ Wcat,
    int64_t ldw, const float* __restrict__ bias_u, const float* __restrict__ bias_vin,
    const float* __restrict__ bias_vout, int n_u, int n_v, float sqrt_dn, int norm_from_input,
    float* __restrict__ Z, int64_t ldz, PairTail pt) {
    constexpr int BM = 64 * TM;
    extern __shared__ __attribute__((aligned(16))) float zt_dyn[];  // PAIRS: BM x (KP + 1) floats
    float* const zt = PAIRS ? zt_dyn : nullptr;
    const int zld = PAIRS ? pt.KP + 1 : 0;
    __shared__ __attribute__((aligned(16))) float smem[gemm_tile::Shape<TM, 2>::SMEM_FLOATS];
    __shared__ float rowsq[2][BM];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, wc = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    f32x16 acc[TM][2];

    float ss[TM][16];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) ss[i][e] = 0.f;
    for (int n0 = 0; n0 < n_u; n0 += 128) {
        gemm_tile::zero<TM, 2>(acc);
        // rows of Wcat from n0 on are zero in their first n0 entries (see plan_create)
        gemm_tile::mainloop<TM, 2>(x, ldx, Wcat, ldw, m0, n0, n, n_u, d_in, smem, acc, n0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + gemm_tile::col_of<2>(j);
            const bool ok = col < n_u;
            const float bu = (ok && bias_u) ? bias_u[col] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = ok ? acc[i][j][e] + bu : 0.f;
                    ss[i][e] = fmaf(v, v, ss[i][e]);
                }
        }
    }
    if (norm_from_input) {
        // identity matrix: |u| is the norm of the input row itself (plus the bias, if any).
        // Sixteen lanes per row, four rows per wave at a time: the rows' loads are independent.
        const int sub = lane >> 4, l16 = lane & 15;
        for (int rr = 4 * wave + sub; rr < BM; rr += 16) {
            const int64_t row = m0 + rr;
            float a2 = 0.f;
            if (row < n)
                for (int kk = l16; kk < d_in; kk += 16) {
                    const float v = x[row * ldx + kk] + (bias_u ? bias_u[kk] : 0.f);
                    a2 = fmaf(v, v, a2);
                }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) a2 += __shfl_xor(a2, off, 64);
            if (l16 == 0) {
                rowsq[0][rr] = a2;
                rowsq[1][rr] = 0.f;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = ss[i][e];
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);  // the 32 columns of a half
                if (r == 0) rowsq[wc][gemm_tile::row_of<TM>(i, e)] = v;
            }
    }
    __syncthreads();
    float sc[TM][16];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int rr = gemm_tile::row_of<TM>(i, e);
            sc[i][e] = sqrt_dn / (sqrtf(rowsq[0][rr] + rowsq[1][rr]) + 1e-10f);
        }
    const float* Wv = Wcat + (int64_t)n_u * ldw;
    // 128-column chunks; a remainder of at most 64 columns runs as a 64-column chunk
    int n0 = 0;
    for (; n_v - n0 > 64; n0 += 128)
        fused_v_chunk<TM, 2>(x, ldx, Wv, ldw, m0, n0, n, n_v, d_in, smem, sc, bias_vin, bias_vout, Z, ldz, zt, zld);
    if (n0 < n_v)
        fused_v_chunk<TM, 1>(x, ldx, Wv, ldw, m0, n0, n, n_v, d_in, smem, sc, bias_vin, bias_vout, Z, ldz, zt, zld);
    if constexpr (PAIRS) {
        using namespace hypmi_detail;
        const int K = pt.K, KP = pt.KP;
        const float *shift_t = pt.col_map, *scale_t = pt.col_map + K, *shift_n = pt.col_map + 2 * K,
                    *scale_n = pt.col_map + 3 * K;
        __syncthreads();  // every column chunk of the rows is in LDS
        // (a) the per-column maps and q, one wavefront per row, in the order book_pack_kernel sums
        for (int rr = wave; rr < BM; rr += 4) {
            float* z = zt + rr * zld;
            double sg = 0.0, sh = 0.0;
            for (int k = lane; k < K; k += 64) {
                const double v = z[k];
                const double gd = (v + (double)shift_t[k]) * (double)scale_t[k];
                const double hv = (v + (double)shift_n[k]) * (double)scale_n[k];
                sg += gd * gd;  // q from the unrounded g, as book_pack_kernel takes it
                sh += hv * hv;
                z[k] = (float)gd;
            }
            const double tg = wave_sum(sg);
            const double q = 0.5 * (tg - wave_sum(sh));
            if (pt.bnd && lane == 0 && m0 + rr < n) {
                pt.bnd[2 * (m0 + rr)] = (float)sqrt(tg);
                pt.bnd[2 * (m0 + rr) + 1] = (float)fabs(pt.side == HYPMI_SIDE_TEST ? q + pt.c : q);
            }
            // coordinates K, K + 1 depend on the side ([q, 1] or [1, q + c]): both q's are kept, the
            // test side's in the spare column behind the row
            for (int k = K + lane; k <= KP; k += 64) z[k] = k == K ? (float)q : (k == KP ? (float)(q + pt.c) : 0.f);
        }
        __syncthreads();
        // (b) packed row -> scaled fp16 pairs: lane = chunk of eight coordinates
        const int hs = split_half_slots(pt.ks16, 2), row_slots = split_row_slots(pt.ks16, 2);
        auto emit = [&](int side, uint4* H, float* G) {
            for (int rr = wave; rr < BM; rr += 4) {
                const int64_t row = m0 + rr;
                if (row >= n) break;  // wave-uniform
                const float* z = zt + rr * zld;
                const int k0 = 8 * lane;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = k0 + e;
                    float t = k < KP ? z[k] : 0.f;
                    if (k == K) t = side == HYPMI_SIDE_ENROL ? t : 1.f;
                    if (k == K + 1) t = side == HYPMI_SIDE_ENROL ? 1.f : z[KP];
                    v[e] = t;
                }
                if (G && k0 < KP) {
                    float4* g = reinterpret_cast<float4*>(G + row * KP + k0);
                    g[0] = make_float4(v[0], v[1], v[2], v[3]);
                    g[1] = make_float4(v[4], v[5], v[6], v[7]);
                }
                float m = pair_chunk_max(v);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
                const int up = pair_scale_exponent(m);
                uint4* out = H + row * row_slots;
                if (k0 < 16 * pt.ks16) pair_chunk_store(v, up, lane, hs, out);
                if (lane == 0) pair_row_tail(up, pt.ks16, hs, row_slots, out);
            }
        };
        emit(pt.side, pt.H, pt.G);
        if (pt.H_other) emit(pt.side == HYPMI_SIDE_ENROL ? HYPMI_SIDE_TEST : HYPMI_SIDE_ENROL, pt.H_other, nullptr);
    }
}

int run_fused_tail(const hypmi_plan* p, const float* in, int64_t ldin, int64_t rows, float* Z, int64_t ldz,
                   hipStream_t stream) {
    const FusedTail& f = p->fused;
    const float* bias_out = p->shared ? p->sfin.bias : p->fin.bias;
    const PairTail none = {};
    if (rows >= 128 * 2048) {
        hipLaunchKernelGGL(prologue_fused_kernel<2>, dim3((unsigned)((rows + 127) / 128)), dim3(256), 0, stream,
                           in, ldin, rows, f.d_in, f.Wcat, f.ldw, f.bias_u, f.bias_vin, bias_out, f.n_u,
                           f.n_v, f.sqrt_dn, f.norm_from_input ? 1 : 0, Z, ldz, none);
    } else {
        hipLaunchKernelGGL(prologue_fused_kernel<1>, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, stream, in,
                           ldin, rows, f.d_in, f.Wcat, f.ldw, f.bias_u, f.bias_vin, bias_out, f.n_u, f.n_v,
                           f.sqrt_dn, f.norm_from_input ? 1 : 0, Z, ldz, none);
    }
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

// The fused tail with the packing and the fp16-pair conversion in its epilogue (shared-basis plans).
int run_fused_tail_pairs(const hypmi_plan* p, const float* in, int64_t ldin, int64_t rows, int side, void* H,
                         void* H_other, float* G, float* bnd, hipStream_t stream) {
    const FusedTail& f = p->fused;
    PairTail pt;
    pt.col_map = p->col_map;
    pt.K = p->K;
    pt.KP = p->KP;
    pt.side = side;
    pt.ks16 = 2 * ((p->KP + 31) / 32);
    pt.c = p->c;
    pt.H = static_cast<uint4*>(H);
    pt.H_other = static_cast<uint4*>(H_other);
    pt.G = G;
    pt.bnd = bnd;
    const int lds1 = 64 * (p->KP + 1) * (int)sizeof(float), lds2 = 2 * lds1;
    const bool big = rows >= 128 * 2048 && lds2 <= 96 * 1024;
    if (big) {
        if (lds2 > 32 * 1024)
            HYPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&prologue_fused_kernel<2, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
        hipLaunchKernelGGL((prologue_fused_kernel<2, true>), dim3((unsigned)((rows + 127) / 128)), dim3(256), lds2,
                           stream, in, ldin, rows, f.d_in, f.Wcat, f.ldw, f.bias_u, f.bias_vin, p->sfin.bias, f.n_u,
                           f.n_v, f.sqrt_dn, f.norm_from_input ? 1 : 0, nullptr, 0, pt);
    } else {
        if (lds1 > 32 * 1024)
            HYPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&prologue_fused_kernel<1, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds1));
        hipLaunchKernelGGL((prologue_fused_kernel<1, true>), dim3((unsigned)((rows + 63) / 64)), dim3(256), lds1,
                           stream, in, ldin, rows, f.d_in, f.Wcat, f.ldw, f.bias_u, f.bias_vin, p->sfin.bias, f.n_u,
                           f.n_v, f.sqrt_dn, f.norm_from_input ? 1 : 0, nullptr, 0, pt);
    }
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

int run_lnorm(float* y, int64_t n, int64_t ldy, int d, hipStream_t stream) {
    if (n == 0) return HYPMI_OK;
    hipLaunchKernelGGL(lnorm_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, y, n, ldy, d);
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

struct Work {
    float* buf[2];
    float* z;
    int64_t ld, ldz, rows;
};

int carve(const hypmi_plan* p, int64_t n, void* ws, int64_t ws_bytes, Work* w) {
    const int64_t need = hypmi_prologue_workspace_bytes(p, n);
    HYPMI_REQUIRE(need == 0 || (ws && aligned16(ws)), "workspace must be a 16-byte aligned device pointer");
    HYPMI_REQUIRE(ws_bytes >= need, "workspace too small: %lld < %lld bytes", (long long)ws_bytes,
                  (long long)need);
    w->rows = n < CHUNK_ROWS ? n : CHUNK_ROWS;
    w->ld = round_up(p->max_dim, 4);
    w->ldz = round_up(2 * p->K, 4);
    float* base = static_cast<float*>(ws);
    w->buf[0] = base;
    w->buf[1] = base + w->rows * w->ld;
    w->z = base + 2 * w->rows * w->ld;
    return HYPMI_OK;
}

// Runs the segments on `rows` rows starting at x.  On return *cur / *ldcur describe where the result
// lives (x itself when the chain is empty).  If `final_out` is given, the last materialising step
// writes straight into it.
int run_segments(const hypmi_plan* p, int n_run, const float* x, int64_t ldx, int64_t rows, const Work& w,
                 float* final_out, int64_t ld_final, const float** cur, int64_t* ldcur,
                 hipStream_t stream) {
    const float* in = x;
    int64_t ldin = ldx;
    int which = 0;
    int last_mat = -1;  // last segment that has to write a buffer
    for (int i = 0; i < n_run; ++i)
        if (p->seg[i].Wt || p->seg[i].lnorm) last_mat = i;
    for (int i = 0; i < n_run; ++i) {
        const Segment& s = p->seg[i];
        if (!s.Wt && !s.lnorm) continue;
        const bool to_final = final_out && i == last_mat;
        float* dst = nullptr;
        int64_t lddst = 0;
        if (s.Wt) {
            dst = to_final ? final_out : w.buf[which];
            lddst = to_final ? ld_final : w.ld;
            int rc = hypmi_launch_gemm_nt(in, ldin, s.Wt, s.ldw, s.bias, dst, lddst, rows, s.d_out,
                                          s.d_in, stream);
            if (rc != HYPMI_OK) return rc;
        } else if (in == x || to_final) {
            // length norm only: never touch the caller's input, normalise a copy
            dst = to_final ? final_out : w.buf[which];
            lddst = to_final ? ld_final : w.ld;
            HYPMI_HIP(hipMemcpy2DAsync(dst, lddst * sizeof(float), in, ldin * sizeof(float),
                                       (size_t)s.d_in * sizeof(float), (size_t)rows,
                                       hipMemcpyDeviceToDevice, stream));
        } else {
            dst = const_cast<float*>(in);  // already one of our scratch buffers
            lddst = ldin;
        }
        if (s.lnorm) {
            int rc = run_lnorm(dst, rows, lddst, s.d_out, stream);
            if (rc != HYPMI_OK) return rc;
        }
        if (dst == w.buf[which]) which ^= 1;
        in = dst;
        ldin = lddst;
    }
    *cur = in;
    *ldcur = ldin;
    return HYPMI_OK;
}

}  // namespace

extern "C" int64_t hypmi_prologue_workspace_bytes(const hypmi_plan* p, int64_t n) {
    if (!p || n <= 0) return 0;
    const int64_t rows = n < CHUNK_ROWS ? n : CHUNK_ROWS;
    const int64_t ld = round_up(p->max_dim, 4);
    const int64_t ldz = round_up(2 * p->K, 4);
    return rows * (2 * ld + ldz) * (int64_t)sizeof(float);
}

extern "C" int hypmi_lnorm_rows(float* y, int64_t n, int64_t ldy, int d, void* stream) {
    HYPMI_REQUIRE(y && n >= 0 && d > 0 && ldy >= d, "lnorm_rows: bad argument");
    return run_lnorm(y, n, ldy, d, static_cast<hipStream_t>(stream));
}

namespace {
int prologue_impl(const hypmi_plan* p, const float* x, int64_t n, int64_t ldx, int side, float* G, void* ws,
                  int64_t ws_bytes, hipStream_t stream);

// Scratch of the entry point that allocates nothing: one array per GPU the library runs on, held
// until the process ends.  The array's name is a host-side stand-in; its device address comes from
// hipGetSymbolAddress, once per device, and that address is what kernels and copies are given.
constexpr int64_t STATIC_SCRATCH_BYTES = 16ll << 20;
constexpr int MAX_DEVICES = 64;
}  // namespace
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];
namespace {
std::mutex scratch_mutex;
void* scratch_ptr[MAX_DEVICES];

int static_scratch(void** out) {
    int dev = -1;
    HYPMI_HIP(hipGetDevice(&dev));
    HYPMI_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(scratch_mutex);
    if (!scratch_ptr[dev]) {
        void* ptr = nullptr;
        HYPMI_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));  // this device's copy of the array
        HYPMI_REQUIRE(ptr != nullptr, "no device address for the static scratch array");
        scratch_ptr[dev] = ptr;
    }
    *out = scratch_ptr[dev];
    return HYPMI_OK;
}
}  // namespace

extern "C" int64_t hypmi_prologue_noalloc_max_rows(const hypmi_plan* p) {
    if (!p) return 0;
    const int64_t per_row = hypmi_prologue_workspace_bytes(p, 1);
    const int64_t rows = per_row > 0 ? STATIC_SCRATCH_BYTES / per_row : 0;
    return rows < CHUNK_ROWS ? rows : CHUNK_ROWS;
}

extern "C" int hypmi_prologue_noalloc(const hypmi_plan* p, const float* x, int64_t n, int64_t ldx, int side,
                                      float* G, void* stream_) {
    HYPMI_REQUIRE(p && p->K > 0, "prologue_noalloc: plan has no PLDA projection");
    const int64_t max_rows = hypmi_prologue_noalloc_max_rows(p);
    HYPMI_REQUIRE(n <= max_rows,
                  "prologue_noalloc: %lld rows exceed the %lld this plan fits into the %lld-byte static scratch; "
                  "use hypmi_prologue with a workspace",
                  (long long)n, (long long)max_rows, (long long)STATIC_SCRATCH_BYTES);
    void* ws = nullptr;
    int rc = static_scratch(&ws);  // resolved when the plan was created on this device: a table look-up
    if (rc != HYPMI_OK) return rc;
    // Every intermediate row is written by its producing kernel before anything reads it and nothing
    // is accumulated in place, so what the previous call left behind needs no clearing.
    return prologue_impl(p, x, n, ldx, side, G, ws, STATIC_SCRATCH_BYTES, static_cast<hipStream_t>(stream_));
}

extern "C" int hypmi_prologue(const hypmi_plan* p, const float* x, int64_t n, int64_t ldx, int side,
                              float* G, void* ws, int64_t ws_bytes, void* stream_) {
    return prologue_impl(p, x, n, ldx, side, G, ws, ws_bytes, static_cast<hipStream_t>(stream_));
}

namespace {
int prologue_impl(const hypmi_plan* p, const float* x, int64_t n, int64_t ldx, int side, float* G, void* ws,
                  int64_t ws_bytes, hipStream_t stream) {
    HYPMI_REQUIRE(p && p->K > 0, "prologue: plan has no PLDA projection");
    HYPMI_REQUIRE(x && G, "prologue: null pointer");
    HYPMI_REQUIRE(n >= 0 && ldx >= round_up(p->d_in, 4) && ldx % 4 == 0 && aligned16(x),
                  "prologue: x must be 16-byte aligned with ldx %% 4 == 0 and ldx >= d_in rounded up to 4");
    HYPMI_REQUIRE(side == HYPMI_SIDE_ENROL || side == HYPMI_SIDE_TEST, "prologue: bad side %d", side);
    if (n == 0) return HYPMI_OK;
    Work w;
    int rc = carve(p, n, ws, ws_bytes, &w);
    if (rc != HYPMI_OK) return rc;
    for (int64_t r0 = 0; r0 < n; r0 += w.rows) {
        const int64_t rows = (n - r0 < w.rows) ? n - r0 : w.rows;
        const float* cur;
        int64_t ldcur;
        const bool fuse = p->fused.enabled;
        rc = run_segments(p, fuse ? p->n_seg - 1 : p->n_seg, x + r0 * ldx, ldx, rows, w, nullptr, 0, &cur,
                          &ldcur, stream);
        if (rc != HYPMI_OK) return rc;
        const Segment& fin = p->shared ? p->sfin : p->fin;
        if (fuse)
            rc = run_fused_tail(p, cur, ldcur, rows, w.z, w.ldz, stream);
        else
            rc = hypmi_launch_gemm_nt(cur, ldcur, fin.Wt, fin.ldw, fin.bias, w.z, w.ldz, rows, fin.d_out,
                                      fin.d_in, stream);
        if (rc != HYPMI_OK) return rc;
        if (p->shared) {
            const float* cm = p->col_map;
            rc = hypmi_book_pack(w.z, w.ldz, p->K, nullptr, rows, cm, cm + p->K, cm + 2 * p->K, cm + 3 * p->K,
                                 side, p->c, G + r0 * p->KP, p->KP, stream);
            if (rc != HYPMI_OK) return rc;
        } else {
            hipLaunchKernelGGL(pack_gq_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, w.z,
                               rows, w.ldz, p->K, p->KP, side, p->c, G + r0 * p->KP);
            HYPMI_HIP(hipGetLastError());
        }
    }
    return HYPMI_OK;
}
}  // namespace

extern "C" int hypmi_prologue_pairs_supported(const hypmi_plan* p) {
    // the fused tail in its shared-basis form, and a packed width the fp16-pair score kernel takes
    return (p && p->K > 0 && p->fused.enabled && p->shared && p->col_map && p->KP % 8 == 0 && p->KP <= 320) ? 1 : 0;
}

extern "C" int hypmi_prologue_pairs_bounds(const hypmi_plan* p, const float* x, int64_t n, int64_t ldx, int side,
                                           void* H, void* H_other, float* G, float* bnd, void* ws,
                                           int64_t ws_bytes, void* stream_);

extern "C" int hypmi_prologue_pairs(const hypmi_plan* p, const float* x, int64_t n, int64_t ldx, int side,
                                    void* H, void* H_other, float* G, void* ws, int64_t ws_bytes, void* stream_) {
    return hypmi_prologue_pairs_bounds(p, x, n, ldx, side, H, H_other, G, nullptr, ws, ws_bytes, stream_);
}

extern "C" int hypmi_prologue_pairs_bounds(const hypmi_plan* p, const float* x, int64_t n, int64_t ldx, int side,
                                           void* H, void* H_other, float* G, float* bnd, void* ws,
                                           int64_t ws_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HYPMI_REQUIRE(hypmi_prologue_pairs_supported(p),
                  "prologue_pairs: this plan has no fused shared-basis tail; use hypmi_prologue + hypmi_pair_rows");
    HYPMI_REQUIRE(x && H, "prologue_pairs: null pointer");
    HYPMI_REQUIRE(n >= 0 && ldx >= round_up(p->d_in, 4) && ldx % 4 == 0 && aligned16(x) && aligned16(H) &&
                      (!G || aligned16(G)) && (!H_other || aligned16(H_other)),
                  "prologue_pairs: buffers must be 16-byte aligned with ldx %% 4 == 0 and ldx >= d_in rounded up to 4");
    HYPMI_REQUIRE(side == HYPMI_SIDE_ENROL || side == HYPMI_SIDE_TEST, "prologue_pairs: bad side %d", side);
    if (n == 0) return HYPMI_OK;
    Work w;
    int rc = carve(p, n, ws, ws_bytes, &w);
    if (rc != HYPMI_OK) return rc;
    // Nothing in front of the fused tail writes a buffer (the usual chain: one affine step, the length
    // norm, one more affine step): the rows need no staging and all of them go in one launch.
    bool staged = false;
    for (int i = 0; i + 1 < p->n_seg; ++i) staged |= p->seg[i].Wt != nullptr || p->seg[i].lnorm;
    if (!staged) w.rows = n;
    const int64_t row_bytes = (int64_t)hypmi_detail::split_row_slots(2 * ((p->KP + 31) / 32), 2) * 16;
    for (int64_t r0 = 0; r0 < n; r0 += w.rows) {
        const int64_t rows = (n - r0 < w.rows) ? n - r0 : w.rows;
        const float* cur;
        int64_t ldcur;
        rc = run_segments(p, p->n_seg - 1, x + r0 * ldx, ldx, rows, w, nullptr, 0, &cur, &ldcur, stream);
        if (rc != HYPMI_OK) return rc;
        rc = run_fused_tail_pairs(p, cur, ldcur, rows, side, static_cast<char*>(H) + r0 * row_bytes,
                                  H_other ? static_cast<char*>(H_other) + r0 * row_bytes : nullptr,
                                  G ? G + r0 * p->KP : nullptr, bnd ? bnd + 2 * r0 : nullptr, stream);
        if (rc != HYPMI_OK) return rc;
    }
    return HYPMI_OK;
}

extern "C" int hypmi_transform(const hypmi_plan* p, const float* x, int64_t n, int64_t ldx, float* y,
                               int64_t ldy, void* ws, int64_t ws_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HYPMI_REQUIRE(p && x && y, "transform: null pointer");
    HYPMI_REQUIRE(n >= 0 && ldx >= round_up(p->d_in, 4) && ldx % 4 == 0 && aligned16(x),
                  "transform: x must be 16-byte aligned with ldx %% 4 == 0");
    HYPMI_REQUIRE(ldy >= p->d_last, "transform: ldy too small");
    if (n == 0) return HYPMI_OK;
    Work w;
    int rc = carve(p, n, ws, ws_bytes, &w);
    if (rc != HYPMI_OK) return rc;
    for (int64_t r0 = 0; r0 < n; r0 += w.rows) {
        const int64_t rows = (n - r0 < w.rows) ? n - r0 : w.rows;
        const float* cur;
        int64_t ldcur;
        rc = run_segments(p, p->n_seg, x + r0 * ldx, ldx, rows, w, y + r0 * ldy, ldy, &cur, &ldcur, stream);
        if (rc != HYPMI_OK) return rc;
        if (cur != y + r0 * ldy) {  // empty chain: plain copy
            HYPMI_HIP(hipMemcpy2DAsync(y + r0 * ldy, ldy * sizeof(float), cur, ldcur * sizeof(float),
                                       (size_t)p->d_last * sizeof(float), (size_t)rows,
                                       hipMemcpyDeviceToDevice, stream));
        }
    }
    return HYPMI_OK;
}

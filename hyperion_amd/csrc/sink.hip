// Score sink: target / non-target score histograms and moments of a device score block, for trial
// sets whose score matrix is too large to leave the device (1e10 .. 1e12 trials).  The reference
// pulls target and non-target scores out of the full matrix with boolean masks
// (hyperion/utils/trial_scores.py:342-357 get_tar_non) and sorts them on the host
// (hyperion/metrics/roc.py:54-102); here a block of scores is reduced where it lies.
//
// The histograms give the ROC up to the bin width; hypmi_score_extract is the second pass that makes it
// exact: it appends the scores of flagged (class, bin) cells to two lists, which the host sorts (the
// cells that can hold a vertex of the ROC convex hull are few; with every cell flagged it is
// get_tar_non itself).
//
// HBM-bound: one read of 4 bytes per score (plus one byte per mask matrix when the labels come as
// masks).  A workgroup owns a strip of 1024 columns and a slab of rows, keeps both histograms in LDS
// (ds_add_u32), and adds its non-empty bins and its moments to the global result once at the end.
#include "common.h"

namespace {

constexpr int SH_THREADS = 256;
constexpr int SH_COLS = 4 * SH_THREADS;  // columns of a strip: four consecutive ones per lane
constexpr int SH_MAX_BINS = 8192;        // 2 x 8192 x 4 B = 64 KB of LDS

enum LabelMode { LABEL_NONE = 0, LABEL_KEYS = 1, LABEL_MASKS = 2 };

// Running moments of one class in one lane: counts as integers (a lane sees fewer than 2^32 scores),
// sums in double.
struct Moments {
    unsigned n, bad;
    double s1, s2;
    __device__ __forceinline__ void add(float s, float lo, float scale, int nbins, unsigned* h) {
        if (!(fabsf(s) <= 3.402823466e38f)) {  // NaN or infinite: counted, kept out of sums and bins
            ++bad;
            return;
        }
        const float f = floorf((s - lo) * scale);
        const int b = f < 0.f ? 0 : (f >= (float)nbins ? nbins - 1 : (int)f);
        atomicAdd(&h[b], 1u);
        ++n;
        const double d = (double)s;
        s1 += d;
        s2 = fma(d, d, s2);
    }
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// cls: 0 non-target, 1 target, anything else: not a trial.  The two classes are separate code paths
// under wave-uniform branches: targets are rare in large trial sets, and a wave without one never
// pays for their double-precision updates.
__device__ __forceinline__ void account(float s, int cls, float lo, float scale, int nbins, unsigned* h,
                                        Moments& non, Moments& tar) {
    if (__ballot(cls == 0)) {
        if (cls == 0) non.add(s, lo, scale, nbins, h);
    }
    if (__ballot(cls == 1)) {
        if (cls == 1) tar.add(s, lo, scale, nbins, h + nbins);
    }
}

// Second pass: a finite score whose (class, bin) cell is flagged goes to the class's list.  One atomic
// per wave and class reserves the places (the order inside a list is arbitrary).
__device__ __forceinline__ void extract(float s, int cls, float lo, float scale, int nbins,
                                        const unsigned* flag, float* __restrict__ out, int64_t cap,
                                        unsigned long long* __restrict__ count) {
    bool want = false;
    if ((cls == 0 || cls == 1) && fabsf(s) <= 3.402823466e38f) {
        const float f = floorf((s - lo) * scale);
        const int b = f < 0.f ? 0 : (f >= (float)nbins ? nbins - 1 : (int)f);
        want = flag[cls * nbins + b] != 0u;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const unsigned long long m = __ballot(want && cls == c);
        if (m == 0ull) continue;
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll((long long)m) - 1;
        unsigned long long base = 0;
        if (lane == leader) base = atomicAdd(&count[c], (unsigned long long)__popcll(m));
        base = __shfl(base, leader);
        if (want && cls == c) {
            const unsigned long long at = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
            if (at < (unsigned long long)cap) out[(int64_t)c * cap + (int64_t)at] = s;
        }
    }
}

template <int MODE, bool EXTRACT>
__global__ __launch_bounds__(SH_THREADS) void score_hist_kernel(
    const float* __restrict__ S, int64_t lds_, int64_t ne, int64_t nt, const int32_t* __restrict__ key_e,
    const int32_t* __restrict__ key_t, const unsigned char* __restrict__ tar,
    const unsigned char* __restrict__ non, int64_t ldm, float lo, float scale, int nbins,
    int64_t rows_per_block, int vec_ok, int mvec_ok, unsigned long long* __restrict__ hist,
    double* __restrict__ mom, const unsigned char* __restrict__ want, float* __restrict__ out, int64_t cap) {
    // histogram pass: h = the two histograms, flushed to `hist` at the end, moments to `mom`;
    // extraction pass: h = the flags of `want`, `hist` = the two list lengths
    extern __shared__ unsigned h[];  // [2][nbins]
    __shared__ double red[SH_THREADS / 64][8];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * nbins; i += SH_THREADS) h[i] = EXTRACT ? (unsigned)want[i] : 0u;
    __syncthreads();

    const int64_t c0 = (int64_t)blockIdx.x * SH_COLS + 4 * tid;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < ne ? r0 + rows_per_block : ne;
    const int ncol = c0 >= nt ? 0 : (nt - c0 < 4 ? (int)(nt - c0) : 4);
    Moments m_non = {0u, 0u, 0.0, 0.0}, m_tar = {0u, 0u, 0.0, 0.0};
    int kt[4] = {-1, -1, -1, -1};
    if (MODE == LABEL_KEYS)
        for (int j = 0; j < ncol; ++j) kt[j] = key_t[c0 + j];

    constexpr int AHEAD = 4;  // rows whose loads are in flight together (memory-level parallelism)
    const bool vec = vec_ok && ncol == 4, mvec = mvec_ok && ncol == 4;
    for (int64_t rb = r0; rb < r1 && ncol > 0; rb += AHEAD) {
        float sv[AHEAD][4];
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) {
            const int64_t r = rb + a < r1 ? rb + a : r1 - 1;  // past the slab: a copy that is not used
            const float* row = S + r * lds_ + c0;
            if (vec) {
                const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row));
                sv[a][0] = v[0]; sv[a][1] = v[1]; sv[a][2] = v[2]; sv[a][3] = v[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) sv[a][j] = j < ncol ? row[j] : 0.f;
            }
        }
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) {
            const int64_t r = rb + a;
            if (r >= r1) break;
            int cls[4] = {0, 0, 0, 0};
            if (MODE == LABEL_KEYS) {
                const int ke = key_e[r];
#pragma unroll
                for (int j = 0; j < 4; ++j) cls[j] = (ke < 0 || kt[j] < 0) ? -1 : (ke == kt[j] ? 1 : 0);
            } else if (MODE == LABEL_MASKS) {
                const int64_t at = r * ldm + c0;
                unsigned char bt[4] = {0, 0, 0, 0}, bn[4] = {1, 1, 1, 1};
                if (mvec) {
                    const uchar4 t4 = *reinterpret_cast<const uchar4*>(tar + at);
                    bt[0] = t4.x; bt[1] = t4.y; bt[2] = t4.z; bt[3] = t4.w;
                    if (non) {
                        const uchar4 n4 = *reinterpret_cast<const uchar4*>(non + at);
                        bn[0] = n4.x; bn[1] = n4.y; bn[2] = n4.z; bn[3] = n4.w;
                    }
                } else {
                    for (int j = 0; j < ncol; ++j) {
                        bt[j] = tar[at + j];
                        if (non) bn[j] = non[at + j];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) cls[j] = bt[j] ? 1 : (bn[j] ? 0 : -1);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (EXTRACT) extract(sv[a][j], j < ncol ? cls[j] : -1, lo, scale, nbins, h, out, cap, hist);
                else account(sv[a][j], j < ncol ? cls[j] : -1, lo, scale, nbins, h, m_non, m_tar);
            }
        }
    }
    if constexpr (EXTRACT) return;
    __syncthreads();
    for (int i = tid; i < 2 * nbins; i += SH_THREADS) {
        const unsigned v = h[i];
        if (v) atomicAdd(&hist[i], (unsigned long long)v);
    }
    double part[8] = {(double)m_non.n, m_non.s1, m_non.s2, (double)m_non.bad,
                      (double)m_tar.n, m_tar.s1, m_tar.s2, (double)m_tar.bad};
#pragma unroll
    for (int i = 0; i < 8; ++i) part[i] = wave_sum(part[i]);
    if ((tid & 63) == 0)
        for (int i = 0; i < 8; ++i) red[tid >> 6][i] = part[i];
    __syncthreads();
    if (tid < 8) {
        double v = 0.0;
        for (int w = 0; w < SH_THREADS / 64; ++w) v += red[w][tid];
        if (v != 0.0) atomicAdd(&mom[tid], v);
    }
}

template <int MODE, bool EXTRACT>
int launch_pass(dim3 grid, size_t lds_bytes, hipStream_t stream, const float* S, int64_t lds_, int64_t ne,
                int64_t nt, const int32_t* key_e, const int32_t* key_t, const unsigned char* tar,
                const unsigned char* non, int64_t ldm, float lo, float scale, int nbins, int64_t rows_per_block,
                int vec_ok, int mvec_ok, unsigned long long* hist, double* mom, const unsigned char* want,
                float* out, int64_t cap) {
    if (lds_bytes > 48 * 1024)
        HYPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&score_hist_kernel<MODE, EXTRACT>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL((score_hist_kernel<MODE, EXTRACT>), grid, dim3(SH_THREADS), lds_bytes, stream, S, lds_, ne,
                       nt, key_e, key_t, tar, non, ldm, lo, scale, nbins, rows_per_block, vec_ok, mvec_ok, hist,
                       mom, want, out, cap);
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

// Checks and launch geometry shared by the two passes.  `counters`: the histograms, or the list lengths.
template <bool EXTRACT>
int run_pass(const char* what, const float* S, int64_t lds_, int64_t ne, int64_t nt, const int32_t* key_e,
             const int32_t* key_t, const unsigned char* tar, const unsigned char* non, int64_t ldm, float lo,
             float hi, int nbins, unsigned long long* counters, double* mom, const unsigned char* want,
             float* out, int64_t cap, hipStream_t stream) {
    HYPMI_REQUIRE(S && counters, "%s: null pointer", what);
    HYPMI_REQUIRE(ne >= 0 && nt >= 0 && lds_ >= nt, "%s: bad shape", what);
    HYPMI_REQUIRE(nbins >= 1 && nbins <= SH_MAX_BINS, "%s: %d bins (1 .. %d supported)", what, nbins, SH_MAX_BINS);
    HYPMI_REQUIRE(hi > lo, "%s: empty score range", what);
    HYPMI_REQUIRE((key_e == nullptr) == (key_t == nullptr), "%s: keys come for both sides or not at all", what);
    HYPMI_REQUIRE(!(key_e && tar), "%s: labels come as keys or as masks, not both", what);
    HYPMI_REQUIRE(!(non && !tar), "%s: a non-target mask needs a target mask", what);
    HYPMI_REQUIRE(!tar || ldm >= nt, "%s: mask pitch smaller than the row", what);
    if (ne == 0 || nt == 0) return HYPMI_OK;
    const int64_t strips = (nt + SH_COLS - 1) / SH_COLS;
    HYPMI_REQUIRE(strips <= 0x7fffffffLL, "%s: too many columns", what);
    // about 2560 workgroups (two rounds of five per CU at 4096 bins), so that the 2 x nbins global
    // updates per workgroup stay small against its share of the scores; a slab is bounded so that an
    // LDS bin cannot wrap
    int64_t slabs = (2560 + strips - 1) / strips;
    if (slabs > ne) slabs = ne;
    if (slabs > 65535) slabs = 65535;
    int64_t rows_per_block = (ne + slabs - 1) / slabs;
    const int64_t max_rows = (int64_t)0xffffffffu / SH_COLS;
    if (rows_per_block > max_rows) rows_per_block = max_rows;
    slabs = (ne + rows_per_block - 1) / rows_per_block;
    HYPMI_REQUIRE(slabs <= 65535, "%s: %lld rows in one call", what, (long long)ne);
    const float scale = (float)nbins / (hi - lo);
    const int vec_ok = aligned16(S) && (lds_ % 4 == 0);
    const auto aligned4 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; };
    const int mvec_ok = tar && aligned4(tar) && (!non || aligned4(non)) && (ldm % 4 == 0);
    const size_t lds_bytes = (size_t)2 * nbins * sizeof(unsigned);
    const dim3 grid((unsigned)strips, (unsigned)slabs);
#define HYPMI_SH(MODE)                                                                                        \
    return launch_pass<MODE, EXTRACT>(grid, lds_bytes, stream, S, lds_, ne, nt, key_e, key_t, tar, non, ldm, \
                                      lo, scale, nbins, rows_per_block, vec_ok, mvec_ok, counters, mom, want,  \
                                      out, cap)
    if (key_e) HYPMI_SH(LABEL_KEYS);
    if (tar) HYPMI_SH(LABEL_MASKS);
    HYPMI_SH(LABEL_NONE);
#undef HYPMI_SH
}

}  // namespace

extern "C" int hypmi_score_hist(const float* S, int64_t lds_, int64_t ne, int64_t nt, const int32_t* key_e,
                                const int32_t* key_t, const unsigned char* tar, const unsigned char* non,
                                int64_t ldm, float lo, float hi, int nbins, unsigned long long* hist,
                                double* mom, void* stream) {
    HYPMI_REQUIRE(mom, "score_hist: null pointer");
    return run_pass<false>("score_hist", S, lds_, ne, nt, key_e, key_t, tar, non, ldm, lo, hi, nbins, hist, mom,
                           nullptr, nullptr, 0, static_cast<hipStream_t>(stream));
}

extern "C" int hypmi_score_extract(const float* S, int64_t lds_, int64_t ne, int64_t nt, const int32_t* key_e,
                                   const int32_t* key_t, const unsigned char* tar, const unsigned char* non,
                                   int64_t ldm, float lo, float hi, int nbins, const unsigned char* want,
                                   float* out, int64_t cap, unsigned long long* count, void* stream) {
    HYPMI_REQUIRE(want && out && cap >= 0, "score_extract: null pointer or negative capacity");
    return run_pass<true>("score_extract", S, lds_, ne, nt, key_e, key_t, tar, non, ldm, lo, hi, nbins, count,
                          nullptr, want, out, cap, static_cast<hipStream_t>(stream));
}

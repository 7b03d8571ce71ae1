// Score normalisation kernels.
//
// Adaptive S-norm works on "rows over the cohort": A_ec[e, c] (enrol vs cohort) and A_tc[t, c]
// (test vs cohort).  For every row the cohort members ranked discard .. discard+k-1 by descending
// score are selected (top-k), then for every (row r of X, list q) pair the mean and standard
// deviation of X[r, list_q] are gathered:
//     Z part:  X = A_ec, lists from A_tc rows  -> stats[e, t]
//     T part:  X = A_tc, lists from A_ec rows  -> stats[t, e]
// and the final kernel combines both with the raw score.
//
//   topk_rows     one workgroup per row; the row lives in registers as order-preserving integer keys
//                 and the boundary keys are found bit by bit; ties at a boundary are resolved by
//                 cohort index (lowest first), so the result is deterministic.
//   gather_stats  one workgroup per group of R rows of X held in LDS (interleaved, so one LDS read
//                 returns the R values of a cohort member); lanes walk different lists and reuse
//                 every index for the R rows.  Cohorts that do not fit LDS two rows at a time are
//                 taken in segments (gather_seg_kernel).
//
// Index lists are uint16 (cohort < 65536) in groups of eight: entry j of list q lives at
// idx8[((j >> 3) * ldi + q) * 8 + (j & 7)], so a lane fetches eight consecutive entries of its list
// with one 16-byte load and a wave's loads are contiguous (1 KiB).
#include <mutex>

#include "common.h"

namespace {


__device__ __forceinline__ unsigned key_of(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending float order == ascending key
}

__device__ __forceinline__ float float_of_key(unsigned key) {
    return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
}

// Device view of hypmi_topk_exact (see hypmi.h): where the exact value of entry (row, c) comes from.
struct ExactSrc {
    int mode;               // 0 none, 1 fp64 matrix, 2 fp64 packed rows
    const double* A64;      // mode 1: [n_rows x lda64]
    int64_t lda64;
    const double* G_rows;   // mode 2: [n_rows x ldg] rows [g, q, |g|]
    const double* G_coh;    //         [nc x ldg]
    int64_t ldg;
    int K;
    const float* bnd_rows;  // mode 2: (|g|, |q|) per row
    float coh_gmax, coh_qmax, err_coeff;
    unsigned long long* counters;  // [0] rows with a re-decided boundary, [1] rows skipped (too many ties)
    // mode 3 (register-resident kernel only): no exact values yet -- rows whose boundary band holds more
    // than the boundary key itself are appended to `todo` (their count in *todo_count) for a second call
    const int* row_map;      // optional: workgroup b handles row row_map[b] of A and of the lists, with
                             // exact values and bounds taken from row b of G_rows / bnd_rows
    int* todo;
    unsigned* todo_count;
};

constexpr int TK_CAND = 64;  // boundary candidates re-decided per boundary

// ---- register-resident top-k ----------------------------------------------------------------------
// One workgroup of 256 (long rows: 512) threads per row; thread t keeps the keys of cohort members t, t + 256, ... in
// ITEMS registers (no LDS copy of the row, no atomics on it).  The key at descending rank `want` is
// found bit by bit from the top: the answer has bit b set iff more than `want` keys are
// >= (prefix | 1 << b).  One step is ITEMS compares per thread, a ballot count per wave and one
// barrier; a step whose threshold lies outside the row's range of keys (known from one pass of three-input
// minima and maxima) is decided without looking at the keys.  As soon as at most TR_CAND keys are left
// inside the bracket [prefix, prefix + 2^b) they are appended to an array in LDS (a place per key from an
// LDS counter), one wave takes all of them (TR_CAND / 64 registers per lane), finishes the remaining bits
// on its own without barriers -- stopping as soon as a single key is left -- and hands the result to the
// others.  (512 measured best for 20 000-member rows: 1024 doubles the compares of the remaining steps,
// 256 costs two more full passes and more often leaves the uncertainty band of the boundary outside the
// bracket.)  Several such workgroups share a CU, so the barriers of one overlap the compares of the
// others.  The kernel is bound by vector-instruction issue (about 70 % of the cycles), so the work per
// key is counted in instructions: row loads through a buffer descriptor with scalar offsets (none), the
// order-preserving key in four (key_of_min1), a range check only for the last items of a thread
// (topk_safe_items), one compare per step, two per key for the candidate test, two for the membership bit
// (shift_in4_ge).
constexpr int TR_THREADS = 256;    // threads per row of up to 24 576 members, twice as many above
constexpr int TR_MAX_WAVES = 8;
constexpr int TR_CAND = 512;     // the bracket is narrow once it holds at most this many keys
constexpr int TR_STAGE = 8192;    // longest list staged in LDS before it is written (entries)
constexpr int TR_ORDER = 2048;    // longest list that leaves ordered by LDS bank class
constexpr int TR_MAX_NC = 128 * 512;  // longest row: 512 threads with 128 keys each

struct TopkShared {
    unsigned cnt[2][TR_MAX_WAVES];         // per-wave counters, alternating between steps
    unsigned ext[2][TR_MAX_WAVES];         // per-wave smallest and largest key
    unsigned res[8];                       // what the wave that finished a selection found
    unsigned ncand;                        // candidates written so far / boundary candidates
    unsigned cand[TR_CAND]; // keys inside the bracket
};

// Sum over the workgroup of a per-wave count (one barrier; the counter slots alternate so that the
// next call may start before every wave has read this one).
template <int WAVES>
__device__ __forceinline__ int block_sum(TopkShared& sh, int& par, unsigned wave_count, int lane, int wave) {
    // The count is pinned in a scalar register here: sunk into the lane-0 branch below, its popcounts
    // would keep every ballot mask of the caller alive across that branch.
    asm volatile("" : "+s"(wave_count));
    if (lane == 0) sh.cnt[par][wave] = wave_count;
    __syncthreads();
    unsigned sum = 0;
#pragma unroll
    for (int q = 0; q < WAVES / 4; ++q) {
        const uint4 v = *reinterpret_cast<const uint4*>(sh.cnt[par] + 4 * q);
        sum += (v.x + v.y) + (v.z + v.w);
    }
    par ^= 1;
    return __builtin_amdgcn_readfirstlane((int)sum);  // the same in every lane: the callers branch on it
}

// Makes the compiler forget what it knows about x (no instruction): keeps it from carrying dozens of
// 64-bit ballot masks across a barrier instead of recomputing them from one register.
__device__ __forceinline__ void opaque(unsigned& x) { asm volatile("" : "+v"(x)); }

// Population counts of four ballot masks, issued where they stand: left to the compiler, the counts
// of an unrolled loop are all moved behind the compares and every mask has to be kept (and spilled);
// one at a time, every compare waits for the previous count to free the mask register.
__device__ __forceinline__ unsigned mask_count4(unsigned long long m0, unsigned long long m1,
                                                unsigned long long m2, unsigned long long m3) {
    unsigned c0, c1, c2, c3;
    asm volatile(
        "s_bcnt1_i32_b64 %0, %4\n\ts_bcnt1_i32_b64 %1, %5\n\ts_bcnt1_i32_b64 %2, %6\n\ts_bcnt1_i32_b64 %3, %7"
        : "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
        : "s"(m0), "s"(m1), "s"(m2), "s"(m3)
        : "scc");
    return (c0 + c1) + (c2 + c3);
}
__device__ __forceinline__ unsigned mask_count(unsigned long long m) {
    unsigned c;
    asm volatile("s_bcnt1_i32_b64 %0, %1" : "=s"(c) : "s"(m) : "scc");
    return c;
}

// key_of(x) raised to at least 1 (key 0 stands for "no element"), in four instructions without a
// compare: the compiler's own sequence takes five and a wait between its compare and its select.
__device__ __forceinline__ unsigned key_of_min1(float f) {
    unsigned k;
    asm("v_ashrrev_i32 %0, 31, %1\n\tv_or_b32 %0, 0x80000000, %0\n\tv_xor_b32 %0, %1, %0\n\tv_max_u32 %0, 1, %0"
        : "=&v"(k)
        : "v"(f));
    return k;
}

// bits = bits << 4 | (k3 >= kb) << 3 | (k2 >= kb) << 2 | (k1 >= kb) << 1 | (k0 >= kb): every compare
// result enters as the carry of bits + bits, so a membership bit costs two vector instructions; the four
// compares stand in front of the four additions because a vector instruction may read a scalar register
// only two instructions after the vector instruction that wrote it.
__device__ __forceinline__ unsigned shift_in4_ge(unsigned bits, unsigned k3, unsigned k2, unsigned k1, unsigned k0,
                                                 unsigned kb) {
    unsigned long long m0, m1, m2, m3;
    asm("v_cmp_ge_u32_e64 %1, %5, %9\n\t"
        "v_cmp_ge_u32_e64 %2, %6, %9\n\t"
        "v_cmp_ge_u32_e64 %3, %7, %9\n\t"
        "v_cmp_ge_u32_e64 %4, %8, %9\n\t"
        "v_addc_co_u32_e64 %0, %1, %0, %0, %1\n\t"
        "v_addc_co_u32_e64 %0, %2, %0, %0, %2\n\t"
        "v_addc_co_u32_e64 %0, %3, %0, %0, %3\n\t"
        "v_addc_co_u32_e64 %0, %4, %0, %0, %4"
        : "+v"(bits), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(k3), "v"(k2), "v"(k1), "v"(k0), "s"(kb));
    return bits;
}

// The lanes named by four masks append their keys to an array in LDS: every such lane adds 4 to the word at
// cnt_addr (which starts at the array's address) and stores its key where the sum stood before.  The four
// atomics are in flight together; the execution mask is set per instruction and restored.
__device__ __forceinline__ void append4(unsigned cnt_addr, unsigned long long m0, unsigned long long m1,
                                        unsigned long long m2, unsigned long long m3, unsigned k0, unsigned k1,
                                        unsigned k2, unsigned k3) {
    unsigned s0, s1, s2, s3;
    unsigned long long save;
    const unsigned four = 4;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "s_mov_b64 exec, %[m0]\n\tds_add_rtn_u32 %[s0], %[ca], %[four]\n\t"
        "s_mov_b64 exec, %[m1]\n\tds_add_rtn_u32 %[s1], %[ca], %[four]\n\t"
        "s_mov_b64 exec, %[m2]\n\tds_add_rtn_u32 %[s2], %[ca], %[four]\n\t"
        "s_mov_b64 exec, %[m3]\n\tds_add_rtn_u32 %[s3], %[ca], %[four]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "ds_write_b32 %[s3], %[k3]\n\t"
        "s_mov_b64 exec, %[m2]\n\tds_write_b32 %[s2], %[k2]\n\t"
        "s_mov_b64 exec, %[m1]\n\tds_write_b32 %[s1], %[k1]\n\t"
        "s_mov_b64 exec, %[m0]\n\tds_write_b32 %[s0], %[k0]\n\t"
        "s_mov_b64 exec, %[save]"
        : [save] "=&s"(save), [s0] "=&v"(s0), [s1] "=&v"(s1), [s2] "=&v"(s2), [s3] "=&v"(s3)
        : [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3), [ca] "v"(cnt_addr), [four] "v"(four), [k0] "v"(k0),
          [k1] "v"(k1), [k2] "v"(k2), [k3] "v"(k3)
        : "memory");
}

// Items of a thread whose cohort index is inside the row for every row the (ITEMS, THREADS) variant is
// launched for (launch_topk_reg checks it): no range check is compiled for them.
constexpr int topk_safe_items(int items, int threads) {
    if (threads == 256) return items <= 4 ? 0 : items <= 8 ? 4 : items <= 16 ? 8 : items - 16;
    return items - (items <= 96 ? 16 : 32);
}

// number of set bits of m below this lane
__device__ __forceinline__ unsigned lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// What select() can tell about the fp32 uncertainty band [t - delta, t + delta] around the key t it
// returns, from the bracket candidates every wave holds anyway: if the band lies inside the bracket,
// `in_band` is the number of keys in it (t itself included).  One key in the band means there is
// nothing to re-decide and the scan of all ITEMS keys for band members is not needed.
struct BandProbe {
    int mode;      // 0: no probe; 1: delta = |t| * 2.4e-7; 2: delta = 2 * bound
    float bound;
    bool known;    // out: the band lies inside the bracket of candidates
    int in_band;   // out: keys with t - delta <= value <= t + delta (valid when known)
};

template <int ITEMS, int THREADS>
struct TopkRegs {
    unsigned key[ITEMS];

    template <class P>
    __device__ __forceinline__ unsigned wave_count(P pred) const {
        unsigned c = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; j += 4)
            c += mask_count4(__ballot(pred(key[j])), __ballot(pred(key[j + 1])), __ballot(pred(key[j + 2])),
                             __ballot(pred(key[j + 3])));
        return c;
    }

    // Smallest and largest real key of the row (key 0, "no element", can only stand in items >= SAFE).
    // Two keys per instruction and extreme (v_min3 / v_max3), one barrier.
    template <int SAFE>
    __device__ __forceinline__ void extremes(TopkShared& sh, int tid, unsigned* kmin, unsigned* kmax) const {
        const int lane = tid & 63, wave = tid >> 6;
        unsigned hi = 0, lo = 0xffffffffu, lo1 = 0xffffffffu;  // lo1: min of key - 1 over the items that may be 0
#pragma unroll
        for (int j = 0; j < ITEMS; j += 2) {
            hi = max(max(hi, key[j]), key[j + 1]);
            if (j + 1 < SAFE) lo = min(min(lo, key[j]), key[j + 1]);
            else lo1 = min(min(lo1, key[j] - 1u), key[j + 1] - 1u);
        }
        if (lo1 != 0xffffffffu) lo = min(lo, lo1 + 1u);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            hi = max(hi, (unsigned)__shfl_xor((int)hi, off, 64));
            lo = min(lo, (unsigned)__shfl_xor((int)lo, off, 64));
        }
        if (lane == 0) {
            sh.ext[0][wave] = lo;
            sh.ext[1][wave] = hi;
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < THREADS / 64; ++w) {
            lo = min(lo, sh.ext[0][w]);
            hi = max(hi, sh.ext[1][w]);
        }
        *kmin = (unsigned)__builtin_amdgcn_readfirstlane((int)lo);
        *kmax = (unsigned)__builtin_amdgcn_readfirstlane((int)hi);
    }

    // Key T at descending rank `want` (0-based), the number of keys > T and the number == T.  kmin / kmax:
    // the row's extreme keys -- a step whose threshold lies above the largest key counts nothing and one
    // whose threshold lies at or below the smallest counts everything, neither looks at the keys (the
    // exponent bits that all of a row's candidates share are most of the steps of the first phase).
    __device__ __forceinline__ void select(int want, int nc, unsigned kmin, unsigned kmax, TopkShared& sh, int& par,
                                           int tid, unsigned* out_key, int* out_gt, int* out_eq,
                                           BandProbe* band = nullptr) {
        if (band) band->known = false;
        const int lane = tid & 63, wave = tid >> 6;
        unsigned prefix = 0;
        int c_lo = nc;  // keys >= prefix
        int c_hi = 0;   // keys >= prefix + 2^(b+1): above the bracket
        int b = 31;
        for (; b >= 0; --b) {
            if (c_lo - c_hi <= TR_CAND && b < 31) break;
            const unsigned cand = prefix | (1u << b);
            int c;
            if (cand > kmax)
                c = 0;
            else if (cand <= kmin)
                c = nc;
            else
                c = block_sum<THREADS / 64>(sh, par, wave_count([cand](unsigned kx) { return kx >= cand; }), lane, wave);
            if (c > want) {
                prefix = cand;
                c_lo = c;
            } else {
                c_hi = c;
            }
        }
        if (b >= 0) {
            // bracket [prefix, prefix + 2^(b+1)) holds c_lo - c_hi <= TR_CAND keys: move them to registers
            const unsigned width = b == 31 ? 0u : (1u << (b + 1));  // b < 31 here
            const unsigned bracket_lo = prefix;
            // one compare per key: the bracket is aligned, so key 0 ("no element") falls outside it unless
            // it starts at 0, in which case it starts at 1 instead
            const unsigned first_key = prefix == 0 ? 1u : prefix, span = width - (first_key - prefix);
            auto inside = [first_key, span](unsigned kx) { return kx - first_key < span; };
            // Every candidate takes its place from a counter in LDS and is written there (append4): two
            // vector instructions per key (the subtraction and the compare) instead of the five of a ballot
            // compaction.  The places depend on the order the atomics arrive in, the set of candidates --
            // all that the later steps look at -- does not.  The counter starts at the array's own address.
            const unsigned cand_addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)sh.cand;
            const unsigned cnt_addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)&sh.ncand;
            if (tid == 0) sh.ncand = cand_addr;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < ITEMS; j += 4)
                append4(cnt_addr, __ballot(inside(key[j])), __ballot(inside(key[j + 1])), __ballot(inside(key[j + 2])),
                        __ballot(inside(key[j + 3])), key[j], key[j + 1], key[j + 2], key[j + 3]);
            // the compiler does not see the stores inside append4 and would let the barrier overtake them
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            // One wave takes all candidates (TR_CAND / 64 per lane) and finishes the remaining bits on its
            // own, without barriers or LDS traffic per step; the others wait for its result (every wave
            // doing the same steps, as before, spent four times the vector instructions on them).  The wave
            // changes from row to row, so that the rows a CU works on do not all load the same SIMD.
            const int n_in = c_lo - c_hi;
            if (__builtin_amdgcn_readfirstlane(wave) == (int)(blockIdx.x & (THREADS / 64 - 1))) {
            unsigned ck[TR_CAND / 64];
#pragma unroll
            for (int u = 0; u < TR_CAND / 64; ++u) {
                const int e = u * 64 + lane;
                ck[u] = e < n_in ? sh.cand[e] : 0u;  // key 0 never counts (candidates are >= 1)
            }
            const int above = c_hi;  // keys above the bracket as it was when the candidates were copied
            for (; b >= 0; --b) {
                if (c_lo - c_hi == 1) {
                    // One key left in the bracket [prefix, prefix + 2^(b+1)): it is the answer, and its
                    // remaining bits (about fourteen steps for distinct keys) need not be found one by one.
                    const unsigned w_now = 2u << b, first_now = prefix == 0 ? 1u : prefix;
                    unsigned mine = 0;
#pragma unroll
                    for (int u = 0; u < TR_CAND / 64; ++u)
                        if (ck[u] >= first_now && ck[u] - prefix < w_now) mine = ck[u];
                    const unsigned long long holder = __ballot(mine != 0);
                    prefix = (unsigned)__builtin_amdgcn_readlane((int)mine, __builtin_ctzll(holder));
                    break;  // c_hi keys lie above it, c_lo = c_hi + 1 at or above
                }
                const unsigned cand = prefix | (1u << b);
                unsigned wc = 0;
#pragma unroll
                for (int u = 0; u < TR_CAND / 64; u += 4)
                    wc += mask_count4(__ballot(ck[u] >= cand), __ballot(ck[u + 1] >= cand),
                                      __ballot(ck[u + 2] >= cand), __ballot(ck[u + 3] >= cand));
                const int c = above + (int)wc;
                if (c > want) {
                    prefix = cand;
                    c_lo = c;
                } else {
                    c_hi = c;
                }
            }
            if (band && band->mode != 0) {
                // the same float comparisons as the full scan, over the candidates only
                const float thr = float_of_key(prefix);
                const float delta = band->mode == 1 ? fabsf(thr) * 2.4e-7f : 2.f * band->bound;
                const float hi = thr + delta, lo = thr - delta;
                const unsigned first = bracket_lo == 0 ? 1u : bracket_lo, last_key = bracket_lo | (width - 1u);
                // finite band whose ends map into the bracket (keys are monotone in the value)
                // (a band end of zero is left to the full scan: -0 and +0 compare equal but have two keys)
                band->known = delta >= 0.f && fabsf(hi) <= 3.402823466e38f && fabsf(lo) <= 3.402823466e38f &&
                              hi != 0.f && lo != 0.f && key_of(lo) >= first && key_of(hi) <= last_key;
                unsigned wb = 0;
#pragma unroll
                for (int u = 0; u < TR_CAND / 64; ++u) {
                    const float v = float_of_key(ck[u]);
                    wb += mask_count(__ballot(ck[u] != 0 && v >= lo && !(v > hi)));
                }
                band->in_band = (int)wb;
            }
            if (lane == 0) {
                sh.res[0] = prefix;
                sh.res[1] = (unsigned)c_lo;
                sh.res[2] = (unsigned)c_hi;
                sh.res[3] = band && band->known ? 1u : 0u;
                sh.res[4] = band ? (unsigned)band->in_band : 0u;
            }
            }
            __syncthreads();
            prefix = (unsigned)__builtin_amdgcn_readfirstlane((int)sh.res[0]);
            c_lo = __builtin_amdgcn_readfirstlane((int)sh.res[1]);
            c_hi = __builtin_amdgcn_readfirstlane((int)sh.res[2]);
            if (band) {
                band->known = __builtin_amdgcn_readfirstlane((int)sh.res[3]) != 0;
                band->in_band = __builtin_amdgcn_readfirstlane((int)sh.res[4]);
            }
        }
        *out_key = prefix;
        *out_gt = c_hi;          // keys >= prefix + 1
        *out_eq = c_lo - c_hi;   // keys == prefix
    }
};

template <int ITEMS, int THREADS>
__device__ __forceinline__ void topk_rows_reg_body(const float* __restrict__ A, int64_t lda, int nc, int k,
                                                   int discard, uint16_t* __restrict__ idx8, int64_t ldi,
                                                   const ExactSrc& ex) {
    __shared__ __attribute__((aligned(16))) TopkShared sh;
    constexpr int TR_WAVES = THREADS / 64;
    __shared__ unsigned tab[TR_WAVES * ((ITEMS + 63) / 64) * 64];  // per (wave, item) counts for the ordered compaction
    __shared__ double exact_s[TK_CAND];
    __shared__ unsigned short cand_idx[2 * TK_CAND];
    __shared__ unsigned char cand_above[2 * TK_CAND];
    __shared__ __attribute__((aligned(16))) unsigned short stage[TR_STAGE];
    __shared__ __attribute__((aligned(16))) unsigned short ordered[TR_ORDER];
    __shared__ unsigned short bank_cnt[THREADS / 32][32];
    __shared__ unsigned short bank_start[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t xrow = blockIdx.x;  // row of the exact values and bounds
    const int64_t row = ex.row_map ? (int64_t)__builtin_amdgcn_readfirstlane(ex.row_map[xrow]) : xrow;
    int par = 0;
    TopkRegs<ITEMS, THREADS> R;
    const float* src = A + row * lda;
    // All loads of the row are issued before the first use, through a buffer descriptor of the row: the
    // address of an item that lies inside the row for every lane is the lane's byte offset plus a scalar,
    // so its load costs no vector instruction; the last items carry their offset in the lane's own (the
    // range check covers the vector offset only) and read zero behind the end of the row instead of the
    // next row or unmapped memory.
    // Key 0 is reserved for "no element": real keys are at least 1 (only the all-ones NaN maps lower).
    {
        constexpr int SAFE = topk_safe_items(ITEMS, THREADS);
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nc * 4, 0x00020000);
        float raw[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            raw[j] = __uint_as_float(j < SAFE ? __builtin_amdgcn_raw_buffer_load_b32(rs, tid * 4, j * THREADS * 4, 2)
                                              : __builtin_amdgcn_raw_buffer_load_b32(rs, (j * THREADS + tid) * 4, 0, 2));  // 2: non-temporal
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const unsigned kx = key_of_min1(raw[j]);
            R.key[j] = j < SAFE || j * THREADS + tid < nc ? kx : 0u;
        }
    }

    const int last = discard + k;
    const bool use_lo = discard > 0;
    unsigned key_hi, key_lo = 0xffffffffu;
    int gt_hi, eq_hi, gt_lo = 0, eq_lo = 0;
    BandProbe probe_hi = {0, 0.f, false, 0}, probe_lo = {0, 0.f, false, 0};
    if (ex.mode != 0) {
        probe_hi.mode = probe_lo.mode = ex.mode == 1 ? 1 : 2;
        if (ex.mode != 1)
            probe_hi.bound = probe_lo.bound =
                ex.err_coeff * (ex.bnd_rows[2 * xrow] * ex.coh_gmax + ex.bnd_rows[2 * xrow + 1] + ex.coh_qmax);
    }
    unsigned kmin, kmax;
    R.template extremes<topk_safe_items(ITEMS, THREADS)>(sh, tid, &kmin, &kmax);
    R.select(last - 1, nc, kmin, kmax, sh, par, tid, &key_hi, &gt_hi, &eq_hi, &probe_hi);
    if (use_lo) R.select(discard - 1, nc, kmin, kmax, sh, par, tid, &key_lo, &gt_lo, &eq_lo, &probe_lo);

    // what fp32 cannot resolve around a boundary value: re-decide those entries from float64 values
    int n_hi = 0, n_lo = 0;
    float thr_hi = 0.f, thr_lo = 0.f, d_hi = 0.f, d_lo = 0.f;
    if (ex.mode == 3) {
        // first of two calls: only note the rows that need exact values
        const bool settled = probe_hi.known && probe_hi.in_band <= 1 &&
                             (!use_lo || (probe_lo.known && probe_lo.in_band <= 1));
        if (!settled && tid == 0) ex.todo[atomicAdd(ex.todo_count, 1u)] = (int)row;
    } else if (ex.mode != 0) {
        thr_hi = float_of_key(key_hi);
        thr_lo = float_of_key(key_lo);
        if (ex.mode == 1) {  // the fp32 matrix is the rounding of the fp64 one
            d_hi = fabsf(thr_hi) * 2.4e-7f;
            d_lo = fabsf(thr_lo) * 2.4e-7f;
        } else {
            d_hi = d_lo = 2.f * probe_hi.bound;
        }
        // Entries above thr + delta are certainly above the boundary "rank < Rk"; m = Rk - (their
        // count) of the candidates within +-delta must be above, chosen by exact value (ties by index).
        auto redecide = [&](int Rk, float thr, float delta, unsigned short* cidx, unsigned char* cabove,
                            const BandProbe& probe) -> int {
            // the boundary key alone in its band (the same in every wave): nothing to decide
            if (probe.known && probe.in_band <= 1) return 0;
            if (tid == 0) sh.ncand = 0;
            const float hi = thr + delta, lo = thr - delta;
            __syncthreads();  // ncand is cleared before anyone appends
            unsigned wc = 0;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) opaque(R.key[j]);  // nothing of this pass is worth keeping
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const unsigned kx = R.key[j];
                const float v = float_of_key(kx);
                const bool above = kx != 0 && v > hi;
                wc += mask_count(__ballot(above));
                if (kx != 0 && !above && v >= lo) {
                    const unsigned pos = atomicAdd(&sh.ncand, 1u);
                    if (pos < TK_CAND) cidx[pos] = (unsigned short)(j * THREADS + tid);
                }
                if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
            const int n_above = block_sum<TR_WAVES>(sh, par, wc, lane, wave);
            __syncthreads();
            const int n_cand = (int)sh.ncand;
            const int m = Rk - n_above;
            if (n_cand > TK_CAND) {
                if (tid == 0 && ex.counters) atomicAdd(&ex.counters[1], 1ull);
                return 0;
            }
            if (m <= 0 || m >= n_cand) return 0;  // all candidates on one side: nothing to decide
            if (ex.mode == 1) {
                if (tid < n_cand) exact_s[tid] = ex.A64[row * ex.lda64 + cidx[tid]];
            } else {
                const int grp = tid >> 4, sub = tid & 15;
                const double* gr = ex.G_rows + xrow * ex.ldg;
                for (int a2 = grp; a2 < n_cand; a2 += THREADS / 16) {
                    const double* gc = ex.G_coh + (int64_t)cidx[a2] * ex.ldg;
                    double acc = 0.0;
                    for (int kk = sub; kk < ex.K; kk += 16) acc = fma(gr[kk], gc[kk], acc);
#pragma unroll
                    for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
                    if (sub == 0) exact_s[a2] = acc + gr[ex.K] + gc[ex.K];
                }
            }
            __syncthreads();
            if (tid < n_cand) {
                const double v = exact_s[tid];
                const unsigned short me = cidx[tid];
                int rank = 0;
                for (int b2 = 0; b2 < n_cand; ++b2) {
                    const double w = exact_s[b2];
                    rank += (w > v || (w == v && cidx[b2] < me)) ? 1 : 0;
                }
                cabove[tid] = rank < m ? 1 : 0;
            }
            __syncthreads();
            if (tid == 0 && ex.counters) atomicAdd(&ex.counters[0], 1ull);
            return n_cand;
        };
        n_hi = redecide(last, thr_hi, d_hi, cand_idx, cand_above, probe_hi);
        if (use_lo) n_lo = redecide(discard, thr_lo, d_lo, cand_idx + TK_CAND, cand_above + TK_CAND, probe_lo);
    }

    // Ordered positions.  Index order is item-major (i = j * 256 + tid), so the number of flagged
    // entries before entry (j, wave, lane) is base(j, wave) + (flagged lower lanes of the ballot), with
    // base(j, wave) = flagged entries of earlier (item, wave) pairs.  A wave counts its ITEMS ballots
    // into the lanes of NW registers (v_writelane), the four waves exchange them through LDS once,
    // and every wave scans the totals itself; base(j, wave) is then read back lane j (v_readlane).
    constexpr int NW = (ITEMS + 63) / 64, IP = NW * 64;
    constexpr int SELW = (ITEMS + 31) / 32;
    auto ordered_bases = [&](const unsigned (&bits)[SELW], unsigned (&basev)[NW]) {
        unsigned cntv[NW];
#pragma unroll
        for (int q = 0; q < NW; ++q) cntv[q] = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const unsigned c = mask_count(__ballot((bits[j >> 5] >> (j & 31)) & 1u));
            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(cntv[j >> 6]) : "s"(c), "i"(j & 63));
        }
        __syncthreads();  // earlier readers of tab are done
#pragma unroll
        for (int q = 0; q < NW; ++q) tab[wave * IP + 64 * q + lane] = cntv[q];
        __syncthreads();
        unsigned carry = 0;
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            unsigned tot = 0, mine_before = 0;
#pragma unroll
            for (int w = 0; w < TR_WAVES; ++w) {
                const unsigned v = tab[w * IP + 64 * q + lane];
                tot += v;
                if (w < wave) mine_before += v;
            }
            unsigned incl = tot;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned v = __shfl_up(incl, off, 64);
                if (lane >= off) incl += v;
            }
            basev[q] = carry + incl - tot + mine_before;
            carry += __shfl(incl, 63, 64);
        }
    };
    auto base_of = [](const unsigned (&basev)[NW], int j) -> unsigned {
        return __builtin_amdgcn_readlane(basev[j >> 6], j & 63);
    };

    // Which entries equal to a boundary key are above it: the first (wanted rank - #greater) of them in
    // index order.  When every tie is above (the usual case: one such entry) no order is needed.
    const int need_hi = last - gt_hi, need_lo = discard - gt_lo;  // ties to accept at each boundary
    const bool order_hi = need_hi < eq_hi, order_lo = use_lo && need_lo < eq_lo;
    unsigned hi_in[SELW], lo_in[SELW];  // bit j: a tie of item j would be above the boundary
#pragma unroll
    for (int w = 0; w < SELW; ++w) hi_in[w] = lo_in[w] = 0xffffffffu;
    auto tie_order = [&](unsigned key_b, int need, unsigned (&bits)[SELW]) {
        unsigned eq[SELW], basev[NW];
#pragma unroll
        for (int w = 0; w < SELW; ++w) eq[w] = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            if (R.key[j] == key_b) eq[j >> 5] |= 1u << (j & 31);
        ordered_bases(eq, basev);
#pragma unroll
        for (int w = 0; w < SELW; ++w) opaque(eq[w]);  // recompute the ballots rather than keep ITEMS masks
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const unsigned long long m = __ballot((eq[j >> 5] >> (j & 31)) & 1u);
            const int before = (int)(base_of(basev, j) + lanes_below(m));
            if (!(before < need)) bits[j >> 5] &= ~(1u << (j & 31));
        }
    };
    if (order_hi) tie_order(key_hi, need_hi, hi_in);
    if (order_lo) tie_order(key_lo, need_lo, lo_in);

    // bit j of above_hi / above_lo: item j is above the boundary "rank < last" / "rank < discard"
    unsigned above_hi[SELW], above_lo[SELW], valid[SELW];
#pragma unroll
    for (int w = 0; w < SELW; ++w) above_hi[w] = above_lo[w] = valid[w] = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) opaque(R.key[j]);
    if (!order_hi && !use_lo) {
        // the usual case -- one boundary, all its ties accepted: a member is a key >= the boundary key
        // (which is at least 1, so "no element" never qualifies); bits shifted in from the top item down
        const unsigned kb = (unsigned)__builtin_amdgcn_readfirstlane((int)key_hi);
#pragma unroll
        for (int w = 0; w < SELW; ++w) {
            unsigned bits = 0;
#pragma unroll
            for (int j = (32 * w + 31 < ITEMS ? 32 * w + 31 : ITEMS - 1); j >= 32 * w; j -= 4)
                bits = shift_in4_ge(bits, R.key[j], R.key[j - 1], R.key[j - 2], R.key[j - 3], kb);
            above_hi[w] = bits;
            valid[w] = 0xffffffffu;
        }
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const unsigned kx = R.key[j];
            const unsigned bit = 1u << (j & 31);
            if (kx > key_hi || (kx == key_hi && (hi_in[j >> 5] & bit))) above_hi[j >> 5] |= bit;
            if (use_lo && (kx > key_lo || (kx == key_lo && (lo_in[j >> 5] & bit)))) above_lo[j >> 5] |= bit;
            if (kx != 0) valid[j >> 5] |= bit;
        }
    }
    // re-decided boundary candidates override the fp32 decision of their owner (item i / 256 of thread i % 256)
    auto override_bits = [&](int n, const unsigned short* cidx, const unsigned char* cabove, unsigned (&bits)[SELW]) {
        for (int a2 = 0; a2 < n; ++a2) {
            const int i = cidx[a2];
            if ((i & (THREADS - 1)) == tid) {
                const int j = i / THREADS;
                const unsigned bit = 1u << (j & 31);
#pragma unroll
                for (int w = 0; w < SELW; ++w)
                    if (w == (j >> 5)) bits[w] = cabove[a2] ? (bits[w] | bit) : (bits[w] & ~bit);
            }
        }
    };
    if (n_hi) override_bits(n_hi, cand_idx, cand_above, above_hi);
    if (n_lo) override_bits(n_lo, cand_idx + TK_CAND, cand_above + TK_CAND, above_lo);
    unsigned selbits[SELW], basev[NW];
#pragma unroll
    for (int w = 0; w < SELW; ++w) selbits[w] = above_hi[w] & ~above_lo[w] & valid[w];
    const int groups = (k + 7) >> 3;

    if (k <= TR_ORDER) {
        // bank_order: the list leaves ordered by (index mod 32) -- the class of LDS banks a member's
        // values fall into in the gather -- and rotated by a (row mod 32)-th of its length, so that the
        // 32 lists a half wave walks in step are in different classes most of the time.  Member
        // j * 256 + tid is in class tid & 31: all members of a thread share one class, so a thread's
        // members go to consecutive places behind those of the threads of its class with a lower
        // number, and the order inside a list is (class, thread, item): the same for the same input.
        unsigned mine = 0;
#pragma unroll
        for (int w = 0; w < SELW; ++w) mine += __builtin_popcount(selbits[w]);
        const int b = tid & 31, part = tid >> 5;
        bank_cnt[part][b] = (unsigned short)mine;
        __syncthreads();
        if (tid < 32) {  // first place of every class: exclusive scan of the class totals
            int tot = 0;
#pragma unroll
            for (int q2 = 0; q2 < THREADS / 32; ++q2) tot += bank_cnt[q2][tid];
            int incl = tot;
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                const int v = __shfl_up(incl, off, 64);
                if (tid >= off) incl += v;
            }
            bank_start[tid] = (unsigned short)(incl - tot);
        }
        if (tid < 8 && k + tid < 8 * groups) ordered[k + tid] = 0;  // tail of the last group
        __syncthreads();
        int at = bank_start[b];
        for (int q2 = 0; q2 < part; ++q2) at += bank_cnt[q2][b];
        at -= (int)(((row & 31) * k) >> 5);  // the rotation
#pragma unroll
        for (int w = 0; w < SELW; ++w) {
            unsigned bits = selbits[w];
            while (bits) {
                const int j = 32 * w + __builtin_ctz(bits);
                bits &= bits - 1;
                const int dst = at < 0 ? at + k : at;
                if (dst < k) ordered[dst] = (unsigned short)(j * THREADS + tid);  // (always: exactly k are selected)
                ++at;
            }
        }
        __syncthreads();
        for (int g = tid; g < groups; g += THREADS)
            reinterpret_cast<uint4*>(idx8)[(int64_t)g * ldi + row] = reinterpret_cast<const uint4*>(ordered)[g];
        return;
    }

    // Longer lists leave in ascending index order.
    ordered_bases(selbits, basev);
#pragma unroll
    for (int w = 0; w < SELW; ++w) opaque(selbits[w]);  // recompute the ballots rather than keep ITEMS masks

    // The list goes through LDS so that it leaves as 16-byte groups (one store per eight entries)
    // instead of k scattered 2-byte stores; lists longer than the staging buffer are written directly.
    const bool staged = k <= TR_STAGE;
    if (tid < 8) {
        if (staged) stage[8 * (groups - 1) + tid] = 0;  // tail of the last group
        else if (k + tid < 8 * groups) idx8[((int64_t)(groups - 1) * ldi + row) * 8 + ((k + tid) & 7)] = 0;
    }
    if (staged) __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const bool sel = (selbits[j >> 5] >> (j & 31)) & 1u;
        const unsigned long long m_sel = __ballot(sel);
        const int pos = (int)(base_of(basev, j) + lanes_below(m_sel));
        if (sel && pos < k) {
            if (staged) stage[pos] = (unsigned short)(j * THREADS + tid);
            else idx8[((int64_t)(pos >> 3) * ldi + row) * 8 + (pos & 7)] = (uint16_t)(j * THREADS + tid);
        }
    }
    if (staged) {
        __syncthreads();
        for (int g = tid; g < groups; g += THREADS)
            reinterpret_cast<uint4*>(idx8)[(int64_t)g * ldi + row] = reinterpret_cast<const uint4*>(stage)[g];
    }
}

// The kernel is bound by instruction issue and its barriers: more resident waves help as long as the keys
// still fit the registers that leaves.  Measured for 20 000-member rows (80 keys per thread): 7.3 ms per
// three chunks at two waves per SIMD (what the compiler chooses on its own), 5.9 at three, 5.3 at four.
template <int ITEMS, int THREADS>
__global__ __launch_bounds__(THREADS) void topk_rows_reg_kernel(const float* __restrict__ A, int64_t lda, int nc,
                                                                int k, int discard, uint16_t* __restrict__ idx8,
                                                                int64_t ldi, ExactSrc ex) {
    topk_rows_reg_body<ITEMS, THREADS>(A, lda, nc, k, discard, idx8, ldi, ex);
}
template <int ITEMS, int THREADS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(3))) void topk_rows_reg_kernel_w3(
    const float* __restrict__ A, int64_t lda, int nc, int k, int discard, uint16_t* __restrict__ idx8, int64_t ldi,
    ExactSrc ex) {
    topk_rows_reg_body<ITEMS, THREADS>(A, lda, nc, k, discard, idx8, ldi, ex);
}
template <int ITEMS, int THREADS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4))) void topk_rows_reg_kernel_w4(
    const float* __restrict__ A, int64_t lda, int nc, int k, int discard, uint16_t* __restrict__ idx8, int64_t ldi,
    ExactSrc ex) {
    topk_rows_reg_body<ITEMS, THREADS>(A, lda, nc, k, discard, idx8, ldi, ex);
}

template <int ITEMS, int THREADS = TR_THREADS>
int launch_topk_reg(const float* A, int64_t n_rows, int nc, int64_t lda, int k, int discard, uint16_t* idx_t,
                     int64_t ldi, const ExactSrc& ex, hipStream_t stream) {
    static_assert(ITEMS % 4 == 0, "the membership bits are made four items at a time");
    HYPMI_REQUIRE(nc > topk_safe_items(ITEMS, THREADS) * THREADS && nc <= ITEMS * THREADS,
                  "topk_rows: variant <%d, %d> does not take rows of %d", ITEMS, THREADS, nc);
    // as many waves per SIMD as the keys leave registers for without spilling
    if constexpr (ITEMS <= 80)
        hipLaunchKernelGGL((topk_rows_reg_kernel_w4<ITEMS, THREADS>), dim3((unsigned)n_rows), dim3(THREADS), 0, stream,
                           A, lda, nc, k, discard, idx_t, ldi, ex);
    else if constexpr (ITEMS <= 96 && THREADS == 256)
        hipLaunchKernelGGL((topk_rows_reg_kernel_w3<ITEMS, THREADS>), dim3((unsigned)n_rows), dim3(THREADS), 0, stream,
                           A, lda, nc, k, discard, idx_t, ldi, ex);
    else
        hipLaunchKernelGGL((topk_rows_reg_kernel<ITEMS, THREADS>), dim3((unsigned)n_rows), dim3(THREADS), 0, stream, A,
                           lda, nc, k, discard, idx_t, ldi, ex);
    return HYPMI_OK;
}

// R rows of X per workgroup, held in LDS interleaved (xr[c * R + rr]); lanes over lists q.
//   mean[r, q] = sum_j X[r, L[j, q]] / norm,  istd[r, q] = 1 / max(sqrt(sum x^2 / norm - mean^2), floor)
// Without a mask the moments are accumulated in fp32 around a per-(row, list) pivot (the list's first
// member): the selected scores lie within a few standard deviations of each other, so the shifted
// sums carry no cancellation, and a constant cohort gives exactly zero variance.  With a mask
// (uint8 rows also in LDS, X already zeroed there) norm is the number of unmasked selected members
// and the sums are kept in double.
constexpr int GS_THREADS = 1024;

template <int R>
struct VecR;
template <>
struct VecR<1> { typedef float type; };
template <>
struct VecR<2> { typedef float2 type; };
template <>
struct VecR<4> { typedef float4 type; };

template <int R>
__device__ __forceinline__ void unpack(const typename VecR<R>::type& v, float (&o)[R]);
template <>
__device__ __forceinline__ void unpack<1>(const float& v, float (&o)[1]) { o[0] = v; }
template <>
__device__ __forceinline__ void unpack<2>(const float2& v, float (&o)[2]) { o[0] = v.x; o[1] = v.y; }
template <>
__device__ __forceinline__ void unpack<4>(const float4& v, float (&o)[4]) {
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}

// IN_LDS = false (R = 1 only) is the variant for cohorts whose row does not fit in LDS: the values
// are gathered from the row in global memory (L2), an order of magnitude slower per element.
template <int R, bool MASK, bool IN_LDS = true>
__global__ __launch_bounds__(GS_THREADS) void gather_stats_kernel(
    const float* __restrict__ X, int64_t ldx, int64_t n_rows, const unsigned char* __restrict__ mask,
    int64_t ldm, int nc, const uint16_t* __restrict__ idx8, int64_t ldi, int k, int64_t Q,
    int64_t q_per_block, float floor_, float* __restrict__ mean, float* __restrict__ istd, int64_t ldo) {
    static_assert(IN_LDS || R == 1, "rows gathered from global memory are taken one at a time");
    typedef typename VecR<R>::type vec_t;
    extern __shared__ __attribute__((aligned(16))) float xr[];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * R;
    if (IN_LDS) {
        unsigned char* mw = reinterpret_cast<unsigned char*>(xr + (size_t)nc * R);
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const bool live = r0 + rr < n_rows;
            const float* src = X + (r0 + rr) * ldx;
            for (int c = tid; c < nc; c += GS_THREADS)
                xr[c * R + rr] = live ? __builtin_nontemporal_load(src + c) : 0.f;
            if (MASK)
                for (int c = tid; c < nc; c += GS_THREADS) mw[c * R + rr] = live ? mask[(r0 + rr) * ldm + c] : 0;
        }
        __syncthreads();
    }
    // the address space of the gathered operand is a compile-time property of the instantiation
    const auto xv = [&]() {
        if constexpr (IN_LDS) return reinterpret_cast<const vec_t*>(xr);
        else return reinterpret_cast<const vec_t*>(X + r0 * ldx);
    }();
    const auto mr = [&]() {
        if constexpr (IN_LDS) return reinterpret_cast<const unsigned char*>(xr + (size_t)nc * R);
        else return mask + (MASK ? r0 * ldm : 0);
    }();
    const int kg = k >> 3, krem = k & 7;
    const int64_t q0 = (int64_t)blockIdx.y * q_per_block;
    const int64_t q1 = (q0 + q_per_block < Q) ? q0 + q_per_block : Q;
    for (int64_t q = q0 + tid; q < q1; q += GS_THREADS) {
        const uint4* lp = reinterpret_cast<const uint4*>(idx8) + q;  // one uint4 = 8 entries of list q
        float out_m[R], out_i[R];
        if (!MASK) {
            float piv[R], s1[R], s2[R];
            {
                const unsigned first = lp[0].x & 0xffffu;
                unpack<R>(xv[first], piv);
            }
#pragma unroll
            for (int rr = 0; rr < R; ++rr) s1[rr] = s2[rr] = 0.f;
            auto add = [&](unsigned c) {
                float v[R];
                unpack<R>(xv[c], v);
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    const float d = v[rr] - piv[rr];
                    s1[rr] += d;
                    s2[rr] = fmaf(d, d, s2[rr]);
                }
            };
#pragma unroll 2
            for (int g = 0; g < kg; ++g) {
                const uint4 pk = lp[(int64_t)g * ldi];
                add(pk.x & 0xffffu); add(pk.x >> 16);
                add(pk.y & 0xffffu); add(pk.y >> 16);
                add(pk.z & 0xffffu); add(pk.z >> 16);
                add(pk.w & 0xffffu); add(pk.w >> 16);
            }
            if (krem) {
                const uint4 pk = lp[(int64_t)kg * ldi];
                const unsigned w[4] = {pk.x, pk.y, pk.z, pk.w};
                for (int j = 0; j < krem; ++j) add((w[j >> 1] >> (16 * (j & 1))) & 0xffffu);
            }
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const double md = (double)s1[rr] / (double)k;
                double var = (double)s2[rr] / (double)k - md * md;
                double sd = sqrt(var > 0.0 ? var : (var == var ? 0.0 : var));
                if (sd < (double)floor_) sd = (double)floor_;
                out_m[rr] = (float)((double)piv[rr] + md);
                out_i[rr] = (float)(1.0 / sd);
            }
        } else {
            double s1[R], s2[R];
            int cnt[R];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                s1[rr] = s2[rr] = 0.0;
                cnt[rr] = 0;
            }
            for (int j = 0; j < k; ++j) {
                const uint4 pk = lp[(int64_t)(j >> 3) * ldi];
                const unsigned w[4] = {pk.x, pk.y, pk.z, pk.w};
                const unsigned c = (w[(j & 7) >> 1] >> (16 * (j & 1))) & 0xffffu;
                float v[R];
                unpack<R>(xv[c], v);
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    s1[rr] += (double)v[rr];
                    s2[rr] += (double)v[rr] * (double)v[rr];
                    cnt[rr] += mr[c * R + rr] ? 1 : 0;
                }
            }
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const double m = s1[rr] / (double)cnt[rr];
                double var = s2[rr] / (double)cnt[rr] - m * m;
                double sd = sqrt(var > 0.0 ? var : (var == var ? 0.0 : var));  // keep NaN (0/0) visible
                if (sd < (double)floor_) sd = (double)floor_;
                out_m[rr] = (float)m;
                out_i[rr] = (float)(1.0 / sd);
            }
        }
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            if (r0 + rr < n_rows) {
                mean[(r0 + rr) * ldo + q] = out_m[rr];
                istd[(r0 + rr) * ldo + q] = out_i[rr];
            }
        }
    }
}

// ---- unmasked, LDS-resident gather (the adaptive S-norm hot loop) -----------------------------------
// Measured on MI355X (tools/probes/gather_rate.hip, lds_rate.hip): the walk is bound by instruction
// issue, not by LDS bytes -- a SIMD spends about 10 clocks on a ds_read_b64, 3 on a plain f32 add or
// fma, 4.5 on an integer multiply-add and 7.7 on a packed f32 instruction (more than two plain ones),
// and these add up; a dependent index load per eight entries stalls the wave unless several are in
// flight; random lists cost 3.5 LDS cycles per half wave instead of 1.  Hence:
//   * the rows are stored in LDS minus a per-row pivot (an estimate of the row's mean), so the loop
//     has no subtraction: one v_mad_u32_u16 (16-bit entry -> byte address), one LDS read, and a plain
//     add + fma per value, written as single instructions so that they are not packed;
//   * every lane keeps GP_RING 16-byte index loads in flight over the flattened (list, group)
//     sequence of its lists;
//   * the lists arrive ordered by LDS bank class and rotated by lane (see bank_order in the top-k
//     kernel), which removes most bank conflicts without any coordination between lists;
//   * the waves of a SIMD are served oldest first, which lets the first of them finish its lists well
//     before the last and leaves the SIMD short of work at the end of every workgroup: each wave moves
//     through the four issue-priority levels, one per list.
// The walk alone runs at 12.7 us per round of 1024 lists on 2 x 20 000 floats (0.315 of the LDS read
// rate); a workgroup pays about 11 us more for its row fill and the finish (tools/gather_overhead.py).
// With a per-row pivot the variance is s2/k - (s1/k)^2 with s2/k up to GP_AMP times the variance; for
// the few (row, list) pairs beyond that (a selection far out in the tail of its row, or constant)
// the lane walks the list again around the list's own first member, which is what the general kernel
// above does for every pair.
constexpr int MAX_LDS_BYTES = 160 * 1024;  // LDS of a CU: what one workgroup of the gather kernels may take
constexpr int GP_RING = 4;
constexpr float GP_AMP = 16.f;
constexpr int GP_XCD_GROUPS = 16;  // row groups that share 128-byte lines of the transposed T-part statistics

// The normalised score from the raw score and the two pairs of cohort statistics (mean, 1 / std):
// one rounding per operation in this order wherever it is computed (no contraction into fma), so that
// the fused and the separate paths give the same bits.
__device__ __forceinline__ float asnorm_value(float s, float mz, float iz, float mt, float it) {
#pragma clang fp contract(off)  // __fmul_rn / __fadd_rn are plain operators here and would fuse into z + (s - mt) it
    const float z = __fmul_rn(__fsub_rn(s, mz), iz);
    const float t = __fmul_rn(__fsub_rn(s, mt), it);
    return __fmul_rn(__fadd_rn(z, t), 0.70710678118654752440f);
}

// What the gather needs to finish the normalisation itself (rows = enrolment models, lists = tests):
// raw scores S [rows, lists], T-part statistics mt / it [lists, rows] and the result [rows, lists].
struct ApplyArgs {
    const float* S;
    int64_t lds;
    const float* mt;
    const float* it;
    int64_t ldt;
    float* out;
    int64_t ldo;
};
constexpr int GP_SCRATCH_FLOATS = 80;  // per-wave partial sums of the pivot estimate (R x 16) + pad

__device__ unsigned long long g_gather_redo;  // (row, list) pairs redone around a list pivot (diagnostic)
// Diagnostic: shader cycles and ticks of the constant 100 MHz clock that the first wave of every workgroup of
// gather_pivot_kernel has spent between its entry and its last list: their ratio is the clock the kernel ran at.
__device__ unsigned long long g_gather_clock[2];

// An LDS read at a byte address of the LDS address space (not relative to an array: the address
// computation already contains the array's base, so that it costs no instruction of its own).
template <int R>
struct LdsVec {
    typedef float type __attribute__((ext_vector_type(R)));
};
template <>
struct LdsVec<1> { typedef float type; };

template <int R>
__device__ __forceinline__ typename LdsVec<R>::type lds_vec(unsigned addr) {
    typedef __attribute__((address_space(3))) const typename LdsVec<R>::type lds_t;
    return *reinterpret_cast<lds_t*>(addr);
}

template <int R>
__device__ __forceinline__ float lds_vec_elem(const typename LdsVec<R>::type& v, int rr) { return v[rr]; }
template <>
__device__ __forceinline__ float lds_vec_elem<1>(const float& v, int) { return v; }

// R (mean, 1 / std) pairs from global memory, 8 R bytes at an address that is a multiple of min(8 R, 16)
template <int R>
__device__ __forceinline__ typename LdsVec<2 * R>::type load_pairs(const float* p) {
    typename LdsVec<2 * R>::type v;
    if constexpr (R == 1) {
        const float2 a = *reinterpret_cast<const float2*>(p);
        v[0] = a.x; v[1] = a.y;
    } else {
#pragma unroll
        for (int h = 0; h < R / 2; ++h) {
            const float4 a = *reinterpret_cast<const float4*>(p + 4 * h);
            v[4 * h] = a.x; v[4 * h + 1] = a.y; v[4 * h + 2] = a.z; v[4 * h + 3] = a.w;
        }
    }
    return v;
}

__device__ __forceinline__ unsigned lds_address(const void* p) {
    return (unsigned)reinterpret_cast<size_t>((__attribute__((address_space(3))) const void*)p);
}

// LDS byte address of 16-bit entry (low or high half of w) of an array of R floats per entry at
// LDS address `base`: one instruction
template <int R, int HI>
__device__ __forceinline__ unsigned entry_addr(unsigned w, unsigned base) {
    unsigned a;
    if (HI) asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(a) : "v"(w), "n"(4 * R), "s"(base));
    else asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(a) : "v"(w), "n"(4 * R), "s"(base));
    return a;
}

// What the gather kernels keep in LDS of a row entry minus the row's pivot: the value rounded to 16 bits of
// mantissa (nearest; the low 8 bits zero).  These kernels run at the clock the chip's power budget leaves
// them -- the same cycles per workgroup at 2.20 - 2.22 GHz on rows of random scores and at 2.38 GHz on constant
// rows -- and the zero bits toggle less in the LDS data path, the adders and the multipliers: 2.26 GHz, both
// gathers 1.1 % (benchmark scores) to 2.7 % (normal random scores) faster.  The rounding errors (7.6e-6 relative
// per value, unbiased) average out over a list -- the mean moves by at most 3e-5 / sqrt(k) standard deviations,
// the variance by GP_AMP * 1.5e-5 / sqrt(k) of itself -- so only lists of at least GP_ROUND_MIN_K members are
// walked on rounded values (2e-6 of a standard deviation, what the fp32 sums themselves carry).  The pairs that
// are walked again around their own pivot take the unrounded values from global memory.
constexpr int GP_ROUND_MIN_K = 256;
__device__ __forceinline__ float lds_rounded(float v, bool on) {
    return on ? __uint_as_float((__float_as_uint(v) + 0x80u) & 0xffffff00u) : v;
}

__device__ __forceinline__ void moment_step(float& s1, float& s2, float v) {
    asm("v_add_f32 %0, %1, %0" : "+v"(s1) : "v"(v));
    asm("v_fma_f32 %0, %1, %1, %0" : "+v"(s2) : "v"(v));
}

// The general walk of one list for R rows, around the list's first member (fp32 shifted moments: exact
// zero variance for a constant selection).  Cold path of gather_pivot_kernel.  The values come from the
// rows in global memory, not from the LDS copy: that copy is already rounded after the subtraction of
// the row pivot, which is too coarse exactly for the selections that end up here (spread far below
// their distance from the row's mean).
template <int R>
struct ShiftedMoments {
    float piv[R], s1[R], s2[R];
};

// Returned by value (registers): taking the caller's accumulators by reference would put them in
// scratch memory for the whole hot loop.
template <int R>
__device__ __noinline__ ShiftedMoments<R> list_moments_shifted(const float* X, int64_t ldx, int64_t r0,
                                                               int64_t n_rows, const uint4* lp, int64_t ldi, int k) {
    ShiftedMoments<R> m;
    const float* row[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        row[rr] = X + (r0 + rr < n_rows ? r0 + rr : n_rows - 1) * ldx;
        m.piv[rr] = row[rr][lp[0].x & 0xffffu];
        m.s1[rr] = m.s2[rr] = 0.f;
    }
    for (int j = 0; j < k; ++j) {
        const uint4 pk = lp[(int64_t)(j >> 3) * ldi];
        const unsigned w = (j & 4) ? ((j & 2) ? pk.w : pk.z) : ((j & 2) ? pk.y : pk.x);
        const unsigned c = (w >> (16 * (j & 1))) & 0xffffu;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const float d = row[rr][c] - m.piv[rr];
            m.s1[rr] += d;
            m.s2[rr] = fmaf(d, d, m.s2[rr]);
        }
    }
    return m;
}

// APPLY: instead of storing (mean, 1 / std) the kernel reads the raw score and the other side's
// statistics of every (row, list) pair and stores the normalised score.  Those statistics are
// transposed ([lists, rows]): a lane reads 4 R bytes of a 128-byte line that the workgroups of the
// neighbouring rows need at about the same time, so the row groups are numbered such that
// GP_XCD_GROUPS consecutive ones run on the same XCD (workgroups go to the XCDs round-robin) and
// share those lines in its L2.  The loads ride in the shadow of the walk, which leaves HBM idle.
// Issue priority of the calling wave: level (turn mod 4).  The four waves of a SIMD take turns at every
// level, so that none of them is served first for its whole life.
__device__ __forceinline__ void rotate_priority(int turn) {
    switch (turn & 3) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}

// MODE 0: (mean, 1 / std) stored, in two arrays or -- istd == nullptr -- as pairs in one ([rows, ldo] pairs);
// 1: APPLY with the other side's statistics in two arrays; 2: APPLY with them as pairs ([lists, ldt] pairs: one
// load of 8 R bytes per lane and list instead of two of 4 R -- a lane's load is a line of its own, and
// every such line costs the CU about a clock of its vector-memory path that the walk cannot use).
template <int R, int MODE>
__global__ __launch_bounds__(GS_THREADS) void gather_pivot_kernel(
    const float* __restrict__ X, int64_t ldx, int64_t n_rows, int nc, const uint16_t* __restrict__ idx8,
    int64_t ldi, int k, int64_t Q, int64_t q_per_block, float floor_, float* __restrict__ mean,
    float* __restrict__ istd, int64_t ldo, ApplyArgs ap, unsigned* __restrict__ front) {
    constexpr bool APPLY = MODE != 0, PAIRS = MODE == 2;
    const unsigned long long cycles_in = __builtin_readcyclecounter(), ticks_in = __builtin_amdgcn_s_memrealtime();
    extern __shared__ __attribute__((aligned(16))) float xr[];
    float* red = xr + (size_t)nc * R;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t grp = blockIdx.x;
    if (APPLY) {  // the grid is padded to whole sets of 8 x GP_XCD_GROUPS row groups
        const int64_t slot = grp >> 3, xcd = grp & 7;
        grp = ((slot / GP_XCD_GROUPS) * 8 + xcd) * GP_XCD_GROUPS + slot % GP_XCD_GROUPS;
        if (grp * R >= n_rows) return;
    }
    const int64_t r0 = grp * R;
    // The rows come in with 16-byte loads that are all issued before anything waits for them (one row
    // of 20 000 floats is five loads per lane; fetched one float at a time, each behind its own wait,
    // the fill took a quarter of the time of a workgroup that walks ten lists per lane).
    // pivot of a row: mean of its first min(nc, 1024) entries (the cohort order carries no meaning)
    float rowpiv[R];
    const bool round_lds = k >= GP_ROUND_MIN_K;
    const bool vec_ok = ldx % 4 == 0 && reinterpret_cast<uintptr_t>(X) % 16 == 0 && nc >= 4;  // uniform
    if (vec_ok) {
        // float4 pieces per lane and row for the largest cohort that fits LDS with R rows
        constexpr int FILL = ((MAX_LDS_BYTES / 4 - GP_SCRATCH_FLOATS) / R / 4 + GS_THREADS - 1) / GS_THREADS;
        const int nc4 = nc >> 2, rem = nc & 3;
        typedef typename LdsVec<4>::type f4;
        f4 v[FILL][R];
        float tail[R];
#pragma unroll
        for (int it = 0; it < FILL; ++it) {
            const int c4 = tid + it * GS_THREADS;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                v[it][rr] = f4{0.f, 0.f, 0.f, 0.f};
                if (c4 < nc4 && r0 + rr < n_rows)
                    v[it][rr] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(X + (r0 + rr) * ldx) + c4);
            }
        }
#pragma unroll
        for (int rr = 0; rr < R; ++rr)
            tail[rr] = (tid < rem && r0 + rr < n_rows) ? X[(r0 + rr) * ldx + 4 * nc4 + tid] : 0.f;
        const int ns4 = nc4 < GS_THREADS / 4 ? nc4 : GS_THREADS / 4;  // the first 4 ns4 entries make the pivot
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            float p = tid < ns4 ? (v[0][rr].x + v[0][rr].y) + (v[0][rr].z + v[0][rr].w) : 0.f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
            if (lane == 0) red[rr * 16 + wave] = p;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < GS_THREADS / 64; ++w) sum += red[rr * 16 + w];
            rowpiv[rr] = sum / (float)(4 * ns4);
        }
#pragma unroll
        for (int it = 0; it < FILL; ++it) {
            const int c4 = tid + it * GS_THREADS;
            if (c4 < nc4) {
                float* dst = xr + (size_t)c4 * 4 * R;
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    dst[0 * R + rr] = lds_rounded(v[it][rr].x - rowpiv[rr], round_lds);
                    dst[1 * R + rr] = lds_rounded(v[it][rr].y - rowpiv[rr], round_lds);
                    dst[2 * R + rr] = lds_rounded(v[it][rr].z - rowpiv[rr], round_lds);
                    dst[3 * R + rr] = lds_rounded(v[it][rr].w - rowpiv[rr], round_lds);
                }
            }
        }
        if (tid < rem) {
#pragma unroll
            for (int rr = 0; rr < R; ++rr) xr[(size_t)(4 * nc4 + tid) * R + rr] = lds_rounded(tail[rr] - rowpiv[rr], round_lds);
        }
    } else {
        const int ns = nc < GS_THREADS ? nc : GS_THREADS;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            float v = (r0 + rr < n_rows && tid < ns) ? X[(r0 + rr) * ldx + tid] : 0.f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0) red[rr * 16 + wave] = v;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < GS_THREADS / 64; ++w) sum += red[rr * 16 + w];
            rowpiv[rr] = sum / (float)ns;
        }
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const bool live = r0 + rr < n_rows;
            const float* src = X + (r0 + rr) * ldx;
            for (int c = tid; c < nc; c += GS_THREADS)
                xr[c * R + rr] = live ? lds_rounded(__builtin_nontemporal_load(src + c) - rowpiv[rr], round_lds) : 0.f;
        }
    }
    __syncthreads();
    const unsigned xbase = __builtin_amdgcn_readfirstlane(lds_address(xr));

    const int G = (k + 7) >> 3, krem = k & 7;
    const int64_t q0 = (int64_t)blockIdx.y * q_per_block;
    const int64_t q1 = (q0 + q_per_block < Q) ? q0 + q_per_block : Q;
    if (q0 >= q1) return;
    // Entries past the end of a list (the last group holds krem of them) are redirected to a slot
    // of zeros behind the rows: a zero adds nothing to either moment.
    if (tid < R) xr[(size_t)nc * R + tid] = 0.f;  // `red` is no longer needed
    // The lists of the block are handed out to the waves in units of 64 (unit u = lists q0 + 64 u ...): a
    // wave starts with units `wave` and `wave + 16` and then takes the next one nobody has, from a counter in
    // LDS.  The waves of a SIMD do not advance at the same rate (arbitration by age, LDS conflicts of their own
    // lists); with a fixed share each the first wave of a workgroup finished well before the last, and a SIMD
    // with one or two waves left cannot fill its issue slots.
    unsigned* unit_counter = reinterpret_cast<unsigned*>(xr + (size_t)nc * R + 8);
    // Where the workgroup starts in its block of lists.  Every row group walks the same lists, and a list
    // block is larger than an L2; workgroups that started at different times would stream different parts
    // of it through the L2 of their XCD at once.  `front` (one counter per XCD class of workgroups and list
    // block, zero at launch) holds the furthest position any of them has reached, in units, counted without
    // wrapping: a new workgroup starts there and wraps around, so that the workgroups of an XCD read the same
    // few rounds of lists (and, with APPLY, the same lines of the other side's statistics) at the same time.
    unsigned* my_front = front ? front + (blockIdx.y * 8 + (blockIdx.x & 7)) : nullptr;
    unsigned* start_word = reinterpret_cast<unsigned*>(xr + (size_t)nc * R + 9);
    if (tid == 0) {
        *unit_counter = 2 * (GS_THREADS / 64);
        *start_word = my_front ? __hip_atomic_load(my_front, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
    __syncthreads();
    unsigned keep[4], fill[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const bool lo_ok = krem == 0 || 2 * w < krem, hi_ok = krem == 0 || 2 * w + 1 < krem;
        keep[w] = (lo_ok ? 0xffffu : 0u) | (hi_ok ? 0xffff0000u : 0u);
        fill[w] = (lo_ok ? 0u : (unsigned)nc) | (hi_ok ? 0u : (unsigned)nc << 16);
    }
    constexpr int WAVES = GS_THREADS / 64;
    const int U = __builtin_amdgcn_readfirstlane((int)((q1 - q0 + 63) >> 6));  // units of the block
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    if (uw >= U) return;  // (no barrier follows)
    const unsigned v_start = __builtin_amdgcn_readfirstlane(*start_word);
    const int u0 = (int)(v_start % (unsigned)U);
    // unit number (the order in which this workgroup takes them) -> position in the block
    auto place = [&](int u) -> int { return u + u0 < U ? u + u0 : u + u0 - U; };
    // A unit is claimed one unit ahead of the index stream, which itself runs GP_RING groups ahead of the
    // arithmetic; with lists of at most GP_RING groups the stream could cross two units before the
    // arithmetic has left one, and the shares stay fixed (unit u is followed by u + 16).
    const bool dynamic = G > GP_RING;
    const uint4* base = reinterpret_cast<const uint4*>(idx8);
    // The index loads of a wave form one stream over (unit, group).  Group g of unit u starts at
    // idx8 + (g ldi + q0 + 64 u) 16 bytes -- a uniform address, stepped with scalar instructions -- plus 16
    // bytes per lane, so the loads go through a buffer descriptor (scalar base, per-lane offset) and the
    // stream costs no vector instruction besides the load.  The descriptor ends behind the last list of the
    // block: lanes of the last unit that have no list read zeros (and store nothing).
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const char* sp0 = reinterpret_cast<const char*>(idx8) + q0 * 16;  // uniform
    const int64_t step_b = ldi * 16;
    const unsigned voff = (unsigned)lane * 16u;
    int u_ld = uw, u_next = uw + WAVES, g_ld = 0;  // uniform: unit and group of the next load, the unit after it
    unsigned claim = 0;                            // lane 0: the unit claimed at the last crossing
    const char* sp = sp0 + (int64_t)place(u_ld) * 1024;
    unsigned nrec =
        (unsigned)((q1 - q0 - 64 * (int64_t)place(u_ld) < 64 ? q1 - q0 - 64 * (int64_t)place(u_ld) : 64) * 16);
    int u_after = u_next;  // the unit the arithmetic takes after its current one
#define HYPMI_NEXT_LOAD(dst)                                                                              \
    {                                                                                                     \
        const v4u w_ = __builtin_amdgcn_raw_buffer_load_b128(                                             \
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sp), 0, nrec, 0x00020000), voff, 0, 0);   \
        dst = make_uint4(w_[0], w_[1], w_[2], w_[3]);                                                     \
        if (++g_ld < G) {                                                                                 \
            sp += step_b;                                                                                 \
        } else if (u_ld < U) {                                                                            \
            /* the stream moves on to the next unit of this wave and claims the one after that */         \
            g_ld = 0;                                                                                     \
            u_ld = u_next;                                                                                \
            u_after = u_ld;                                                                               \
            if (dynamic) {                                                                                \
                u_next = __builtin_amdgcn_readfirstlane(claim);                                           \
                if (lane == 0) claim = atomicAdd(unit_counter, 1u);                                       \
            } else {                                                                                      \
                u_next = u_ld + WAVES;                                                                    \
            }                                                                                             \
            if (u_ld < U) {                                                                               \
                const int at_ = place(u_ld);                                                              \
                const int64_t left = q1 - q0 - 64 * (int64_t)at_;                                         \
                sp = sp0 + (int64_t)at_ * 1024;                                                           \
                nrec = (unsigned)((left < 64 ? left : 64) * 16);                                          \
                /* one wave per workgroup reports how far the workgroup has come */                      \
                if (my_front && uw == 0 && lane == 0)                                                     \
                    __hip_atomic_fetch_max(my_front, v_start + (unsigned)u_ld, __ATOMIC_RELAXED,          \
                                           __HIP_MEMORY_SCOPE_AGENT);                                     \
            } else {                                                                                      \
                sp -= (int64_t)(G - 1) * step_b; /* the stream is over: further loads re-read a unit */   \
            }                                                                                             \
        } else {                                                                                          \
            g_ld = 0;                                                                                     \
            sp -= (int64_t)(G - 1) * step_b;                                                              \
        }                                                                                                 \
    }
    // the first claim: it is picked up when the stream crosses into the wave's second unit
    if (dynamic && lane == 0) claim = atomicAdd(unit_counter, 1u);
    // APPLY: what the end of a list needs from global memory -- the raw scores of the R rows and the R
    // transposed statistics (one 4 R-byte load each) -- is requested when the list begins, so that the
    // requests of the sixteen waves (which reach the ends of their lists together) have a whole list's
    // walk to drain instead of stalling everyone at once.
    typename LdsVec<R>::type pm, pi;
    typename LdsVec<2 * R>::type pmi;  // PAIRS: (mean, 1 / std) of row 0, of row 1, ...
    float ps[R];
#define HYPMI_APPLY_PREFETCH(UNIT)                                                                  \
    if (APPLY) {                                                                                    \
        int64_t qq = q0 + lane + (int64_t)place(UNIT) * 64;                                         \
        qq = qq < q1 ? qq : q1 - 1;                                                                 \
        if (PAIRS) {                                                                                \
            pmi = load_pairs<R>(ap.mt + (qq * ap.ldt + r0) * 2);                                    \
        } else {                                                                                    \
            pm = *reinterpret_cast<const typename LdsVec<R>::type*>(ap.mt + qq * ap.ldt + r0);      \
            pi = *reinterpret_cast<const typename LdsVec<R>::type*>(ap.it + qq * ap.ldt + r0);      \
        }                                                                                           \
        _Pragma("unroll") for (int rr = 0; rr < R; ++rr) {                                          \
            const int64_t e = r0 + rr < n_rows ? r0 + rr : n_rows - 1;                              \
            ps[rr] = __builtin_nontemporal_load(ap.S + e * ap.lds + qq);                            \
        }                                                                                           \
    }
    int u_cp = uw;  // the unit of the arithmetic
    HYPMI_APPLY_PREFETCH(u_cp)
    uint4 ring[GP_RING];
#pragma unroll
    for (int d = 0; d < GP_RING; ++d) HYPMI_NEXT_LOAD(ring[d])

    float s1[R], s2[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) s1[rr] = s2[rr] = 0.f;
    int g = 0, l = 0;
    const float inv_k = 1.0f / (float)k, floor2 = floor_ * floor_, inv_floor = 1.0f / floor_;
    // One group of eight entries from ring slot D: addresses, then all eight reads, then the arithmetic
    // (the fences keep the reads of a group in flight together instead of one wait per read).  The
    // slot is reloaded once its contents have become addresses, so that the new group can take the
    // registers of the old one; every slot is reloaded unconditionally (behind the end of the stream the
    // loads repeat its last unit), which lets the compiler count the loads in flight instead of waiting
    // for all of them.
#define HYPMI_GROUP(D)                                                                                          \
    {                                                                                                           \
        unsigned a[8];                                                                                          \
        if (krem != 0 && g == G - 1) {                                                                          \
            const unsigned w0 = (ring[D].x & keep[0]) | fill[0], w1 = (ring[D].y & keep[1]) | fill[1],          \
                           w2 = (ring[D].z & keep[2]) | fill[2], w3 = (ring[D].w & keep[3]) | fill[3];          \
            a[0] = entry_addr<R, 0>(w0, xbase); a[1] = entry_addr<R, 1>(w0, xbase);                             \
            a[2] = entry_addr<R, 0>(w1, xbase); a[3] = entry_addr<R, 1>(w1, xbase);                             \
            a[4] = entry_addr<R, 0>(w2, xbase); a[5] = entry_addr<R, 1>(w2, xbase);                             \
            a[6] = entry_addr<R, 0>(w3, xbase); a[7] = entry_addr<R, 1>(w3, xbase);                             \
        } else {                                                                                                \
            a[0] = entry_addr<R, 0>(ring[D].x, xbase); a[1] = entry_addr<R, 1>(ring[D].x, xbase);               \
            a[2] = entry_addr<R, 0>(ring[D].y, xbase); a[3] = entry_addr<R, 1>(ring[D].y, xbase);               \
            a[4] = entry_addr<R, 0>(ring[D].z, xbase); a[5] = entry_addr<R, 1>(ring[D].z, xbase);               \
            a[6] = entry_addr<R, 0>(ring[D].w, xbase); a[7] = entry_addr<R, 1>(ring[D].w, xbase);               \
        }                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        HYPMI_NEXT_LOAD(ring[D])                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        typename LdsVec<R>::type val[8];                                                                        \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) val[e] = lds_vec<R>(a[e]);                                \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                           \
            _Pragma("unroll") for (int rr = 0; rr < R; ++rr)                                                    \
                moment_step(s1[rr], s2[rr], lds_vec_elem<R>(val[e], rr));                                       \
        if (++g == G) {                                                                                         \
            /* The sums carry fp32 rounding (about 1e-6 relative); so does this arithmetic. */                  \
            const int64_t q = q0 + lane + (int64_t)place(u_cp) * 64;                                            \
            const bool live_q = q < q1;                                                                         \
            float lp[R]; /* what the moments are taken around: the row pivot, or after a redo the list's own */ \
            bool redo = false;                                                                                  \
            _Pragma("unroll") for (int rr = 0; rr < R; ++rr) {                                                  \
                lp[rr] = rowpiv[rr];                                                                            \
                const float md = s1[rr] * inv_k, ms = s2[rr] * inv_k;                                           \
                redo |= fmaf(-md, md, ms) * GP_AMP < ms; /* false for NaN: those stay NaN */                    \
            }                                                                                                   \
            if (redo && live_q) {                                                                               \
                const ShiftedMoments<R> m = list_moments_shifted<R>(X, ldx, r0, n_rows, base + q, ldi, k);      \
                _Pragma("unroll") for (int rr = 0; rr < R; ++rr) {                                              \
                    lp[rr] = m.piv[rr];                                                                         \
                    s1[rr] = m.s1[rr];                                                                          \
                    s2[rr] = m.s2[rr];                                                                          \
                }                                                                                               \
                atomicAdd(&g_gather_redo, 1ull);                                                                \
            }                                                                                                   \
            _Pragma("unroll") for (int rr = 0; rr < R; ++rr) {                                                  \
                const float md = s1[rr] * inv_k;                                                                \
                float var = fmaf(-md, md, s2[rr] * inv_k);                                                      \
                var = var > 0.f ? var : (var == var ? 0.f : var); /* NaN stays NaN */                           \
                /* 1 / max(sqrt(var), floor): v_rsq_f32 is good to one unit in the last place */                \
                const float is = var < floor2 ? inv_floor : __builtin_amdgcn_rsqf(var);                         \
                if (live_q && r0 + rr < n_rows) {                                                               \
                    /* one explicit fma: left as `lp + md` the compiler contracts it in some variants of the    \
                       kernel and not in others (__fadd_rn is a plain `+` here), and they are compared bit for bit */ \
                    const float m = fmaf(s1[rr], inv_k, lp[rr]);                                                \
                    if (PAIRS) {                                                                                \
                        ap.out[(r0 + rr) * ap.ldo + q] = asnorm_value(ps[rr], m, is, pmi[2 * rr], pmi[2 * rr + 1]); \
                    } else if (APPLY) {                                                                         \
                        ap.out[(r0 + rr) * ap.ldo + q] =                                                        \
                            asnorm_value(ps[rr], m, is, lds_vec_elem<R>(pm, rr), lds_vec_elem<R>(pi, rr));      \
                    } else if (istd == nullptr) {                                                               \
                        reinterpret_cast<float2*>(mean)[(r0 + rr) * ldo + q] = make_float2(m, is);              \
                    } else {                                                                                    \
                        mean[(r0 + rr) * ldo + q] = m;                                                          \
                        istd[(r0 + rr) * ldo + q] = is;                                                         \
                    }                                                                                           \
                }                                                                                               \
                s1[rr] = s2[rr] = 0.f;                                                                          \
            }                                                                                                   \
            g = 0;                                                                                              \
            ++l;                                                                                                \
            u_cp = dynamic ? u_after : u_cp + WAVES;                                                            \
            if (u_cp >= U) break; /* nothing left for this wave */                                              \
            rotate_priority((wave >> 2) + l);                                                                   \
            HYPMI_APPLY_PREFETCH(u_cp)                                                                          \
        }                                                                                                       \
    }
    static_assert(GP_RING == 4, "the walk below names the four ring slots");
    rotate_priority(wave >> 2);
    for (;;) {
        HYPMI_GROUP(0)
        HYPMI_GROUP(1)
        HYPMI_GROUP(2)
        HYPMI_GROUP(3)
    }
    if (tid == 0) {
        atomicAdd(&g_gather_clock[0], __builtin_readcyclecounter() - cycles_in);
        atomicAdd(&g_gather_clock[1], __builtin_amdgcn_s_memrealtime() - ticks_in);
    }
}

#undef HYPMI_GROUP
#undef HYPMI_NEXT_LOAD
#undef HYPMI_APPLY_PREFETCH

// ---- segmented gather: cohorts whose rows do not fit LDS two at a time --------------------------------
// The cohort is cut into S segments of seg_len members (a multiple of 32, so that a member keeps its LDS
// bank class) that do fit with R = 2, the lists are partitioned by segment, and a workgroup walks its
// lists once per segment with the moments of SG_NL lists per lane carried in registers.  Cost against an
// undivided cohort: the lanes of a wave wait for the longest of their 64 list parts (about 1.15 x with two
// segments, 1.2 x with three), and the rows are fetched once per SG_NL rounds of lists instead of once
// per 64 or 128.
//
// Segmented lists (segment_lists_kernel): list q keeps its layout (groups of eight 16-bit entries, entry
// j at seg8[((j >> 3) ldi + q) 8 + (j & 7)]), its members ordered by segment (stable: the bank order
// inside a segment is the one the selection kernel produced) and stored as 1 + (member - segment start);
// every segment's part is padded to whole groups with entries 0, which is the LDS entry that holds zeros.
// gstart[s ldi + q] (s = 0 .. S) is the first group of segment s of list q.
constexpr int SG_NL = 12;        // lists per lane of one workgroup (their moments: 4 R registers each)
constexpr int SG_MAX_SEG = 8;

__global__ __launch_bounds__(256) void segment_lists_kernel(const uint16_t* __restrict__ idx8, int64_t ldi, int k,
                                                            int64_t Q, int S, int seg_len,
                                                            uint16_t* __restrict__ seg8,
                                                            uint16_t* __restrict__ gstart) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;  // wave-uniform
    auto entry = [&](int j) -> unsigned { return idx8[(((int64_t)(j >> 3)) * ldi + q) * 8 + (j & 7)]; };
    auto seg_of = [&](unsigned c) -> int {
        int s = 0;
        for (int b = 1; b < S; ++b) s += c >= (unsigned)(b * seg_len) ? 1 : 0;
        return s;
    };
    int cnt[SG_MAX_SEG];
#pragma unroll
    for (int s = 0; s < SG_MAX_SEG; ++s) cnt[s] = 0;
    for (int j0 = 0; j0 < k; j0 += 64) {
        const int j = j0 + lane;
        const int sg = j < k ? seg_of(entry(j)) : -1;
#pragma unroll
        for (int s = 0; s < SG_MAX_SEG; ++s) cnt[s] += __popcll(__ballot(sg == s));
    }
    int pos[SG_MAX_SEG];  // next entry position of every segment
    int g = 0;
#pragma unroll
    for (int s = 0; s < SG_MAX_SEG; ++s) {
        if (s <= S && lane == 0) gstart[(int64_t)s * ldi + q] = (uint16_t)g;
        pos[s] = 8 * g;
        if (s < S) {
            // the padding of this segment's last group
            const int p = 8 * g + cnt[s] + lane;
            g += (cnt[s] + 7) >> 3;
            if (lane < 8 && p < 8 * g) seg8[(((int64_t)(p >> 3)) * ldi + q) * 8 + (p & 7)] = 0;
        }
    }
    if (S == SG_MAX_SEG && lane == 0) gstart[(int64_t)S * ldi + q] = (uint16_t)g;
    const uint64_t below = (1ull << lane) - 1ull;
    for (int j0 = 0; j0 < k; j0 += 64) {
        const int j = j0 + lane;
        const unsigned c = j < k ? entry(j) : 0u;
        const int sg = j < k ? seg_of(c) : -1;
#pragma unroll
        for (int s = 0; s < SG_MAX_SEG; ++s) {
            const uint64_t m = __ballot(sg == s);
            if (sg == s) {
                const int p = pos[s] + __popcll(m & below);
                seg8[(((int64_t)(p >> 3)) * ldi + q) * 8 + (p & 7)] = (uint16_t)(c - (unsigned)(s * seg_len) + 1u);
            }
            pos[s] += __popcll(m);
        }
    }
}

// MODE as in gather_pivot_kernel (0: statistics stored, as pairs when istd == nullptr; 1 / 2: normalised
// scores stored, the other side's statistics in two arrays / as pairs).  R = 2.
template <int MODE>
__global__ __launch_bounds__(GS_THREADS) void gather_seg_kernel(
    const float* __restrict__ X, int64_t ldx, int64_t n_rows, int nc, const uint16_t* __restrict__ seg8,
    const uint16_t* __restrict__ gstart, int64_t ldi, unsigned seg_bytes, const uint16_t* __restrict__ idx8, int k,
    int64_t Q, int S, int seg_len, float floor_, float* __restrict__ mean, float* __restrict__ istd, int64_t ldo,
    ApplyArgs ap) {
    constexpr int R = 2;
    constexpr bool APPLY = MODE != 0, PAIRS = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) float xr[];  // entry 0: zeros; entry 1 + c: member c of the segment
    float* red = xr + (size_t)(seg_len + 1) * R;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t grp = blockIdx.x;
    if (APPLY) {  // the grid is padded to whole sets of 8 x GP_XCD_GROUPS row groups
        const int64_t slot = grp >> 3, xcd = grp & 7;
        grp = ((slot / GP_XCD_GROUPS) * 8 + xcd) * GP_XCD_GROUPS + slot % GP_XCD_GROUPS;
        if (grp * R >= n_rows) return;
    }
    const int64_t r0 = grp * R;
    const int64_t q0 = (int64_t)blockIdx.y * (SG_NL * GS_THREADS);
    const int64_t q1 = (q0 + SG_NL * GS_THREADS < Q) ? q0 + SG_NL * GS_THREADS : Q;
    if (q0 >= q1) return;
    const int nl = (int)((q1 - q0 + GS_THREADS - 1) / GS_THREADS);  // rounds of lists in this block
    // pivot of a row: mean of its first min(nc, 1024) entries, as in gather_pivot_kernel
    float rowpiv[R];
    const bool round_lds = k >= GP_ROUND_MIN_K;
    {
        const int ns = nc < GS_THREADS ? nc : GS_THREADS;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            float v = (r0 + rr < n_rows && tid < ns) ? X[(r0 + rr) * ldx + tid] : 0.f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0) red[rr * 16 + wave] = v;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < GS_THREADS / 64; ++w) sum += red[rr * 16 + w];
            rowpiv[rr] = sum / (float)ns;
        }
    }
    const unsigned xbase = __builtin_amdgcn_readfirstlane(lds_address(xr));
    const bool vec_ok = ldx % 4 == 0 && reinterpret_cast<uintptr_t>(X) % 16 == 0;  // uniform (seg_len % 32 == 0)
    // moments of the lane's lists: [0] is the list being walked, the arrays rotate at the end of every list
    float s1[SG_NL][R], s2[SG_NL][R];
#pragma unroll
    for (int l = 0; l < SG_NL; ++l)
#pragma unroll
        for (int rr = 0; rr < R; ++rr) s1[l][rr] = s2[l][rr] = 0.f;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const unsigned step = (unsigned)(ldi * 16);  // bytes from a group of a list to its next (checked at launch)
    for (int s = 0; s < S; ++s) {
        __syncthreads();  // the previous segment (or the pivot scratch) is no longer read
        const int base = s * seg_len;
        const int len = nc - base < seg_len ? nc - base : seg_len;
        if (vec_ok) {
            typedef typename LdsVec<4>::type f4;
            const int len4 = len >> 2;
            for (int c4 = tid; c4 < len4; c4 += GS_THREADS) {
                f4 v[R];
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    v[rr] = f4{0.f, 0.f, 0.f, 0.f};
                    if (r0 + rr < n_rows)
                        v[rr] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(X + (r0 + rr) * ldx + base) + c4);
                }
                float2* dst = reinterpret_cast<float2*>(xr) + 1 + 4 * c4;
                dst[0] = make_float2(lds_rounded(v[0].x - rowpiv[0], round_lds), lds_rounded(v[1].x - rowpiv[1], round_lds));
                dst[1] = make_float2(lds_rounded(v[0].y - rowpiv[0], round_lds), lds_rounded(v[1].y - rowpiv[1], round_lds));
                dst[2] = make_float2(lds_rounded(v[0].z - rowpiv[0], round_lds), lds_rounded(v[1].z - rowpiv[1], round_lds));
                dst[3] = make_float2(lds_rounded(v[0].w - rowpiv[0], round_lds), lds_rounded(v[1].w - rowpiv[1], round_lds));
            }
            for (int c = 4 * len4 + tid; c < len; c += GS_THREADS)
#pragma unroll
                for (int rr = 0; rr < R; ++rr)
                    xr[(size_t)(c + 1) * R + rr] =
                        r0 + rr < n_rows ? lds_rounded(X[(r0 + rr) * ldx + base + c] - rowpiv[rr], round_lds) : 0.f;
        } else {
            for (int c = tid; c < len; c += GS_THREADS)
#pragma unroll
                for (int rr = 0; rr < R; ++rr)
                    xr[(size_t)(c + 1) * R + rr] =
                        r0 + rr < n_rows ? lds_rounded(X[(r0 + rr) * ldx + base + c] - rowpiv[rr], round_lds) : 0.f;
        }
        if (tid < R) xr[tid] = 0.f;
        __syncthreads();
        // group range of the lane's first list in this segment; those of the next list are fetched a list ahead
        auto range_of = [&](int l, unsigned& first, int& count) {
            const int64_t q = q0 + (int64_t)l * GS_THREADS + tid;
            first = 0;
            count = 0;
            if (l < nl && q < q1) {
                first = gstart[(int64_t)s * ldi + q];
                count = (int)gstart[(int64_t)(s + 1) * ldi + q] - (int)first;
            }
        };
        unsigned g_first, g_first_next;
        int n_grp, n_grp_next;
        range_of(0, g_first, n_grp);
        for (int l = 0; l < SG_NL; ++l) {
            range_of(l + 1, g_first_next, n_grp_next);
            int nmax = n_grp;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const int o = __shfl_xor(nmax, off, 64);
                nmax = o > nmax ? o : nmax;
            }
            nmax = __builtin_amdgcn_readfirstlane(nmax);
            // byte offset of the lane's first group of this part (the whole list array is one buffer)
            const unsigned voff0 = (unsigned)((((int64_t)g_first * ldi) + q0 + (int64_t)l * GS_THREADS + tid) * 16);
            uint4 ring[GP_RING];
            // group i of the part: lanes whose part is shorter read past the end of the buffer and get zeros,
            // i.e. eight times the entry of zeros
#define HYPMI_SEG_LOAD(dst, I)                                                                            \
    {                                                                                                     \
        const unsigned vo_ = (I) < n_grp ? voff0 + (unsigned)(I) * step : 0xfffffff0u;                   \
        const v4u w_ = __builtin_amdgcn_raw_buffer_load_b128(                                             \
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(seg8), 0, seg_bytes, 0x00020000), vo_, 0, 0); \
        dst = make_uint4(w_[0], w_[1], w_[2], w_[3]);                                                     \
    }
#define HYPMI_SEG_GROUP(D, RELOAD, I)                                                                     \
    {                                                                                                     \
        unsigned a[8];                                                                                    \
        a[0] = entry_addr<R, 0>(ring[D].x, xbase); a[1] = entry_addr<R, 1>(ring[D].x, xbase);             \
        a[2] = entry_addr<R, 0>(ring[D].y, xbase); a[3] = entry_addr<R, 1>(ring[D].y, xbase);             \
        a[4] = entry_addr<R, 0>(ring[D].z, xbase); a[5] = entry_addr<R, 1>(ring[D].z, xbase);             \
        a[6] = entry_addr<R, 0>(ring[D].w, xbase); a[7] = entry_addr<R, 1>(ring[D].w, xbase);             \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        if (RELOAD) HYPMI_SEG_LOAD(ring[D], I)                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        typename LdsVec<R>::type val[8];                                                                  \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) val[e] = lds_vec<R>(a[e]);                          \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                     \
            _Pragma("unroll") for (int rr = 0; rr < R; ++rr)                                              \
                moment_step(s1[0][rr], s2[0][rr], lds_vec_elem<R>(val[e], rr));                           \
    }
            if (nmax > 0) {
#pragma unroll
                for (int d = 0; d < GP_RING; ++d) HYPMI_SEG_LOAD(ring[d], d)
                int i = 0;
                for (; i + GP_RING <= nmax; i += GP_RING) {
                    HYPMI_SEG_GROUP(0, true, i + 4)
                    HYPMI_SEG_GROUP(1, true, i + 5)
                    HYPMI_SEG_GROUP(2, true, i + 6)
                    HYPMI_SEG_GROUP(3, true, i + 7)
                }
                if (i < nmax) HYPMI_SEG_GROUP(0, false, 0)
                if (i + 1 < nmax) HYPMI_SEG_GROUP(1, false, 0)
                if (i + 2 < nmax) HYPMI_SEG_GROUP(2, false, 0)
            }
#undef HYPMI_SEG_GROUP
#undef HYPMI_SEG_LOAD
            rotate_priority((wave >> 2) + l + 1);
            // the next list's moments move to the front
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const float t1 = s1[0][rr], t2 = s2[0][rr];
#pragma unroll
                for (int m = 0; m + 1 < SG_NL; ++m) {
                    s1[m][rr] = s1[m + 1][rr];
                    s2[m][rr] = s2[m + 1][rr];
                }
                s1[SG_NL - 1][rr] = t1;
                s2[SG_NL - 1][rr] = t2;
            }
            g_first = g_first_next;
            n_grp = n_grp_next;
        }
    }
    // after S x SG_NL rotations the moments of list l are back in place l
    const float inv_k = 1.0f / (float)k, floor2 = floor_ * floor_, inv_floor = 1.0f / floor_;
    const uint4* base = reinterpret_cast<const uint4*>(idx8);
    // The lists are finished in batches: what a batch needs from memory (APPLY: raw scores and the other
    // side's statistics) is requested for all its lists before the first is used -- one round trip per
    // batch, not one per list.
    constexpr int LB = SG_NL % 4 == 0 ? 4 : (SG_NL % 3 == 0 ? 3 : 1);
#pragma unroll
    for (int l0 = 0; l0 < SG_NL; l0 += LB) {
        typename LdsVec<2 * R>::type pmi[LB];
        typename LdsVec<R>::type pm[LB], pi[LB];
        float sc[LB][R];
        if (APPLY) {
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                int64_t q = q0 + (int64_t)(l0 + u) * GS_THREADS + tid;
                q = q < q1 ? q : q1 - 1;
                if (PAIRS) {
                    pmi[u] = load_pairs<R>(ap.mt + (q * ap.ldt + r0) * 2);
                } else {
                    pm[u] = *reinterpret_cast<const typename LdsVec<R>::type*>(ap.mt + q * ap.ldt + r0);
                    pi[u] = *reinterpret_cast<const typename LdsVec<R>::type*>(ap.it + q * ap.ldt + r0);
                }
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    const int64_t e = r0 + rr < n_rows ? r0 + rr : n_rows - 1;
                    sc[u][rr] = __builtin_nontemporal_load(ap.S + e * ap.lds + q);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int l = l0 + u;
            const int64_t q = q0 + (int64_t)l * GS_THREADS + tid;
            if (l >= nl || q >= q1) continue;
            float lp[R], m1[R], m2[R];
            bool redo = false;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                lp[rr] = rowpiv[rr];
                m1[rr] = s1[l][rr];
                m2[rr] = s2[l][rr];
                const float md = m1[rr] * inv_k, ms = m2[rr] * inv_k;
                redo |= fmaf(-md, md, ms) * GP_AMP < ms;  // false for NaN: those stay NaN
            }
            if (redo) {  // a selection far from the row pivot: again around the list's own first member
                const ShiftedMoments<R> m = list_moments_shifted<R>(X, ldx, r0, n_rows, base + q, ldi, k);
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    lp[rr] = m.piv[rr];
                    m1[rr] = m.s1[rr];
                    m2[rr] = m.s2[rr];
                }
                atomicAdd(&g_gather_redo, 1ull);
            }
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                if (r0 + rr >= n_rows) continue;
                const float md = m1[rr] * inv_k;
                float var = fmaf(-md, md, m2[rr] * inv_k);
                var = var > 0.f ? var : (var == var ? 0.f : var);  // NaN stays NaN
                const float is = var < floor2 ? inv_floor : __builtin_amdgcn_rsqf(var);
                const float m = fmaf(m1[rr], inv_k, lp[rr]);
                if (PAIRS) {
                    ap.out[(r0 + rr) * ap.ldo + q] = asnorm_value(sc[u][rr], m, is, pmi[u][2 * rr], pmi[u][2 * rr + 1]);
                } else if (APPLY) {
                    ap.out[(r0 + rr) * ap.ldo + q] = asnorm_value(sc[u][rr], m, is, lds_vec_elem<R>(pm[u], rr),
                                                                  lds_vec_elem<R>(pi[u], rr));
                } else if (istd == nullptr) {
                    reinterpret_cast<float2*>(mean)[(r0 + rr) * ldo + q] = make_float2(m, is);
                } else {
                    mean[(r0 + rr) * ldo + q] = m;
                    istd[(r0 + rr) * ldo + q] = is;
                }
            }
        }
    }
}

// out[e, t] = ((S - mz[e,t]) * iz[e,t] + (S - mt[t,e]) * it[t,e]) / sqrt(2); the T-part statistics
// arrive transposed and go through a 32 x 33 LDS tile.
__global__ __launch_bounds__(256) void asnorm_apply_kernel(const float* __restrict__ S, int64_t lds_,
                                                           int64_t ne, int64_t nt,
                                                           const float* __restrict__ mz,
                                                           const float* __restrict__ iz, int64_t ldz,
                                                           const float* __restrict__ mt,
                                                           const float* __restrict__ it, int64_t ldt,
                                                           float* __restrict__ out, int64_t ldo) {
    __shared__ float tm[32][33], ti[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int64_t e0 = (int64_t)blockIdx.y * 32, t0 = (int64_t)blockIdx.x * 32;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int64_t t = t0 + ty + 8 * p, e = e0 + tx;
        const bool ok = t < nt && e < ne;
        tm[ty + 8 * p][tx] = ok ? mt[t * ldt + e] : 0.f;
        ti[ty + 8 * p][tx] = ok ? it[t * ldt + e] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int64_t e = e0 + ty + 8 * p, t = t0 + tx;
        if (e < ne && t < nt) {
            out[e * ldo + t] = asnorm_value(S[e * lds_ + t], mz[e * ldz + t], iz[e * ldz + t], tm[tx][ty + 8 * p],
                                            ti[tx][ty + 8 * p]);
        }
    }
}

// 32 x 32 tiled transpose: out[c, r] = in[r, c]
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int64_t ldi,
                                                        int64_t rows, int64_t cols,
                                                        float* __restrict__ out, int64_t ldo) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int64_t r = r0 + ty + 8 * p, c = c0 + tx;
        tile[ty + 8 * p][tx] = (r < rows && c < cols) ? in[r * ldi + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int64_t c = c0 + ty + 8 * p, r = r0 + tx;
        if (c < cols && r < rows) out[c * ldo + r] = tile[tx][ty + 8 * p];
    }
}

// One wavefront per row: mean and 1 / max(std, floor) over a whole row (non-adaptive Z/T-norm).
// With a mask, masked entries count as zero and the moments are divided by the unmasked fraction.
__global__ __launch_bounds__(256) void row_meanstd_kernel(const float* __restrict__ X, int64_t ldx,
                                                          int64_t n_rows, int nc,
                                                          const unsigned char* __restrict__ mask,
                                                          int64_t ldm, float floor_,
                                                          float* __restrict__ mean,
                                                          float* __restrict__ istd) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const float* src = X + r * ldx;
    double s1 = 0.0, s2 = 0.0, cnt = 0.0;
    for (int i = lane; i < nc; i += 64) {
        const bool on = mask ? mask[r * ldm + i] != 0 : true;
        const double v = on ? (double)src[i] : 0.0;
        s1 += v;
        s2 += v * v;
        cnt += on ? 1.0 : 0.0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off, 64);
        s2 += __shfl_xor(s2, off, 64);
        cnt += __shfl_xor(cnt, off, 64);
    }
    if (lane == 0) {
        double m, var;
        if (mask) {
            m = s1 / cnt;
            var = s2 / cnt - m * m;
        } else {
            // two-pass would need the row twice; the shifted one-pass form in double is exact enough
            m = s1 / (double)nc;
            var = s2 / (double)nc - m * m;
        }
        double sd = sqrt(var > 0.0 ? var : (var == var ? 0.0 : var));
        if (sd < (double)floor_) sd = (double)floor_;
        mean[r] = (float)m;
        istd[r] = (float)(1.0 / sd);
    }
}

// out = sum over enabled parts of (S - m) * i, divided by sqrt(2) when both are enabled
__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ S, int64_t lds_,
                                                         int64_t ne, int64_t nt,
                                                         const float* __restrict__ mz,
                                                         const float* __restrict__ iz,
                                                         const float* __restrict__ mt,
                                                         const float* __restrict__ it,
                                                         float* __restrict__ out, int64_t ldo) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t e = blockIdx.y;
    if (t >= nt || e >= ne) return;
    const float s = S[e * lds_ + t];
    float acc = 0.f;
    if (mz) acc += (s - mz[e]) * iz[e];
    if (mt) acc += (s - mt[t]) * it[t];
    if (mz && mt) acc *= 0.70710678118654752440f;
    out[e * ldo + t] = acc;
}

// X[i, j] = 0 where mask[i, j] == 0
__global__ __launch_bounds__(256) void mask_zero_kernel(float* __restrict__ X, int64_t ldx, int64_t rows,
                                                        int64_t cols,
                                                        const unsigned char* __restrict__ mask,
                                                        int64_t ldm) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (i < rows && j < cols && mask[i * ldm + j] == 0) X[i * ldx + j] = 0.f;
}

}  // namespace

extern "C" int64_t hypmi_topk_workspace_bytes(int64_t n_rows, int nc) {
    (void)n_rows;
    (void)nc;
    return 0;  // every cohort the 16-bit lists can name is selected from registers
}

extern "C" int64_t hypmi_topk_list_bytes(int k, int64_t ldi) {
    return (int64_t)((k + 7) / 8) * ldi * 8 * (int64_t)sizeof(uint16_t);
}

extern "C" int hypmi_topk_rows(const float* A, int64_t n_rows, int nc, int64_t lda, int k, int discard,
                               uint16_t* idx_t, int64_t ldi, const hypmi_topk_exact* exact, void* workspace,
                               int64_t workspace_bytes, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HYPMI_REQUIRE(A && idx_t, "topk_rows: null pointer");
    HYPMI_REQUIRE(n_rows >= 0 && nc > 0 && lda >= nc && ldi >= n_rows, "topk_rows: bad shape");
    HYPMI_REQUIRE(k > 0 && discard >= 0 && discard + k <= nc,
                  "topk_rows: need 0 < k and discard + k <= cohort size (k=%d discard=%d nc=%d)", k,
                  discard, nc);
    HYPMI_REQUIRE(n_rows <= 0x7fffffffLL, "topk_rows: too many rows");
    HYPMI_REQUIRE(nc <= 65536, "topk_rows: cohort of %d exceeds the 16-bit index lists", nc);
    HYPMI_REQUIRE(aligned16(idx_t), "topk_rows: index lists must be 16-byte aligned");
    ExactSrc ex = {};
    if (exact && exact->mode != 0) {
        HYPMI_REQUIRE(exact->mode >= 1 && exact->mode <= 3, "topk_rows: unknown exact mode %d", exact->mode);
        if (exact->mode == 1) {
            HYPMI_REQUIRE(exact->A64 && exact->lda64 >= nc, "topk_rows: exact matrix missing or too narrow");
        } else if (exact->mode == 2) {
            HYPMI_REQUIRE(exact->G_rows && exact->G_coh && exact->bnd_rows && exact->K > 0 &&
                              exact->ldg >= exact->K + 2,
                          "topk_rows: exact packed rows missing");
        } else {
            HYPMI_REQUIRE(exact->bnd_rows && exact->todo && exact->todo_count,
                          "topk_rows: mode 3 needs the rows' bounds and the list of rows to fill");
        
        }
        HYPMI_REQUIRE(!exact->row_map || exact->mode == 2, "topk_rows: a row map goes with mode 2");
        ex.mode = exact->mode;
        ex.A64 = exact->A64;
        ex.lda64 = exact->lda64;
        ex.G_rows = exact->G_rows;
        ex.G_coh = exact->G_coh;
        ex.ldg = exact->ldg;
        ex.K = exact->K;
        ex.bnd_rows = exact->bnd_rows;
        ex.coh_gmax = exact->coh_gmax;
        ex.coh_qmax = exact->coh_qmax;
        ex.err_coeff = exact->err_coeff;
        ex.counters = exact->counters;
        ex.row_map = exact->row_map;
        ex.todo = exact->todo;
        ex.todo_count = exact->todo_count;
    }
    if (n_rows == 0) return HYPMI_OK;
    // Threads per row: 256 while a thread's keys leave registers for three or four waves per SIMD, then 512
    // (up to 80 keys each still run four waves per SIMD; 65 536 rows of 40 960 members: 7.6 ms
    // with 160 keys in each of 256 threads, one wave per SIMD; 3.6 ms with 80 in each of 512).  More threads
    // at the same keys per thread do not pay: 20 000 members take 1.33 ms with 256 threads, 1.81 with 512.
    int rc = HYPMI_OK;
    if (nc <= 96 * 256) {
        const int items = (nc + TR_THREADS - 1) / TR_THREADS;
#define HYPMI_TOPK_CASE(N) \
    if (items <= N) { rc = launch_topk_reg<N>(A, n_rows, nc, lda, k, discard, idx_t, ldi, ex, stream); } else
        HYPMI_TOPK_CASE(4) HYPMI_TOPK_CASE(8) HYPMI_TOPK_CASE(16) HYPMI_TOPK_CASE(32) HYPMI_TOPK_CASE(48)
        HYPMI_TOPK_CASE(64) HYPMI_TOPK_CASE(80)
        { rc = launch_topk_reg<96>(A, n_rows, nc, lda, k, discard, idx_t, ldi, ex, stream); }
#undef HYPMI_TOPK_CASE
    } else if (nc <= 64 * 512) {
        rc = launch_topk_reg<64, 512>(A, n_rows, nc, lda, k, discard, idx_t, ldi, ex, stream);
    } else if (nc <= 80 * 512) {
        rc = launch_topk_reg<80, 512>(A, n_rows, nc, lda, k, discard, idx_t, ldi, ex, stream);
    } else if (nc <= 96 * 512) {  // (1024 threads with 48 or 64 keys each are no faster than these two)
        rc = launch_topk_reg<96, 512>(A, n_rows, nc, lda, k, discard, idx_t, ldi, ex, stream);
    } else if (nc <= TR_MAX_NC) {
        rc = launch_topk_reg<128, 512>(A, n_rows, nc, lda, k, discard, idx_t, ldi, ex, stream);
    }
    if (rc != HYPMI_OK) return rc;
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

namespace {
// One workgroup walks up to max_rounds rounds of lists for its rows: the index lists of a block (600 bytes
// each at 300 members) are read by every row group, and when they are too many they no longer stay in the
// caches the row groups share (262 144 lists in one block: the launch took 1.45 times as long per list).
// 128 rounds (79 MB of lists) run at the rate of 64 in the plain kernel and pay the row fill half as often;
// the kernel that also normalises, whose loads of scores and statistics share those caches, is 1 %
// faster with 64 (both builds alternating on one GPU).  With
// too few row groups to fill the chip the lists are split further over blockIdx.y (every block of lists
// reloads the rows, so blocks are kept as large as that allows).
inline void gather_grid(int64_t groups, int64_t n_lists, int64_t* chunks_out, int64_t* q_per_block_out,
                        int max_rounds = 128) {
    int64_t chunks = (n_lists + (int64_t)max_rounds * GS_THREADS - 1) / ((int64_t)max_rounds * GS_THREADS);
    if (chunks < 1) chunks = 1;
    while (groups * chunks < 1024 && (n_lists + chunks - 1) / chunks > 4 * GS_THREADS) chunks *= 2;
    if (chunks > 65535) chunks = 65535;
    const int64_t q_per_block = round_up((n_lists + chunks - 1) / chunks, GS_THREADS);
    *chunks_out = (n_lists + q_per_block - 1) / q_per_block;
    *q_per_block_out = q_per_block;
}

template <int R, bool MASK, bool IN_LDS = true>
int launch_gather(const float* X, int64_t n_rows, int nc, int64_t ldx, const unsigned char* mask,
                  int64_t ldm, const uint16_t* idx8, int64_t ldi, int k, int64_t n_lists, float std_floor,
                  float* mean, float* istd, int64_t ldo, hipStream_t stream) {
    const int64_t lds_need =
        IN_LDS ? round_up((int64_t)nc * R * 4 + (MASK ? (int64_t)nc * R : 0), 16) : 16;
    if (lds_need > 48 * 1024)
        HYPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_stats_kernel<R, MASK, IN_LDS>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_need));
    const int64_t groups = (n_rows + R - 1) / R;
    HYPMI_REQUIRE(groups <= 0x7fffffffLL, "gather_stats: too many rows");
    int64_t chunks, q_per_block;
    gather_grid(groups, n_lists, &chunks, &q_per_block);
    hipLaunchKernelGGL((gather_stats_kernel<R, MASK, IN_LDS>), dim3((unsigned)groups, (unsigned)chunks),
                       dim3(GS_THREADS), lds_need, stream, X, ldx, n_rows, mask, ldm, nc, idx8, ldi, k,
                       n_lists, q_per_block, std_floor, mean, istd, ldo);
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

inline int64_t pivot_lds_bytes(int nc, int R) {
    return round_up(((int64_t)nc * R + GP_SCRATCH_FLOATS) * 4, 16);
}

}  // namespace
// The `front` counters of the gather launches (see gather_pivot_kernel): a pool of sets, one set per launch
// in turn (launches that overlap on different streams must not share a set; FRONT_SETS launches apart they may).
constexpr int FRONT_SETS = 32, FRONT_BLOCKS = 16;
static __device__ unsigned g_gather_front[FRONT_SETS * FRONT_BLOCKS * 8];
namespace {
std::mutex front_mutex;
unsigned* front_base[64];  // per device
unsigned front_seq = 0;

// zeroed counters for a launch with `blocks` list blocks on the current device (nullptr: none, the launch
// has more list blocks than a set covers -- its workgroups all start at the first list)
int next_front_set(int64_t blocks, hipStream_t stream, unsigned** set) {
    *set = nullptr;
    if (blocks > FRONT_BLOCKS) return HYPMI_OK;
    int dev = -1;
    HYPMI_HIP(hipGetDevice(&dev));
    HYPMI_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    unsigned* at;
    {
        std::lock_guard<std::mutex> lock(front_mutex);
        if (!front_base[dev]) {
            void* ptr = nullptr;
            HYPMI_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_gather_front)));
            HYPMI_REQUIRE(ptr != nullptr, "no device address for the gather front counters");
            front_base[dev] = static_cast<unsigned*>(ptr);
        }
        at = front_base[dev] + (size_t)(front_seq++ % FRONT_SETS) * FRONT_BLOCKS * 8;
    }
    HYPMI_HIP(hipMemsetAsync(at, 0, (size_t)blocks * 8 * sizeof(unsigned), stream));
    *set = at;
    return HYPMI_OK;
}

template <int R, int MODE = 0>
int launch_gather_pivot(const float* X, int64_t n_rows, int nc, int64_t ldx, const uint16_t* idx8, int64_t ldi,
                        int k, int64_t n_lists, float std_floor, float* mean, float* istd, int64_t ldo,
                        hipStream_t stream, const ApplyArgs& ap = ApplyArgs{}) {
    const int64_t lds_need = pivot_lds_bytes(nc, R);
    if (lds_need > 48 * 1024)
        HYPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_pivot_kernel<R, MODE>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_need));
    const int64_t groups = (n_rows + R - 1) / R;
    int64_t chunks, q_per_block;
    gather_grid(groups, n_lists, &chunks, &q_per_block, MODE != 0 ? 64 : 128);
    // with APPLY the row groups are renumbered in sets of 8 x GP_XCD_GROUPS: pad the grid to whole sets
    const int64_t gx = MODE != 0 ? round_up(groups, 8 * GP_XCD_GROUPS) : groups;
    HYPMI_REQUIRE(gx <= 0x7fffffffLL, "gather_stats: too many rows");
    HYPMI_REQUIRE((q_per_block / GS_THREADS) * ((k + 7) / 8) < 0x7fffffffLL, "gather_stats: lists too long");
    unsigned* front = nullptr;
    if (q_per_block >= 4 * GS_THREADS) {  // (short blocks stay in L2 whatever the order)
        const int rc = next_front_set(chunks, stream, &front);
        if (rc != HYPMI_OK) return rc;
    }
    hipLaunchKernelGGL((gather_pivot_kernel<R, MODE>), dim3((unsigned)gx, (unsigned)chunks), dim3(GS_THREADS),
                       lds_need, stream, X, ldx, n_rows, nc, idx8, ldi, k, n_lists, q_per_block, std_floor, mean,
                       istd, ldo, ap, front);
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

// rows per workgroup of the LDS-resident unmasked gather for this shape; 0: the rows do not fit
inline int pivot_rows_per_group(int nc, int64_t n_rows) {
    if ((int64_t)nc * 4 > MAX_LDS_BYTES) return 0;
    int R = 1;
    if (pivot_lds_bytes(nc, 4) <= MAX_LDS_BYTES && n_rows >= 4) R = 4;
    else if (pivot_lds_bytes(nc, 2) <= MAX_LDS_BYTES && n_rows >= 2) R = 2;
    return pivot_lds_bytes(nc, R) <= MAX_LDS_BYTES ? R : 0;
}
}  // namespace

extern "C" int hypmi_gather_stats(const float* X, int64_t n_rows, int nc, int64_t ldx,
                                  const unsigned char* mask, int64_t ldm, const uint16_t* idx_t,
                                  int64_t ldi, int k, int64_t n_lists, float std_floor, float* mean,
                                  float* istd, int64_t ldo, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HYPMI_REQUIRE(X && idx_t && mean && istd, "gather_stats: null pointer");
    HYPMI_REQUIRE(n_rows >= 0 && nc > 0 && nc <= 65536 && ldx >= nc && k > 0 && n_lists >= 0 &&
                      ldi >= n_lists && ldo >= n_lists,
                  "gather_stats: bad shape");
    HYPMI_REQUIRE(aligned16(idx_t), "gather_stats: index lists must be 16-byte aligned");
    if (n_rows == 0 || n_lists == 0) return HYPMI_OK;
    const int64_t per_row = (int64_t)nc * 4 + (mask ? nc : 0);
    if (per_row > MAX_LDS_BYTES)  // one row does not fit in LDS: gather it from global memory
        return mask ? launch_gather<1, true, false>(X, n_rows, nc, ldx, mask, ldm, idx_t, ldi, k, n_lists,
                                                    std_floor, mean, istd, ldo, stream)
                    : launch_gather<1, false, false>(X, n_rows, nc, ldx, mask, ldm, idx_t, ldi, k, n_lists,
                                                     std_floor, mean, istd, ldo, stream);
    // as many rows per workgroup as LDS holds (each index fetched is reused for all of them)
    if (!mask) {
        const int R = pivot_rows_per_group(nc, n_rows);
        if (R == 4)
            return launch_gather_pivot<4>(X, n_rows, nc, ldx, idx_t, ldi, k, n_lists, std_floor, mean, istd, ldo,
                                          stream);
        if (R == 2)
            return launch_gather_pivot<2>(X, n_rows, nc, ldx, idx_t, ldi, k, n_lists, std_floor, mean, istd, ldo,
                                          stream);
        if (R == 1)
            return launch_gather_pivot<1>(X, n_rows, nc, ldx, idx_t, ldi, k, n_lists, std_floor, mean, istd, ldo,
                                          stream);
        return launch_gather<1, false>(X, n_rows, nc, ldx, mask, ldm, idx_t, ldi, k, n_lists, std_floor, mean,
                                       istd, ldo, stream);
    }
    int R = 1;
    if (4 * per_row <= MAX_LDS_BYTES && n_rows >= 4) R = 4;
    else if (2 * per_row <= MAX_LDS_BYTES && n_rows >= 2) R = 2;
    if (R == 4)
        return launch_gather<4, true>(X, n_rows, nc, ldx, mask, ldm, idx_t, ldi, k, n_lists, std_floor, mean, istd,
                                      ldo, stream);
    if (R == 2)
        return launch_gather<2, true>(X, n_rows, nc, ldx, mask, ldm, idx_t, ldi, k, n_lists, std_floor, mean, istd,
                                      ldo, stream);
    return launch_gather<1, true>(X, n_rows, nc, ldx, mask, ldm, idx_t, ldi, k, n_lists, std_floor, mean, istd, ldo,
                                  stream);
}

extern "C" int hypmi_gather_apply_supported(int nc, int64_t n_rows) {
    return nc > 0 && nc <= 65536 && pivot_rows_per_group(nc, n_rows) != 0 ? 1 : 0;
}

extern "C" int hypmi_gather_apply(const float* X, int64_t n_rows, int nc, int64_t ldx, const uint16_t* idx_t,
                                  int64_t ldi, int k, int64_t n_lists, float std_floor, const float* S,
                                  int64_t lds_, const float* mt, const float* it, int64_t ldt, float* out,
                                  int64_t ldo, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HYPMI_REQUIRE(X && idx_t && S && mt && it && out, "gather_apply: null pointer");
    HYPMI_REQUIRE(n_rows >= 0 && nc > 0 && nc <= 65536 && ldx >= nc && k > 0 && n_lists >= 0 && ldi >= n_lists &&
                      lds_ >= n_lists && ldo >= n_lists && ldt >= n_rows,
                  "gather_apply: bad shape");
    HYPMI_REQUIRE(aligned16(idx_t), "gather_apply: index lists must be 16-byte aligned");
    if (n_rows == 0 || n_lists == 0) return HYPMI_OK;
    const int R = pivot_rows_per_group(nc, n_rows);
    HYPMI_REQUIRE(R != 0, "gather_apply: a cohort of %d does not fit in LDS (use gather_stats + asnorm_apply)", nc);
    // the R statistics of a row group are read with one load
    HYPMI_REQUIRE(ldt % R == 0 && ldt >= round_up(n_rows, R) &&
                      (reinterpret_cast<uintptr_t>(mt) | reinterpret_cast<uintptr_t>(it)) % (4 * R) == 0,
                  "gather_apply: T-part statistics need a leading dimension that is a multiple of %d (and covers "
                  "the rows rounded up to it) and %d-byte alignment", R, 4 * R);
    const ApplyArgs ap = {S, lds_, mt, it, ldt, out, ldo};
    if (R == 4)
        return launch_gather_pivot<4, 1>(X, n_rows, nc, ldx, idx_t, ldi, k, n_lists, std_floor, nullptr, nullptr,
                                            0, stream, ap);
    if (R == 2)
        return launch_gather_pivot<2, 1>(X, n_rows, nc, ldx, idx_t, ldi, k, n_lists, std_floor, nullptr, nullptr,
                                            0, stream, ap);
    return launch_gather_pivot<1, 1>(X, n_rows, nc, ldx, idx_t, ldi, k, n_lists, std_floor, nullptr, nullptr, 0,
                                        stream, ap);
}

extern "C" int hypmi_gather_stats_pairs(const float* X, int64_t n_rows, int nc, int64_t ldx, const uint16_t* idx_t,
                                        int64_t ldi, int k, int64_t n_lists, float std_floor, float* stats,
                                        int64_t ldo, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HYPMI_REQUIRE(X && idx_t && stats, "gather_stats_pairs: null pointer");
    HYPMI_REQUIRE(n_rows >= 0 && nc > 0 && nc <= 65536 && ldx >= nc && k > 0 && n_lists >= 0 && ldi >= n_lists &&
                      ldo >= n_lists,
                  "gather_stats_pairs: bad shape");
    HYPMI_REQUIRE(aligned16(idx_t), "gather_stats_pairs: index lists must be 16-byte aligned");
    HYPMI_REQUIRE(reinterpret_cast<uintptr_t>(stats) % 8 == 0, "gather_stats_pairs: statistics must be 8-byte aligned");
    if (n_rows == 0 || n_lists == 0) return HYPMI_OK;
    const int R = pivot_rows_per_group(nc, n_rows);
    HYPMI_REQUIRE(R != 0, "gather_stats_pairs: a cohort of %d does not fit in LDS (use gather_stats)", nc);
    if (R == 4)
        return launch_gather_pivot<4>(X, n_rows, nc, ldx, idx_t, ldi, k, n_lists, std_floor, stats, nullptr, ldo, stream);
    if (R == 2)
        return launch_gather_pivot<2>(X, n_rows, nc, ldx, idx_t, ldi, k, n_lists, std_floor, stats, nullptr, ldo, stream);
    return launch_gather_pivot<1>(X, n_rows, nc, ldx, idx_t, ldi, k, n_lists, std_floor, stats, nullptr, ldo, stream);
}

extern "C" int hypmi_gather_apply_pairs(const float* X, int64_t n_rows, int nc, int64_t ldx, const uint16_t* idx_t,
                                        int64_t ldi, int k, int64_t n_lists, float std_floor, const float* S,
                                        int64_t lds_, const float* stats_t, int64_t ldt, float* out, int64_t ldo,
                                        void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HYPMI_REQUIRE(X && idx_t && S && stats_t && out, "gather_apply_pairs: null pointer");
    HYPMI_REQUIRE(n_rows >= 0 && nc > 0 && nc <= 65536 && ldx >= nc && k > 0 && n_lists >= 0 && ldi >= n_lists &&
                      lds_ >= n_lists && ldo >= n_lists && ldt >= n_rows,
                  "gather_apply_pairs: bad shape");
    HYPMI_REQUIRE(aligned16(idx_t), "gather_apply_pairs: index lists must be 16-byte aligned");
    if (n_rows == 0 || n_lists == 0) return HYPMI_OK;
    const int R = pivot_rows_per_group(nc, n_rows);
    HYPMI_REQUIRE(R != 0, "gather_apply_pairs: a cohort of %d does not fit in LDS (use gather_stats + asnorm_apply)", nc);
    // the R pairs of a row group are read with one load (two of 16 bytes for four rows)
    HYPMI_REQUIRE(ldt % R == 0 && ldt >= round_up(n_rows, R) &&
                      reinterpret_cast<uintptr_t>(stats_t) % (R == 4 ? 16 : 8 * R) == 0,
                  "gather_apply_pairs: T-part statistics need a leading dimension (in pairs) that is a multiple of "
                  "%d (and covers the rows rounded up to it) and %d-byte alignment", R, R == 4 ? 16 : 8 * R);
    const ApplyArgs ap = {S, lds_, stats_t, nullptr, ldt, out, ldo};
    if (R == 4)
        return launch_gather_pivot<4, 2>(X, n_rows, nc, ldx, idx_t, ldi, k, n_lists, std_floor, nullptr, nullptr, 0,
                                         stream, ap);
    if (R == 2)
        return launch_gather_pivot<2, 2>(X, n_rows, nc, ldx, idx_t, ldi, k, n_lists, std_floor, nullptr, nullptr, 0,
                                         stream, ap);
    return launch_gather_pivot<1, 2>(X, n_rows, nc, ldx, idx_t, ldi, k, n_lists, std_floor, nullptr, nullptr, 0,
                                     stream, ap);
}

namespace {
constexpr int SG_SEG_CAP = ((MAX_LDS_BYTES / 4 - GP_SCRATCH_FLOATS) / 2 - 1) / 32 * 32;  // members per segment (R = 2)

// segments, members per segment and list groups for a cohort of nc members and lists of k; false when the
// undivided kernel takes the cohort two rows at a time (or nothing here applies)
inline bool seg_plan(int nc, int k, int* n_seg, int* seg_len, int* groups) {
    if (nc <= 0 || nc > 65535 || k <= 0 || pivot_lds_bytes(nc, 2) <= MAX_LDS_BYTES) return false;
    const int S = (nc + SG_SEG_CAP - 1) / SG_SEG_CAP;
    if (S > SG_MAX_SEG) return false;
    *n_seg = S;
    *seg_len = (int)round_up((nc + S - 1) / S, 32);
    *groups = (k + 7) / 8 + S;
    return true;
}

template <int MODE>
int launch_gather_seg(const float* X, int64_t n_rows, int nc, int64_t ldx, const uint16_t* idx8,
                      const uint16_t* seg8, const uint16_t* gstart, int64_t ldi, int k, int64_t n_lists,
                      float std_floor, float* mean, float* istd, int64_t ldo, hipStream_t stream,
                      const ApplyArgs& ap = ApplyArgs{}) {
    int S, seg_len, groups_per_list;
    HYPMI_REQUIRE(seg_plan(nc, k, &S, &seg_len, &groups_per_list),
                  "gather (segmented): a cohort of %d does not take the segmented kernel", nc);
    const int64_t seg_bytes = (int64_t)groups_per_list * ldi * 16;
    HYPMI_REQUIRE(seg_bytes <= 0xffffff00LL, "gather (segmented): %lld bytes of lists exceed one buffer",
                  (long long)seg_bytes);
    const int64_t lds_need = round_up(((int64_t)(seg_len + 1) * 2 + GP_SCRATCH_FLOATS) * 4, 16);
    HYPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_seg_kernel<MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_need));
    const int64_t groups = (n_rows + 1) / 2;
    const int64_t gx = MODE != 0 ? round_up(groups, 8 * GP_XCD_GROUPS) : groups;
    const int64_t gy = (n_lists + SG_NL * GS_THREADS - 1) / (SG_NL * GS_THREADS);
    HYPMI_REQUIRE(gx <= 0x7fffffffLL && gy <= 65535, "gather (segmented): too many rows or lists");
    hipLaunchKernelGGL((gather_seg_kernel<MODE>), dim3((unsigned)gx, (unsigned)gy), dim3(GS_THREADS), lds_need,
                       stream, X, ldx, n_rows, nc, seg8, gstart, ldi, (unsigned)seg_bytes, idx8, k, n_lists, S, seg_len,
                       std_floor, mean, istd, ldo, ap);
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}
}  // namespace

extern "C" int hypmi_gather_seg_plan(int nc, int k, int* n_seg, int* seg_len, int* groups) {
    int a = 0, b = 0, c = 0;
    const bool ok = seg_plan(nc, k, &a, &b, &c);
    if (n_seg) *n_seg = ok ? a : 0;
    if (seg_len) *seg_len = ok ? b : 0;
    if (groups) *groups = ok ? c : 0;
    return ok ? 1 : 0;
}

extern "C" int hypmi_segment_lists(const uint16_t* idx_t, int64_t ldi, int k, int64_t n_lists, int nc,
                                   uint16_t* seg_idx, uint16_t* gstart, void* stream_) {
    HYPMI_REQUIRE(idx_t && seg_idx && gstart, "segment_lists: null pointer");
    HYPMI_REQUIRE(n_lists >= 0 && ldi >= n_lists && aligned16(idx_t) && aligned16(seg_idx), "segment_lists: bad shape");
    int S, seg_len, groups;
    HYPMI_REQUIRE(seg_plan(nc, k, &S, &seg_len, &groups), "segment_lists: a cohort of %d is not segmented", nc);
    if (n_lists == 0) return HYPMI_OK;
    hipLaunchKernelGGL(segment_lists_kernel, dim3((unsigned)((n_lists + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), idx_t, ldi, k, n_lists, S, seg_len, seg_idx, gstart);
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

extern "C" int hypmi_gather_stats_seg(const float* X, int64_t n_rows, int nc, int64_t ldx, const uint16_t* idx_t,
                                      const uint16_t* seg_idx, const uint16_t* gstart, int64_t ldi, int k,
                                      int64_t n_lists, float std_floor, float* mean, float* istd, int64_t ldo,
                                      void* stream_) {
    HYPMI_REQUIRE(X && idx_t && seg_idx && gstart && mean, "gather_stats_seg: null pointer");
    HYPMI_REQUIRE(n_rows >= 0 && ldx >= nc && k > 0 && n_lists >= 0 && ldi >= n_lists && ldo >= n_lists,
                  "gather_stats_seg: bad shape");
    HYPMI_REQUIRE(aligned16(idx_t) && aligned16(seg_idx), "gather_stats_seg: index lists must be 16-byte aligned");
    HYPMI_REQUIRE(istd != nullptr || reinterpret_cast<uintptr_t>(mean) % 8 == 0,
                  "gather_stats_seg: paired statistics must be 8-byte aligned");
    if (n_rows == 0 || n_lists == 0) return HYPMI_OK;
    return launch_gather_seg<0>(X, n_rows, nc, ldx, idx_t, seg_idx, gstart, ldi, k, n_lists, std_floor, mean, istd,
                                ldo, static_cast<hipStream_t>(stream_));
}

extern "C" int hypmi_gather_apply_seg(const float* X, int64_t n_rows, int nc, int64_t ldx, const uint16_t* idx_t,
                                      const uint16_t* seg_idx, const uint16_t* gstart, int64_t ldi, int k,
                                      int64_t n_lists, float std_floor, const float* S, int64_t lds_,
                                      const float* stats_t, int64_t ldt, float* out, int64_t ldo, void* stream_) {
    HYPMI_REQUIRE(X && idx_t && seg_idx && gstart && S && stats_t && out, "gather_apply_seg: null pointer");
    HYPMI_REQUIRE(n_rows >= 0 && ldx >= nc && k > 0 && n_lists >= 0 && ldi >= n_lists && lds_ >= n_lists &&
                      ldo >= n_lists && ldt >= n_rows,
                  "gather_apply_seg: bad shape");
    HYPMI_REQUIRE(aligned16(idx_t) && aligned16(seg_idx), "gather_apply_seg: index lists must be 16-byte aligned");
    HYPMI_REQUIRE(ldt % 2 == 0 && ldt >= round_up(n_rows, 2) && reinterpret_cast<uintptr_t>(stats_t) % 16 == 0,
                  "gather_apply_seg: T-part statistics need an even leading dimension (in pairs) that covers the "
                  "rows rounded up to it, and 16-byte alignment");
    if (n_rows == 0 || n_lists == 0) return HYPMI_OK;
    const ApplyArgs ap = {S, lds_, stats_t, nullptr, ldt, out, ldo};
    return launch_gather_seg<2>(X, n_rows, nc, ldx, idx_t, seg_idx, gstart, ldi, k, n_lists, std_floor, nullptr,
                                nullptr, 0, static_cast<hipStream_t>(stream_), ap);
}

extern "C" int hypmi_gather_redo_count(unsigned long long* count, int reset) {
    HYPMI_REQUIRE(count, "gather_redo_count: null pointer");
    HYPMI_HIP(hipMemcpyFromSymbol(count, HIP_SYMBOL(g_gather_redo), sizeof(unsigned long long)));
    if (reset) {
        const unsigned long long zero = 0;
        HYPMI_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gather_redo), &zero, sizeof(zero)));
    }
    return HYPMI_OK;
}

extern "C" int hypmi_gather_clock(double* ghz, int reset) {
    HYPMI_REQUIRE(ghz, "gather_clock: null pointer");
    unsigned long long c[2] = {0, 0};
    HYPMI_HIP(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_gather_clock), sizeof(c)));
    *ghz = c[1] ? (double)c[0] / (double)c[1] * 0.1 : 0.0;  // cycles per 10 ns tick
    if (reset) {
        const unsigned long long zero[2] = {0, 0};
        HYPMI_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gather_clock), zero, sizeof(zero)));
    }
    return HYPMI_OK;
}

extern "C" int hypmi_asnorm_apply(const float* S, int64_t lds_, int64_t ne, int64_t nt, const float* mz,
                                  const float* iz, int64_t ldz, const float* mt, const float* it,
                                  int64_t ldt, float* out, int64_t ldo, void* stream_) {
    HYPMI_REQUIRE(S && mz && iz && mt && it && out, "asnorm_apply: null pointer");
    HYPMI_REQUIRE(ne >= 0 && nt >= 0 && lds_ >= nt && ldz >= nt && ldt >= ne && ldo >= nt,
                  "asnorm_apply: bad shape");
    if (ne == 0 || nt == 0) return HYPMI_OK;
    const int64_t gy = (ne + 31) / 32, gx = (nt + 31) / 32;
    HYPMI_REQUIRE(gy <= 65535, "asnorm_apply: more than 2M enrol rows per call");
    hipLaunchKernelGGL(asnorm_apply_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), S, lds_, ne, nt, mz, iz, ldz, mt, it, ldt, out,
                       ldo);
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

extern "C" int hypmi_transpose(const float* in, int64_t rows, int64_t cols, int64_t ldi, float* out,
                               int64_t ldo, void* stream_) {
    HYPMI_REQUIRE(in && out && rows >= 0 && cols >= 0 && ldi >= cols && ldo >= rows,
                  "transpose: bad argument");
    if (rows == 0 || cols == 0) return HYPMI_OK;
    const int64_t gy = (rows + 31) / 32, gx = (cols + 31) / 32;
    HYPMI_REQUIRE(gy <= 65535, "transpose: more than 2M rows per call");
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), in, ldi, rows, cols, out, ldo);
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

extern "C" int hypmi_row_meanstd(const float* X, int64_t n_rows, int nc, int64_t ldx,
                                 const unsigned char* mask, int64_t ldm, float std_floor, float* mean,
                                 float* istd, void* stream_) {
    HYPMI_REQUIRE(X && mean && istd && n_rows >= 0 && nc > 0 && ldx >= nc, "row_meanstd: bad argument");
    if (n_rows == 0) return HYPMI_OK;
    hipLaunchKernelGGL(row_meanstd_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream_), X, ldx, n_rows, nc, mask, ldm, std_floor, mean,
                       istd);
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

extern "C" int hypmi_norm_apply(const float* S, int64_t lds_, int64_t ne, int64_t nt, const float* mz,
                                const float* iz, const float* mt, const float* it, float* out,
                                int64_t ldo, void* stream_) {
    HYPMI_REQUIRE(S && out && ne >= 0 && nt >= 0 && lds_ >= nt && ldo >= nt, "norm_apply: bad argument");
    HYPMI_REQUIRE((mz == nullptr) == (iz == nullptr) && (mt == nullptr) == (it == nullptr) && (mz || mt),
                  "norm_apply: statistics must come in (mean, 1/std) pairs");
    if (ne == 0 || nt == 0) return HYPMI_OK;
    for (int64_t e0 = 0; e0 < ne; e0 += 65535) {
        const int64_t rows = ne - e0 < 65535 ? ne - e0 : 65535;
        hipLaunchKernelGGL(norm_apply_kernel, dim3((unsigned)((nt + 255) / 256), (unsigned)rows), dim3(256),
                           0, static_cast<hipStream_t>(stream_), S + e0 * lds_, lds_, rows, nt,
                           mz ? mz + e0 : nullptr, iz ? iz + e0 : nullptr, mt, it, out + e0 * ldo, ldo);
    }
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

extern "C" int hypmi_mask_zero(float* X, int64_t rows, int64_t cols, int64_t ldx, const unsigned char* mask,
                               int64_t ldm, void* stream_) {
    HYPMI_REQUIRE(X && mask && rows >= 0 && cols >= 0 && ldx >= cols && ldm >= cols,
                  "mask_zero: bad argument");
    if (rows == 0 || cols == 0) return HYPMI_OK;
    for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
        const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
        hipLaunchKernelGGL(mask_zero_kernel, dim3((unsigned)((cols + 255) / 256), (unsigned)nr), dim3(256), 0,
                           static_cast<hipStream_t>(stream_), X + r0 * ldx, ldx, nr, cols, mask + r0 * ldm,
                           ldm);
    }
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

// Trial score matrix on the gfx950 16-bit matrix cores at fp32 accuracy, in two operand formats.
//
// (1) bf16 triples, exact:
// Every fp32 value of a packed row is written as the exact sum of three bf16 numbers
// x = x0 + x1 + x2 (x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1); 3 x 8 significand bits
// plus the signs of the residuals cover the 24 bits of an fp32).  The inner product of two rows is
// then the sum of nine bf16 products per coordinate, of which the three smallest (x1 y2, x2 y1,
// x2 y2 <= 2^-27 |x y|, below the rounding of an fp32 product chain) are dropped:
//     x y  ~=  x0 y0  +  (x0 y1 + x1 y0)  +  (x1 y1 + x0 y2 + x2 y0)
// bf16 x bf16 products are exact in fp32 and v_mfma_f32_16x16x32_bf16 accumulates in fp32, so six
// of these MFMAs replace eight fp32 MFMAs at 16 times the rate per instruction-flop: 2.7 times the
// arithmetic throughput of the fp32 kernel in score.hip at the same accuracy.  The three magnitude
// classes are accumulated separately and added once per tile, smallest first.
//
// (2) fp16 pairs, scaled (the default path): every row is multiplied by a power of two that puts its
// largest magnitude in [2^14, 2^15) and each scaled value is written as x0 + x1 with x0 = fp16(x),
// x1 = fp16(x - x0): 2 x 11 significand bits plus the sign of the residual, |x - x0 - x1| <= 2^-22 |x|
// (and <= 2^-28 of the row maximum where x1 leaves the normal fp16 range).  Three products
//     x y  ~=  x0 y0  +  (x0 y1 + x1 y0)
// are accumulated in fp32 by v_mfma_f32_16x16x32_f16 (fp16 x fp16 is exact in fp32) in two classes;
// the dropped x1 y1 is <= 2^-22 |x y|.  Measured against float64 the result is closer than an fp32
// product chain over the same rows (fewer roundings of partial sums), at half the matrix
// instructions and two thirds of the operand traffic of format (1).  The inverse scale of a row
// travels in the last (padding) slot of its split row and is applied when a tile is summed.
//
// Split rows (hypmi_split_rows, hypmi_pair_rows): coordinates are handled in chunks of 8 (one
// 16-byte slot = 8 values of one part).  With NP parts and KS16 = 2 * ceil(KP / 32) chunk pairs, a row is
//     [ HS slots: even chunks ][ NP * KS16 slots: odd chunks ][ pad to an odd slot count ]
// where HS = NP * KS16 rounded up to 16 and slot NP * j + s of a half holds part s of chunk 2 * j + h
// (coordinates 16 * j + 8 * h .. + 7).  v_mfma_f32_16x16x32_bf16 takes chunk `lane >> 4` of a k-step
// of 32 from every lane, so the lanes of one ds_read_b128 bank group read either the same slot
// offset or offsets HS apart: with HS a multiple of 16 slots and an odd row pitch all 16 lanes of a
// group land on different 16-byte bank slots.  At K = 150: 1008 bytes per row of triples, 848 of pairs.
//
// The kernel has the structure of score_rs_kernel: the split rows of 32 enrol vectors (48 for pairs
// up to 160 coordinates) stay in registers for the life of a wave (row tiles of 16), test rows stream
// through two LDS buffers by
// LDS-DMA in tiles of 32 rows (2 column tiles of 16), and a tile's scores leave through one buffer
// descriptor after v_permlane16_swap has put the two column tiles of a row side by side (128-byte
// segments).  The 16x16x32 shape is used because it sustains about 14 % more flop/s than 32x32x16
// under the chip's power limit, which is what bounds this kernel.
#include "score_tile.h"
#include "pair_rows.h"
#include <type_traits>

namespace hypmi_detail {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Operand type and matrix instruction of a format, by its number of parts.
template <int NP>
struct SplitFmt;
template <>
struct SplitFmt<3> {
    typedef bf16x8 vec;
    static __device__ __forceinline__ f32x4 mfma(vec a, vec b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <>
struct SplitFmt<2> {
    typedef f16x8 vec;
    static __device__ __forceinline__ f32x4 mfma(vec a, vec b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

__device__ __forceinline__ unsigned bf16_rn_bits(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// x -> (x0, x1, x2) as bf16 bit patterns.  Non-finite values keep their class in x0 alone.
__device__ __forceinline__ void split3(float x, unsigned& b0, unsigned& b1, unsigned& b2) {
    const unsigned u = __float_as_uint(x);
    if ((u & 0x7f800000u) == 0x7f800000u) {
        b0 = (u >> 16) | ((u & 0x007fffffu) ? 0x40u : 0u);
        b1 = b2 = 0;
        return;
    }
    b0 = bf16_rn_bits(x);
    if ((b0 & 0x7f80u) == 0x7f80u) {  // rounded up to infinity: stay finite
        b0 -= 1;
    }
    const float r1 = x - __uint_as_float(b0 << 16);
    b1 = bf16_rn_bits(r1);
    const float r2 = r1 - __uint_as_float(b1 << 16);
    b2 = bf16_rn_bits(r2);
}

// One thread per (row, chunk pair, half): 8 coordinates -> 3 slots; the first thread of a row also
// clears the padding slots.
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ G, int64_t n, int kp,
                                                         int ks16, uint4* __restrict__ H) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int per_row = 2 * ks16;
    const int64_t row = idx / per_row;
    if (row >= n) return;
    const int rem = (int)(idx - row * per_row);
    const int j = rem >> 1, h = rem & 1;
    const int k0 = 16 * j + 8 * h;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    const float* src = G + row * kp + k0;
    if (k0 < kp) {  // kp is a multiple of 8: all eight or none
        const float4 lo = *reinterpret_cast<const float4*>(src);
        const float4 hi = *reinterpret_cast<const float4*>(src + 4);
        v[0] = lo.x, v[1] = lo.y, v[2] = lo.z, v[3] = lo.w;
        v[4] = hi.x, v[5] = hi.y, v[6] = hi.z, v[7] = hi.w;
    }
    unsigned b[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split3(v[e], b[0][e], b[1][e], b[2][e]);
    const int hs = split_half_slots(ks16), row_slots = split_row_slots(ks16);
    uint4* out = H + row * row_slots;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        uint4 w;
        w.x = b[s][0] | (b[s][1] << 16);
        w.y = b[s][2] | (b[s][3] << 16);
        w.z = b[s][4] | (b[s][5] << 16);
        w.w = b[s][6] | (b[s][7] << 16);
        out[h * hs + 3 * j + s] = w;
    }
    if (rem == 0) {
        for (int p = 3 * ks16; p < hs; ++p) out[p] = make_uint4(0, 0, 0, 0);
        for (int p = hs + 3 * ks16; p < row_slots; ++p) out[p] = make_uint4(0, 0, 0, 0);
    }
}

// fp16 pairs: one wavefront per row (lane = chunk of 8 coordinates, kp <= 512).  The row maximum
// (finite values only) fixes the power of two; the last slot of the row keeps the inverse scale.
__global__ __launch_bounds__(256) void pair_rows_kernel(const float* __restrict__ G, int64_t n, int kp,
                                                        int ks16, uint4* __restrict__ H) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int k0 = 8 * lane;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (k0 < kp) {  // kp is a multiple of 8: all eight or none
        const float* src = G + row * kp + k0;
        const float4 lo = *reinterpret_cast<const float4*>(src);
        const float4 hi = *reinterpret_cast<const float4*>(src + 4);
        v[0] = lo.x, v[1] = lo.y, v[2] = lo.z, v[3] = lo.w;
        v[4] = hi.x, v[5] = hi.y, v[6] = hi.z, v[7] = hi.w;
    }
    float m = pair_chunk_max(v);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const int up = pair_scale_exponent(m);
    const int hs = split_half_slots(ks16, 2), row_slots = split_row_slots(ks16, 2);
    uint4* out = H + row * row_slots;
    if (k0 < 16 * ks16) pair_chunk_store(v, up, lane, hs, out);
    if (lane == 0) pair_row_tail(up, ks16, hs, row_slots, out);
}

// LDS-DMA piece with a wave-uniform source: lane copies the 16 bytes at sbase + voff to LDS address
// lds_dst + 16 * lane (no per-lane address arithmetic; see lds_dma_16 for why it is inline asm).
__device__ __forceinline__ void lds_dma_16_uniform(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}

// Tile loop.  In a "steady" wave (all of its rows and all columns of its chunk exist) nothing but
// MFMAs depends on the MFMAs of the current tile: the LDS-DMA of the next tile and the stores of the
// PREVIOUS tile's scores are spread between the MFMA groups of the current one, so the matrix pipe
// only pauses for the three-way sum at the end of a tile.  Other waves take the plain order
// (stage, multiply, guarded stores).
// Up to 160 coordinates (KS2 <= 5) a wave needs fewer than 256 registers and two workgroups share a
// CU; wider rows keep up to 240 registers of stationary operand alone and run one wave per SIMD.
// Pairs hold two thirds of that per row: two workgroups per CU up to KS2 = 7.
// RT row tiles of 16 enrol rows per wave: three where the registers allow it (pairs up to KS2 = 5),
// which spreads the per-tile barrier, LDS-DMA wait and operand reads over half as many more MFMAs.
template <int NP, int KS2, int RT>  // parts per value; k-steps of 32 coordinates; row tiles per wave
__global__ __launch_bounds__(256, (KS2 <= (NP == 3 ? 5 : 7) ? 2 : 1)) void score_split_kernel(const uint4* __restrict__ H1,
                                                             const uint4* __restrict__ H2,
                                                             float* __restrict__ S, int64_t n1,
                                                             int64_t n2, int64_t ldo,
                                                             int tiles_per_chunk, int gx, int64_t n_wg,
                                                             int64_t wg_per_xcd) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef SplitFmt<NP> Fmt;
    typedef typename Fmt::vec vec_t;
    constexpr int KS16 = 2 * KS2;
    constexpr int HS = split_half_slots(KS16, NP);
    constexpr int ROW_SLOTS = split_row_slots(KS16, NP);
    constexpr int TILE_SLOTS = 32 * ROW_SLOTS;
    constexpr int PIECES = (TILE_SLOTS + 63) / 64;  // LDS-DMA instructions per tile
    constexpr int BUF_SLOTS = PIECES * 64;          // the last piece runs past the tile (see hypmi_split_bytes)
    constexpr int NPW = (PIECES + 3) / 4;           // pieces per wave
    extern __shared__ __attribute__((aligned(16))) uint4 tile[];  // 2 * BUF_SLOTS (up to 128 KiB)

    // Workgroup -> (enrol block bx, test chunk by).  The (chunk, block) pairs are numbered chunk-major
    // and cut into eight equal contiguous ranges; workgroups are handed to the eight XCDs round robin
    // (a placement that is assumed for speed only), so launch slot w works on entry w / 8 of range
    // w % 8: the workgroups that run on one XCD at a time read the same chunk of test rows, which
    // then crosses the fabric once instead of once per XCD.
    const int64_t t_wg = (int64_t)(blockIdx.x & 7) * wg_per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= wg_per_xcd || t_wg >= n_wg) return;
    const int bx = (int)(t_wg % gx);
    const int64_t by = t_wg / gx;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int my_slot = (kq & 1) * HS + NP * (kq >> 1);  // + 2 * NP * ks + s
    const int64_t i0 = (int64_t)bx * (64 * RT) + wave * (16 * RT);

    // stationary operand: a[rt][s][ks] = part s of H1[i0 + 16 rt + r16], coordinates 32 ks + 8 kq .. + 7
    vec_t a[RT][NP][KS2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        int64_t row = i0 + 16 * rt + r16;
        if (row > n1 - 1) row = n1 - 1;
        const uint4* p = H1 + row * ROW_SLOTS + my_slot;
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
            for (int s = 0; s < NP; ++s) a[rt][s][ks] = __builtin_bit_cast(vec_t, p[2 * NP * ks + s]);
    }
    // pairs: inverse scales of the wave's 16 RT enrol rows, parked in LDS behind the tile buffers (a
    // lane needs those of rows 16 rt + 4 (lane >> 4) + e only when a tile is summed; keeping them in
    // registers across the MFMA loop costs 4 RT registers the three-tile variant does not have)
    float* const row_scale = reinterpret_cast<float*>(tile + 2 * BUF_SLOTS) + wave * (16 * RT);
    if constexpr (NP == 2) {
        if (lane < 16 * RT) {
            int64_t row = i0 + lane;
            if (row > n1 - 1) row = n1 - 1;
            row_scale[lane] = __uint_as_float(H1[row * ROW_SLOTS + ROW_SLOTS - 1].x);
        }
    }

    const int64_t tile0 = by * tiles_per_chunk;
    const int64_t ntiles_all = (n2 + 31) / 32;
    int ntiles = tiles_per_chunk;
    if (tile0 + ntiles > ntiles_all) ntiles = (int)(ntiles_all - tile0);
    if (ntiles <= 0) return;
    const int64_t h2_slots = n2 * ROW_SLOTS;  // valid slots of H2
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)tile;
    const unsigned lane16 = (unsigned)lane * 16u;

    // After the swap a register holds, for lane half u = lane >> 5, row 8 u + (row of the register)
    // and columns lane & 31: the row part of a store address is wave-uniform (soffset), the lane
    // part is one 32-bit voffset.
    const int u = lane >> 5, c32 = lane & 31;
    const int vb = (int)(((unsigned)(8 * u) * (unsigned)ldo + (unsigned)c32) * 4u);
    const int row_bytes = (int)ldo * 4;
    const unsigned span = (unsigned)(16 * RT) * (unsigned)ldo * 4u;
    float* const s_rows = S + i0 * ldo;
    // register p of a stored tile: rt = p >> 3, reg = (p >> 1) & 3, second of the swapped pair = p & 1
    auto prow = [](int p) { return 16 * (p >> 3) + ((p >> 1) & 3) + 4 * (p & 1); };

    // A wave whose rows and whose chunk's columns are all in range issues exactly 8 RT stores per
    // tile, so "all but the youngest 8 RT memory operations" is exactly "the next tile has landed".
    constexpr int NST = 8 * RT;
    const bool steady = (n1 - i0 >= 16 * RT) && ((tile0 + ntiles) * 32 <= n2);

    auto run = [&](auto steady_tag) {
        constexpr bool ST = decltype(steady_tag)::value;
        constexpr int NSIDE = ST ? NPW + NST : 0;  // DMA pieces, then stores, between the MFMA groups
        constexpr int NSLOT = 2 * NP * KS2;

        auto stage_piece = [&](int it, int buf, int p) {
            const int q = 4 * p + wave;
            if (4 * p + 3 < PIECES || q < PIECES) {
                const int64_t slot0 = (tile0 + it) * TILE_SLOTS + q * 64;
                const unsigned dst = lds_base + (unsigned)(buf * BUF_SLOTS + q * 64) * 16u;
                if constexpr (ST) {
                    lds_dma_16_uniform(H2 + slot0, lane16, dst);
                } else {
                    int64_t src = slot0 + lane;
                    if (src > h2_slots - 1) src = h2_slots - 1;  // short last tile: re-read a valid slot
                    lds_dma_16(H2 + src, dst);
                }
            }
        };

        float prev[NST];
#pragma unroll
        for (int p = 0; p < NST; ++p) prev[p] = 0.f;

#pragma unroll
        for (int p = 0; p < NPW; ++p) stage_piece(0, 0, p);

        for (int it = 0; it < ntiles; ++it) {
            const int buf = it & 1;
            if (ST && it > 0) {
                if constexpr (RT == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();  // tile `it` has landed; every wave is done with the other buffer
            asm volatile("" ::: "memory");
            const bool more = it + 1 < ntiles;
            if constexpr (!ST) {
                if (more) {
#pragma unroll
                    for (int p = 0; p < NPW; ++p) stage_piece(it + 1, buf ^ 1, p);
                }
            }
            // block of the previous tile; no records for the first tile, so its stores are dropped
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                s_rows + (tile0 + it - 1) * 32, 0, it > 0 ? span : 0u, 0x00020000);

            f32x4 hi[RT][2], mid[RT][2], lo[RT][2];  // pairs use hi and lo only
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int e = 0; e < 4; ++e) hi[rt][ct][e] = mid[rt][ct][e] = lo[rt][ct][e] = 0.f;

            // bt[(16 ct) * ROW_SLOTS + 2 NP ks + s]: part s of test row 16 ct + r16, this lane's chunk of k-step ks.
            // The lane's position is taken afresh from the lane counter here (two instructions per tile):
            // kept in a register across the tile loop it is the value the register allocator spills at
            // 256 registers, and the reload's wait would drain the previous tile's stores before the
            // first MFMA of every tile.
            unsigned ln;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
            const int r16 = (int)(ln & 15u), kq = (int)(ln >> 4);
            const int my_slot = (kq & 1) * HS + NP * (kq >> 1);
            const uint4* bt = tile + buf * BUF_SLOTS + r16 * ROW_SLOTS + my_slot;
            vec_t b[NP];
#pragma unroll
            for (int s = 0; s < NP; ++s) b[s] = __builtin_bit_cast(vec_t, bt[s]);
            // pairs: inverse scales of this lane's two columns (test rows r16 and 16 + r16 of the tile)
            float cs[2] = {1.f, 1.f};
            if constexpr (NP == 2) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    cs[ct] = __uint_as_float(tile[buf * BUF_SLOTS + (16 * ct + r16) * ROW_SLOTS + ROW_SLOTS - 1].x);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    // operands of the next half step: the other column tile, or the next k-step
                    const int nct = ct ^ 1, nks = ct ? (ks + 1 < KS2 ? ks + 1 : ks) : ks;
                    vec_t nb[NP];
#pragma unroll
                    for (int g = 0; g < NP; ++g) {
                        nb[g] = __builtin_bit_cast(vec_t, bt[16 * nct * ROW_SLOTS + 2 * NP * nks + g]);
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            if constexpr (NP == 3) {
                                if (g == 0) {
                                    mid[rt][ct] = Fmt::mfma(a[rt][1][ks], b[0], mid[rt][ct]);
                                    lo[rt][ct] = Fmt::mfma(a[rt][2][ks], b[0], lo[rt][ct]);
                                } else if (g == 1) {
                                    hi[rt][ct] = Fmt::mfma(a[rt][0][ks], b[0], hi[rt][ct]);
                                    lo[rt][ct] = Fmt::mfma(a[rt][1][ks], b[1], lo[rt][ct]);
                                } else {
                                    mid[rt][ct] = Fmt::mfma(a[rt][0][ks], b[1], mid[rt][ct]);
                                    lo[rt][ct] = Fmt::mfma(a[rt][0][ks], b[2], lo[rt][ct]);
                                }
                            } else {
                                if (g == 0) {
                                    lo[rt][ct] = Fmt::mfma(a[rt][1][ks], b[0], lo[rt][ct]);
                                    hi[rt][ct] = Fmt::mfma(a[rt][0][ks], b[0], hi[rt][ct]);
                                } else {
                                    lo[rt][ct] = Fmt::mfma(a[rt][0][ks], b[1], lo[rt][ct]);
                                }
                            }
                        }
                        // side work of this slot
                        const int j = (2 * ks + ct) * NP + g;
#pragma unroll
                        for (int idx = j * NSIDE / NSLOT; idx < (j + 1) * NSIDE / NSLOT; ++idx) {
                            if (idx < NPW) {
                                if (more) stage_piece(it + 1, buf ^ 1, idx);
                            } else {
                                const int p = idx - NPW;
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(prev[p]), rsrc, vb,
                                                                      prow(p) * row_bytes, 2);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int s = 0; s < NP; ++s) b[s] = nb[s];
                }
            }
            // sum of the classes (and un-scaling), then the two column tiles of a row side by side:
            // X' = [X.q0 Y.q0 X.q2 Y.q2], Y' = [X.q1 Y.q1 X.q3 Y.q3]  (q = 16-lane group = 4 rows)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 x, y;
                if constexpr (NP == 3) {
                    x = hi[rt][0] + (mid[rt][0] + lo[rt][0]);
                    y = hi[rt][1] + (mid[rt][1] + lo[rt][1]);
                } else {
                    const f32x4 rs = *reinterpret_cast<const f32x4*>(row_scale + 16 * rt + 4 * kq);
                    x = ((hi[rt][0] + lo[rt][0]) * rs) * cs[0];
                    y = ((hi[rt][1] + lo[rt][1]) * rs) * cs[1];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[e]), __float_as_uint(y[e]),
                                                               false, false);
                    prev[8 * rt + 2 * e] = __uint_as_float(sw[0]);
                    prev[8 * rt + 2 * e + 1] = __uint_as_float(sw[1]);
                }
            }

            if constexpr (!ST) {
                const int c32 = (int)(ln & 31u), u = (int)(ln >> 5);
                const int64_t col = (tile0 + it) * 32 + c32;
                const int64_t rows_left = n1 - i0;  // wave-uniform
                if (col < n2) {
#pragma unroll
                    for (int p = 0; p < NST; ++p) {
                        const int rr = prow(p) + 8 * u;
                        if (rr < rows_left) __builtin_nontemporal_store(prev[p], s_rows + rr * ldo + col);
                    }
                }
            }
        }
        if constexpr (ST) {  // scores of the last tile
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                s_rows + (tile0 + ntiles - 1) * 32, 0, span, 0x00020000);
#pragma unroll
            for (int p = 0; p < NST; ++p)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(prev[p]), rsrc, vb, prow(p) * row_bytes, 2);
        }
    };
    if (steady) run(std::true_type{});
    else run(std::false_type{});
#endif
}

template <int NP, int KS2>
static int launch_split(const uint4* H1, int64_t n1, const uint4* H2, int64_t n2, float* S, int64_t lds,
                        hipStream_t stream) {
    constexpr int PIECES = (32 * split_row_slots(2 * KS2, NP) + 63) / 64;
    constexpr int LDS_BYTES = 2 * PIECES * 64 * 16 + 1024;  // two tile buffers + the enrol rows' scales
    static_assert(LDS_BYTES <= 160 * 1024, "tile buffers exceed the LDS");
    constexpr int RT = (NP == 2 && KS2 <= 5) ? 3 : 2;
    LaunchGeom g;
    int rc = launch_geometry(n1, n2, 64 * RT, &g);
    if (rc != HYPMI_OK) return rc;
    if (LDS_BYTES > 48 * 1024)
        HYPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&score_split_kernel<NP, KS2, RT>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    const int64_t n_wg = g.gx * g.gy, wg_per_xcd = (n_wg + 7) / 8;
    HYPMI_REQUIRE(g.gx <= 0x7fffffffLL && 8 * wg_per_xcd <= 0x7fffffffLL, "score: too many workgroups");
    hipLaunchKernelGGL((score_split_kernel<NP, KS2, RT>), dim3((unsigned)(8 * wg_per_xcd)), dim3(256), LDS_BYTES,
                       stream, H1, H2, S, n1, n2, lds, g.tiles_per_chunk, (int)g.gx, n_wg, wg_per_xcd);
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

constexpr int SPLIT_MAX_KS2 = 10;

}  // namespace hypmi_detail

using namespace hypmi_detail;

// k-steps of 32 of the split kernel for a packed width, 0 if there is no instantiation for it
static int split_ksteps(int kp) {
    if (kp <= 0 || kp % 8 != 0) return 0;
    const int ks2 = (kp + 31) / 32;
    return ks2 <= SPLIT_MAX_KS2 ? ks2 : 0;
}

namespace {

int64_t row_bytes_of(int kp, int np) {
    const int ks2 = split_ksteps(kp);
    return ks2 ? (int64_t)split_row_slots(2 * ks2, np) * 16 : 0;
}

// n rows plus 1 KiB: the last LDS-DMA piece of a tile copies up to 63 slots beyond it
int64_t buffer_bytes_of(int kp, int np, int64_t n) {
    const int64_t rb = row_bytes_of(kp, np);
    return (rb && n >= 0) ? n * rb + 1024 : 0;
}

int convert_rows(const char* what, int np, const float* G, int64_t n, int kp, void* H, hipStream_t stream) {
    const int ks = 2 * split_ksteps(kp);  // chunk pairs
    HYPMI_REQUIRE(ks > 0, "%s: no split kernel for packed width %d", what, kp);
    HYPMI_REQUIRE(n >= 0, "%s: bad row count", what);
    if (n == 0) return HYPMI_OK;
    HYPMI_REQUIRE(G && H, "%s: null pointer", what);
    HYPMI_REQUIRE(aligned16(G) && aligned16(H), "%s: buffers must be 16-byte aligned", what);
    const int64_t blocks = np == 3 ? (n * 2 * ks + 255) / 256 : (n + 3) / 4;
    HYPMI_REQUIRE(blocks <= 0x7fffffffLL, "%s: too many rows", what);
    if (np == 3)
        hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, G, n, kp, ks,
                           static_cast<uint4*>(H));
    else
        hipLaunchKernelGGL(pair_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, G, n, kp, ks,
                           static_cast<uint4*>(H));
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

template <int NP>
int score_rows(const char* what, const void* H1_, int64_t n1, const void* H2_, int64_t n2, int kp, float* S,
               int64_t lds, hipStream_t stream) {
    const int ks = split_ksteps(kp);
    HYPMI_REQUIRE(ks > 0, "%s: no split kernel for packed width %d", what, kp);
    HYPMI_REQUIRE(n1 >= 0 && n2 >= 0, "%s: bad shape n1=%lld n2=%lld", what, (long long)n1, (long long)n2);
    HYPMI_REQUIRE(lds >= n2, "%s: output leading dimension %lld < n2 %lld", what, (long long)lds, (long long)n2);
    if (n1 == 0 || n2 == 0) return HYPMI_OK;
    HYPMI_REQUIRE(H1_ && H2_ && S, "%s: null pointer", what);
    HYPMI_REQUIRE(aligned16(H1_) && aligned16(H2_), "%s: split rows must be 16-byte aligned", what);
    const uint4* H1 = static_cast<const uint4*>(H1_);
    const uint4* H2 = static_cast<const uint4*>(H2_);
    switch (ks) {
        case 1: return launch_split<NP, 1>(H1, n1, H2, n2, S, lds, stream);
        case 2: return launch_split<NP, 2>(H1, n1, H2, n2, S, lds, stream);
        case 3: return launch_split<NP, 3>(H1, n1, H2, n2, S, lds, stream);
        case 4: return launch_split<NP, 4>(H1, n1, H2, n2, S, lds, stream);
        case 5: return launch_split<NP, 5>(H1, n1, H2, n2, S, lds, stream);
        case 6: return launch_split<NP, 6>(H1, n1, H2, n2, S, lds, stream);
        case 7: return launch_split<NP, 7>(H1, n1, H2, n2, S, lds, stream);
        case 8: return launch_split<NP, 8>(H1, n1, H2, n2, S, lds, stream);
        case 9: return launch_split<NP, 9>(H1, n1, H2, n2, S, lds, stream);
        default: return launch_split<NP, 10>(H1, n1, H2, n2, S, lds, stream);
    }
}

}  // namespace

extern "C" int64_t hypmi_split_row_bytes(int kp) { return row_bytes_of(kp, 3); }
extern "C" int64_t hypmi_split_bytes(int kp, int64_t n) { return buffer_bytes_of(kp, 3, n); }
extern "C" int hypmi_split_rows(const float* G, int64_t n, int kp, void* H, void* stream) {
    return convert_rows("split_rows", 3, G, n, kp, H, static_cast<hipStream_t>(stream));
}
extern "C" int hypmi_score_split(const void* H1, int64_t n1, const void* H2, int64_t n2, int kp, float* S,
                                 int64_t lds, void* stream) {
    return score_rows<3>("score_split", H1, n1, H2, n2, kp, S, lds, static_cast<hipStream_t>(stream));
}

extern "C" int64_t hypmi_pair_row_bytes(int kp) { return row_bytes_of(kp, 2); }
extern "C" int64_t hypmi_pair_bytes(int kp, int64_t n) { return buffer_bytes_of(kp, 2, n); }
extern "C" int hypmi_pair_rows(const float* G, int64_t n, int kp, void* H, void* stream) {
    return convert_rows("pair_rows", 2, G, n, kp, H, static_cast<hipStream_t>(stream));
}
extern "C" int hypmi_score_pair(const void* H1, int64_t n1, const void* H2, int64_t n2, int kp, float* S,
                                int64_t lds, void* stream) {
    return score_rows<2>("score_pair", H1, n1, H2, n2, kp, S, lds, static_cast<hipStream_t>(stream));
}

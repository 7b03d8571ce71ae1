// Trial score matrix on the gfx950 bf16 matrix cores at fp32 accuracy.
//
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
// Split rows (hypmi_split_rows): coordinates are handled in chunks of 8 (one 16-byte slot = 8 bf16
// of one part).  With KS16 = 2 * ceil(KP / 32) chunk pairs, a row is
//     [ HS slots: even chunks ][ 3 * KS16 slots: odd chunks ][ pad to an odd slot count ]
// where HS = 3 * KS16 rounded up to 16 and slot 3 * j + s of a half holds part s of chunk 2 * j + h
// (coordinates 16 * j + 8 * h .. + 7).  v_mfma_f32_16x16x32_bf16 takes chunk `lane >> 4` of a k-step
// of 32 from every lane, so the lanes of one ds_read_b128 bank group read either the same slot
// offset or offsets HS apart: with HS a multiple of 16 slots and an odd row pitch all 16 lanes of a
// group land on different 16-byte bank slots.  1008 bytes per row at K = 150.
//
// The kernel has the structure of score_rs_kernel: the split rows of 32 enrol vectors stay in
// registers for the life of a wave (2 row tiles of 16), test rows stream through two LDS buffers by
// LDS-DMA in tiles of 32 rows (2 column tiles of 16), and a tile's scores leave through one buffer
// descriptor after v_permlane16_swap has put the two column tiles of a row side by side (128-byte
// segments).  The 16x16x32 shape is used because it sustains about 14 % more flop/s than 32x32x16
// under the chip's power limit, which is what bounds this kernel.
#include "score_tile.h"
#include <type_traits>

namespace hypmi_detail {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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

__host__ __device__ constexpr int split_half_slots(int ks16) { return (3 * ks16 + 15) / 16 * 16; }
__host__ __device__ constexpr int split_row_slots(int ks16) { return (split_half_slots(ks16) + 3 * ks16) | 1; }

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
template <int KS2>  // k-steps of 32 coordinates
__global__ __launch_bounds__(256, (KS2 <= 5 ? 2 : 1)) void score_split_kernel(const uint4* __restrict__ H1,
                                                             const uint4* __restrict__ H2,
                                                             float* __restrict__ S, int64_t n1,
                                                             int64_t n2, int64_t ldo,
                                                             int tiles_per_chunk) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KS16 = 2 * KS2;
    constexpr int HS = split_half_slots(KS16);
    constexpr int ROW_SLOTS = split_row_slots(KS16);
    constexpr int TILE_SLOTS = 32 * ROW_SLOTS;
    constexpr int PIECES = (TILE_SLOTS + 63) / 64;  // LDS-DMA instructions per tile
    constexpr int BUF_SLOTS = PIECES * 64;          // the last piece runs past the tile (see hypmi_split_bytes)
    constexpr int NPW = (PIECES + 3) / 4;           // pieces per wave
    extern __shared__ __attribute__((aligned(16))) uint4 tile[];  // 2 * BUF_SLOTS (up to 128 KiB)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int my_slot = (kq & 1) * HS + 3 * (kq >> 1);  // + 6 * ks + s
    const int64_t i0 = (int64_t)blockIdx.x * 128 + wave * 32;

    // stationary operand: a[rt][s][ks] = part s of H1[i0 + 16 rt + r16], coordinates 32 ks + 8 kq .. + 7
    bf16x8 a[2][3][KS2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        int64_t row = i0 + 16 * rt + r16;
        if (row > n1 - 1) row = n1 - 1;
        const uint4* p = H1 + row * ROW_SLOTS + my_slot;
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks)
#pragma unroll
            for (int s = 0; s < 3; ++s) a[rt][s][ks] = __builtin_bit_cast(bf16x8, p[6 * ks + s]);
    }

    const int64_t tile0 = (int64_t)blockIdx.y * tiles_per_chunk;
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
    const unsigned span = 32u * (unsigned)ldo * 4u;
    float* const s_rows = S + i0 * ldo;
    // register p of a stored tile: rt = p >> 3, reg = (p >> 1) & 3, second of the swapped pair = p & 1
    auto prow = [](int p) { return 16 * (p >> 3) + ((p >> 1) & 3) + 4 * (p & 1); };

    // A wave whose rows and whose chunk's columns are all in range issues exactly 16 stores per
    // tile, so "all but the youngest 16 memory operations" is exactly "the next tile has landed".
    const bool steady = (n1 - i0 >= 32) && ((tile0 + ntiles) * 32 <= n2);

    auto run = [&](auto steady_tag) {
        constexpr bool ST = decltype(steady_tag)::value;
        constexpr int NSIDE = ST ? NPW + 16 : 0;  // DMA pieces, then stores, between the MFMA groups
        constexpr int NSLOT = 6 * KS2;

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

        float prev[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) prev[p] = 0.f;

#pragma unroll
        for (int p = 0; p < NPW; ++p) stage_piece(0, 0, p);

        for (int it = 0; it < ntiles; ++it) {
            const int buf = it & 1;
            if (ST && it > 0) {
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
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

            f32x4 hi[2][2], mid[2][2], lo[2][2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int e = 0; e < 4; ++e) hi[rt][ct][e] = mid[rt][ct][e] = lo[rt][ct][e] = 0.f;

            // bt[(16 ct) * ROW_SLOTS + 6 ks + s]: part s of test row 16 ct + r16, this lane's chunk of k-step ks
            const uint4* bt = tile + buf * BUF_SLOTS + r16 * ROW_SLOTS + my_slot;
            bf16x8 b[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) b[s] = __builtin_bit_cast(bf16x8, bt[s]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    // operands of the next half step: the other column tile, or the next k-step
                    const int nct = ct ^ 1, nks = ct ? (ks + 1 < KS2 ? ks + 1 : ks) : ks;
                    bf16x8 nb[3];
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        nb[g] = __builtin_bit_cast(bf16x8, bt[16 * nct * ROW_SLOTS + 6 * nks + g]);
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) {
                            if (g == 0) {
                                mid[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][1][ks], b[0], mid[rt][ct], 0, 0, 0);
                                lo[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][2][ks], b[0], lo[rt][ct], 0, 0, 0);
                            } else if (g == 1) {
                                hi[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][0][ks], b[0], hi[rt][ct], 0, 0, 0);
                                lo[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][1][ks], b[1], lo[rt][ct], 0, 0, 0);
                            } else {
                                mid[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][0][ks], b[1], mid[rt][ct], 0, 0, 0);
                                lo[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[rt][0][ks], b[2], lo[rt][ct], 0, 0, 0);
                            }
                        }
                        // side work of this slot
                        const int j = (2 * ks + ct) * 3 + g;
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
                    for (int s = 0; s < 3; ++s) b[s] = nb[s];
                }
            }
            // three-way sum, then the two column tiles of a row side by side:
            // X' = [X.q0 Y.q0 X.q2 Y.q2], Y' = [X.q1 Y.q1 X.q3 Y.q3]  (q = 16-lane group = 4 rows)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const f32x4 x = hi[rt][0] + (mid[rt][0] + lo[rt][0]);
                const f32x4 y = hi[rt][1] + (mid[rt][1] + lo[rt][1]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[e]), __float_as_uint(y[e]),
                                                               false, false);
                    prev[8 * rt + 2 * e] = __uint_as_float(sw[0]);
                    prev[8 * rt + 2 * e + 1] = __uint_as_float(sw[1]);
                }
            }

            if constexpr (!ST) {
                const int64_t col = (tile0 + it) * 32 + c32;
                const int64_t rows_left = n1 - i0;  // wave-uniform
                if (col < n2) {
#pragma unroll
                    for (int p = 0; p < 16; ++p) {
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
            for (int p = 0; p < 16; ++p)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(prev[p]), rsrc, vb, prow(p) * row_bytes, 2);
        }
    };
    if (steady) run(std::true_type{});
    else run(std::false_type{});
#endif
}

template <int KS2>
static int launch_split(const uint4* H1, int64_t n1, const uint4* H2, int64_t n2, float* S, int64_t lds,
                        hipStream_t stream) {
    constexpr int PIECES = (32 * split_row_slots(2 * KS2) + 63) / 64;
    constexpr int LDS_BYTES = 2 * PIECES * 64 * 16;
    static_assert(LDS_BYTES <= 160 * 1024, "tile buffers exceed the LDS");
    LaunchGeom g;
    int rc = launch_geometry(n1, n2, 128, &g);
    if (rc != HYPMI_OK) return rc;
    if (LDS_BYTES > 48 * 1024)
        HYPMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&score_split_kernel<KS2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipLaunchKernelGGL((score_split_kernel<KS2>), dim3((unsigned)g.gx, (unsigned)g.gy), dim3(256), LDS_BYTES,
                       stream, H1, H2, S, n1, n2, lds, g.tiles_per_chunk);
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

extern "C" int64_t hypmi_split_row_bytes(int kp) {
    const int ks2 = split_ksteps(kp);
    return ks2 ? (int64_t)split_row_slots(2 * ks2) * 16 : 0;
}

// n rows plus 1 KiB: the last LDS-DMA piece of a tile copies up to 63 slots beyond it
extern "C" int64_t hypmi_split_bytes(int kp, int64_t n) {
    const int64_t rb = hypmi_split_row_bytes(kp);
    return (rb && n >= 0) ? n * rb + 1024 : 0;
}

extern "C" int hypmi_split_rows(const float* G, int64_t n, int kp, void* H, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int ks = 2 * split_ksteps(kp);  // chunk pairs
    HYPMI_REQUIRE(ks > 0, "split_rows: no split kernel for packed width %d", kp);
    HYPMI_REQUIRE(n >= 0, "split_rows: bad row count");
    if (n == 0) return HYPMI_OK;
    HYPMI_REQUIRE(G && H, "split_rows: null pointer");
    HYPMI_REQUIRE(aligned16(G) && aligned16(H), "split_rows: buffers must be 16-byte aligned");
    const int64_t threads = n * 2 * ks;
    const int64_t blocks = (threads + 255) / 256;
    HYPMI_REQUIRE(blocks <= 0x7fffffffLL, "split_rows: too many rows");
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, G, n, kp, ks,
                       static_cast<uint4*>(H));
    HYPMI_HIP(hipGetLastError());
    return HYPMI_OK;
}

extern "C" int hypmi_score_split(const void* H1_, int64_t n1, const void* H2_, int64_t n2, int kp,
                                 float* S, int64_t lds, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int ks = split_ksteps(kp);
    HYPMI_REQUIRE(ks > 0, "score_split: no split kernel for packed width %d", kp);
    HYPMI_REQUIRE(n1 >= 0 && n2 >= 0, "score_split: bad shape n1=%lld n2=%lld", (long long)n1,
                  (long long)n2);
    HYPMI_REQUIRE(lds >= n2, "score_split: output leading dimension %lld < n2 %lld", (long long)lds,
                  (long long)n2);
    if (n1 == 0 || n2 == 0) return HYPMI_OK;
    HYPMI_REQUIRE(H1_ && H2_ && S, "score_split: null pointer");
    HYPMI_REQUIRE(aligned16(H1_) && aligned16(H2_), "score_split: split rows must be 16-byte aligned");
    const uint4* H1 = static_cast<const uint4*>(H1_);
    const uint4* H2 = static_cast<const uint4*>(H2_);
    switch (ks) {
        case 1: return launch_split<1>(H1, n1, H2, n2, S, lds, stream);
        case 2: return launch_split<2>(H1, n1, H2, n2, S, lds, stream);
        case 3: return launch_split<3>(H1, n1, H2, n2, S, lds, stream);
        case 4: return launch_split<4>(H1, n1, H2, n2, S, lds, stream);
        case 5: return launch_split<5>(H1, n1, H2, n2, S, lds, stream);
        case 6: return launch_split<6>(H1, n1, H2, n2, S, lds, stream);
        case 7: return launch_split<7>(H1, n1, H2, n2, S, lds, stream);
        case 8: return launch_split<8>(H1, n1, H2, n2, S, lds, stream);
        case 9: return launch_split<9>(H1, n1, H2, n2, S, lds, stream);
        default: return launch_split<10>(H1, n1, H2, n2, S, lds, stream);
    }
}

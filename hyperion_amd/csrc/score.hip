// Trial score matrix  S[i,j] = <G1[i,:], G2[j,:]>  on the gfx950 fp32 matrix cores.
//
// G1/G2 are packed rows of width KP (a multiple of 8): the K projected coordinates followed by
// [q, 1] on the enrol side and [1, q + c] on the test side, so the dot product already contains the
// quadratic terms and the constant and the kernel has no epilogue arithmetic at all.
//
// Register-stationary design (K is tiny, the matrix is huge):
//   * a wave keeps the packed rows of 32*RT enrol vectors in VGPRs for its whole life, already in
//     the A-operand layout of v_mfma_f32_32x32x2_f32 (lane l: row l&31, k-parity l>>5);
//   * test vectors stream through LDS in tiles of 32 rows; a tile is one contiguous 128*KP-byte
//     piece of G2, copied by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) into one
//     of two buffers while the previous tile is being multiplied;
//   * per tile a wave issues KP/8 ds_read_b128 and KP*RT/2 MFMAs and stores a (32*RT) x 32 block
//     of S through one buffer descriptor (scalar row offset + one 32-bit lane offset, so a store
//     costs one SALU instruction and the store); every store writes two full 128-byte lines.
// Measured at 100k x 100k, KP = 152 (MI355X): 22.2 ms = 135 TFLOP/s; the same loop without the
// stores runs 20.8 ms, with 64-bit per-lane store addresses 24.0 ms; raising s_setprio per
// workgroup or per tile half, four waves per SIMD (RT = 1) and 16-byte stores were all within 1.5 %.
// The k order inside one 16-byte read is shared by both operands (lane half h owns k = 8s+4h+e),
// which is all an inner product needs.
#include "score_tile.h"

namespace hypmi_detail {

template <int KP, int RT>
__global__ __launch_bounds__(256, 2) void score_rs_kernel(const float* __restrict__ G1,
                                                          const float* __restrict__ G2,
                                                          float* __restrict__ S, int64_t n1,
                                                          int64_t n2, int64_t ldo,
                                                          int tiles_per_chunk) {
    static_assert(KP % 8 == 0, "packed width must be a multiple of 8");
    constexpr int KS = KP / 8;        // 16-byte steps per row = LDS-DMA instructions per tile
    constexpr int TILE_F4 = 8 * KP;   // float4 per 32-row tile
    __shared__ __attribute__((aligned(16))) float4 tile[2 * TILE_F4];  // two tile buffers

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int64_t i0 = (int64_t)blockIdx.x * (128 * RT) + wave * (32 * RT);

    // stationary operand: a[t][s] = G1[i0 + 32 t + r][8 s + 4 h .. +3]
    float4 a[RT][KS];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        int64_t row = i0 + 32 * t + r;
        if (row > n1 - 1) row = n1 - 1;
        const float4* p = reinterpret_cast<const float4*>(G1 + row * KP) + h;
#pragma unroll
        for (int s = 0; s < KS; ++s) a[t][s] = p[2 * s];
    }

    const int64_t tile0 = (int64_t)blockIdx.y * tiles_per_chunk;
    const int64_t ntiles_all = (n2 + 31) / 32;
    int ntiles = tiles_per_chunk;
    if (tile0 + ntiles > ntiles_all) ntiles = (int)(ntiles_all - tile0);
    const int64_t g2_f4 = n2 * (KP / 4);  // valid float4 count of G2
    const float4* G2v = reinterpret_cast<const float4*>(G2);

    // LDS-DMA of tile `it` into buffer `buf`: instruction q covers float4 [64 q, 64 q + 64);
    // the wait for it is placed by hand below.
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)tile;
    auto stage = [&](int it, int buf) {
        const int64_t base = (tile0 + it) * TILE_F4;
#pragma unroll
        for (int q0 = 0; q0 < KS; q0 += 4) {
            const int q = q0 + wave;
            if (q < KS) {
                int64_t src = base + q * 64 + lane;
                if (src > g2_f4 - 1) src = g2_f4 - 1;  // last tile may be short: re-read a valid row
                const unsigned dst = lds_base + (unsigned)(buf * TILE_F4 + q * 64) * 16u;
                lds_dma_16(G2v + src, dst);
            }
        }
    };

    // A wave whose 32*RT rows and whose chunk's columns are all in range issues exactly 16*RT stores
    // per tile, so "all but the youngest 16*RT memory operations" is exactly "the tile has landed".
    // Any other wave drains everything.
    const bool steady = (n1 - i0 >= 32 * RT) && ((tile0 + ntiles) * 32 <= n2);

    if (ntiles > 0) stage(0, 0);
    for (int it = 0; it < ntiles; ++it) {
        const int buf = it & 1;
        if (steady) {
            if constexpr (RT == 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // tile `it` has landed; every wave is done with the other buffer
        asm volatile("" ::: "memory");
        if (it + 1 < ntiles) stage(it + 1, buf ^ 1);

        f32x16 acc[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

        const float4* bt = tile + buf * TILE_F4 + r * (KP / 4) + h;
        float4 b = bt[0];
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 bn = bt[s + 1 < KS ? 2 * (s + 1) : 2 * s];  // prefetch the next step
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s].x, b.x, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s].y, b.y, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s].z, b.z, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s].w, b.w, acc[t], 0, 0, 0);
            b = bn;
            // keep the LDS read of step s+1 ahead of the MFMAs of step s
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * RT, 0);
        }

        // D layout: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
        // Row offsets are wave-uniform (scalar base per store), the lane part is one 32-bit offset.
        const int64_t col0 = (tile0 + it) * 32;
        const unsigned voff = (unsigned)(4 * h) * (unsigned)ldo + (unsigned)r;
        // recompute the scalar row bases per tile instead of keeping 32 of them live across the loop
        int64_t ldo_t = ldo;
        asm volatile("" : "+s"(ldo_t));
        const int64_t rows_left = n1 - i0;  // wave-uniform
        if (col0 + r < n2) {
            if (rows_left >= 32 * RT) {
                store_tile_buffer<RT>(acc, S + i0 * ldo_t + col0, (unsigned)ldo_t, voff);
            } else {
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int rr = 32 * t + (e & 3) + 8 * (e >> 2);
                        float* base = S + (i0 + rr) * ldo_t + col0;
                        if (rr + 4 * h < rows_left) __builtin_nontemporal_store(acc[t][e], base + voff);
                    }
            }
        }
    }
}

// One plain launcher per instantiation.  The tile buffers are static LDS (up to 76 KiB at KP = 304),
// so no per-kernel attribute call is needed before the launch.
#define HYPMI_DEFINE_LAUNCH(KP, RT)                                                               \
    static int launch_##KP(const float* G1, int64_t n1, const float* G2, int64_t n2, float* S,          \
                           int64_t lds, hipStream_t stream) {                                          \
        LaunchGeom g;                                                                                  \
        int rc = launch_geometry(n1, n2, 128 * RT, &g);                                                \
        if (rc != HYPMI_OK) return rc;                                                                 \
        hipLaunchKernelGGL((score_rs_kernel<KP, RT>), dim3((unsigned)g.gx, (unsigned)g.gy),            \
                           dim3(256), 0, stream, G1, G2, S, n1, n2, lds, g.tiles_per_chunk);           \
        HYPMI_HIP(hipGetLastError());                                                                  \
        return HYPMI_OK;                                                                               \
    }

HYPMI_DEFINE_LAUNCH(16, 2)
HYPMI_DEFINE_LAUNCH(32, 2)
HYPMI_DEFINE_LAUNCH(64, 2)
HYPMI_DEFINE_LAUNCH(96, 2)
HYPMI_DEFINE_LAUNCH(128, 2)
HYPMI_DEFINE_LAUNCH(152, 2)
HYPMI_DEFINE_LAUNCH(192, 1)
HYPMI_DEFINE_LAUNCH(232, 1)
HYPMI_DEFINE_LAUNCH(304, 1)
#undef HYPMI_DEFINE_LAUNCH

}  // namespace hypmi_detail

using namespace hypmi_detail;

// Packed widths with a register-stationary instantiation; wider models use the K-loop GEMM.
int hypmi_kp_for_k(int K) {
    static const int widths[] = {16, 32, 64, 96, 128, 152, 192, 232, 304};
    const int need = K + 2;
    for (int w : widths)
        if (w >= need) return w;
    return (int)round_up(need, 8);
}

extern "C" int hypmi_packed_width(int K) { return K > 0 ? hypmi_kp_for_k(K) : 0; }

extern "C" int hypmi_score(const float* G1, int64_t n1, const float* G2, int64_t n2, int kp, float* S,
                           int64_t lds, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HYPMI_REQUIRE(G1 && G2 && S, "score: null pointer");
    HYPMI_REQUIRE(n1 >= 0 && n2 >= 0 && kp > 0 && kp % 8 == 0, "score: bad shape n1=%lld n2=%lld kp=%d",
                  (long long)n1, (long long)n2, kp);
    HYPMI_REQUIRE(lds >= n2, "score: output leading dimension %lld < n2 %lld", (long long)lds,
                  (long long)n2);
    HYPMI_REQUIRE(aligned16(G1) && aligned16(G2), "score: packed rows must be 16-byte aligned");
    if (n1 == 0 || n2 == 0) return HYPMI_OK;
    switch (kp) {
        case 16: return launch_16(G1, n1, G2, n2, S, lds, stream);
        case 32: return launch_32(G1, n1, G2, n2, S, lds, stream);
        case 64: return launch_64(G1, n1, G2, n2, S, lds, stream);
        case 96: return launch_96(G1, n1, G2, n2, S, lds, stream);
        case 128: return launch_128(G1, n1, G2, n2, S, lds, stream);
        case 152: return launch_152(G1, n1, G2, n2, S, lds, stream);
        case 192: return launch_192(G1, n1, G2, n2, S, lds, stream);
        case 232: return launch_232(G1, n1, G2, n2, S, lds, stream);
        case 304: return launch_304(G1, n1, G2, n2, S, lds, stream);
        default:
            return hypmi_launch_gemm_nt(G1, kp, G2, kp, nullptr, S, lds, n1, n2, kp, stream);
    }
}

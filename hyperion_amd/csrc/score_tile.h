// Pieces shared by the score kernels (score.hip, score_split.hip): the buffer-descriptor store of
// one accumulator block, the LDS-DMA of one 1 KiB piece and the launch geometry.
#pragma once
#include "common.h"

namespace hypmi_detail {

// Stores a (32*RT) x 32 accumulator block through one buffer descriptor: the row offsets are scalar
// (soffset), the lane part is one 32-bit voffset, so a store costs one SALU instruction and the store.
// The buffer builtins exist only in the device pass of the compiler.
template <int RT>
__device__ __forceinline__ void store_tile_buffer(const f32x16 (&acc)[RT], float* tile_base, unsigned ldo,
                                                  unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned span = (unsigned)(32 * RT) * ldo * 4u;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(tile_base, 0, span, 0x00020000);
    const int vb = (int)(voff * 4u);
    const int row_bytes = (int)ldo * 4;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int rr = 32 * t + (e & 3) + 8 * (e >> 2);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[t][e]), rsrc, vb, rr * row_bytes, 2);
        }
#endif
}

// One wave-instruction of LDS-DMA: every lane copies the 16 bytes at `src` to LDS address
// lds_dst + 16 * lane.  Issued through inline asm so that the compiler neither counts it nor drains
// it with a vmcnt(0) in front of the LDS reads; the caller places the wait by hand.
__device__ __forceinline__ void lds_dma_16(const void* src, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_dst)
        : "memory");
}

struct LaunchGeom {
    int64_t gx, gy;
    int tiles_per_chunk;
};

// enough workgroups to fill 256 CUs several times over, but long enough streams that loading the
// stationary rows (one pass over rows_per_wg * KP floats) stays a small fraction
static int launch_geometry(int64_t n1, int64_t n2, int rows_per_wg, LaunchGeom* g) {
    const int64_t gx = (n1 + rows_per_wg - 1) / rows_per_wg;
    const int64_t ntiles = (n2 + 31) / 32;
    // 64 tiles (2048 test rows, 1.7 MB of fp16 pairs: half an XCD's L2) per stream.  Measured on
    // 100k x 100k, K = 150, fp16 pairs, with the chunk-major workgroup order of the split kernel:
    // 32 tiles 9.61 ms, 64 9.34, 128 9.45, 256 9.64 (the kernel is bound by its stores and the clock it
    // holds, not by these re-reads: 3.7 GB of L2 misses per launch at 128 tiles against 40 GB written)
    int64_t tiles_per_chunk = 64;
    while (tiles_per_chunk > 8 && gx * ((ntiles + tiles_per_chunk - 1) / tiles_per_chunk) < 2048)
        tiles_per_chunk /= 2;
    int64_t gy = (ntiles + tiles_per_chunk - 1) / tiles_per_chunk;
    while (gy > 65535) {
        tiles_per_chunk *= 2;
        gy = (ntiles + tiles_per_chunk - 1) / tiles_per_chunk;
    }
    HYPMI_REQUIRE(gx <= 0x7fffffffLL, "score: too many enrol rows");
    g->gx = gx;
    g->gy = gy;
    g->tiles_per_chunk = (int)tiles_per_chunk;
    return HYPMI_OK;
}

}  // namespace hypmi_detail

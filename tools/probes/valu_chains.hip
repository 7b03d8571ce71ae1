// How many independent accumulation chains a wave needs before plain fp32 adds / fmas issue at the
// SIMD's rate (2 clocks per wave64 instruction), for 4 and 8 resident waves per SIMD.
// Standalone (no torch).   hipcc -O3 --offload-arch=gfx950 valu_chains.hip -o valu_chains
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// CH chains; every pass over the chains issues one add and one fma per pair of chains, the way the
// gather accumulates a sum and a sum of squares
template <int CH, int THREADS>
__global__ __launch_bounds__(THREADS) void chains(float* out, int iters) {
    float a[CH];
    float b[4] = {0.5f, 0.25f, 0.125f, 0.75f};
#pragma unroll
    for (int c = 0; c < CH; ++c) a[c] = threadIdx.x * 0.001f + c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / CH; ++r) {
#pragma unroll
            for (int c = 0; c < CH; c += 2) {
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[c]) : "v"(b[c & 3]));
                asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a[c + 1]) : "v"(b[c & 3]));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += a[c];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int CH, int THREADS>
static void run(const char* what, int blocks, float* out, double ghz, int ncu) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((chains<CH, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    CK(hipGetLastError());
    // wave-instructions per SIMD: 64 per iteration and wave; waves per SIMD = blocks * THREADS / 64 / (4 ncu)
    const double per_simd = 64.0 * iters * ((double)blocks * THREADS / 64.0) / (4.0 * ncu);
    printf("%-34s chains %2d  %8.3f ms  %5.2f clk per wave-instruction per SIMD\n", what, CH, best,
           best * 1e-3 * ghz * 1e9 / per_simd);
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    const double ghz = 2.4;
    float* out;
    CK(hipMalloc(&out, (size_t)4 * ncu * 1024 * 4));
    run<2, 1024>("16 waves/CU (one block of 1024)", ncu, out, ghz, ncu);
    run<4, 1024>("16 waves/CU (one block of 1024)", ncu, out, ghz, ncu);
    run<8, 1024>("16 waves/CU (one block of 1024)", ncu, out, ghz, ncu);
    run<16, 1024>("16 waves/CU (one block of 1024)", ncu, out, ghz, ncu);
    run<4, 1024>("32 waves/CU (two blocks of 1024)", 2 * ncu, out, ghz, ncu);
    run<8, 1024>("32 waves/CU (two blocks of 1024)", 2 * ncu, out, ghz, ncu);
    run<4, 256>("4 waves/CU (one block of 256)", ncu, out, ghz, ncu);
    run<8, 256>("4 waves/CU (one block of 256)", ncu, out, ghz, ncu);
    run<4, 512>("8 waves/CU (one block of 512)", ncu, out, ghz, ncu);
    run<8, 512>("8 waves/CU (one block of 512)", ncu, out, ghz, ncu);
    return 0;
}

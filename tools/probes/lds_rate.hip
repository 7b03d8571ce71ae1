// Issue-rate probe: conflict-free ds_read_b64 / ds_read_b128 streams with 0, 1, 2 or 4 plain VALU
// instructions between consecutive reads, 16 waves per CU.  Prints nominal clocks per DS
// wave-instruction per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NV, int WIDE>
__global__ __launch_bounds__(1024) void lds_rate(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float xr[];
    for (int i = threadIdx.x; i < 40000; i += 1024) xr[i] = i * 0.5f;
    __syncthreads();
    // lane l of a half wave reads bank pair l (b64) / bank quad l % 16 (b128): conflict free
    unsigned a0 = (threadIdx.x & 63) * (WIDE ? 16 : 8) + (threadIdx.x >> 6) * 4096, a1 = a0 + 1024, a2 = a0 + 2048, a3 = a0 + 3072;
    float f0 = 1.f, f1 = 2.f, f2 = 3.f, f3 = 4.f, g = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#define RD(A) if (WIDE) asm volatile("ds_read_b128 v[20:23], %0" :: "v"(A) : "v20", "v21", "v22", "v23"); \
              else asm volatile("ds_read_b64 v[20:21], %0" :: "v"(A) : "v20", "v21");
#define VA(N) if (NV >= N) asm volatile("v_fma_f32 %0, %4, %4, %0\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g));
#define VB(N) if (NV >= N) asm volatile("v_fma_f32 %1, %4, %4, %1\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g));
#define VC(N) if (NV >= N) asm volatile("v_fma_f32 %2, %4, %4, %2\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g));
#define VD(N) if (NV >= N) asm volatile("v_fma_f32 %3, %4, %4, %3\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(g));
            RD(a0) VA(1) VB(2) VC(3) VD(4) VA(5) VB(6)
            RD(a1) VA(1) VB(2) VC(3) VD(4) VA(5) VB(6)
            RD(a2) VA(1) VB(2) VC(3) VD(4) VA(5) VB(6)
            RD(a3) VA(1) VB(2) VC(3) VD(4) VA(5) VB(6)
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    out[blockIdx.x * 1024 + threadIdx.x] = f0 + f1 + f2 + f3;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const double ghz = prop.clockRate / 1e6;
    const int ncu = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* out;
    CK(hipMalloc(&out, sizeof(float) * 1024 * ncu));
    const int iters = 4000;
#define RUN(NV, W)                                                                                     \
    {                                                                                                  \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_rate<NV, W>),                        \
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160000));                   \
        hipLaunchKernelGGL((lds_rate<NV, W>), dim3(ncu), dim3(1024), 160000, 0, out, 10);              \
        CK(hipEventRecord(e0));                                                                        \
        hipLaunchKernelGGL((lds_rate<NV, W>), dim3(ncu), dim3(1024), 160000, 0, out, iters);           \
        CK(hipEventRecord(e1));                                                                        \
        CK(hipEventSynchronize(e1));                                                                   \
        float ms;                                                                                      \
        CK(hipEventElapsedTime(&ms, e0, e1));                                                          \
        const double ds = 16.0 * iters * 16; /* DS wave-instructions per CU */                         \
        printf("%s + %d v_fma per read: %7.3f ms  %5.2f nominal clk per DS wave-instruction per CU (%5.1f B/clk/CU)\n", \
               W ? "ds_read_b128" : "ds_read_b64 ", NV, ms, ms * 1e-3 * ghz * 1e9 / ds,                \
               (W ? 1024.0 : 512.0) * ds / (ms * 1e-3 * ghz * 1e9));                                   \
    }
    RUN(0, 0) RUN(1, 0) RUN(2, 0) RUN(4, 0) RUN(6, 0) RUN(0, 1) RUN(2, 1) RUN(4, 1) RUN(6, 1)
    return 0;
}

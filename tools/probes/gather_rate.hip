// What bounds the cohort-statistics gather: VALU issue, LDS bank conflicts or the index stream.
// Standalone (no torch).  Part 1 times bare VALU loops (16 waves per CU) for the instruction kinds
// the gather's inner loop can be built from.  Part 2 times gather loops of increasing leanness on
// three index patterns: uniformly random lists, lists whose entry j of list q falls in LDS bank
// pair (q + j) mod 32 (no two lanes of a half wave on one bank pair), and all entries equal.
//   hipcc -O3 --offload-arch=gfx950 gather_rate.hip -o gather_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------- part 1: VALU issue rates
template <int MODE>
__global__ __launch_bounds__(1024) void valu_rate(float* out, int iters) {
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    float b0 = 0.5f, b1 = 0.25f, b2 = 0.125f, b3 = 0.75f;
    unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7, three = 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE == 0)  // plain fma, 4 independent chains
                asm volatile("v_fma_f32 %0, %4, %4, %0\n v_fma_f32 %1, %5, %5, %1\n v_fma_f32 %2, %6, %6, %2\n v_fma_f32 %3, %7, %7, %3"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
            if (MODE == 1) {  // packed fma, 2 independent chains of 2
                asm volatile("v_pk_fma_f32 %0, %2, %2, %0\n v_pk_fma_f32 %1, %3, %3, %1\n v_pk_fma_f32 %0, %2, %2, %0\n v_pk_fma_f32 %1, %3, %3, %1"
                             : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&b0), "v"(*(double*)&b2));
            }
            if (MODE == 2)  // packed add
                asm volatile("v_pk_add_f32 %0, %2, %0\n v_pk_add_f32 %1, %3, %1\n v_pk_add_f32 %0, %2, %0\n v_pk_add_f32 %1, %3, %1"
                             : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&b0), "v"(*(double*)&b2));
            if (MODE == 3)  // plain add
                asm volatile("v_add_f32 %0, %4, %0\n v_add_f32 %1, %5, %1\n v_add_f32 %2, %6, %2\n v_add_f32 %3, %7, %3"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
            if (MODE == 4)  // and / bfe
                asm volatile("v_and_b32 %0, 0xffff, %0\n v_bfe_u32 %1, %1, 16, 16\n v_and_b32 %2, 0xffff, %2\n v_bfe_u32 %3, %3, 16, 16"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            if (MODE == 5)  // lshl_add
                asm volatile("v_lshl_add_u32 %0, %0, 3, 0\n v_lshl_add_u32 %1, %1, 3, 0\n v_lshl_add_u32 %2, %2, 3, 0\n v_lshl_add_u32 %3, %3, 3, 0"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            if (MODE == 6)  // mad_u32_u16 with op_sel
                asm volatile("v_mad_u32_u16 %0, %0, 8, 0\n v_mad_u32_u16 %1, %1, 8, 0 op_sel:[1,0,0,0]\n v_mad_u32_u16 %2, %2, 8, 0\n v_mad_u32_u16 %3, %3, 8, 0 op_sel:[1,0,0,0]"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
            if (MODE == 7)  // sdwa shift of a 16-bit half
                asm volatile("v_lshlrev_b32_sdwa %0, %4, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                             "v_lshlrev_b32_sdwa %1, %4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                             "v_lshlrev_b32_sdwa %2, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
                             "v_lshlrev_b32_sdwa %3, %4, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(three));
            if (MODE == 8)  // fma_mix (fp16 halves into fp32)
                asm volatile("v_fma_mix_f32 %0, %4, %4, %0 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %1, %5, %5, %1 op_sel:[1,1,0] op_sel_hi:[1,1,0]\n"
                             "v_fma_mix_f32 %2, %6, %6, %2 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %3, %7, %7, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0]"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));
            if (MODE == 9)  // sdwa convert of a 16-bit half to float
                asm volatile("v_cvt_f32_u32_sdwa %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                             "v_cvt_f32_u32_sdwa %1, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                             "v_cvt_f32_u32_sdwa %2, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                             "v_cvt_f32_u32_sdwa %3, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(u0 + u1 + u2 + u3);
}

// ---------------------------------------------------------------- part 2: gather loops
constexpr int GS = 1024;
// shader clock (s_memtime) and constant 100 MHz clock (s_memrealtime) around the walk of one workgroup in the
// middle of the grid: their ratio is the clock the chip holds under this load
__device__ unsigned long long g_clk[4];

// VAR 0: the shipped loop (per-list pivot, compiler-packed math)
// VAR 1: pivot subtracted when the rows are copied to LDS; plain adds and fmas; and/bfe + lshl_add
// VAR 2: as 1 with one v_mad_u32_u16 per address
// VAR 3: as 2 with packed add / fma
// VAR 4: as 2 without the first moment (sum of squares only)
// VAR 5: address + LDS read only (values OR-ed together): the LDS + index floor
template <int VAR>
__global__ __launch_bounds__(GS) void gather(const float* __restrict__ X, int64_t ldx, int nc,
                                             const uint16_t* __restrict__ idx8, int64_t ldi, int k, int64_t Q,
                                             float* __restrict__ mean, float* __restrict__ istd, int64_t ldo) {
    extern __shared__ __attribute__((aligned(16))) float xr[];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 2;
    float rp0 = 0.f, rp1 = 0.f;
    if (VAR >= 1) { rp0 = X[r0 * ldx]; rp1 = X[(r0 + 1) * ldx]; }
    for (int c = tid; c < nc; c += GS) {
        xr[c * 2] = X[r0 * ldx + c] - rp0;
        xr[c * 2 + 1] = X[(r0 + 1) * ldx + c] - rp1;
    }
    __syncthreads();
    const char* xb = reinterpret_cast<const char*>(xr);
    const int kg = k >> 3;
    for (int64_t q = tid; q < Q; q += GS) {
        const uint4* lp = reinterpret_cast<const uint4*>(idx8) + q;
        float s1a = 0.f, s1b = 0.f, s2a = 0.f, s2b = 0.f, pa = 0.f, pb = 0.f;
        unsigned acc_or = 0;
        if (VAR == 0) {
            const float2 p = *reinterpret_cast<const float2*>(xb + (lp[0].x & 0xffffu) * 8);
            pa = p.x; pb = p.y;
        }
        auto add = [&](unsigned w, int hi) {
            unsigned addr;
            if (VAR <= 1) addr = (hi ? (w >> 16) : (w & 0xffffu)) * 8;
            else if (hi) asm("v_mad_u32_u16 %0, %1, 8, 0 op_sel:[1,0,0,0]" : "=v"(addr) : "v"(w));
            else asm("v_mad_u32_u16 %0, %1, 8, 0" : "=v"(addr) : "v"(w));
            const float2 v = *reinterpret_cast<const float2*>(xb + addr);
            if (VAR == 0) {
                const float da = v.x - pa, db = v.y - pb;
                s1a += da; s1b += db;
                s2a = fmaf(da, da, s2a); s2b = fmaf(db, db, s2b);
            } else if (VAR == 1 || VAR == 2) {
                asm("v_add_f32 %0, %1, %0" : "+v"(s1a) : "v"(v.x));
                asm("v_add_f32 %0, %1, %0" : "+v"(s1b) : "v"(v.y));
                asm("v_fma_f32 %0, %1, %1, %0" : "+v"(s2a) : "v"(v.x));
                asm("v_fma_f32 %0, %1, %1, %0" : "+v"(s2b) : "v"(v.y));
            } else if (VAR == 3) {
                s1a += v.x; s1b += v.y;
                s2a = fmaf(v.x, v.x, s2a); s2b = fmaf(v.y, v.y, s2b);
            } else if (VAR == 4) {
                asm("v_fma_f32 %0, %1, %1, %0" : "+v"(s2a) : "v"(v.x));
                asm("v_fma_f32 %0, %1, %1, %0" : "+v"(s2b) : "v"(v.y));
            } else {
                acc_or |= __float_as_uint(v.x) ^ __float_as_uint(v.y);
            }
        };
#pragma unroll 2
        for (int g = 0; g < kg; ++g) {
            const uint4 pk = lp[(int64_t)g * ldi];
            add(pk.x, 0); add(pk.x, 1); add(pk.y, 0); add(pk.y, 1);
            add(pk.z, 0); add(pk.z, 1); add(pk.w, 0); add(pk.w, 1);
        }
        if (VAR == 5) { s1a = __uint_as_float(acc_or); }
        const float m0 = s1a / k, m1 = s1b / k;
        mean[r0 * ldo + q] = pa + m0 + rp0;
        mean[(r0 + 1) * ldo + q] = pb + m1 + rp1;
        istd[r0 * ldo + q] = rsqrtf(fmaxf(s2a / k - m0 * m0, 1e-10f));
        istd[(r0 + 1) * ldo + q] = rsqrtf(fmaxf(s2b / k - m1 * m1, 1e-10f));
    }
}

// Flattened (list, group) stream per lane with a ring of D index loads in flight; pivot subtracted at
// the fill; plain adds / fmas; one v_mad_u32_u16 per address.  MODE 0: both moments, 1: sum of
// squares only, 2: LDS reads only, 3: the sums of squares on the matrix pipe (v_mfma_f32_4x4x1: a lane's
// own square is the diagonal entry of its 4 x 4 block), 4: one of the two sums of squares there.
template <int D, int MODE>
__global__ __launch_bounds__(GS) void gather2(const float* __restrict__ X, int64_t ldx, int nc,
                                              const uint16_t* __restrict__ idx8, int64_t ldi, int k, int64_t Q,
                                              float* __restrict__ mean, float* __restrict__ istd, int64_t ldo,
                                              int confine) {
    extern __shared__ __attribute__((aligned(16))) float xr[];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 2;
    const float rp0 = X[r0 * ldx], rp1 = X[(r0 + 1) * ldx];
    for (int c = tid; c < nc; c += GS) {
        xr[c * 2] = X[r0 * ldx + c] - rp0;
        xr[c * 2 + 1] = X[(r0 + 1) * ldx + c] - rp1;
    }
    __syncthreads();
    const char* xb = reinterpret_cast<const char*>(xr);
    const int G = k >> 3;
    const int n = (int)(Q / GS) * G;  // Q is a multiple of GS here
    const uint4* p = reinterpret_cast<const uint4*>(idx8) + tid;
    const int64_t wrap = (confine ? 0 : (int64_t)GS) - (int64_t)G * ldi;  // confine: re-read the first 1024 lists
    int g_ld = 0;
    uint4 ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        ring[d] = *p;
        p += ldi;
        if (++g_ld == G) { g_ld = 0; p += wrap; }
    }
    const bool report = blockIdx.x == gridDim.x / 2 && tid == 0;
    if (report) { g_clk[0] = clock64(); g_clk[1] = wall_clock64(); }
    float s1a = 0.f, s1b = 0.f, s2a = 0.f, s2b = 0.f;
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v ma = {0.f, 0.f, 0.f, 0.f}, mb = {0.f, 0.f, 0.f, 0.f};
    unsigned acc_or = 0;
    int g = 0;
    int64_t q = tid;
    auto add = [&](unsigned w, int hi) {
        unsigned addr;
        if (hi) asm("v_mad_u32_u16 %0, %1, 8, 0 op_sel:[1,0,0,0]" : "=v"(addr) : "v"(w));
        else asm("v_mad_u32_u16 %0, %1, 8, 0" : "=v"(addr) : "v"(w));
        const float2 v = *reinterpret_cast<const float2*>(xb + addr);
        if (MODE == 0 || MODE >= 3) {
            asm("v_add_f32 %0, %1, %0" : "+v"(s1a) : "v"(v.x));
            asm("v_add_f32 %0, %1, %0" : "+v"(s1b) : "v"(v.y));
        }
        if (MODE == 3) {
            ma = __builtin_amdgcn_mfma_f32_4x4x1f32(v.x, v.x, ma, 0, 0, 0);
            mb = __builtin_amdgcn_mfma_f32_4x4x1f32(v.y, v.y, mb, 0, 0, 0);
        } else if (MODE == 4) {
            asm("v_fma_f32 %0, %1, %1, %0" : "+v"(s2a) : "v"(v.x));
            mb = __builtin_amdgcn_mfma_f32_4x4x1f32(v.y, v.y, mb, 0, 0, 0);
        } else if (MODE <= 1) {
            asm("v_fma_f32 %0, %1, %1, %0" : "+v"(s2a) : "v"(v.x));
            asm("v_fma_f32 %0, %1, %1, %0" : "+v"(s2b) : "v"(v.y));
        } else {
            acc_or |= __float_as_uint(v.x) ^ __float_as_uint(v.y);
        }
    };
    for (int i = 0; i < n; i += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (i + d < n) {
                const uint4 pk = ring[d];
                if (i + d + D < n) {
                    ring[d] = *p;
                    p += ldi;
                    if (++g_ld == G) { g_ld = 0; p += wrap; }
                }
                add(pk.x, 0); add(pk.x, 1); add(pk.y, 0); add(pk.y, 1);
                add(pk.z, 0); add(pk.z, 1); add(pk.w, 0); add(pk.w, 1);
                if (++g == G) {
                    if (MODE == 2) s1a = __uint_as_float(acc_or);
                    if (MODE >= 3) {
                        const int j = tid & 3;
                        s2b = j == 0 ? mb[0] : j == 1 ? mb[1] : j == 2 ? mb[2] : mb[3];
                        if (MODE == 3) s2a = j == 0 ? ma[0] : j == 1 ? ma[1] : j == 2 ? ma[2] : ma[3];
                        ma = mb = f4v{0.f, 0.f, 0.f, 0.f};
                    }
                    const float m0 = s1a / k, m1 = s1b / k;
                    mean[r0 * ldo + q] = m0 + rp0;
                    mean[(r0 + 1) * ldo + q] = m1 + rp1;
                    istd[r0 * ldo + q] = rsqrtf(fmaxf(s2a / k - m0 * m0, 1e-10f));
                    istd[(r0 + 1) * ldo + q] = rsqrtf(fmaxf(s2b / k - m1 * m1, 1e-10f));
                    s1a = s1b = s2a = s2b = 0.f;
                    acc_or = 0;
                    g = 0;
                    q += GS;
                }
            }
        }
    }
    if (report) { g_clk[2] = clock64(); g_clk[3] = wall_clock64(); }
}

// As gather2 (both moments) with list entries stored as 32-bit LDS byte offsets (four per 16-byte load):
// no address instruction, twice the index bytes.
template <int D>
__global__ __launch_bounds__(GS) void gather3(const float* __restrict__ X, int64_t ldx, int nc,
                                              const uint32_t* __restrict__ idx4, int64_t ldi, int k, int64_t Q,
                                              float* __restrict__ mean, float* __restrict__ istd, int64_t ldo) {
    extern __shared__ __attribute__((aligned(16))) float xr[];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 2;
    const float rp0 = X[r0 * ldx], rp1 = X[(r0 + 1) * ldx];
    for (int c = tid; c < nc; c += GS) {
        xr[c * 2] = X[r0 * ldx + c] - rp0;
        xr[c * 2 + 1] = X[(r0 + 1) * ldx + c] - rp1;
    }
    __syncthreads();
    const char* xb = reinterpret_cast<const char*>(xr);
    const int G = k >> 2;
    const int n = (int)(Q / GS) * G;
    const uint4* p = reinterpret_cast<const uint4*>(idx4) + tid;
    const int64_t wrap = (int64_t)GS - (int64_t)G * ldi;
    int g_ld = 0;
    uint4 ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        ring[d] = *p;
        p += ldi;
        if (++g_ld == G) { g_ld = 0; p += wrap; }
    }
    float s1a = 0.f, s1b = 0.f, s2a = 0.f, s2b = 0.f;
    int g = 0;
    int64_t q = tid;
    auto add = [&](unsigned addr) {
        const float2 v = *reinterpret_cast<const float2*>(xb + addr);
        asm("v_add_f32 %0, %1, %0" : "+v"(s1a) : "v"(v.x));
        asm("v_add_f32 %0, %1, %0" : "+v"(s1b) : "v"(v.y));
        asm("v_fma_f32 %0, %1, %1, %0" : "+v"(s2a) : "v"(v.x));
        asm("v_fma_f32 %0, %1, %1, %0" : "+v"(s2b) : "v"(v.y));
    };
    for (int i = 0; i < n; i += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (i + d < n) {
                const uint4 pk = ring[d];
                if (i + d + D < n) {
                    ring[d] = *p;
                    p += ldi;
                    if (++g_ld == G) { g_ld = 0; p += wrap; }
                }
                add(pk.x); add(pk.y); add(pk.z); add(pk.w);
                if (++g == G) {
                    const float m0 = s1a / k, m1 = s1b / k;
                    mean[r0 * ldo + q] = m0 + rp0;
                    mean[(r0 + 1) * ldo + q] = m1 + rp1;
                    istd[r0 * ldo + q] = rsqrtf(fmaxf(s2a / k - m0 * m0, 1e-10f));
                    istd[(r0 + 1) * ldo + q] = rsqrtf(fmaxf(s2b / k - m1 * m1, 1e-10f));
                    s1a = s1b = s2a = s2b = 0.f;
                    g = 0;
                    q += GS;
                }
            }
        }
    }
}

static uint64_t rng_state = 88172645463325252ull;
static inline uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 11); }

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 4096, nc = 20000, k = 304;
    const int64_t Q = argc > 2 ? atoll(argv[2]) : 65536;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const double ghz = prop.clockRate / 1e6;
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, %.2f GHz nominal\n", prop.name, ncu, ghz);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* out;
    CK(hipMalloc(&out, sizeof(float) * 1024 * ncu * 4));
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_add_f32", "v_and/v_bfe", "v_lshl_add_u32",
                           "v_mad_u32_u16", "v_lshlrev_sdwa", "v_fma_mix_f32", "v_cvt_f32_u32_sdwa"};
    const int iters = 2000;
#define RATE(M)                                                                                         \
    {                                                                                                   \
        hipLaunchKernelGGL(valu_rate<M>, dim3(ncu), dim3(1024), 0, 0, out, 10);                         \
        CK(hipEventRecord(e0));                                                                         \
        hipLaunchKernelGGL(valu_rate<M>, dim3(ncu), dim3(1024), 0, 0, out, iters);                      \
        CK(hipEventRecord(e1));                                                                         \
        CK(hipEventSynchronize(e1));                                                                    \
        float ms;                                                                                       \
        CK(hipEventElapsedTime(&ms, e0, e1));                                                           \
        const double insts = 64.0 * iters * 4; /* wave instructions per SIMD (4 waves x 64 per iter) */ \
        printf("%-20s %7.3f ms  %5.2f nominal clk per wave-instruction per SIMD\n", names[M], ms,       \
               ms * 1e-3 * ghz * 1e9 / insts);                                                          \
    }
    RATE(0) RATE(1) RATE(2) RATE(3) RATE(4) RATE(5) RATE(6) RATE(7) RATE(8) RATE(9)

    // gather inputs
    const int64_t ldx = nc, ldi = Q, ldo = Q;
    std::vector<float> hx((size_t)rows * nc);
    for (auto& v : hx) v = (float)(rnd() & 0xffff) / 4096.f - 8.f;
    float *X, *mean, *istd;
    CK(hipMalloc(&X, hx.size() * 4));
    CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&mean, (size_t)rows * Q * 4));
    CK(hipMalloc(&istd, (size_t)rows * Q * 4));
    const int kg = k / 8;
    std::vector<uint16_t> hi((size_t)kg * Q * 8);
    uint16_t* idx;
    CK(hipMalloc(&idx, hi.size() * 2));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gather<0>), hipFuncAttributeMaxDynamicSharedMemorySize, nc * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gather<1>), hipFuncAttributeMaxDynamicSharedMemorySize, nc * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gather<2>), hipFuncAttributeMaxDynamicSharedMemorySize, nc * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gather<3>), hipFuncAttributeMaxDynamicSharedMemorySize, nc * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gather<4>), hipFuncAttributeMaxDynamicSharedMemorySize, nc * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gather<5>), hipFuncAttributeMaxDynamicSharedMemorySize, nc * 8));
    const char* pats[] = {"random", "bank-pair (q+j)%32", "all equal", "random, each list ordered by bank pair and rotated by lane"};
    for (int pat = 0; pat < 4; ++pat) {
        for (int64_t q = 0; q < Q; ++q)
            for (int j = 0; j < k; ++j) {
                uint32_t c;
                if (pat == 0) c = rnd() % nc;
                else if (pat == 1) c = (rnd() % (nc / 32)) * 32 + ((q + j) & 31);
                else if (pat == 2) c = 1234;
                else continue;
                hi[((size_t)(j >> 3) * ldi + q) * 8 + (j & 7)] = (uint16_t)c;
            }
        if (pat == 3) {
            std::vector<uint32_t> tmp(k), srt(k);
            for (int64_t q = 0; q < Q; ++q) {
                int cnt[33] = {0};
                for (int j = 0; j < k; ++j) { tmp[j] = rnd() % nc; cnt[(tmp[j] & 31) + 1]++; }
                for (int b = 0; b < 32; ++b) cnt[b + 1] += cnt[b];
                for (int j = 0; j < k; ++j) srt[cnt[tmp[j] & 31]++] = tmp[j];
                const int rot = (int)((q & 31) * k / 32);
                for (int j = 0; j < k; ++j)
                    hi[((size_t)(j >> 3) * ldi + q) * 8 + (j & 7)] = (uint16_t)srt[(j + rot) % k];
            }
        }
        CK(hipMemcpy(idx, hi.data(), hi.size() * 2, hipMemcpyHostToDevice));
#define GATHER(V)                                                                                              \
    {                                                                                                          \
        float best = 1e9f;                                                                                     \
        for (int rep = 0; rep < 3; ++rep) {                                                                    \
            CK(hipEventRecord(e0));                                                                            \
            hipLaunchKernelGGL(gather<V>, dim3(rows / 2), dim3(GS), nc * 8, 0, X, ldx, nc, idx, ldi, k, Q,     \
                               mean, istd, ldo);                                                               \
            CK(hipEventRecord(e1));                                                                            \
            CK(hipEventSynchronize(e1));                                                                       \
            float ms;                                                                                          \
            CK(hipEventElapsedTime(&ms, e0, e1));                                                              \
            if (ms < best) best = ms;                                                                          \
        }                                                                                                      \
        CK(hipGetLastError());                                                                                 \
        const double el = (double)rows * Q * k;                                                                \
        printf("  var %d  %8.3f ms  %6.2f elements/clk/CU  %5.2f clk per wave-entry per CU  index %5.2f TB/s\n", V, best, \
               el / (best * 1e-3) / ncu / (ghz * 1e9), 128.0 * ncu * ghz * 1e9 * best * 1e-3 / el,             \
               (rows / 2.0) * Q * kg * 16 / (best * 1e-3) / 1e12);                                             \
    }
#define GATHER2(D, M, CF)                                                                                          \
    {                                                                                                          \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gather2<D, M>),                                  \
                               hipFuncAttributeMaxDynamicSharedMemorySize, nc * 8));                           \
        float best = 1e9f;                                                                                     \
        for (int rep = 0; rep < 3; ++rep) {                                                                    \
            CK(hipEventRecord(e0));                                                                            \
            hipLaunchKernelGGL((gather2<D, M>), dim3(rows / 2), dim3(GS), nc * 8, 0, X, ldx, nc, idx, ldi, k,  \
                               Q, mean, istd, ldo, CF);                                                        \
            CK(hipEventRecord(e1));                                                                            \
            CK(hipEventSynchronize(e1));                                                                       \
            float ms;                                                                                          \
            CK(hipEventElapsedTime(&ms, e0, e1));                                                              \
            if (ms < best) best = ms;                                                                          \
        }                                                                                                      \
        CK(hipGetLastError());                                                                                 \
        const double el = (double)rows * Q * k;                                                                \
        printf("  ring D=%d mode %d confine " #CF "  %8.3f ms  %6.2f elements/clk/CU  %5.2f clk per wave-entry per CU  index %5.2f TB/s\n", D, M, best, \
               el / (best * 1e-3) / ncu / (ghz * 1e9), 128.0 * ncu * ghz * 1e9 * best * 1e-3 / el,             \
               (rows / 2.0) * Q * kg * 16 / (best * 1e-3) / 1e12);                                             \
        unsigned long long clk[4];                                                                             \
        CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk)));                                          \
        printf("    shader clock under this load: %.0f MHz (walk of one workgroup: %.1f us)\n",                \
               (double)(clk[2] - clk[0]) / ((double)(clk[3] - clk[1]) / 100.0), (double)(clk[3] - clk[1]) / 100.0); \
    }
        printf("pattern: %s (%d rows x %lld lists x %d)\n", pats[pat], rows, (long long)Q, k);
        GATHER(0)
        GATHER2(4, 0, 0) GATHER2(8, 0, 0) GATHER2(8, 1, 0) GATHER2(8, 2, 0) GATHER2(4, 3, 0) GATHER2(4, 4, 0)
        {   // 32-bit byte-offset lists
            std::vector<uint32_t> h4((size_t)(k / 4) * Q * 4);
            for (int64_t q = 0; q < Q; ++q)
                for (int j = 0; j < k; ++j)
                    h4[((size_t)(j >> 2) * ldi + q) * 4 + (j & 3)] = 8u * hi[((size_t)(j >> 3) * ldi + q) * 8 + (j & 7)];
            uint32_t* idx4;
            CK(hipMalloc(&idx4, h4.size() * 4));
            CK(hipMemcpy(idx4, h4.data(), h4.size() * 4, hipMemcpyHostToDevice));
#define GATHER3(D)                                                                                             \
    {                                                                                                          \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gather3<D>),                                     \
                               hipFuncAttributeMaxDynamicSharedMemorySize, nc * 8));                           \
        float best = 1e9f;                                                                                     \
        for (int rep = 0; rep < 3; ++rep) {                                                                    \
            CK(hipEventRecord(e0));                                                                            \
            hipLaunchKernelGGL((gather3<D>), dim3(rows / 2), dim3(GS), nc * 8, 0, X, ldx, nc, idx4, ldi, k, Q, \
                               mean, istd, ldo);                                                               \
            CK(hipEventRecord(e1));                                                                            \
            CK(hipEventSynchronize(e1));                                                                       \
            float ms;                                                                                          \
            CK(hipEventElapsedTime(&ms, e0, e1));                                                              \
            if (ms < best) best = ms;                                                                          \
        }                                                                                                      \
        CK(hipGetLastError());                                                                                 \
        const double el = (double)rows * Q * k;                                                                \
        printf("  32-bit offsets, ring D=%d  %8.3f ms  %5.2f clk per wave-entry per CU\n", D, best,             \
               128.0 * ncu * ghz * 1e9 * best * 1e-3 / el);                                                    \
    }
            GATHER3(4) GATHER3(8)
            CK(hipFree(idx4));
        }
    }
    return 0;
}

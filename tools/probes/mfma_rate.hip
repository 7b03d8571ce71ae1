// Sustained matrix-core rate under the chip's power limit for the MFMA shapes a split-precision
// score kernel could use.  Operands live in registers and are random; every wave runs ITER rounds
// of 12 independent-accumulator MFMAs.  Prints achieved ops/s and the implied time for the
// 100k x 100k x 160 score matrix with 6 (bf16) or 10 (i8) partial products.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

__device__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ unsigned rnd_bf16_pair(unsigned s) {  // two bf16 in [-2, 2) with random mantissas
    unsigned h = hash(s);
    unsigned a = (h & 0x807f) | 0x3f80, b = ((h >> 16) & 0x807f) | 0x3f80;
    return a | (b << 16);
}

__device__ unsigned rnd_f16_pair(unsigned s) {  // two fp16 in [-2, 2) with random 10-bit mantissas
    unsigned h = hash(s);
    unsigned a = (h & 0x83ff) | 0x3c00, b = ((h >> 16) & 0x83ff) | 0x3c00;
    return a | (b << 16);
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    uint4 av[6], bv[6];
    for (int i = 0; i < 6; ++i) {
        if (MODE == 4) {
            av[i] = make_uint4(rnd_f16_pair(tid * 97 + i), rnd_f16_pair(tid * 89 + i + 7), rnd_f16_pair(tid * 83 + i + 11), rnd_f16_pair(tid * 79 + i + 13));
            bv[i] = make_uint4(rnd_f16_pair(tid * 73 + i), rnd_f16_pair(tid * 71 + i + 17), rnd_f16_pair(tid * 67 + i + 19), rnd_f16_pair(tid * 61 + i + 23));
        } else if (MODE < 2) {
            av[i] = make_uint4(rnd_bf16_pair(tid * 97 + i), rnd_bf16_pair(tid * 89 + i + 7), rnd_bf16_pair(tid * 83 + i + 11), rnd_bf16_pair(tid * 79 + i + 13));
            bv[i] = make_uint4(rnd_bf16_pair(tid * 73 + i), rnd_bf16_pair(tid * 71 + i + 17), rnd_bf16_pair(tid * 67 + i + 19), rnd_bf16_pair(tid * 61 + i + 23));
        } else {
            av[i] = make_uint4(hash(tid * 97 + i), hash(tid * 89 + i + 7), hash(tid * 83 + i + 11), hash(tid * 79 + i + 13));
            bv[i] = make_uint4(hash(tid * 73 + i), hash(tid * 71 + i + 17), hash(tid * 67 + i + 19), hash(tid * 61 + i + 23));
        }
    }
    float r = 0.f;
    if (MODE == 0) {  // bf16 32x32x16, 3 accumulators
        f32x16 c0 = {}, c1 = {}, c2 = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[i]), __builtin_bit_cast(bf16x8, bv[i]), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[i]), __builtin_bit_cast(bf16x8, bv[5 - i]), c1, 0, 0, 0);
                if (i & 1) c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[5 - i]), __builtin_bit_cast(bf16x8, bv[i]), c2, 0, 0, 0);
            }
        }
        for (int e = 0; e < 16; ++e) r += c0[e] + c1[e] + c2[e];
    } else if (MODE == 1) {  // bf16 16x16x32, 12 accumulators
        f32x4 c[12] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                c[2 * i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[i]), __builtin_bit_cast(bf16x8, bv[i]), c[2 * i], 0, 0, 0);
                c[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[i]), __builtin_bit_cast(bf16x8, bv[5 - i]), c[2 * i + 1], 0, 0, 0);
                if (i & 1) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[5 - i]), __builtin_bit_cast(bf16x8, bv[i]), c[i], 0, 0, 0);
            }
        }
        for (int i = 0; i < 12; ++i) for (int e = 0; e < 4; ++e) r += c[i][e];
    } else if (MODE == 2) {  // i8 32x32x32
        i32x16 c0 = {}, c1 = {}, c2 = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4v, av[i]), __builtin_bit_cast(i32x4v, bv[i]), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4v, av[i]), __builtin_bit_cast(i32x4v, bv[5 - i]), c1, 0, 0, 0);
                if (i & 1) c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4v, av[5 - i]), __builtin_bit_cast(i32x4v, bv[i]), c2, 0, 0, 0);
            }
        }
        for (int e = 0; e < 16; ++e) r += (float)(c0[e] + c1[e] + c2[e]);
    } else if (MODE == 4) {  // f16 16x16x32, 12 accumulators
        f32x4 c[12] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                c[2 * i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[i]), __builtin_bit_cast(f16x8, bv[i]), c[2 * i], 0, 0, 0);
                c[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[i]), __builtin_bit_cast(f16x8, bv[5 - i]), c[2 * i + 1], 0, 0, 0);
                if (i & 1) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[5 - i]), __builtin_bit_cast(f16x8, bv[i]), c[i], 0, 0, 0);
            }
        }
        for (int i = 0; i < 12; ++i) for (int e = 0; e < 4; ++e) r += c[i][e];
    } else {  // i8 16x16x64
        i32x4 c[12] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                c[2 * i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4v, av[i]), __builtin_bit_cast(i32x4v, bv[i]), c[2 * i], 0, 0, 0);
                c[2 * i + 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4v, av[i]), __builtin_bit_cast(i32x4v, bv[5 - i]), c[2 * i + 1], 0, 0, 0);
                if (i & 1) c[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4v, av[5 - i]), __builtin_bit_cast(i32x4v, bv[i]), c[i], 0, 0, 0);
            }
        }
        for (int i = 0; i < 12; ++i) for (int e = 0; e < 4; ++e) r += (float)c[i][e];
    }
    if (r == 12345.678f) out[tid] = r;
}

template <int MODE>
void run(const char* name, double ops_per_mfma, int mfma_per_iter, double job_mfma_ops) {
    float* out;
    hipMalloc(&out, 4 << 20);
    const int blocks = 256 * 2 * 4, iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(out, iters / 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        probe<MODE><<<blocks, 256>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double ops = (double)blocks * 4 * iters * mfma_per_iter * ops_per_mfma;
    const double rate = ops / (best * 1e-3);
    printf("%-16s %8.3f ms  %7.1f Tops/s  -> 100k x 100k x 160 job: %.2f ms\n", name, best, rate / 1e12, job_mfma_ops / rate * 1e3);
    hipFree(out);
}

int main() {
    const double mk = 1e10 * 160 * 2;  // one full product over K = 160
    run<0>("bf16 32x32x16", 32768.0, 15, 6 * mk);
    run<1>("bf16 16x16x32", 16384.0, 30, 6 * mk);
    run<4>("f16 16x16x32 x4", 16384.0, 30, 4 * mk);
    run<2>("i8 32x32x32", 65536.0, 15, 10 * mk);
    run<3>("i8 16x16x64", 32768.0, 30, 10 * mk);
    run<0>("bf16 32x32x16", 32768.0, 15, 6 * mk);
    return 0;
}

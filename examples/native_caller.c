/*
 * A caller of libhypmi.so that is neither Python nor C++: plain C99 against include/hypmi.h and the
 * HIP runtime's C API.  It does what a native host of the reference's scoring path would do:
 * builds a folded model (one length-normalised pre-processing segment and the two K-dimensional
 * projections of a PLDA model, hyperion/pdfs/plda/splda.py:297-315), creates a plan, runs the
 * prologue for an enrol and a test set, converts both to fp16 pairs, scores every trial and checks
 * the matrix against the same closed form evaluated in double precision on the host:
 *     y = sqrt(d) (x A + b) / |x A + b|,  g = y At + bt,  h = y An + bn,
 *     q = (|g|^2 - |h|^2) / 2,  S[i, j] = g1_i . g2_j + q1_i + q2_j + c.
 * Build: make -C examples (gcc; links libhypmi.so and libamdhip64.so).  Exit status 0 = agreement.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "hypmi.h"

#define D_IN 24
#define D_MID 16
#define KDIM 5
#define N1 70
#define N2 131

#define CHECK_HIP(call)                                                                 \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                  \
            return 2;                                                                   \
        }                                                                               \
    } while (0)
#define CHECK_HYPMI(call)                                                               \
    do {                                                                                \
        if ((call) != HYPMI_OK) {                                                       \
            fprintf(stderr, "%s: %s\n", #call, hypmi_last_error());                     \
            return 3;                                                                   \
        }                                                                               \
    } while (0)

static unsigned long long rng_state = 88172645463325252ULL;
static double uniform(void) { /* xorshift64: a fixed input set without any library */
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0 - 0.5;
}

static void host_packed(const float* x, int n, const double* A, const double* b, const double* At,
                        const double* bt, const double* An, const double* bn, double* g, double* q) {
    for (int i = 0; i < n; ++i) {
        double u[D_MID], norm = 0.0;
        for (int j = 0; j < D_MID; ++j) {
            u[j] = b[j];
            for (int k = 0; k < D_IN; ++k) u[j] += (double)x[i * D_IN + k] * A[k * D_MID + j];
            norm += u[j] * u[j];
        }
        const double s = sqrt((double)D_MID) / (sqrt(norm) + 1e-10);
        double gg = 0.0, hh = 0.0;
        for (int c = 0; c < KDIM; ++c) {
            double gv = bt[c], hv = bn[c];
            for (int j = 0; j < D_MID; ++j) {
                gv += s * u[j] * At[j * KDIM + c];
                hv += s * u[j] * An[j * KDIM + c];
            }
            g[i * KDIM + c] = gv;
            gg += gv * gv;
            hh += hv * hv;
        }
        q[i] = 0.5 * (gg - hh);
    }
}

int main(void) {
    static double A[D_IN * D_MID], b[D_MID], At[D_MID * KDIM], bt[KDIM], An[D_MID * KDIM], bn[KDIM];
    static float x1[N1 * D_IN], x2[N2 * D_IN];
    const double c = -1.75;
    for (int i = 0; i < D_IN * D_MID; ++i) A[i] = uniform();
    for (int i = 0; i < D_MID; ++i) b[i] = 0.3 * uniform();
    for (int i = 0; i < D_MID * KDIM; ++i) {
        At[i] = uniform();
        An[i] = 0.8 * At[i] + 0.2 * uniform();
    }
    for (int i = 0; i < KDIM; ++i) {
        bt[i] = 0.1 * uniform();
        bn[i] = 0.1 * uniform();
    }
    for (int i = 0; i < N1 * D_IN; ++i) x1[i] = (float)(2.0 * uniform());
    for (int i = 0; i < N2 * D_IN; ++i) x2[i] = (float)(2.0 * uniform());

    int ndev = 0;
    CHECK_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) {
        fprintf(stderr, "no HIP device\n");
        return 2;
    }
    CHECK_HIP(hipSetDevice(0));

    hypmi_plan_desc desc;
    hypmi_plan* plan = NULL;
    desc.d_in = D_IN;
    desc.n_seg = 1;
    desc.seg[0].A = A;
    desc.seg[0].b = b;
    desc.seg[0].d_out = D_MID;
    desc.seg[0].lnorm = 1;
    desc.At = At;
    desc.bt = bt;
    desc.An = An;
    desc.bn = bn;
    desc.K = KDIM;
    desc.c = c;
    desc.shift_t = desc.scale_t = desc.shift_n = desc.scale_n = NULL; /* two-projection form */
    CHECK_HYPMI(hypmi_plan_create(&desc, &plan));
    const int kp = hypmi_plan_kp(plan);
    if (hypmi_plan_k(plan) != KDIM || hypmi_plan_d_in(plan) != D_IN || kp < KDIM + 2 || kp % 8 != 0) {
        fprintf(stderr, "plan reports KDIM %d d_in %d kp %d\n", hypmi_plan_k(plan), hypmi_plan_d_in(plan), kp);
        return 4;
    }

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    float *dx1, *dx2, *G1, *G2, *S;
    void *ws, *H1, *H2;
    const int64_t ws_bytes = hypmi_prologue_workspace_bytes(plan, N2);
    const int64_t lds = (N2 + 31) / 32 * 32;
    CHECK_HIP(hipMalloc((void**)&dx1, sizeof(x1)));
    CHECK_HIP(hipMalloc((void**)&dx2, sizeof(x2)));
    CHECK_HIP(hipMalloc((void**)&G1, (size_t)N1 * kp * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&G2, (size_t)N2 * kp * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&S, (size_t)N1 * lds * sizeof(float)));
    CHECK_HIP(hipMalloc(&ws, ws_bytes > 0 ? (size_t)ws_bytes : 16));
    CHECK_HIP(hipMemcpyAsync(dx1, x1, sizeof(x1), hipMemcpyHostToDevice, stream));
    CHECK_HIP(hipMemcpyAsync(dx2, x2, sizeof(x2), hipMemcpyHostToDevice, stream));
    CHECK_HYPMI(hypmi_prologue(plan, dx1, N1, D_IN, HYPMI_SIDE_ENROL, G1, ws, ws_bytes, stream));
    CHECK_HYPMI(hypmi_prologue(plan, dx2, N2, D_IN, HYPMI_SIDE_TEST, G2, ws, ws_bytes, stream));

    static float got_pair[N1 * N2], got_f32[N1 * N2];
    /* fp32 matrix cores */
    CHECK_HYPMI(hypmi_score(G1, N1, G2, N2, kp, S, lds, stream));
    CHECK_HIP(hipMemcpy2DAsync(got_f32, N2 * sizeof(float), S, lds * sizeof(float), N2 * sizeof(float), N1,
                               hipMemcpyDeviceToHost, stream));
    /* fp16 pairs, the default path of the Python host */
    if (hypmi_pair_row_bytes(kp) == 0) {
        fprintf(stderr, "no pair kernel for kp %d\n", kp);
        return 4;
    }
    CHECK_HIP(hipMalloc(&H1, (size_t)hypmi_pair_bytes(kp, N1)));
    CHECK_HIP(hipMalloc(&H2, (size_t)hypmi_pair_bytes(kp, N2)));
    CHECK_HYPMI(hypmi_pair_rows(G1, N1, kp, H1, stream));
    CHECK_HYPMI(hypmi_pair_rows(G2, N2, kp, H2, stream));
    CHECK_HYPMI(hypmi_score_pair(H1, N1, H2, N2, kp, S, lds, stream));
    CHECK_HIP(hipMemcpy2DAsync(got_pair, N2 * sizeof(float), S, lds * sizeof(float), N2 * sizeof(float), N1,
                               hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));

    static double g1[N1 * KDIM], q1[N1], g2[N2 * KDIM], q2[N2];
    host_packed(x1, N1, A, b, At, bt, An, bn, g1, q1);
    host_packed(x2, N2, A, b, At, bt, An, bn, g2, q2);
    double worst_pair = 0.0, worst_f32 = 0.0;
    for (int i = 0; i < N1; ++i)
        for (int j = 0; j < N2; ++j) {
            double s = q1[i] + q2[j] + c;
            for (int k = 0; k < KDIM; ++k) s += g1[i * KDIM + k] * g2[j * KDIM + k];
            const double scale = fabs(s) > 1.0 ? fabs(s) : 1.0;
            const double e1 = fabs((double)got_pair[i * N2 + j] - s) / scale;
            const double e2 = fabs((double)got_f32[i * N2 + j] - s) / scale;
            if (e1 > worst_pair) worst_pair = e1;
            if (e2 > worst_f32) worst_f32 = e2;
        }
    /* an argument error must come back as a code and a message, not as a crash */
    const int rc = hypmi_score(G1, N1, G2, N2, kp, S, N2 - 1, stream);
    printf("native caller: %d x %d trials, kp %d, max error pairs %.3g fp32 %.3g (relative to max(|s|, 1)); "
           "bad ld -> %d \"%s\"\n", N1, N2, kp, worst_pair, worst_f32, rc, hypmi_last_error());
    hypmi_plan_destroy(plan);
    hipFree(dx1); hipFree(dx2); hipFree(G1); hipFree(G2); hipFree(S); hipFree(ws); hipFree(H1); hipFree(H2);
    hipStreamDestroy(stream);
    return (worst_pair <= 1e-4 && worst_f32 <= 1e-4 && rc == HYPMI_EINVAL) ? 0 : 1;
}

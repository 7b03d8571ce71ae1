// Device-resident folded model shared by the fp32 prologue (prologue.hip) and the fp64 refinement
// path (refine.hip).
#pragma once
#include "common.h"

struct Segment {
    float* Wt = nullptr;      // [d_out x ldw] transposed weight, zero padded (nullptr = identity)
    float* bias = nullptr;    // [d_out] or nullptr
    double* Wt64 = nullptr;   // the same in fp64
    double* bias64 = nullptr;
    int64_t ldw = 0;
    int d_in = 0, d_out = 0;
    bool lnorm = false;

    void release() {
        if (Wt) (void)hipFree(Wt);
        if (bias) (void)hipFree(bias);
        if (Wt64) (void)hipFree(Wt64);
        if (bias64) (void)hipFree(bias64);
        Wt = bias = nullptr;
        Wt64 = bias64 = nullptr;
    }
};

// Last length-normalised segment folded with the PLDA projection (one kernel, see prologue.hip):
//   u = x A1 + b1 (n_u columns, only its row norm is needed), v = x (A1 A2) + b1 A2 (n_v = 2K columns),
//   z = sqrt(n_norm) / (|u| + 1e-10) * v + b2.   Wcat = [A1 Q | A1 A2]^T, [n_u + n_v] x ldw, with Q
//   orthogonal and chosen so that A1 Q is lower trapezoidal (|u Q| = |u|); bias_u = b1 Q.
struct FusedTail {
    bool enabled = false;
    float* Wcat = nullptr;
    float* bias_u = nullptr;    // b1 Q, n_u entries (b1 itself without a matrix; nullptr = 0)
    float* bias_vin = nullptr;  // b1 A2, n_v entries (nullptr = 0)
    int64_t ldw = 0;
    int n_u = 0, n_v = 0, d_in = 0;
    float sqrt_dn = 0.f;        // sqrt of the normalised dimension
    bool norm_from_input = false;  // identity A1: |u| is the norm of the input row (+ b1)

    void release() {
        if (Wcat) (void)hipFree(Wcat);
        if (bias_u) (void)hipFree(bias_u);
        if (bias_vin) (void)hipFree(bias_vin);
        Wcat = bias_u = bias_vin = nullptr;
        enabled = false;
    }
};

struct hypmi_plan {
    int d_in = 0, n_seg = 0, d_last = 0, K = 0, KP = 0;
    Segment seg[HYPMI_MAX_SEG];
    Segment fin;  // [At | An]^T, 2K outputs (shared-basis plans: the float64 path only)
    // shared-basis form: one K-column projection s, then g_k = (s_k + shift_t[k]) scale_t[k] and
    // h_k = (s_k + shift_n[k]) scale_n[k]
    bool shared = false;
    Segment sfin;  // K outputs
    float* col_map = nullptr;  // shift_t | scale_t | shift_n | scale_n, K floats each
    FusedTail fused;
    double c = 0.0;
    int max_dim = 0;
};

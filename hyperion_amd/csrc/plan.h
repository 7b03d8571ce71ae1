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

struct hypmi_plan {
    int d_in = 0, n_seg = 0, d_last = 0, K = 0, KP = 0;
    Segment seg[HYPMI_MAX_SEG];
    Segment fin;  // [At | An]^T, 2K outputs
    double c = 0.0;
    int max_dim = 0;
};

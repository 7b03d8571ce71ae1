/*
 * hypmi — C ABI of the MI355X (gfx950) PLDA back-end scoring library.
 *
 * The reference is pure Python/numpy and has no FFI layer of its own; each entry point below names
 * the numpy expression(s) of the reference it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - every function returns 0 on success or a negative HYPMI_E* code; hypmi_last_error() gives a
 *     thread-local message for the last failure on the calling thread;
 *   - all data pointers are DEVICE pointers owned by the caller (the Python shim hands over
 *     torch-allocated storage); the library owns only opaque hypmi_plan objects; the hypmi_ark_*
 *     loader functions are the exception: they run on the host and take host pointers;
 *   - all matrices are row-major fp32 unless stated; leading dimensions are in elements;
 *     row starts must be 16-byte aligned (base pointer 16-byte aligned, ld a multiple of 4);
 *   - calls are stream-ordered on `stream` (a hipStream_t passed as void*) and do not block the
 *     host; plan creation/destruction are the only calls that allocate or synchronise;
 *   - there is no CPU fallback: without a HIP device every compute entry point fails.
 */
#ifndef HYPMI_H
#define HYPMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HYPMI_OK 0
#define HYPMI_EINVAL (-1)  /* bad argument (shape, alignment, null pointer) */
#define HYPMI_EHIP (-2)    /* a HIP runtime call failed                     */
#define HYPMI_ENOMEM (-3)  /* device allocation failed                      */
#define HYPMI_EIO (-4)     /* a file could not be read as asked             */

#define HYPMI_MAX_SEG 4

/* Which side of a trial a packed vector is for: enrol-side rows carry [g, q, 1], test-side rows
 * carry [g, 1, q + c], so that a plain dot product of the two is the log-likelihood ratio. */
#define HYPMI_SIDE_ENROL 0
#define HYPMI_SIDE_TEST 1

typedef struct hypmi_plan hypmi_plan;

/* One pre-processing segment: y = (x A + b), then optionally y <- sqrt(d_out) y / (|y| + 1e-10).
 * A is [d_in x d_out] row-major fp64 on the HOST (NULL = identity, d_out must equal d_in),
 * b is [d_out] fp64 on the host (NULL = 0). */
typedef struct {
    const double* A;
    const double* b;
    int32_t d_out;
    int32_t lnorm;
} hypmi_segment;

/* Folded model: a chain of segments (hyperion/transforms/transform_list.py:41-44 with
 * cent_whiten.py:25-33, lnorm.py:14-17, lda.py:31-34, pca.py:39-42 folded by the host) followed by
 * the two K-dimensional projections of the PLDA model,
 *   g = y At + bt   (the reference's gamma_tar, e.g. hyperion/pdfs/plda/splda.py:306-307)
 *   h = y An + bn   (gamma_non, splda.py:300-301)
 * and the additive constant c (splda.py:314-315: (2 logLnon - logLtar)/2).  At/An are
 * [d_last x K] row-major fp64 host arrays, bt/bn [K] (NULL = 0).  The host computes them in fp64
 * (Cholesky and triangular solves act on K x K matrices only).
 *
 * Shared-basis form (scale_t != NULL): in the basis that diagonalises the model's L matrices
 * (I + n V W V^T, or B + n W) the two projections are one projection s = y At + bt followed by a
 * map per column,
 *   g_k = (s_k + shift_t[k]) scale_t[k],   h_k = (s_k + shift_n[k]) scale_n[k]
 * (the by-the-book scoring of splda.py:334-375 for one segment a side), so the device projects onto
 * K columns instead of 2K.  An/bn are then ignored and may be NULL; the four vectors are [K] fp64
 * host arrays.  The scores are those of the two-projection form: g differs by a rotation. */
typedef struct {
    int32_t d_in;
    int32_t n_seg;
    hypmi_segment seg[HYPMI_MAX_SEG];
    const double* At;
    const double* bt;
    const double* An;
    const double* bn;
    int32_t K;
    double c;
    const double* shift_t;
    const double* scale_t;
    const double* shift_n;
    const double* scale_n;
} hypmi_plan_desc;

const char* hypmi_last_error(void);
int hypmi_version(void);

/* Create / destroy a device-resident plan on the current HIP device. */
int hypmi_plan_create(const hypmi_plan_desc* desc, hypmi_plan** out);
void hypmi_plan_destroy(hypmi_plan* plan);

/* Width (in floats, a multiple of 8) of one packed row produced by hypmi_prologue for this plan:
 * K + 2 rounded up to the width of the score kernel that will be used. */
int hypmi_plan_kp(const hypmi_plan* plan);
int hypmi_plan_k(const hypmi_plan* plan);
int hypmi_plan_d_in(const hypmi_plan* plan);
int hypmi_plan_d_out(const hypmi_plan* plan); /* dimension after the last segment */

/* Packed row width for a K-dimensional projection that is packed outside a plan (book scoring). */
int hypmi_packed_width(int K);

/* Bytes of scratch hypmi_prologue / hypmi_transform need for n rows. */
int64_t hypmi_prologue_workspace_bytes(const hypmi_plan* plan, int64_t n);

/* Pre-processing chain + PLDA projection for n embeddings.
 * Replaces TransformList.predict followed by the `np.dot(x - mu, WV)` / mult_icholL* / Q* lines of
 * llr_1vs1 (hyperion/pdfs/plda/splda.py:297-310, frplda.py:278-292, plda.py:453-466).
 * x: [n x d_in], ldx >= d_in rounded up to 4, pad columns zero.  G: [n x KP] packed rows. */
int hypmi_prologue(const hypmi_plan* plan, const float* x, int64_t n, int64_t ldx, int side,
                   float* G, void* workspace, int64_t workspace_bytes, void* stream);

/* hypmi_prologue followed by hypmi_pair_rows as ONE kernel per call for plans whose last
 * length-normalised step is fused with a shared-basis projection (hypmi_prologue_pairs_supported):
 * the per-column maps, q, the packed row and its conversion to scaled fp16 pairs happen on the
 * workgroup's LDS copy of the projected rows, which never reach memory.  Same reference lines as
 * hypmi_prologue.  H: hypmi_pair_bytes(kp, n) bytes; H_other (optional, same size) receives the same
 * vectors packed for the OTHER side of a trial (adaptive S-norm scores every test vector both as a
 * row of the cohort panel and as a column of the trial matrix: one pass over the chain instead of
 * two); G (optional, [n x kp]) also receives the packed fp32 rows of `side`, bit-identical to
 * hypmi_prologue's; workspace as for hypmi_prologue. */
int hypmi_prologue_pairs_supported(const hypmi_plan* plan);
int hypmi_prologue_pairs(const hypmi_plan* plan, const float* x, int64_t n, int64_t ldx, int side, void* H,
                         void* H_other, float* G, void* workspace, int64_t workspace_bytes, void* stream);
/* The same with one more optional result: bnd [n x 2] floats (|g|, |q|) of side `side` per row, the
 * two terms a row contributes to the fp32 error bound of its scores (what hypmi_prologue_f64 returns
 * as `bnd`, here from the fp32 rows) -- the input of hypmi_topk_rows mode 3. */
int hypmi_prologue_pairs_bounds(const hypmi_plan* plan, const float* x, int64_t n, int64_t ldx, int side,
                                void* H, void* H_other, float* G, float* bnd, void* workspace,
                                int64_t workspace_bytes, void* stream);

/* hypmi_prologue for callers that cannot allocate device memory or pass a workspace (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc): the intermediate rows live
 * in a 16 MiB array inside the library's device image, one per GPU, held until the process ends.
 * No counterpart in the reference.  Same arguments, result and stream ordering as hypmi_prologue;
 * n may not exceed hypmi_prologue_noalloc_max_rows(plan) (3167 rows for a 512 -> 512 -> 200 chain
 * with K = 150), larger n is refused with an error.  It calls no hipMalloc / hipFree.  The array's
 * device address is resolved by hypmi_plan_create on the device that is current there (once per
 * device); a call on another device than the plan's resolves it on first use, which must then
 * happen outside a stream capture.  All calls on
 * one GPU share that GPU's array: calls that may overlap there (different streams or threads)
 * race; keep them on one stream or order them yourself. */
int64_t hypmi_prologue_noalloc_max_rows(const hypmi_plan* plan);
int hypmi_prologue_noalloc(const hypmi_plan* plan, const float* x, int64_t n, int64_t ldx, int side,
                           float* G, void* stream);

/* Only the segments (TransformList.predict, transform_list.py:41-44): y is [n x d_out]. */
int hypmi_transform(const hypmi_plan* plan, const float* x, int64_t n, int64_t ldx, float* y,
                    int64_t ldy, void* workspace, int64_t workspace_bytes, void* stream);

/* Trial score matrix S[i,j] = <G1[i,:], G2[j,:]>  (= g1.g2 + q1 + q2 + c).
 * Replaces `2*np.dot(gamma_tar_1, gamma_tar_2.T)` and the three broadcast passes of
 * hyperion/pdfs/plda/splda.py:312-315 (frplda.py:294-302, plda.py:468-471).
 * G1: [n1 x kp] enrol-side packed rows, G2: [n2 x kp] test-side packed rows, S: [n1 x n2], lds>=n2.
 */
int hypmi_score(const float* G1, int64_t n1, const float* G2, int64_t n2, int kp, float* S,
                int64_t lds, void* stream);

/* The same score matrix (same reference lines as hypmi_score) from split rows, on the bf16 matrix
 * cores: each fp32 of a packed row is stored as the exact sum of three bf16 numbers and the six
 * largest of the nine partial products are accumulated in fp32, which keeps the accuracy of an
 * fp32 product chain at 2.7 times its arithmetic rate.
 * hypmi_split_row_bytes: bytes of one split row for packed width kp, 0 when kp has no split kernel
 * (then hypmi_score is the only path).  hypmi_split_bytes: size of a buffer of n split rows (the
 * rows plus 1 KiB that the kernel may read but never uses).  hypmi_split_rows: G [n x kp] -> H;
 * either side's packed rows may be converted, the format is the same.  hypmi_score_split: S as
 * above, H1 / H2 in buffers of at least hypmi_split_bytes. */
int64_t hypmi_split_row_bytes(int kp);
int64_t hypmi_split_bytes(int kp, int64_t n);
int hypmi_split_rows(const float* G, int64_t n, int kp, void* H, void* stream);
int hypmi_score_split(const void* H1, int64_t n1, const void* H2, int64_t n2, int kp, float* S,
                      int64_t lds, void* stream);

/* The same again with fp16 PAIRS (the default path of the Python host): every row is scaled by a
 * power of two (largest magnitude in [2^14, 2^15)) and each scaled value stored as x0 + x1 in fp16
 * (|x - x0 - x1| <= 2^-22 |x|); three of the four partial products run on the fp16 matrix cores with
 * fp32 accumulation and the inverse scales, kept in the padding slot of each row, are applied when
 * a tile is summed.  Closer to float64 than an fp32 product chain on the same rows, half the matrix
 * instructions and two thirds of the operand bytes of the bf16 triples.  Same calling sequence:
 * hypmi_pair_row_bytes (0: no kernel for this width), hypmi_pair_bytes, hypmi_pair_rows,
 * hypmi_score_pair.  The two formats cannot be mixed in one product. */
int64_t hypmi_pair_row_bytes(int kp);
int64_t hypmi_pair_bytes(int kp, int64_t n);
int hypmi_pair_rows(const float* G, int64_t n, int kp, void* H, void* stream);
int hypmi_score_pair(const void* H1, int64_t n1, const void* H2, int64_t n2, int kp, float* S,
                     int64_t lds, void* stream);

/* Generic y = x Wt^T + bias (one np.dot(x - mu, T) of the reference, with -mu T in bias).
 * Wt is [d_out x ldw] (the transposed weight, rows padded with zeros to ldw). */
int hypmi_affine(const float* x, int64_t n, int64_t ldx, int d_in, const float* Wt, int64_t ldw,
                 const float* bias, int d_out, float* y, int64_t ldy, void* stream);

/* In-place length normalisation of rows: y <- sqrt(d) y / (|y| + 1e-10)  (lnorm.py:16-17). */
int hypmi_lnorm_rows(float* y, int64_t n, int64_t ldy, int d, void* stream);

/* ---- double-precision refinement ------------------------------------------------------------ */

/* fp64 embedding prologue (same plan, fp64 weights, f64 matrix cores): the float64 arithmetic of
 * hyperion/pdfs/plda/splda.py:297-310 without the fp32 rounding of the fast path.
 * x: [n x d_in] fp64.  G64: [n x ldg] rows [g(0..K-1), q (+ c on the test side), |g|], ldg >= K + 2.
 * bnd: [n x 2] floats (|g|, |q|) used for error bounds. */
int64_t hypmi_prologue_f64_workspace_bytes(const hypmi_plan* plan, int64_t n);
int hypmi_prologue_f64(const hypmi_plan* plan, const double* x, int64_t n, int64_t ldx, int side,
                       double* G64, int64_t ldg, float* bnd, void* workspace, int64_t workspace_bytes,
                       void* stream);

/* Replace by their fp64 value the scores whose fp32 error bound is not below rel_tol * |s|:
 *   err_coeff * (|g1_i| |g2_j| + |q1_i| + |q2_j|) > rel_tol * |S[i,j]|
 * (scores that cancel to ~0; with err_coeff = 8 * 2^-24 about 1-2 % of a calibrated back-end's
 * non-target scores).  counter (optional, device) accumulates the number of refined entries. */
int hypmi_refine_scores(float* S, int64_t lds, int64_t n1, int64_t n2, const double* G1_64,
                        const double* G2_64, int64_t ldg, int K, const float* bnd1, const float* bnd2,
                        float rel_tol, float err_coeff, unsigned long long* counter, void* stream);

/* ---- multi-session enrolment (hyperion/pdfs/plda/plda_base.py:249-351) ---------------------- */

/* Class pooling: out[g,:] = sum (mean != 0: average) of x[order[r],:], r in [offsets[g], offsets[g+1]).
 * Replaces the per-class boolean-mask loop of compute_stats_hard (plda_base.py:259-266) and the
 * F / N divisions of llr_NvsM_vavg / _savg (plda_base.py:307-308, 318-321).  `order` (int32) lists
 * row numbers grouped by class, `offsets` (int64, n_groups + 1) delimits the groups; both are device
 * arrays.  counts (optional) receives the class sizes as floats.  An empty class averages to NaN,
 * as F / N does for N = 0. */
int hypmi_pool_rows(const float* x, int64_t ldx, int d, const int32_t* order, const int64_t* offsets,
                    int64_t n_groups, int mean, float* out, int64_t ldo, float* counts, void* stream);

/* x[i,:] /= denom[i] in place: class sums to class means, F / N (plda_base.py:307-308). */
int hypmi_div_rows(float* x, int64_t ldx, int64_t n, int d, const float* denom, void* stream);

/* By-the-book packing of one (own count, total count) group (splda.py:334-375, frplda.py:320-362):
 * from eigen-basis statistics s[n x K], g_k = (s_k + shift_t[k]) scale_t[k],
 * h_k = (s_k + shift_n[k]) scale_n[k], q = (|g|^2 - |h|^2)/2, packed as in hypmi_prologue.
 * rows (optional int32[n]) selects which rows of s are packed. */
int hypmi_book_pack(const float* s, int64_t lds, int K, const int32_t* rows, int64_t n,
                    const float* shift_t, const float* scale_t, const float* shift_n,
                    const float* scale_n, int side, double c, float* G, int kp, void* stream);

/* The same from float64 statistics and float64 maps (same reference lines): for classes of many
 * segments |g|^2 - |h|^2 cancels to about 1 / count of either term, which needs the maps in double;
 * q is taken from the double g and h, g is rounded to fp32 once. */
int hypmi_book_pack_f64(const double* s, int64_t lds, int K, const int32_t* rows, int64_t n,
                        const double* shift_t, const double* scale_t, const double* shift_n,
                        const double* scale_n, int side, double c, float* G, int kp, void* stream);

/* hypmi_book_pack_f64 that also writes the rows hypmi_refine_scores works from (same reference lines,
 * kept in float64 as splda.py:334-375 computes them): G64[n x ldg] = [g (unrounded), q (+ c on the test
 * side) from the unrounded g, |g|], ldg >= K + 2, and bnd[n x 2] = (|g|, |q|) as floats.  With G64 ==
 * NULL it is hypmi_book_pack_f64. */
int hypmi_book_pack_f64_rows(const double* s, int64_t lds, int K, const int32_t* rows, int64_t n,
                             const double* shift_t, const double* scale_t, const double* shift_n,
                             const double* scale_n, int side, double c, float* G, int kp, double* G64,
                             int64_t ldg, float* bnd, void* stream);

/* out[ri[a], ci[b]] = blk[a, b]: the `scores[np.ix_(i, j)] = scores_ij` of splda.py:376. */
int hypmi_scatter_block(const float* blk, int64_t ldb, const int32_t* ri, int64_t nr, const int32_t* ci,
                        int64_t nc, float* out, int64_t ldo, void* stream);

/* ---- score normalisation (hyperion/score_norm) ---------------------------------------------- */

/* Per-row top-k over the cohort: for every row of A[n_rows x nc] (nc <= 65536) the cohort indices
 * ranked discard .. discard+k-1 by descending score.  Replaces np.argsort + flip + slice
 * (adapt_s_norm.py:42-44, 65-67); ties at a boundary are resolved towards the lower cohort index
 * (numpy leaves that order unspecified).  The ORDER of the members inside a list is not their rank
 * (the statistics taken over a list do not depend on it): lists of up to 2048 members leave ordered
 * by (index mod 32), then (index mod 256), then index, and rotated by (row mod 32) / 32 of their
 * length, which spreads the lanes of a half wave of
 * hypmi_gather_stats over the LDS banks; longer ones leave in ascending index order.  The same
 * input always gives the same list.
 * List layout (uint16, groups of eight so that one 16-byte load fetches eight entries of a list):
 *   entry j of the list of row q is idx[((j >> 3) * ldi + q) * 8 + (j & 7)],  ldi >= n_rows,
 * hypmi_topk_list_bytes(k, ldi) bytes in all, 16-byte aligned.
 * Every cohort the 16-bit lists can name is selected from registers (256 threads per row up to 24 576
 * members, 512 above); `workspace` is not used and may be NULL (hypmi_topk_workspace_bytes returns 0).
 *
 * exact (optional): where the float64 value of an entry comes from.  Cohort members whose fp32 score
 * cannot be separated from a selection boundary (about 0.05 % of rows have one; the wrong member
 * moves every normalised score of that row by up to 1e-2) are then re-decided from exact values:
 *   mode 1: A64 is the float64 matrix A was rounded from ([n_rows x lda64]);
 *   mode 2: entry (r, c) = <G_rows[r,:K], G_coh[c,:K]> + G_rows[r,K] + G_coh[c,K] from the fp64 packed
 *           rows of hypmi_prologue_f64; bnd_rows are that call's (|g|, |q|) pairs, coh_gmax / coh_qmax
 *           their maxima over the cohort, err_coeff the fp32 error coefficient (8 * 2^-24).
 *   mode 3: the exact rows are not computed yet.  The call selects in fp32 as usual and appends to
 *           `todo` (device int32, room for n_rows; *todo_count, device uint32, zeroed by the caller) every
 *           row whose boundary has another entry inside its fp32 uncertainty band -- a few per cent of
 *           the rows.  bnd_rows / coh_gmax / coh_qmax / err_coeff as in mode 2, except that the rows'
 *           (|g|, |q|) may come from the fp32 prologue (hypmi_prologue_pairs_bounds).  The caller then
 *           makes the fp64 packed rows of just those rows and calls again with mode 2 and
 *           row_map = todo, n_rows = *todo_count: workgroup b re-selects row row_map[b] of A (and
 *           overwrites its list) with exact values and bounds from row b of G_rows / bnd_rows.
 * counters (optional, device, 2 x uint64): rows with a re-decided boundary, rows skipped because more
 * than 64 entries tie with a boundary. */
typedef struct {
    int32_t mode;
    const double* A64;
    int64_t lda64;
    const double* G_rows;
    const double* G_coh;
    int64_t ldg;
    int32_t K;
    const float* bnd_rows;
    float coh_gmax, coh_qmax, err_coeff;
    unsigned long long* counters;
    const int32_t* row_map;
    int32_t* todo;
    uint32_t* todo_count;
} hypmi_topk_exact;

int64_t hypmi_topk_workspace_bytes(int64_t n_rows, int nc);
int64_t hypmi_topk_list_bytes(int k, int64_t ldi);
int hypmi_topk_rows(const float* A, int64_t n_rows, int nc, int64_t lda, int k, int discard,
                    uint16_t* idx, int64_t ldi, const hypmi_topk_exact* exact, void* workspace,
                    int64_t workspace_bytes, void* stream);

/* Cohort statistics over selected members (adapt_s_norm.py:46-63 and 69-86): for every row r of
 * X[n_rows x nc] and list q (layout above): mean[r,q] and istd[r,q] = 1 / max(std, std_floor) of
 * X[r, list_q] (population std).  With mask (uint8 [n_rows x nc], 0 = masked, X already zeroed
 * there) the moments are divided by the unmasked fraction as adapt_s_norm.py:54-60 does.
 * Unmasked rows of up to 20 440 cohort members are gathered from LDS two (or four) at a time; up to
 * 40 880 one at a time (callers that can partition the lists take hypmi_gather_stats_seg instead, which
 * keeps two rows per workgroup); masked rows fit LDS up to 32 768 members; longer ones are gathered
 * from global memory, an order of magnitude slower. */
int hypmi_gather_stats(const float* X, int64_t n_rows, int nc, int64_t ldx, const unsigned char* mask,
                       int64_t ldm, const uint16_t* idx, int64_t ldi, int k, int64_t n_lists,
                       float std_floor, float* mean, float* istd, int64_t ldo, void* stream);

/* Cohorts whose rows do not fit LDS two at a time (more than 20 440 members; the reference accepts any
 * cohort, adapt_s_norm.py:30-44) without a mask: the cohort is cut into n_seg segments of seg_len members
 * that do fit, the lists are partitioned by segment once (hypmi_segment_lists) and the statistics kernels
 * walk every list once per segment with its moments carried in registers -- the same values as
 * hypmi_gather_stats to fp32 rounding (the members of a list are added in another order).
 * hypmi_gather_seg_plan returns 1 and the partition when this path applies to a cohort of nc members and
 * lists of k (0: use hypmi_gather_stats / _pairs / _apply).  Segmented lists: `groups` groups of eight
 * uint16 per list in the layout of hypmi_topk_rows (groups * ldi * 16 bytes, below 4 GiB), members as
 * 1 + offset in their segment, every segment's part padded to whole groups with zeros; gstart holds
 * (n_seg + 1) * ldi uint16, gstart[s * ldi + q] = first group of segment s of list q. */
int hypmi_gather_seg_plan(int nc, int k, int* n_seg, int* seg_len, int* groups);
int hypmi_segment_lists(const uint16_t* idx, int64_t ldi, int k, int64_t n_lists, int nc, uint16_t* seg_idx,
                        uint16_t* gstart, void* stream);
/* hypmi_gather_stats over segmented lists (idx: the lists before segmentation, walked for the rare
 * selections that lie far from their row's bulk); istd == NULL: (mean, 1 / std) pairs in `mean`, ldo in pairs. */
int hypmi_gather_stats_seg(const float* X, int64_t n_rows, int nc, int64_t ldx, const uint16_t* idx,
                           const uint16_t* seg_idx, const uint16_t* gstart, int64_t ldi, int k, int64_t n_lists,
                           float std_floor, float* mean, float* istd, int64_t ldo, void* stream);
/* hypmi_gather_apply_pairs over segmented lists. */
int hypmi_gather_apply_seg(const float* X, int64_t n_rows, int nc, int64_t ldx, const uint16_t* idx,
                           const uint16_t* seg_idx, const uint16_t* gstart, int64_t ldi, int k, int64_t n_lists,
                           float std_floor, const float* S, int64_t lds, const float* stats_t, int64_t ldt,
                           float* out, int64_t ldo, void* stream);

/* Diagnostic: the number of (row, list) pairs hypmi_gather_stats has recomputed around the list's
 * own first member since the last reset, because their selection lies so far from the row's mean
 * that the one-pass variance about the row pivot would lose more than four bits (on the current
 * device; synchronises with it).  No counterpart in the reference. */
int hypmi_gather_redo_count(unsigned long long* count, int reset);

/* Diagnostic: the shader clock (GHz) the LDS-resident gather kernels of hypmi_gather_stats / _pairs / _apply*
 * (cohorts that fit LDS undivided, no mask) have run at since the last reset, on the current device: shader
 * cycles over ticks of the constant 100 MHz
 * clock, summed over the first wave of every workgroup (0 if none has run; synchronises with the device).
 * These kernels run at the clock the chip's power budget leaves them, which differs between score
 * distributions and between boxes; the number tells a slow box from a slow build.  No counterpart in the
 * reference. */
int hypmi_gather_clock(double* ghz, int reset);

/* out[e,t] = ((S - mz[e,t]) iz[e,t] + (S - mt[t,e]) it[t,e]) / sqrt(2)  (adapt_s_norm.py:63,86,88). */
int hypmi_asnorm_apply(const float* S, int64_t lds, int64_t ne, int64_t nt, const float* mz,
                       const float* iz, int64_t ldz, const float* mt, const float* it, int64_t ldt,
                       float* out, int64_t ldo, void* stream);

/* Float64 refinement of adaptively normalised scores: the reference computes adapt_s_norm.py:46-88 in
 * float64 throughout, and a normalised score is a difference that cancels to about zero for most
 * non-target trials, where no fp32 evaluation holds a relative tolerance.  out [ne x nt] (fp32, from
 * hypmi_asnorm_apply over the statistics mz / iz [ne x nt] and mt / it [nt x ne]) is scanned; an entry
 * whose fp32 error estimate err_coeff * ((|x| + |mz|) iz + (|x| + |mt|) it + 8 + 16 (|z| + |t|)) exceeds
 * rel_tol * |out| is recomputed from the float64 matrices S64 [ne x nt], E64 [ne x nc] (enrol-vs-cohort)
 * and T64 [nt x nc] (test-vs-cohort) over the same selections t_idx / e_idx (layout of hypmi_topk_rows),
 * two-pass mean and deviation, deviation clipped at std_floor, and stored rounded to fp32.  counter
 * (device, may be NULL) is incremented by the number of entries recomputed.  Unmasked scores only. */
int hypmi_asnorm_refine(const double* S64, int64_t lds64, const double* E64, int64_t lde, const double* T64,
                        int64_t ldt64, int64_t ne, int64_t nt, int nc, const uint16_t* t_idx, int64_t ldti,
                        const uint16_t* e_idx, int64_t ldei, int k, const float* mz, const float* iz, int64_t ldz,
                        const float* mt, const float* it, int64_t ldt, float rel_tol, float err_coeff,
                        float std_floor, float* out, int64_t ldo, unsigned long long* counter, void* stream);

/* hypmi_gather_stats (without mask) and hypmi_asnorm_apply in one launch (adapt_s_norm.py:46-63 and
 * 88): X = enrol-vs-cohort scores [n_rows x nc], lists = the tests' cohort selections, S the raw
 * scores [n_rows x n_lists], mt / it the T-part statistics [n_lists x n_rows] of an earlier
 * hypmi_gather_stats over the test-vs-cohort scores; out[r,q] = the normalised score.  The Z-part
 * statistics stay in registers.  Same bits as the two separate calls.  Only for cohorts whose rows
 * fit in LDS: hypmi_gather_apply_supported(nc, n_rows) tells (1 / 0). */
int hypmi_gather_apply_supported(int nc, int64_t n_rows);
int hypmi_gather_apply(const float* X, int64_t n_rows, int nc, int64_t ldx, const uint16_t* idx, int64_t ldi,
                       int k, int64_t n_lists, float std_floor, const float* S, int64_t lds, const float* mt,
                       const float* it, int64_t ldt, float* out, int64_t ldo, void* stream);

/* The same two steps with the T-part statistics as (mean, 1 / std) PAIRS in one array instead of two
 * arrays: hypmi_gather_stats_pairs writes stats[(r ldo + q) 2 + {0, 1}] (ldo counted in pairs; unmasked
 * rows that fit in LDS only, hypmi_gather_apply_supported(nc, n_rows) tells), hypmi_gather_apply_pairs
 * reads them as stats_t[(q ldt + r) 2 + {0, 1}] (adapt_s_norm.py:69-86, then 46-63 and 88).  Same bits
 * as the calls above.  A lane of hypmi_gather_apply reads the statistics of its list from a memory line
 * of its own; with pairs it touches one such line per list instead of two, which is worth 3 % of the
 * launch. */
int hypmi_gather_stats_pairs(const float* X, int64_t n_rows, int nc, int64_t ldx, const uint16_t* idx,
                             int64_t ldi, int k, int64_t n_lists, float std_floor, float* stats, int64_t ldo,
                             void* stream);
int hypmi_gather_apply_pairs(const float* X, int64_t n_rows, int nc, int64_t ldx, const uint16_t* idx,
                             int64_t ldi, int k, int64_t n_lists, float std_floor, const float* S, int64_t lds,
                             const float* stats_t, int64_t ldt, float* out, int64_t ldo, void* stream);

/* X[i,j] = 0 where mask[i,j] == 0: `scores[mask == False] = 0` (adapt_s_norm.py:37-40,
 * t_norm.py:21, z_norm.py:22), applied to the library's device copy, never to caller arrays. */
int hypmi_mask_zero(float* X, int64_t rows, int64_t cols, int64_t ldx, const unsigned char* mask,
                    int64_t ldm, void* stream);

/* out[c, r] = in[r, c] (cohort-major score matrices to rows-over-cohort). */
int hypmi_transpose(const float* in, int64_t rows, int64_t cols, int64_t ldi, float* out, int64_t ldo,
                    void* stream);

/* Whole-row mean and 1 / max(std, floor) (z_norm.py:18-29, t_norm.py:17-28). */
int hypmi_row_meanstd(const float* X, int64_t n_rows, int nc, int64_t ldx, const unsigned char* mask,
                      int64_t ldm, float std_floor, float* mean, float* istd, void* stream);

/* out[e,t] = (S - mz[e]) iz[e]  and/or  (S - mt[t]) it[t]; both: sum / sqrt(2)
 * (z_norm.py:31, t_norm.py:30, s_norm.py:31-34).  Pass NULL pairs to disable a part. */
int hypmi_norm_apply(const float* S, int64_t lds, int64_t ne, int64_t nt, const float* mz,
                     const float* iz, const float* mt, const float* it, float* out, int64_t ldo,
                     void* stream);

/* ---- training statistics ---------------------------------------------------------------------- */

/* C = sum_s w_s x_s x_s^T (d x d, both halves) and optionally F = sum_s w_s x_s over n float64 rows
 * X [n x ldx]; w (n, optional) are row weights.  f64 matrix cores.  Replaces np.dot(x.T, x) /
 * np.dot(wx.T, x) and np.sum(x, axis=0) of hyperion/pdfs/core/normal.py:102-113 and
 * hyperion/pdfs/plda/plda_base.py:268-272; with w = 1 / N_class it is the sum over classes of the
 * per-class scatter loop of hyperion/transforms/sb_sw.py:36-43 before the class means are taken
 * out.  C and F are overwritten. */
int hypmi_gram_f64(const double* X, int64_t n, int64_t ldx, int d, const double* w, double* C,
                   int64_t ldc, double* F, void* stream);

/* The same sums added in a fixed order (same bits from run to run, like the reference's serial sums):
 * every wave writes its partial block to its own slot of the workspace and a second kernel adds the
 * slots in order.  workspace: hypmi_gram_f64_workspace_bytes(n, d) bytes of device memory. */
int64_t hypmi_gram_f64_workspace_bytes(int64_t n, int d);
int hypmi_gram_f64_ordered(const double* X, int64_t n, int64_t ldx, int d, const double* w, double* C,
                           int64_t ldc, double* F, void* workspace, int64_t workspace_bytes, void* stream);

/* Float64 class sums for training: out[g] = sum over the rows of class g of w_r x_r (w NULL: 1), rows
 * listed by order / offsets as in hypmi_pool_rows.  F of PLDABase.compute_stats_hard
 * (hyperion/pdfs/plda/plda_base.py:258-266) at the reference's precision. */
int hypmi_pool_rows_f64(const double* x, int64_t ldx, int d, const int32_t* order, const int64_t* offsets,
                        int64_t n_groups, const double* w, double* out, int64_t ldo, void* stream);

/* C[m x n] = A[m x k] B[n x k]^T + bias[n] in float64 on the f64 matrix cores (bias may be NULL).
 * Training-side products over the class statistics: gamma = (F - mu) W V^T
 * (hyperion/pdfs/plda/splda.py:77-93) and the change of basis of the posterior means. */
int hypmi_gemm_nt_f64(const double* A, int64_t m, int64_t lda, const double* B, int64_t n, int64_t ldb,
                      int k, const double* bias, double* C, int64_t ldc, void* stream);

/* Posterior of the SPLDA speaker factors given class statistics, in the eigenbasis of
 * V W V^T = Q diag(lam) Q^T (there L_i = I + N_i V W V^T is diagonal for every class at once, so the
 * loop over the distinct N_i with one Cholesky each, hyperion/pdfs/plda/splda.py:112-142, becomes one
 * pass):  T [m x ldt] holds gamma Q on entry and the posterior means y Q on return;
 * logpy[i] (optional) = -K/2 log 2pi + 1/2 (log|L_i| - y_i . gamma_i)  (splda.py:106,142);
 * *logpy_sum (optional) = sum_i logpy[i].  N are the m class counts (any positive reals). */
int hypmi_splda_posterior(double* T, int64_t m, int64_t ldt, int K, const double* N, const double* lam,
                          double* logpy, double* logpy_sum, void* stream);

/* ---- x-vector loading (HOST functions: host pointers, no device, no stream) ------------------- */

/* Shape of the record at byte `offset` of a Kaldi binary archive (the position an scp line names,
 * hyperion/utils/scp_list.py:160-201; header layout hyperion/utils/kaldi_matrix.py:163-178):
 * vectors report rows = 1, is_matrix = 0; elem_bytes is 4 (float32) or 8 (float64). */
int hypmi_ark_probe(const char* archive, int64_t offset, int32_t* rows, int32_t* cols,
                    int32_t* elem_bytes, int32_t* is_matrix);

/* Reads n vector records (or one-row matrices) of `dim` elements into the rows of out[n x ld]
 * (float32, or float64 when out_f64 != 0; records of the other width are converted).  Record i lives
 * in archives[archive_of[i]] at offsets[i] and goes to row rows[i] (NULL: row i; rows < n), so that a
 * caller can pass the records in file order, where neighbours are fetched with one read, whatever
 * the order of its key list.  Replaces the per-key loop of
 * RandomAccessArkDataReader.read + the stacking of DataReader._squeeze
 * (hyperion/io/ark_data_reader.py:554-622, data_reader.py:59-83) for x-vector files; `out` is meant
 * to be page-locked so that one hipMemcpyAsync follows.  n_threads workers read disjoint ranges with
 * pread.  Any record that is missing, truncated, compressed, text, not a vector or of another
 * dimension fails the call (HYPMI_EIO, the message names the record). */
int hypmi_ark_read_vectors(const char* const* archives, int32_t n_archives, const int32_t* archive_of,
                           const int64_t* offsets, const int64_t* rows, int64_t n, int32_t dim,
                           int32_t out_f64, void* out, int64_t ld, int32_t n_threads);

/* ---- score sink (hyperion/utils/trial_scores.py, hyperion/metrics) ---------------------------- */

/* Target / non-target histograms and moments of a device score block S[ne x nt], ADDED to hist and
 * mom (the caller zeroes them before the first block; any number of blocks of one trial set may be
 * added).  Stands in for TrialScores.get_tar_non (trial_scores.py:342-357: scores[key.tar],
 * scores[key.non]) followed by the sort of all scores (metrics/roc.py:67-72) when the matrix is too
 * large to leave the device.  Labels, one of:
 *   keys : key_e[ne], key_t[nt] int32 (both or neither); a trial is a target when the keys are equal
 *          and is no trial when either key is negative;
 *   masks: tar (and optionally non) uint8 [ne x ldm] as in TrialKey.tar / .non; without `non` every
 *          trial that is not a target is a non-target;
 *   none : every score counts as a non-target.
 * Bin of a score s: floor((s - lo) * (float)nbins / (hi - lo)) evaluated in float32, clamped to
 * 0 .. nbins-1 (1 <= nbins <= 8192).  hist: uint64 [2 x nbins], non-targets first.  mom: float64
 * [2 x 4], per class: count, sum, sum of squares, number of non-finite scores (which are left out of
 * the bins and the sums). */
int hypmi_score_hist(const float* S, int64_t lds, int64_t ne, int64_t nt, const int32_t* key_e,
                     const int32_t* key_t, const unsigned char* tar, const unsigned char* non,
                     int64_t ldm, float lo, float hi, int nbins, unsigned long long* hist, double* mom,
                     void* stream);

/* Second pass over the same blocks, with the same labels, range and bins: every finite score whose
 * (class, bin) cell is flagged in want (uint8 [2 x nbins], non-targets first) is appended to the
 * list of its class: out [2 x cap] floats (non-targets first), count [2] uint64 list lengths (ADDED
 * to; a count above cap means the list overflowed and the scores beyond cap were dropped).  The order
 * inside a list is arbitrary.  With the cells that can hold a vertex of the ROC convex hull flagged,
 * sorting the two lists makes the hull of the first pass exact; with every cell flagged it is
 * TrialScores.get_tar_non (trial_scores.py:342-357) on the device. */
int hypmi_score_extract(const float* S, int64_t lds, int64_t ne, int64_t nt, const int32_t* key_e,
                        const int32_t* key_t, const unsigned char* tar, const unsigned char* non,
                        int64_t ldm, float lo, float hi, int nbins, const unsigned char* want, float* out,
                        int64_t cap, unsigned long long* count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HYPMI_H */

/* oracle/verify_ref.c — TEST INFRASTRUCTURE / CPU BASELINE ONLY (never linked by the product).
 *
 * Plain-C restatement of the verify body the reference runs per Jacobi iteration:
 *   greedy = torch.argmax(logits, dim=-1)            MB:476, SB:197, JD:567
 *   accepted = (cumsum(draft[:,1:] != greedy[:,:-1]) == 0).sum(-1) + 1   MB:482-486, JD:253-293
 * torch.argmax semantics: first index of the maximum, NaN is the maximum, -0.0 == +0.0.
 * Checked against oracle/jacobi_oracle.py and the golden vectors by tests/test_oracle_c.py.
 * Parallel over rows with OpenMP so the timed CPU baseline uses every host core.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline float bf16_to_f32(uint16_t b) {
    uint32_t u = ((uint32_t)b) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static int64_t argmax_f32(const float *x, int64_t V) {
    int64_t bi = 0;
    float bv = x[0];
    if (isnan(bv)) return 0;
    for (int64_t i = 1; i < V; ++i) {
        const float v = x[i];
        if (isnan(v)) return i;          /* first NaN wins */
        if (v > bv) { bv = v; bi = i; }  /* strict: first index on ties, -0.0 == +0.0 */
    }
    return bi;
}

static int64_t argmax_bf16(const uint16_t *x, int64_t V) {
    int64_t bi = 0;
    float bv = bf16_to_f32(x[0]);
    if (isnan(bv)) return 0;
    for (int64_t i = 1; i < V; ++i) {
        const float v = bf16_to_f32(x[i]);
        if (isnan(v)) return i;
        if (v > bv) { bv = v; bi = i; }
    }
    return bi;
}

/* dtype: 0 = f32, 1 = bf16; logits [R, V] with row stride in elements */
void ref_argmax_rows(const void *logits, int dtype, int64_t R, int64_t V, int64_t stride, int64_t *greedy) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r) {
        greedy[r] = dtype == 0 ? argmax_f32((const float *)logits + r * stride, V)
                               : argmax_bf16((const uint16_t *)logits + r * stride, V);
    }
}

/* draft [draft_rows, L] (draft_rows == 1 broadcasts), greedy [B, gstride] */
void ref_accept_lengths(const int64_t *draft, int draft_rows, const int64_t *greedy, int64_t gstride, int B, int L,
                        int32_t *accepted, int32_t *best_idx) {
    int best = -1, bi = 0;
    for (int b = 0; b < B; ++b) {
        const int64_t *d = draft + (int64_t)(draft_rows == 1 ? 0 : b) * L;
        const int64_t *g = greedy + (int64_t)b * gstride;
        int k = 0;
        while (k < L - 1 && d[k + 1] == g[k]) ++k;
        accepted[b] = L == 0 ? 0 : k + 1;
        if (accepted[b] > best) { best = accepted[b]; bi = b; }
    }
    if (best_idx) *best_idx = bi;
}

int ref_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

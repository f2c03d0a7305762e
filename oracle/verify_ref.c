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

/* Order-preserving key of an fp32 bit pattern: -0.0 folds onto +0.0, NaN is flagged separately.  Integer max over the keys
 * vectorises (vpmaxud); the first index holding the winning key is found in a second, short pass over one block. */
static inline uint32_t key_of(uint32_t u) {
    u = (u == 0x80000000u) ? 0u : u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline int is_nan_bits(uint32_t u) { return (u & 0x7fffffffu) > 0x7f800000u; }

#define BLK 4096
#define ARGMAX_BODY(LOAD)                                                                              \
    uint32_t best = 0;                                                                                 \
    int64_t best_blk = 0, nan_blk = -1;                                                                \
    for (int64_t b0 = 0; b0 < V; b0 += BLK) {                                                          \
        const int64_t e = b0 + BLK < V ? b0 + BLK : V;                                                 \
        uint32_t m = 0;                                                                                \
        int nan = 0;                                                                                   \
        _Pragma("omp simd reduction(max : m) reduction(| : nan)")                                      \
        for (int64_t i = b0; i < e; ++i) {                                                             \
            const uint32_t u = LOAD(i);                                                                \
            nan |= is_nan_bits(u);                                                                     \
            const uint32_t k = key_of(u);                                                              \
            m = k > m ? k : m;                                                                         \
        }                                                                                              \
        if (nan) { nan_blk = b0; break; }      /* first NaN wins; no earlier block holds one */         \
        if (m > best) { best = m; best_blk = b0; }                                                     \
    }                                                                                                  \
    if (nan_blk >= 0) {                                                                                \
        for (int64_t i = nan_blk;; ++i) if (is_nan_bits(LOAD(i))) return i;                            \
    }                                                                                                  \
    const int64_t e = best_blk + BLK < V ? best_blk + BLK : V;                                         \
    for (int64_t i = best_blk; i < e; ++i) if (key_of(LOAD(i)) == best) return i;   /* first index on ties */ \
    return 0;

static int64_t argmax_f32(const float *x, int64_t V) {
    const uint32_t *w = (const uint32_t *)x;
#define LOAD_F32(i) (w[i])
    ARGMAX_BODY(LOAD_F32)
}

static int64_t argmax_bf16(const uint16_t *x, int64_t V) {
#define LOAD_BF16(i) (((uint32_t)x[i]) << 16)
    ARGMAX_BODY(LOAD_BF16)
}

/* dtype: 0 = f32, 1 = bf16; logits [R, V] with row stride in elements */
void ref_argmax_rows(const void *logits, int dtype, int64_t R, int64_t V, int64_t stride, int64_t *greedy) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r) {
        greedy[r] = dtype == 0 ? argmax_f32((const float *)logits + r * stride, V)
                               : argmax_bf16((const uint16_t *)logits + r * stride, V);
    }
}

/* Fill bf16 logits [R, V] with a cheap hash pattern using the SAME static row partition as ref_argmax_rows, so that on a
 * multi-socket host every row's pages are first touched (and therefore placed) on the NUMA node of the thread that scans it. */
void ref_fill_rows_bf16(uint16_t *x, int64_t R, int64_t V, int64_t stride, uint32_t seed) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < R; ++r) {
        uint32_t h = seed ^ (uint32_t)(r * 2654435761u);
        uint16_t *row = x + r * stride;
        for (int64_t i = 0; i < V; ++i) {
            h = h * 1664525u + 1013904223u;
            row[i] = (uint16_t)(0x3C00u + ((h >> 20) & 0x3FFu));      /* finite values around 0.01 .. 2 */
        }
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

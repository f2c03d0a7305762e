/* A plain-C client of the C ABI: proves include/jacobiforcing.h is valid C (no C++/torch types in the signatures) and that the
 * shared library links and answers without a GPU.  Built and run by tests/test_kernels.py::test_plain_c_client.
 * With -DJF_ABI_GPU (tests/test_kernels.py::test_plain_c_client_launches_kernels, -m gpu) it also allocates device memory
 * through the HIP runtime's C API and runs the argmax, the accept scan and one single-block step from C. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jacobiforcing.h"

#ifdef JF_ABI_GPU
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %s\n", (int)e_, #x); return 2; } } while (0)
#define JF(x) do { int r_ = (x); if (r_ != JF_OK) { printf("jf error %d: %s\n", r_, jf_last_error()); return 3; } } while (0)

static int gpu_part(void) {
    enum { R = 6, V = 5000 };
    float *h = (float *)malloc(sizeof(float) * R * V);
    int64_t want[R];
    for (int r = 0; r < R; ++r) {
        for (int v = 0; v < V; ++v) h[r * V + v] = (float)((v * 2654435761u + r * 40503u) & 0xFFFF) / 65536.0f;
        want[r] = (r * 997 + 13) % V;
        h[r * V + want[r]] = 2.0f;
    }
    h[2 * V + 4000] = 2.0f; if (want[2] > 4000) want[2] = 4000;          /* a tie: the first index wins (torch.argmax) */
    float *d_logits; uint64_t *d_packed; int64_t *d_greedy, *d_draft, *d_acc; int32_t *d_accepted, *d_best; jf_sb_desc *d_desc;
    CK(hipMalloc((void **)&d_logits, sizeof(float) * R * V));
    CK(hipMalloc((void **)&d_packed, sizeof(uint64_t) * R));
    CK(hipMalloc((void **)&d_greedy, sizeof(int64_t) * R));
    CK(hipMalloc((void **)&d_draft, sizeof(int64_t) * R));
    CK(hipMalloc((void **)&d_acc, sizeof(int64_t) * R));
    CK(hipMalloc((void **)&d_accepted, sizeof(int32_t)));
    CK(hipMalloc((void **)&d_best, sizeof(int32_t)));
    CK(hipMalloc((void **)&d_desc, sizeof(jf_sb_desc)));
    CK(hipMemcpy(d_logits, h, sizeof(float) * R * V, hipMemcpyHostToDevice));
    CK(hipMemset(d_packed, 0, sizeof(uint64_t) * R));
    JF(jf_argmax_rows(d_logits, JF_F32, R, V, V, d_packed, d_greedy, NULL));
    int64_t got[R];
    CK(hipMemcpy(got, d_greedy, sizeof(got), hipMemcpyDeviceToHost));
    for (int r = 0; r < R; ++r) if (got[r] != want[r]) { printf("argmax row %d: %lld != %lld\n", r, (long long)got[r], (long long)want[r]); return 4; }
    /* accept scan (MB:482-486): draft[i+1] == greedy[i] for i < 3, then a mismatch -> accepted = 4 */
    int64_t draft[R] = {7, want[0], want[1], want[2], 123456, want[4]};
    CK(hipMemcpy(d_draft, draft, sizeof(draft), hipMemcpyHostToDevice));
    JF(jf_accept_lengths(d_draft, 1, d_greedy, R, 1, R, d_accepted, d_best, NULL));
    int32_t acc = 0;
    CK(hipMemcpy(&acc, d_accepted, sizeof(acc), hipMemcpyDeviceToHost));
    if (acc != 4) { printf("accepted = %d, expected 4\n", acc); return 5; }
    /* one single-block step (SB:197-273) over the same rows: 4 accepted, re-draft of 2 tokens, cache cut back by 2 */
    JF(jf_argmax_partial(d_logits, JF_F32, R, V, V, d_packed, NULL));
    CK(hipMemcpy(d_acc, draft, sizeof(draft), hipMemcpyHostToDevice));
    JF(jf_sb_step(d_draft, R, d_packed, -1, 0, R, d_acc, 100, d_desc, NULL));
    jf_sb_desc ds;
    CK(hipMemcpy(&ds, d_desc, sizeof(ds), hipMemcpyDeviceToHost));
    int64_t nd[2];
    CK(hipMemcpy(nd, d_draft, sizeof(nd), hipMemcpyDeviceToHost));
    if (ds.raw != 4 || ds.total != 4 || ds.kv_len != 104 || ds.next_len != 2 || ds.next_token != (int32_t)want[3] || nd[0] != want[3] || nd[1] != want[4]) {
        printf("sb_step: raw=%d total=%d kv=%d next_len=%d next=%d\n", ds.raw, ds.total, ds.kv_len, ds.next_len, ds.next_token);
        return 6;
    }
    printf("gpu=ok argmax=%lld accepted=%d sb_raw=%d\n", (long long)got[0], acc, ds.raw);
    free(h);
    return 0;
}
#endif

int main(void) {
    jf_mb_params p;
    memset(&p, 0, sizeof p);
    p.n = 32; p.K = 2; p.spawn_threshold = 28; p.pool_size = 4; p.eos_id = -1; p.pad_id = 0; p.max_iter = 128; p.max_blocks = 3;
    p.lookahead_start_ratio = 0.0;
    printf("version=%d state_ints=%lld max_rows=%d max_tokens=%d desc=%zu params=%zu\n", jf_version(),
           (long long)jf_mb_state_ints(&p), (int)jf_mb_max_rows(&p), (int)jf_mb_max_tokens(&p), sizeof(jf_mb_desc), sizeof(jf_mb_params));
    /* argument validation happens before any HIP call: a NULL buffer is JF_E_INVALID with a message */
    int rc = jf_argmax_partial(NULL, JF_BF16, 4, 152064, 152064, NULL, NULL);
    printf("rc=%d err=%s\n", rc, jf_last_error());
    if (!(rc == JF_E_INVALID && jf_version() == JF_VERSION)) return 1;
#ifdef JF_ABI_GPU
    return gpu_part();
#else
    return 0;
#endif
}

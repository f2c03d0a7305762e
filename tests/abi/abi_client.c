/* A plain-C client of the C ABI: proves include/jacobiforcing.h is valid C (no C++/torch types in the signatures) and that the
 * shared library links and answers without a GPU.  Built and run by tests/test_kernels.py::test_plain_c_client.
 * With -DJF_ABI_GPU (tests/test_kernels.py::test_plain_c_client_launches_kernels, -m gpu) it also allocates device memory
 * through the HIP runtime's C API and runs the argmax, the accept scan and one single-block step from C; given a case file
 * (argv[1], written by the test from a golden record of the unmodified reference) it then drives the HOT PATH from C:
 * jf_mb_begin -> [jf_mb_pack -> logits built in C from the recorded greedy rows -> jf_mb_verify] until done -> jf_mb_read_ret,
 * comparing every forward's rows and every call's ret / next_token / iters / kv_len in C. */
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

/* ---- the multiblock hot path from C against a golden record ------------------------------------------------------ */
static int rd(FILE *f, long long *v) { return fscanf(f, "%lld", v) == 1; }

static int hot_path(const char *path) {
    FILE *f = fopen(path, "r");
    if (!f) { printf("cannot open %s\n", path); return 10; }
    long long v, V, ncalls;
    jf_mb_params p;
    memset(&p, 0, sizeof p);
    rd(f, &v); p.n = (int32_t)v; rd(f, &v); p.K = (int32_t)v; rd(f, &v); p.spawn_threshold = (int32_t)v; rd(f, &v); p.pool_size = (int32_t)v;
    rd(f, &v); p.eos_id = (int32_t)v; rd(f, &v); p.pad_id = (int32_t)v; rd(f, &v); p.max_iter = (int32_t)v; rd(f, &v); p.max_blocks = (int32_t)v;
    p.lookahead_start_ratio = 0.0;
    rd(f, &V); rd(f, &ncalls);
    const int64_t state_ints = jf_mb_state_ints(&p);
    const int RMAX = jf_mb_max_rows(&p), TMAX = jf_mb_max_tokens(&p), n = p.n;
    if (state_ints <= 0 || RMAX <= 0 || TMAX <= 0) { printf("bad params: %s\n", jf_last_error()); return 11; }
    int32_t *d_states, *d_pos, *d_rp, *d_rl, *d_kv; int64_t *d_ids, *d_in, *d_ret; uint64_t *d_packed; jf_mb_desc *d_desc; float *d_logits;
    const size_t cells = (size_t)RMAX * TMAX;
    CK(hipMalloc((void **)&d_states, sizeof(int32_t) * state_ints)); CK(hipMemset(d_states, 0, sizeof(int32_t) * state_ints));
    CK(hipMalloc((void **)&d_ids, sizeof(int64_t) * cells)); CK(hipMalloc((void **)&d_pos, sizeof(int32_t) * cells));
    CK(hipMalloc((void **)&d_rp, sizeof(int32_t) * RMAX)); CK(hipMalloc((void **)&d_rl, sizeof(int32_t) * RMAX));
    const size_t pcap = (size_t)JF_MB_PACKED_ENTRIES(cells);        /* positions + room for the launch's per-chunk slots */
    CK(hipMalloc((void **)&d_packed, sizeof(uint64_t) * pcap)); CK(hipMemset(d_packed, 0, sizeof(uint64_t) * pcap));
    CK(hipMalloc((void **)&d_desc, sizeof(jf_mb_desc))); CK(hipMalloc((void **)&d_kv, sizeof(int32_t)));
    CK(hipMalloc((void **)&d_in, sizeof(int64_t) * n)); CK(hipMalloc((void **)&d_ret, sizeof(int64_t) * (TMAX + 2)));
    const size_t lcap = (size_t)8 * 4 * n * V;                     /* logits of one forward: at most 8 rows x 4n tokens here */
    CK(hipMalloc((void **)&d_logits, sizeof(float) * lcap));
    float *h_logits = (float *)malloc(sizeof(float) * lcap);
    int64_t *h_ids = (int64_t *)malloc(sizeof(int64_t) * cells), *h_in = (int64_t *)malloc(sizeof(int64_t) * n);
    int64_t *h_ret = (int64_t *)malloc(sizeof(int64_t) * (TMAX + 2));
    long long forwards = 0;
    for (long long c = 0; c < ncalls; ++c) {
        long long kv_before, nfw;
        rd(f, &kv_before);
        for (int i = 0; i < n; ++i) { rd(f, &v); h_in[i] = v; }
        rd(f, &nfw);
        const int32_t kv32 = (int32_t)kv_before;
        CK(hipMemcpy(d_in, h_in, sizeof(int64_t) * n, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_kv, &kv32, sizeof(kv32), hipMemcpyHostToDevice));
        JF(jf_mb_begin(d_states, state_ints, 1, &p, d_in, d_kv, d_desc, NULL));
        jf_mb_desc ds;
        CK(hipMemcpy(&ds, d_desc, sizeof(ds), hipMemcpyDeviceToHost));
        for (long long it = 0; it < nfw; ++it) {
            long long B, T;
            rd(f, &B); rd(f, &T);
            if (ds.error || ds.done || ds.B != B || ds.T != T) { printf("call %lld forward %lld: desc B=%d T=%d done=%d err=%d, record B=%lld T=%lld\n", c, it, ds.B, ds.T, ds.done, ds.error, B, T); return 12; }
            if ((size_t)(B * T * V) > lcap) { printf("forward too large for the client\n"); return 13; }
            JF(jf_mb_pack(d_states, state_ints, 1, (int32_t)T, 0, d_ids, d_pos, d_rp, d_rl, NULL, 1, NULL));
            CK(hipMemcpy(h_ids, d_ids, sizeof(int64_t) * B * T, hipMemcpyDeviceToHost));
            for (long long i = 0; i < B * T; ++i) { rd(f, &v); if (h_ids[i] != v) { printf("call %lld forward %lld: out[%lld] = %lld, record %lld\n", c, it, i, (long long)h_ids[i], v); return 14; } }
            memset(h_logits, 0, sizeof(float) * B * T * V);
            for (long long i = 0; i < B * T; ++i) { rd(f, &v); h_logits[i * V + v] = 1.0f; }     /* the recorded greedy token wins its row */
            CK(hipMemcpy(d_logits, h_logits, sizeof(float) * B * T * V, hipMemcpyHostToDevice));
            JF(jf_mb_verify(d_logits, JF_F32, B * T, V, V, NULL, d_states, state_ints, 1, d_packed, B * T, (int64_t)pcap, (int32_t)T, d_desc, &p, NULL));
            CK(hipMemcpy(&ds, d_desc, sizeof(ds), hipMemcpyDeviceToHost));
            ++forwards;
        }
        long long ret_len, next_token, iters, kv_len;
        rd(f, &ret_len);
        if (!ds.done || ds.error || ds.ret_len != ret_len) { printf("call %lld: done=%d err=%d ret_len=%d, record %lld\n", c, ds.done, ds.error, ds.ret_len, ret_len); return 15; }
        JF(jf_mb_read_ret(d_states, state_ints, 1, d_ret, TMAX + 2, NULL));
        CK(hipMemcpy(h_ret, d_ret, sizeof(int64_t) * (TMAX + 2), hipMemcpyDeviceToHost));
        for (long long i = 0; i < ret_len; ++i) { rd(f, &v); if (h_ret[i] != v) { printf("call %lld: ret[%lld] = %lld, record %lld\n", c, i, (long long)h_ret[i], v); return 16; } }
        rd(f, &next_token); rd(f, &iters); rd(f, &kv_len);
        if (ds.next_token != next_token || ds.iters != iters || ds.kv_len != kv_len) {
            printf("call %lld: next=%d iters=%d kv=%d, record %lld %lld %lld\n", c, ds.next_token, ds.iters, ds.kv_len, next_token, iters, kv_len);
            return 17;
        }
    }
    fclose(f);
    printf("hot_path=ok calls=%lld forwards=%lld\n", ncalls, forwards);
    return 0;
}
#endif

int main(int argc, char **argv) {
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
    rc = gpu_part();
    if (rc) return rc;
    for (int i = 1; i < argc; ++i) { rc = hot_path(argv[i]); if (rc) return rc; }
    return 0;
#else
    (void)argc; (void)argv;
    return 0;
#endif
}

// tests/hostsim/hostsim.cpp — TEST INFRASTRUCTURE ONLY (never loaded by jacobiforcing_amd).
//
// Compiles the device state-machine source (jacobiforcing_amd/csrc/jf_mb_core.h) with a
// single-lane policy so its control logic can be checked against the golden vectors on a
// machine without a GPU.  On the GPU box the same source runs as one wavefront per prompt
// (jf_multiblock.hip) and is checked again through the C ABI by the `-m gpu` tests.
#include <stdint.h>
#include <string.h>

#include "jacobiforcing.h"
#include "jf_mb_core.h"

struct HostLanes {
    static constexpr bool WAVE64 = false;
    int shfl(int v, int) const { return v; }
    int lane() const { return 0; }
    int count() const { return 1; }
    void sync() const {}
    int reduce_min(int v) const { return v; }
    int reduce_sum(int v) const { return v; }
    int first_true(bool pred) const { return pred ? 0 : 1; }
    int count_true(bool pred) const { return pred ? 1 : 0; }
    int prefix_count(bool) const { return 0; }
    void mail(int32_t *word, int32_t v) const { *word = v; }
    void publish(int32_t *word, int32_t v, bool = false) const { *word = v; }
};

static int g_fast = 1;
extern "C" {
int hs_set_fast_path(int on) { const int old = g_fast; g_fast = on ? 1 : 0; return old; }

int64_t hs_mb_state_ints(const jf_mb_params *p) { return jfmb::make_layout(p->n, p->K, p->pool_size, p->max_blocks).total; }
int32_t hs_mb_max_rows(const jf_mb_params *p) { return jfmb::make_layout(p->n, p->K, p->pool_size, p->max_blocks).RMAX; }
int32_t hs_mb_max_tokens(const jf_mb_params *p) { return jfmb::make_layout(p->n, p->K, p->pool_size, p->max_blocks).TMAX; }

int hs_mb_begin(int32_t *states, int64_t state_ints, int P, const jf_mb_params *params, const int64_t *input_ids,
                const int32_t *kv_len, jf_mb_desc *desc) {
    for (int p = 0; p < P; ++p) jfmb::mb_begin_body(HostLanes{}, p, states, state_ints, *params, input_ids, kv_len, desc);
    return 0;
}
int hs_mb_pack(int32_t *states, int64_t state_ints, int P, int32_t Tpad, int64_t pad_fill, int64_t *input_ids,
               int32_t *positions, int32_t *row_prompt, int32_t *row_len, int32_t *valid_index, int32_t valid_align) {
    const jfmb::PackOut o{input_ids, positions, row_prompt, row_len, valid_index, nullptr, nullptr};
    for (int p = 0; p < P; ++p)
        jfmb::mb_pack_body(HostLanes{}, p, P, states, state_ints, nullptr, Tpad, 1, Tpad, pad_fill, 0, 1, o,
                           valid_align < 1 ? 1 : valid_align, jfmb::LoopDev{}, 0);
    return 0;
}

// ---- the loop API (jf_mb_loop_*): same bodies, prompts one after the other, the "last to finish" is the last of the loop
static void hs_loop_pack(const jf_mb_loop *lp, const jfmb::LoopDev &d) {
    const jfmb::PackOut o{lp->input_ids, lp->positions, lp->row_prompt, lp->row_len, lp->valid_index, lp->row_cand, lp->row_kv_len};
    for (int p = 0; p <= lp->P; ++p)                       // index P: the summary for the host
        jfmb::mb_pack_body(HostLanes{}, p, lp->P, lp->states, lp->state_ints, lp->desc, 0, lp->t_align < 1 ? 1 : lp->t_align, lp->t_cap,
                           lp->pad_fill, lp->order ? 1 : 0, lp->cand_rows, o, lp->valid_align < 1 ? 1 : lp->valid_align, d, 2);
}
int hs_mb_loop_begin(const jf_mb_loop *lp, int32_t seq, const jf_mb_params *params, const int64_t *input_ids, const int32_t *kv_len) {
    const jfmb::LoopDev d = jfmb::make_loop_dev(lp, seq, params);
    for (int p = 0; p < lp->P; ++p)
        jfmb::mb_begin_body(HostLanes{}, p, lp->states, lp->state_ints, *params, input_ids, kv_len, lp->desc, d.kv_len);
    hs_loop_pack(lp, d);
    return 0;
}
// the step half of jf_mb_loop_iterate (the caller has filled packed[] with the argmax stand-in)
int hs_mb_loop_pack(const jf_mb_loop *lp, int32_t seq, const jf_mb_params *params) { hs_loop_pack(lp, jfmb::make_loop_dev(lp, seq, params)); return 0; }
int hs_mb_loop_step(const jf_mb_loop *lp, int32_t seq, const jf_mb_params *params, int32_t Rtot, int32_t Tpad, int queue_pack) {
    const jfmb::LoopDev d = jfmb::make_loop_dev(lp, seq, params);
    for (int p = 0; p < lp->P; ++p)
        jfmb::mb_step_body(HostLanes{}, p, lp->states, lp->state_ints, lp->packed, (int64_t)Rtot * Tpad, lp->desc, d, true, g_fast != 0);
    if (queue_pack) hs_loop_pack(lp, d);
    return 0;
}
int hs_mb_step(int32_t *states, int64_t state_ints, int P, uint64_t *packed, int64_t packed_len, jf_mb_desc *desc) {
    for (int p = 0; p < P; ++p) jfmb::mb_step_body(HostLanes{}, p, states, state_ints, packed, packed_len, desc, jfmb::LoopDev{}, false, g_fast != 0);
    return 0;
}
int hs_mb_read_ret(const int32_t *states, int64_t state_ints, int P, int64_t *ret, int32_t ret_cap) {
    for (int p = 0; p < P; ++p) jfmb::mb_read_ret_body(HostLanes{}, p, states, state_ints, ret, ret_cap);
    return 0;
}

int hs_sb_step(int64_t *out, int L, uint64_t *packed, int32_t eos_id, int32_t total, int32_t cap, int64_t *acc_buf,
               int32_t kv_before, jf_sb_desc *desc) {
    jfmb::sb_step_body(HostLanes{}, out, L, packed, eos_id, total, cap, acc_buf, kv_before, desc);
    return 0;
}

// the loop around the engine steps, same contract as jf_engine_loop_commit
int hs_engine_loop_commit(const jf_engine_loop *lp, int32_t seq) {
    jfmb::engine_loop_commit_body(HostLanes{}, *lp, seq);
    return 0;
}

// engine step for a batch of rows, same contract as jf_engine_step
int hs_engine_step(const int64_t *draft, int B, int L, uint64_t *packed, int32_t eos_id, const int32_t *remaining,
                   int64_t *new_tokens, int64_t *next_draft, const int64_t *pad_stream, int64_t pad_len,
                   int64_t *pad_cursor, jf_engine_row *rows) {
    for (int b = 0; b < B; ++b) {
        const uint64_t *pk = packed + (int64_t)b * (L - 1);
        auto G = [pk](int i) { return jfmb::decode_packed(pk[i]); };
        jfmb::EngineRowOut o = jfmb::engine_row_body(HostLanes{}, draft + (int64_t)b * L, L, G, eos_id, remaining[b],
                                                     new_tokens + (int64_t)b * L, next_draft + (int64_t)b * L);
        rows[b].acc_len = o.acc_len; rows[b].n_new = o.n_new; rows[b].eos = o.eos; rows[b].active_next = o.active_next;
        rows[b].n_pads = o.active_next ? (L - 1 - o.copy_len) : 0;
        rows[b].rsv[0] = o.copy_len;
    }
    int64_t run = *pad_cursor;
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < rows[b].n_pads; ++i)
            next_draft[(int64_t)b * L + 1 + rows[b].rsv[0] + i] = pad_stream[pad_len > 0 ? ((run + i) % pad_len) : 0];
        run += rows[b].n_pads;
    }
    *pad_cursor = run;
    for (int64_t i = 0; i < (int64_t)B * (L - 1); ++i) packed[i] = 0;
    return 0;
}
}

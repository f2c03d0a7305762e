// jf_mb_core.h — the multiblock Jacobi state machine (one generation call of the reference's
// jacobi_forward_greedy_multiblock, MB:227-740), written once against a small "lanes" policy:
//
//   * DevLanes  (jf_common.h): one 64-lane wavefront per prompt; token rows are compared,
//     searched and copied 64 tokens per instruction, reductions are wave shuffles.
//   * a single-lane policy used ONLY by tests/hostsim (CPU CI of this logic; never shipped).
//
// Control flow is wave-uniform: every lane holds the same scalars in registers; token arrays live
// in the state block and phases that exchange data through it are separated
// by lanes.sync().
//
// MB = modeling/cllm2_qwen2_modeling_kv_terminate_on_eos_improved_multiblock_lookahead_unified.py
#pragma once
#include <stdint.h>

#include "jacobiforcing.h"

#if defined(__HIPCC__)
#define JF_HD __host__ __device__ __forceinline__
#define JF_UNROLL _Pragma("unroll")
#else
#define JF_HD inline
#define JF_UNROLL
#endif

#ifndef JF_STAMP
#define JF_STAMP(k) do { } while (0)   // experiment builds define it to record a clock at phase k
#endif
#define JF_FAIL(code) do { err = (code); err_line = __LINE__; } while (0)

namespace jfmb {

// ---- header slots (int32 indices into the state block) ---------------------------------------
enum : int {
    H_N = 0, H_K, H_SPAWN_THR, H_POOL_SIZE, H_EOS, H_PAD, H_MAX_ITER, H_NB,      // params
    H_RMAX, H_TMAX, H_LPOOL, H_LOOK_LO, H_LOOK_HI,
    H_NUM_BLOCKS, H_ACTIVE, H_RA, H_LEN_LISTS, H_LNT, H_HAS_LNT, H_ITERS, H_DONE, // MB:249-262
    H_RET_EARLY, H_ERR, H_PROMPT_LEN, H_KV_LEN, H_POOL_COUNT, H_POOL_HEAD,
    H_RET_LEN, H_NEXT_TOK, H_B, H_T, H_NSPANS, H_ROW_BASE, H_TPAD,
    H_LOOK_THR,              // smallest block total t with (double)t / n >= lookahead_start_ratio (MB:577), INT32_MAX if none
    H_CAND_BASE,             // forward row of this prompt's candidate row 1 (row 0 sits at H_ROW_BASE); set by the pack step
    H_SPANS = 40             // 3 ints per span (block, start, L), NB spans follow the fixed header
};
constexpr int MAX_NB = 4096; // sanity bound only; the block lists can grow by one entry per iteration (Q3/Q4 with K >= 3)

enum : int { EVT_SPAWN = 1, EVT_SWITCH = 2, EVT_EARLY = 4,
             EVT_CALL_END = 8,     // resident driver: a generation call ended in this step (its results are in the driver block)
             EVT_STOPPED = 16,     // resident driver: ... and the prompt stopped (EOS / budget / calls / cache row full)
             EVT_FAST = 32,        // this step ran as Machine::step_fast (diagnostic)
             EVT_SLOW_NEXT = 64 }; // the prompt's NEXT step cannot be the straight-line one (more than one block in flight, a spawn
                                   // or the block's end within reach): the loop's pack step lists such prompts first, so that
                                   // their steps run under the logits stream and the launch ends behind a short one

struct Layout {
    int n, NB, RMAX, TMAX, LPOOL, pool_size;
    int hdr_ints;     // fixed header + 3 * NB span slots
    int blk_stride;   // per block: [need_reverify, total_acc, acc_len, draft_rows, draft_len, rsv*3] + out_acc[n+1] + draft[RMAX][n]
    int off_blocks, off_pool, off_out, off_ret, total;
};

JF_HD int imax(int a, int b) { return a > b ? a : b; }
JF_HD int wrap(int x, int m) { return x >= m ? x - m : x; }   // x mod m for 0 <= x < 2m (ring indices: no integer division)
JF_HD int imin(int a, int b) { return a < b ? a : b; }

// every offset follows from (n, NB, RMAX, TMAX, LPOOL, pool_size): the state block in HBM uses the capacities the
// parameters ask for (make_layout); the fused verify launch runs the SAME machine on a compact image in LDS
// (compact_layout) and falls back to the HBM block when a step does not fit it
JF_HD Layout make_layout_dims(int n, int NB, int RMAX, int TMAX, int LPOOL, int pool_size) {
    Layout L;
    L.n = n;
    L.NB = NB;
    L.RMAX = RMAX;
    L.TMAX = TMAX;
    L.LPOOL = LPOOL;
    L.pool_size = pool_size;
    L.blk_stride = 8 + (n + 1) + L.RMAX * n;
    L.hdr_ints = H_SPANS + 3 * L.NB;
    L.off_blocks = L.hdr_ints;
    L.off_pool = L.off_blocks + L.NB * L.blk_stride;
    L.off_out = L.off_pool + imax(pool_size, 0) * (1 + L.LPOOL);
    L.off_ret = L.off_out + L.RMAX * L.TMAX;
    L.total = L.off_ret + L.TMAX + 2;
    L.total = (L.total + 3) & ~3;         // keep 16-byte multiples
    return L;
}

JF_HD Layout make_layout(int n, int K, int pool_size, int max_blocks) {
    const int NB = imin(imax(max_blocks, K), MAX_NB);
    // RMAX: 1 + (pool_size-1) recycled candidates (MB:74, 579-582); TMAX: RA draft + (acc ⧺ tail) of every pseudo block
    // (MB:317-377); LPOOL: concat of all blocks (MB:387-411)
    return make_layout_dims(n, NB, imax(1, pool_size), NB * n, NB * n, pool_size);
}

// What a step touches when the block counters behave (K blocks in flight, Q3's stale list entries included): K + 2 list
// entries, rows of (K + 2) * n tokens, pool entries of 2 * (K + 2) * n tokens.
JF_HD Layout compact_layout(const Layout &g, int K) {
    const int NB = imin(g.NB, K + 2);
    return make_layout_dims(g.n, NB, g.RMAX, imin(g.TMAX, NB * g.n), imin(g.LPOOL, 2 * NB * g.n), g.pool_size);
}

JF_HD Layout layout_of(const int32_t *S) {
    return make_layout(S[H_N], S[H_K], S[H_POOL_SIZE], S[H_NB]);
}

enum : int { B_NEED = 0, B_TOTAL, B_ACCLEN, B_DROWS, B_DLEN, B_HDR = 8 };

// ---- the machine -----------------------------------------------------------------------------
// GreedyFn: int operator()(int row, int t)  -> argmax token of logits[row, t] of this prompt.
template <class Lanes>
struct Machine {
    int32_t *S;
    Layout L;
    Lanes lanes;

    // scalar state mirrored in registers (uniform across lanes)
    int n, K, eos, pad, num_blocks, active, RA, len_lists, lnt, has_lnt, iters, done, ret_early, err;
    int prompt_len, kv_len, pool_count, pool_head;
    int events, ra_accepted, kv_src_row, kv_copy_dst, kv_copy_len, err_line, err_aux;
    bool allow_fast = true;   // step_fast() for the steady-state iteration (JF_MB_FAST=0 / the tests switch it off for A/B)

    JF_HD Machine(int32_t *s, Lanes l, const Layout &lay) : S(s), L(lay), lanes(l) {}

    JF_HD int32_t *blk(int b) const { return S + L.off_blocks + b * L.blk_stride; }
    JF_HD int32_t *acc(int b) const { return blk(b) + B_HDR; }
    JF_HD int32_t *draft(int b, int r) const { return blk(b) + B_HDR + (L.n + 1) + r * L.n; }
    JF_HD int32_t *pool_entry(int i) const {   // i-th oldest, 0 <= i < pool_count
        int slot = wrap(pool_head + i, L.pool_size);
        return S + L.off_pool + slot * (1 + L.LPOOL);
    }
    JF_HD int32_t *out_row(int r) const { return S + L.off_out + r * L.TMAX; }
    JF_HD int32_t *ret() const { return S + L.off_ret; }

    JF_HD void load_scalars() {
        n = S[H_N]; K = S[H_K]; eos = S[H_EOS]; pad = S[H_PAD];
        num_blocks = S[H_NUM_BLOCKS]; active = S[H_ACTIVE]; RA = S[H_RA]; len_lists = S[H_LEN_LISTS];
        lnt = S[H_LNT]; has_lnt = S[H_HAS_LNT]; iters = S[H_ITERS]; done = S[H_DONE];
        ret_early = S[H_RET_EARLY]; err = S[H_ERR]; prompt_len = S[H_PROMPT_LEN]; kv_len = S[H_KV_LEN];
        pool_count = S[H_POOL_COUNT]; pool_head = S[H_POOL_HEAD];
        events = 0; ra_accepted = 0; kv_src_row = 0; kv_copy_dst = 0; kv_copy_len = 0; err_line = 0; err_aux = 0;
    }
    JF_HD void store_scalars() {
        lanes.sync();
        if (lanes.lane() == 0) {
            S[H_NUM_BLOCKS] = num_blocks; S[H_ACTIVE] = active; S[H_RA] = RA; S[H_LEN_LISTS] = len_lists;
            S[H_LNT] = lnt; S[H_HAS_LNT] = has_lnt; S[H_ITERS] = iters; S[H_DONE] = done;
            S[H_RET_EARLY] = ret_early; S[H_ERR] = err; S[H_KV_LEN] = kv_len;
            S[H_POOL_COUNT] = pool_count; S[H_POOL_HEAD] = pool_head;
        }
        lanes.sync();
    }

    // ---- lane-parallel primitives -------------------------------------------------------------
    JF_HD void copy(int32_t *dst, const int32_t *src, int len) const {
        for (int i = lanes.lane(); i < len; i += lanes.count()) dst[i] = src[i];
    }
    JF_HD void fill(int32_t *dst, int v, int len) const {
        for (int i = lanes.lane(); i < len; i += lanes.count()) dst[i] = v;
    }
    JF_HD int find_first_eq(const int32_t *a, int len, int tok) const {
        for (int i0 = 0; i0 < len; i0 += lanes.count()) {
            const int i = i0 + lanes.lane();
            const int f = lanes.first_true(i < len && a[i] == tok);
            if (f < lanes.count()) return i0 + f;
        }
        return len;
    }

    // ---- MB:264-271 ---------------------------------------------------------------------------
    JF_HD int committed_len(int cur_RA) const {
        int c = prompt_len;
        for (int b = 0; b < num_blocks; ++b)
            if (b != cur_RA && !blk(b)[B_NEED]) c += blk(b)[B_ACCLEN];
        return c + blk(cur_RA)[B_ACCLEN];
    }

    // ---- pool (deque(maxlen=pool_size), MB:238) -------------------------------------------------
    // reserve the slot for a new newest entry; returns its storage (caller fills tokens, then sync)
    JF_HD int32_t *pool_push_slot(int len) {
        if (L.pool_size <= 0) return nullptr;
        if (pool_count == L.pool_size) { pool_head = wrap(pool_head + 1, L.pool_size); pool_count--; }
        int slot = wrap(pool_head + pool_count, L.pool_size);
        pool_count++;
        int32_t *e = S + L.off_pool + slot * (1 + L.LPOOL);
        if (lanes.lane() == 0) e[0] = len;
        return e;
    }

    // ---- MB:317-377 build_out_and_spans; returns T (0 = empty) ----------------------------------
    JF_HD int build_out(int &B_out, int &nspans_out) {
        lanes.sync();
        int32_t *ra = blk(RA);
        int B_ra = ra[B_DROWS];
        int L_ra = ra[B_DLEN];
        int cursor = 1, nsp = 0;
        int tpos = 0;   // write position in out rows
        if (L_ra > 0) {
            for (int r = 0; r < B_ra; ++r) copy(out_row(r), draft(RA, r), L_ra);
            if (lanes.lane() == 0) { S[H_SPANS + 0] = RA; S[H_SPANS + 1] = cursor; S[H_SPANS + 2] = L_ra; }
            nsp = 1; cursor += L_ra; tpos += L_ra;
        }
        for (int b = 0; b < num_blocks; ++b) {
            int32_t *bb = blk(b);
            if (b == RA || !bb[B_NEED]) continue;
            int L_acc = bb[B_ACCLEN];
            if (L_acc > 0) {
                if (tpos + L_acc > L.TMAX) { JF_FAIL(JF_E_CAPACITY); break; }
                for (int r = 0; r < B_ra; ++r) copy(out_row(r) + tpos, acc(b), L_acc);   // broadcast (MB:357)
                cursor += L_acc; tpos += L_acc;
            }
            int L_tail = bb[B_DLEN];
            if (L_tail > 0) {
                if (tpos + L_tail > L.TMAX || nsp >= L.NB) { JF_FAIL(JF_E_CAPACITY); break; }
                int rows_b = bb[B_DROWS];
                for (int r = 0; r < B_ra; ++r) {                                         // MB:288-312
                    int sr = (rows_b == B_ra) ? r : (rows_b == 1 ? 0 : (r % rows_b));
                    copy(out_row(r) + tpos, draft(b, sr), L_tail);
                }
                if (lanes.lane() == 0) {
                    S[H_SPANS + 3 * nsp + 0] = b; S[H_SPANS + 3 * nsp + 1] = cursor; S[H_SPANS + 3 * nsp + 2] = L_tail;
                }
                nsp++; cursor += L_tail; tpos += L_tail;
            }
        }
        lanes.sync();
        B_out = B_ra; nspans_out = nsp;
        return tpos;
    }

    // ---- MB:533-547 / 602-614 in-loop return, MB:723-740 finalize -----------------------------------
    JF_HD void collect_ret(bool in_loop, int kv_cur, int next_tok) {
        lanes.sync();
        int pos = 0;
        const int cap = L.TMAX + 2;                             // the ret region (a compact image has a smaller one)
        for (int b = 0; b < num_blocks; ++b) {
            int32_t *bb = blk(b);
            if (b != RA && !bb[B_NEED] && bb[B_ACCLEN] > 0) {
                if (pos + bb[B_ACCLEN] > cap) { JF_FAIL(JF_E_CAPACITY); break; }
                copy(ret() + pos, acc(b), bb[B_ACCLEN]); pos += bb[B_ACCLEN];
            }
        }
        if (!err && blk(RA)[B_ACCLEN] > 0) {
            if (pos + blk(RA)[B_ACCLEN] > cap) JF_FAIL(JF_E_CAPACITY);
            else { copy(ret() + pos, acc(RA), blk(RA)[B_ACCLEN]); pos += blk(RA)[B_ACCLEN]; }
        }
        int final_committed = prompt_len + pos;
        kv_len = kv_cur > final_committed ? final_committed : kv_cur;      // trim only when td > 0
        done = 1;
        if (in_loop) ret_early = 1;
        lanes.sync();
        if (lanes.lane() == 0) { S[H_RET_LEN] = pos; S[H_NEXT_TOK] = next_tok; }
    }

    // ---- start of a call: MB:230-262 ------------------------------------------------------------
    // look_thr_known: H_LOOK_THR of an earlier call with the same parameters (rolling restarts skip the search below)
    template <class TokFn>
    JF_HD void begin(const jf_mb_params &p, TokFn input_tok, int kv0, jf_mb_desc *d, int look_thr_known = -1) {
        fill(S, 0, L.hdr_ints);                              // header + span table (3 ints per possible block), lanes in parallel
        lanes.sync();
        if (lanes.lane() == 0) {
            S[H_N] = p.n; S[H_K] = p.K; S[H_SPAWN_THR] = p.spawn_threshold; S[H_POOL_SIZE] = p.pool_size;
            S[H_EOS] = p.eos_id; S[H_PAD] = p.pad_id; S[H_MAX_ITER] = p.max_iter; S[H_NB] = L.NB;
            S[H_RMAX] = L.RMAX; S[H_TMAX] = L.TMAX; S[H_LPOOL] = L.LPOOL;
            union { double d; int32_t i[2]; } u; u.d = p.lookahead_start_ratio;
            S[H_LOOK_LO] = u.i[0]; S[H_LOOK_HI] = u.i[1];
            // MB:577 compares total_accepted / n >= ratio in double: monotone in the total, so the step compares integers
            int thr = look_thr_known;
            if (thr < 0) {
                thr = INT32_MAX;
                for (int t = 2 * p.n + 2; t >= 0; --t) if ((double)t / (double)p.n >= p.lookahead_start_ratio) thr = t;
            }
            S[H_LOOK_THR] = thr;
            S[H_NUM_BLOCKS] = 1; S[H_ACTIVE] = 1; S[H_RA] = 0; S[H_LEN_LISTS] = 1;
            S[H_LNT] = -1; S[H_HAS_LNT] = 0; S[H_PROMPT_LEN] = kv0; S[H_KV_LEN] = kv0; S[H_NEXT_TOK] = -1;
            int32_t *b0 = blk(0);
            b0[B_NEED] = 0; b0[B_TOTAL] = 0; b0[B_ACCLEN] = 0; b0[B_DROWS] = 1; b0[B_DLEN] = p.n;
        }
        lanes.sync();
        for (int i = lanes.lane(); i < p.n; i += lanes.count()) draft(0, 0)[i] = (int32_t)input_tok(i);
        lanes.sync();
        load_scalars();
        if (kv0 < 0) { done = 1; kv_len = 0; }               // JF_MB_INACTIVE: a finished prompt rides along with B = 0
        next_iteration(d);
    }

    // step_fast()'s conditions as far as they can be told before the next forward: more than one block in flight, or the
    // spawn threshold / the end of the block within n/4 accepted tokens (a guess that may be wrong either way: it only
    // orders the position list)
    JF_HD int slow_next(int nsp, int total0) const {
        if (done || err) return 0;
        if (num_blocks != 1 || RA != 0 || len_lists != 1 || nsp != 1) return EVT_SLOW_NEXT;
        const int reach = total0 + imax(1, (n + 3) / 4);
        return ((reach >= S[H_SPAWN_THR] && active < K) || reach >= n) ? EVT_SLOW_NEXT : 0;
    }

    // ---- MB:414-419 loop head + the descriptor --------------------------------------------------
    JF_HD void next_iteration(jf_mb_desc *d) {
        int B = 0, T = 0, nsp = 0;
        if (!done && !err) {
            if (iters >= S[H_MAX_ITER]) {
                collect_ret(false, kv_len, has_lnt ? lnt : -1);
            } else {
                iters++;
                T = build_out(B, nsp);
                JF_STAMP(10);
                if (err) { done = 1; T = 0; B = 0; }
                else if (T == 0) { collect_ret(false, kv_len, has_lnt ? lnt : -1); B = 0; }
            }
        }
        store_scalars();
        events |= slow_next(nsp, blk(0)[B_TOTAL]);
        if (lanes.lane() == 0) {
            JF_STAMP(11);
            S[H_B] = done ? 0 : B; S[H_T] = done ? 0 : T; S[H_NSPANS] = done ? 0 : nsp;
            if (d) {
                d->B = done ? 0 : B; d->T = done ? 0 : T; d->done = done; d->error = err; d->iters = iters;
                d->kv_len = kv_len; d->ret_len = S[H_RET_LEN]; d->next_token = S[H_NEXT_TOK];
                d->kv_src_row = kv_src_row; d->kv_copy_dst = kv_copy_dst; d->kv_copy_len = kv_copy_len;
                d->events = events; d->accepted = ra_accepted; d->nspans = done ? 0 : nsp; d->rsv0 = err_line; d->rsv1 = err_aux;
            }
        }
        lanes.sync();
    }

    // ---- the steady-state iteration of MB:467-721 as straight-line code ----------------------------------------
    // One block in flight (num_blocks == 1, RA == 0, one span), some tokens accepted and the rest rejected, and none of
    // the events that change the shape of the state this iteration: no EOS (in the accepted prefix or as next token), no
    // spawn, no promotion / early stop, no full accept, iteration budget not exhausted.  Everything is decided from
    // read-only data first; when a condition does not hold NOTHING has been written and step() runs the general code.
    // What it does is what step() does for that case, line by line (accept scan over the B candidate rows MB:482-489,
    // commit MB:526-528, re-draft MB:550-558, the two pool pushes MB:564-573, candidates MB:577-585, KV bookkeeping
    // MB:617-626), in ~1/4 of the instructions: the last prompt's step is the serial tail of the convergence launch.
    template <class GreedyFn>
    JF_HD bool step_fast(GreedyFn G, jf_mb_desc *d) {
        if (num_blocks != 1 || RA != 0 || len_lists != 1) return false;
        const int B = S[H_B], T = S[H_T];
        if (S[H_NSPANS] != 1 || S[H_SPANS] != 0 || iters >= S[H_MAX_ITER]) return false;
        const int start = S[H_SPANS + 1], Ls = S[H_SPANS + 2];
        int32_t *bb = blk(0);
        if (Ls < 2 || bb[B_DROWS] != B || B < 1) return false;
        const int cmp = Ls - 1;
        int best_idx = 0, acc_len = 0;
        for (int r = 0; r < B; ++r) {                                   // MB:482-489: first row with the longest accepted prefix
            const int32_t *dr = draft(0, r);
            int m = cmp;
            for (int i0 = 0; i0 < cmp; i0 += lanes.count()) {
                const int i = i0 + lanes.lane();
                const int f = lanes.first_true(i < cmp ? dr[i + 1] != G(r, start - 1 + i) : false);
                if (f < lanes.count()) { m = i0 + f; break; }
            }
            if (m + 1 > acc_len) { acc_len = m + 1; best_idx = r; }
        }
        if (acc_len >= Ls) return false;                                // everything accepted: the call may end (MB:590-593)
        const int32_t *drow = draft(0, best_idx);
        if (eos >= 0 && find_first_eq(drow, acc_len, eos) < acc_len) return false;   // MB:513-521
        const int nxt = G(best_idx, start - 1 + acc_len - 1);           // MB:550
        if (eos >= 0 && nxt == eos) return false;                       // MB:599-614
        const int old_acclen = bb[B_ACCLEN];
        const int new_acclen = old_acclen + acc_len, new_total = bb[B_TOTAL] + acc_len;
        const int newL = Ls - acc_len;
        if (new_acclen > n + 1) return false;                           // capacity error: reported by the general code
        if (new_total >= S[H_SPAWN_THR] && active < K) return false;    // MB:629-653 spawn
        if (new_total >= n) return false;                               // MB:656-721 promote / early stop
        if (L.pool_size > 0 && new_acclen + newL > L.LPOOL) return false;
        // ---- nothing below can fail -------------------------------------------------------------------------
        const int kv_before = kv_len;
        int kv_cur = kv_before + T;
        copy(acc(0) + old_acclen, drow, acc_len);                        // MB:526-528
        lanes.sync();                                                   // drow may be d0 itself
        int32_t *d0 = draft(0, 0);
        for (int j = lanes.lane(); j < newL; j += lanes.count())        // [nxt] + greedy[acc_len:-1] (MB:553-558)
            d0[j] = G(best_idx, start - 1 + acc_len - 1 + j);
        if (lanes.lane() == 0) { bb[B_ACCLEN] = new_acclen; bb[B_TOTAL] = new_total; bb[B_DROWS] = 1; bb[B_DLEN] = newL; }
        lanes.sync();
        ra_accepted += acc_len;
        if (L.pool_size > 0) {                                          // MB:564-573
            const int slot = (pool_count == L.pool_size) ? pool_head : wrap(pool_head + pool_count, L.pool_size);
            int32_t *e = S + L.off_pool + slot * (1 + L.LPOOL);
            int clen = 0;
            const int tot = new_acclen + newL;
            for (int i0 = 0; i0 < tot; i0 += lanes.count()) {           // PAD-stripped acc ⧺ draft, order preserved (MB:405-407)
                const int i = i0 + lanes.lane();
                int tok = 0; bool keep = false;
                if (i < tot) { tok = i < new_acclen ? acc(0)[i] : d0[i - new_acclen]; keep = !(pad >= 0 && tok == pad); }
                const int before = lanes.prefix_count(keep);
                if (keep) e[1 + clen + before] = tok;
                clen += lanes.count_true(keep);
            }
            if (clen > 0) pool_push_slot(clen);
            lanes.sync();
            const int tlen = newL - 1;                                  // the rejected greedy tail
            if (tlen > 0) {
                int32_t *t = pool_push_slot(tlen);
                for (int i = lanes.lane(); i < tlen; i += lanes.count()) t[1 + i] = d0[1 + i];
            }
            lanes.sync();
        }
        if (new_total >= S[H_LOOK_THR]) {                               // MB:577-585: every pool entry but the newest, newest first
            int C = 0;
            for (int i1 = pool_count - 2; i1 >= 0; --i1) {
                const int32_t *e = pool_entry(i1);
                const int elen = e[0];
                const int pos = find_first_eq(e + 1, elen, nxt);
                if (pos >= elen) continue;
                if (1 + C >= L.RMAX) { JF_FAIL(JF_E_CAPACITY); break; }
                int32_t *c = draft(0, 1 + C);
                const int avail = elen - pos;
                for (int j = lanes.lane(); j < newL; j += lanes.count()) c[j] = j < avail ? e[1 + pos + j] : d0[j];   // MB:82-86
                C++;
            }
            lanes.sync();
            if (C > 1 && lanes.lane() == 0) bb[B_DROWS] = 1 + C;        // a single recycled candidate is ignored (Q5)
            lanes.sync();
        }
        lnt = nxt; has_lnt = 1;
        events |= EVT_FAST;
        { const int c = prompt_len + new_acclen; if (kv_cur > c) kv_cur = c; }   // MB:617-626 (committed_len with one block)
        kv_len = kv_cur;
        if (err) done = 1;
        if (best_idx != 0 && kv_len > kv_before) { kv_src_row = best_idx; kv_copy_dst = kv_before; kv_copy_len = kv_len - kv_before; }
        next_iteration(d);
        return true;
    }

    // ---- step_fast for a 64-lane wavefront, one token per lane ---------------------------------------------------
    // Same case, same conditions, same effects as step_fast (rows of at most 64 tokens, n <= 63), but the rows live in
    // registers: the accepted prefix, the re-draft and the pool entries come out of lane shuffles of the draft / greedy row
    // instead of strided loops over the image, the next forward's rows are written while the drafts are, and only the header
    // words that change are stored.  The step is the serial tail of the convergence launch: ~1 020 instructions as step_fast.
    // What it costs (3.2 us warm, profiles/verify_step_stages_r04.txt) is ISSUE time, one wavefront getting one instruction
    // every four or five cycles, not image round trips: a version that read every stage's words in one batch (header, four
    // candidate rows, three pool entries per round) measured the same to 0.05 us and was not kept.
    template <class GreedyFn>
    JF_HD bool step_fast64(GreedyFn G, jf_mb_desc *d) {
        if (num_blocks != 1 || RA != 0 || len_lists != 1) return false;
        const int B = S[H_B], T = S[H_T];
        if (S[H_NSPANS] != 1 || S[H_SPANS] != 0 || iters >= S[H_MAX_ITER]) return false;
        const int start = S[H_SPANS + 1], Ls = S[H_SPANS + 2];
        int32_t *bb = blk(0);
        if (Ls < 2 || Ls > 64 || n > 63 || bb[B_DROWS] != B || B < 1 || B > L.RMAX) return false;
        const int l = lanes.lane();
        const int cmp = Ls - 1;
        int best_idx = 0, acc_len = 0;
        for (int r = 0; r < B; ++r) {                                   // MB:482-489
            const int f = lanes.first_true(l < cmp ? draft(0, r)[l + 1] != G(r, start - 1 + l) : false);
            const int m = f < 64 ? f : cmp;
            if (m + 1 > acc_len) { acc_len = m + 1; best_idx = r; }
        }
        JF_STAMP(12);
        if (acc_len >= Ls) return false;
        const int dtok = l < Ls ? draft(0, best_idx)[l] : 0;            // the winning draft row and its greedy row, a token per lane
        const int gtok = l < Ls ? G(best_idx, start - 1 + l) : 0;
        if (eos >= 0 && lanes.first_true(l < acc_len && dtok == eos) < 64) return false;    // MB:513-521
        const int nxt = lanes.shfl(gtok, acc_len - 1);                  // MB:550
        if (eos >= 0 && nxt == eos) return false;                       // MB:599-614
        const int old_acclen = bb[B_ACCLEN];
        const int new_acclen = old_acclen + acc_len, new_total = bb[B_TOTAL] + acc_len;
        const int newL = Ls - acc_len;
        if (new_acclen > n + 1) return false;
        if (new_total >= S[H_SPAWN_THR] && active < K) return false;    // MB:629-653
        if (new_total >= n) return false;                               // MB:656-721
        if (L.pool_size > 0 && new_acclen + newL > L.LPOOL) return false;
        // ---- nothing below can fail; no store above ---------------------------------------------------------
        JF_STAMP(13);
        const int kv_before = kv_len;
        const int nd = lanes.shfl(gtok, l + acc_len - 1 < 64 ? l + acc_len - 1 : 63);   // re-draft [nxt] + greedy[acc_len:-1], lane j < newL
        int32_t *d0 = draft(0, 0), *o0 = out_row(0);
        if (l < acc_len) acc(0)[old_acclen + l] = dtok;                  // MB:526-528 (from registers: no hazard with d0 below)
        if (l < newL) { d0[l] = nd; o0[l] = nd; }                       // MB:553-558 and the next forward's row 0 (MB:317-340)
        ra_accepted += acc_len;
        int C = 0;
        if (L.pool_size > 0) {                                          // MB:564-573
            const int slot = (pool_count == L.pool_size) ? pool_head : wrap(pool_head + pool_count, L.pool_size);
            int32_t *e = S + L.off_pool + slot * (1 + L.LPOOL);
            const int tot = new_acclen + newL;                          // <= 2n + 1 <= 127: two passes of 64
            int clen = 0;
            for (int i0 = 0; i0 < tot; i0 += 64) {
                const int i = i0 + l;
                // position i of acc_old ⧺ accepted prefix ⧺ re-draft: image, or a shuffle out of the rows in registers
                const int from_d = lanes.shfl(dtok, (i - old_acclen) & 63);
                const int from_n = lanes.shfl(nd, (i - new_acclen) & 63);
                int tok = 0; bool keep = false;
                if (i < tot) {
                    tok = i < old_acclen ? acc(0)[i] : (i < new_acclen ? from_d : from_n);
                    keep = !(pad >= 0 && tok == pad);
                }
                const int before = lanes.prefix_count(keep);
                if (keep) e[1 + clen + before] = tok;
                clen += lanes.count_true(keep);
            }
            if (clen > 0) pool_push_slot(clen);
            const int tlen = newL - 1;                                  // the rejected greedy tail
            if (tlen > 0) {
                int32_t *t = pool_push_slot(tlen);
                if (l >= 1 && l < newL) t[l] = nd;
            }
            lanes.sync();                                               // the entries are searched below
        }
        JF_STAMP(14);
        if (new_total >= S[H_LOOK_THR]) {                               // MB:577-585
            for (int i1 = pool_count - 2; i1 >= 0; --i1) {
                const int32_t *e = pool_entry(i1);
                const int elen = e[0];
                const int pos = find_first_eq(e + 1, elen, nxt);
                if (pos >= elen) continue;
                if (1 + C >= L.RMAX) { JF_FAIL(JF_E_CAPACITY); break; }
                const int avail = elen - pos;
                if (l < newL) {
                    const int v = l < avail ? e[1 + pos + l] : nd;      // MB:82-86
                    draft(0, 1 + C)[l] = v;
                    out_row(1 + C)[l] = v;
                }
                C++;
            }
        }
        if (err) { lanes.sync(); done = 1; if (lanes.lane() == 0) { bb[B_ACCLEN] = new_acclen; bb[B_TOTAL] = new_total; bb[B_DROWS] = 1; bb[B_DLEN] = newL; } next_iteration(d); return true; }
        JF_STAMP(15);
        const int rows = C > 1 ? 1 + C : 1;                             // a single recycled candidate is ignored (Q5)
        lnt = nxt; has_lnt = 1;
        events |= EVT_FAST | slow_next(1, new_total);
        int kv_cur = kv_before + T;
        { const int c = prompt_len + new_acclen; if (kv_cur > c) kv_cur = c; }   // MB:617-626
        kv_len = kv_cur;
        if (best_idx != 0 && kv_len > kv_before) { kv_src_row = best_idx; kv_copy_dst = kv_before; kv_copy_len = kv_len - kv_before; }
        iters++;
        if (l == 0) {                                                   // the block entry, the header words that change, the descriptor
            bb[B_ACCLEN] = new_acclen; bb[B_TOTAL] = new_total; bb[B_DROWS] = rows; bb[B_DLEN] = newL;
            S[H_SPANS + 2] = newL;
            S[H_LNT] = lnt; S[H_HAS_LNT] = 1; S[H_ITERS] = iters; S[H_KV_LEN] = kv_len;
            S[H_POOL_COUNT] = pool_count; S[H_POOL_HEAD] = pool_head;
            S[H_B] = rows; S[H_T] = newL;
            if (d) {
                d->B = rows; d->T = newL; d->done = 0; d->error = 0; d->iters = iters;
                d->kv_len = kv_len; d->ret_len = S[H_RET_LEN]; d->next_token = S[H_NEXT_TOK];
                d->kv_src_row = kv_src_row; d->kv_copy_dst = kv_copy_dst; d->kv_copy_len = kv_copy_len;
                d->events = events; d->accepted = ra_accepted; d->nspans = 1; d->rsv0 = 0; d->rsv1 = 0;
            }
        }
        lanes.sync();
        return true;
    }

    // ---- MB:467-721 ------------------------------------------------------------------------------
    template <class GreedyFn>
    JF_HD void step(GreedyFn G, jf_mb_desc *d) {
        load_scalars();
        if (done || err) { next_iteration(d); return; }
        JF_STAMP(1);
        if constexpr (Lanes::WAVE64) { if (allow_fast && step_fast64(G, d)) return; }
        if (allow_fast && step_fast(G, d)) return;
        const int B = S[H_B], T = S[H_T], nspans = S[H_NSPANS];
        const int kv_before = kv_len;
        int kv_cur = kv_before + T;          // DynamicCache.update appended every forwarded token
        int best_row = 0;                    // physical candidate row the logical cache was narrowed to
        bool returned = false;
        const int look_thr = S[H_LOOK_THR];

        for (int s = 0; s < nspans && !returned; ++s) {
            const int b = S[H_SPANS + 3 * s], start = S[H_SPANS + 3 * s + 1], Ls = S[H_SPANS + 3 * s + 2];
            int32_t *bb = blk(b);
            const int rows_d = bb[B_DROWS];
            // accepted[r] (MB:482-486); RA: best = first max (MB:489), pseudo: row 0 (MB:491)
            int best_idx = 0, acc_raw = 0;
            const int nrows = (b == RA) ? B : 1;
            const int cmp = Ls - 1;          // comparisons per row; m == cmp <=> no mismatch found (yet)
            // (the steady-state iteration never gets here — step_fast; this loop favours code size over overlapped loads)
            for (int r = 0; r < nrows; ++r) {
                const int32_t *dr = draft(b, rows_d == 1 ? 0 : r);
                int m = cmp;
                for (int i0 = 0; i0 < cmp; i0 += lanes.count()) {
                    const int i = i0 + lanes.lane();
                    const int f = lanes.first_true(i < cmp ? dr[i + 1] != G(r, start - 1 + i) : false);
                    if (f < lanes.count()) { m = i0 + f; break; }
                }
                if (m + 1 > acc_raw) { acc_raw = m + 1; best_idx = r; }
            }
            JF_STAMP(2);
            if (rows_d != 1 && B != 1 && rows_d != B) {                                   // torch broadcast raises (MB:482)
                JF_FAIL(JF_E_SHAPE);
                err_aux = (rows_d << 16) | (B & 0xFFFF);                                  // the two sizes of its message
                break;
            }
            const int32_t *drow = draft(b, rows_d == 1 ? 0 : best_idx);
            if (s == 0 || b == RA) best_row = (b == RA) ? best_idx : best_row;   // MB:500-502
            if (Ls == 0) continue;
            int acc_len = acc_raw;
            bool eos_reached = false;
            if (eos >= 0 && b == RA && acc_len > 0) {                          // MB:513-521
                int e = find_first_eq(drow, acc_len, eos);
                if (e < acc_len) { acc_len = e + 1; eos_reached = true; }
            }
            const bool has_rejected = acc_len < Ls;
            // MB:526-528
            if (bb[B_ACCLEN] + acc_len > n + 1) { JF_FAIL(JF_E_CAPACITY); break; }
            copy(acc(b) + bb[B_ACCLEN], drow, acc_len);
            const int last_acc_tok = drow[acc_len - 1];
            lanes.sync();
            const int new_acclen = bb[B_ACCLEN] + acc_len, new_total = bb[B_TOTAL] + acc_len;
            lanes.sync();
            if (lanes.lane() == 0) { bb[B_ACCLEN] = new_acclen; bb[B_TOTAL] = new_total; }
            lanes.sync();
            JF_STAMP(3);
            if (b == RA) ra_accepted += acc_len;
            if (eos_reached && b == RA) {                                       // MB:531-547
                collect_ret(true, kv_cur, last_acc_tok);
                returned = true;
                break;
            }
            int nxt;
            if (has_rejected) {                                                 // MB:550-558
                nxt = G(best_idx, start - 1 + (acc_len - 1 > 0 ? acc_len - 1 : 0));
                const int newL = Ls - acc_len;                                  // [nxt] + greedy[acc_len:-1]
                lanes.sync();
                int32_t *d0 = draft(b, 0);
                for (int i = lanes.lane(); i < newL; i += lanes.count())
                    d0[i] = (i == 0) ? nxt : G(best_idx, start - 1 + acc_len + i - 1);
                lanes.sync();
                if (lanes.lane() == 0) { bb[B_DROWS] = 1; bb[B_DLEN] = newL; }
                lanes.sync();
                JF_STAMP(4);
                if (b == RA) {
                    // MB:564-573: pool.append(concat of all blocks), pool.append(rejected greedy tail)
                    {
                        // PAD-stripped concat (MB:405-407) compacted straight into the slot the push will take (the oldest
                        // entry's storage when the deque is full); the push is committed only if something was kept
                        if (L.pool_size > 0) {
                            const int slot = (pool_count == L.pool_size) ? pool_head : wrap(pool_head + pool_count, L.pool_size);
                            int32_t *e = S + L.off_pool + slot * (1 + L.LPOOL);
                            int clen = 0;
                            for (int q = 0; q < num_blocks && !err; ++q) {
                                int32_t *qb = blk(q);
                                const int la = qb[B_ACCLEN], ld = qb[B_DLEN];
                                for (int i0 = 0; i0 < la + ld; i0 += lanes.count()) {   // order-preserving, 64 tokens per pass
                                    const int i = i0 + lanes.lane();
                                    int tok = 0; bool keep = false;
                                    if (i < la + ld) {
                                        tok = i < la ? acc(q)[i] : draft(q, 0)[i - la];
                                        keep = !(pad >= 0 && tok == pad);
                                    }
                                    const int before = lanes.prefix_count(keep);      // # kept lanes below me
                                    const int cnt = lanes.count_true(keep);
                                    if (clen + cnt > L.LPOOL) { JF_FAIL(JF_E_CAPACITY); break; }
                                    if (keep) e[1 + clen + before] = tok;
                                    clen += cnt;
                                }
                            }
                            if (err) break;
                            if (clen > 0) pool_push_slot(clen);
                        }
                        lanes.sync();
                        const int tlen = newL - 1;                                // greedy[acc_len:-1]
                        if (tlen > 0) {
                            int32_t *e = pool_push_slot(tlen);
                            if (e) for (int i = lanes.lane(); i < tlen; i += lanes.count()) e[1 + i] = d0[1 + i];
                        }
                        lanes.sync();
                        JF_STAMP(5);
                    }
                    // MB:577-585 candidates
                    if (new_total >= look_thr) {
                        int C = 0;
                        // reversed(list(pool)[:-1]): every entry but the newest, newest first
                        for (int i1 = pool_count - 2; i1 >= 0 && !err; --i1) {
                            const int32_t *e = pool_entry(i1);
                            const int elen = e[0];
                            const int pos = find_first_eq(e + 1, elen, nxt);
                            if (pos >= elen) continue;
                            if (1 + C >= L.RMAX) { JF_FAIL(JF_E_CAPACITY); continue; }
                            int32_t *c = draft(b, 1 + C);
                            const int avail = elen - pos;
                            for (int j = lanes.lane(); j < newL; j += lanes.count())
                                c[j] = j < avail ? e[1 + pos + j] : d0[j];   // MB:82-86
                            C++;
                        }
                        if (err) break;
                        lanes.sync();
                        if (C > 1 && lanes.lane() == 0) bb[B_DROWS] = 1 + C;        // MB:579-582 (Q5)
                        lanes.sync();
                        JF_STAMP(6);
                    }
                }
            } else {                                                            // MB:590-593
                nxt = G(best_idx, start - 1 + Ls - 1);
                lanes.sync();
                if (lanes.lane() == 0) { bb[B_DROWS] = 1; bb[B_DLEN] = 0; }
                lanes.sync();
            }
            if (b == RA) { lnt = nxt; has_lnt = 1; }
            if (eos >= 0 && b == RA && lnt == eos) {                             // MB:599-614
                if (bb[B_ACCLEN] + 1 > n + 1) { JF_FAIL(JF_E_CAPACITY); break; }
                const int al = bb[B_ACCLEN];
                lanes.sync();
                if (lanes.lane() == 0) { acc(b)[al] = lnt; bb[B_ACCLEN] = al + 1; }
                lanes.sync();
                collect_ret(true, kv_cur, lnt);
                returned = true;
                break;
            }
        }

        if (!returned && !err) {
            // MB:617-626
            JF_STAMP(7);
            { int c = committed_len(RA); if (kv_cur > c) kv_cur = c; }
            kv_len = kv_cur;
            // MB:629-653 spawn
            {
                const int newest = num_blocks - 1;
                if (blk(newest)[B_TOTAL] >= S[H_SPAWN_THR] && active < K) {
                    if (pad < 0) { JF_FAIL(JF_E_INVALID); }
                    else if (len_lists >= L.NB) { JF_FAIL(JF_E_CAPACITY); }
                    else {
                        events |= EVT_SPAWN;
                        int32_t *ra = blk(RA), *nb = blk(len_lists);
                        const int rows = ra[B_DROWS], L_ra = ra[B_DLEN];
                        for (int r = 0; r < rows; ++r) {
                            int32_t *dst = draft(len_lists, r);
                            const int32_t *src = draft(RA, r);
                            for (int i = lanes.lane(); i < n; i += lanes.count()) dst[i] = i < L_ra ? src[i] : pad;
                        }
                        lanes.sync();
                        if (lanes.lane() == 0) {
                            nb[B_NEED] = 1; nb[B_TOTAL] = 0; nb[B_ACCLEN] = 0; nb[B_DROWS] = rows;
                            nb[B_DLEN] = L_ra > n ? L_ra : n;
                        }
                        lanes.sync();
                        len_lists++; num_blocks++; active++;
                    }
                }
            }
            // MB:656-716 promote
            if (!err && blk(RA)[B_TOTAL] >= n) {
                for (int b = 0; b < num_blocks; ++b) {
                    int32_t *bb = blk(b);
                    if (bb[B_NEED] && bb[B_TOTAL] > 0) {
                        events |= EVT_SWITCH;
                        const int a = bb[B_ACCLEN], t = bb[B_DLEN];
                        if (a + t != n || bb[B_DROWS] != 1 || !has_lnt) { JF_FAIL(JF_E_INVALID); break; }  // MB:667 assert
                        // q_full = acc ⧺ tail ; new draft = [last_next_token] ⧺ q_full[1:]
                        lanes.sync();
                        int32_t *d0 = draft(b, 0);
                        // shift tail right by a (backwards-safe: stage through the out buffer)
                        int32_t *tmp = out_row(0);
                        copy(tmp, d0, t);
                        lanes.sync();
                        for (int i = lanes.lane(); i < n; i += lanes.count()) {
                            int v = i < a ? acc(b)[i] : tmp[i - a];
                            if (i == 0) v = lnt;
                            d0[i] = v;
                        }
                        lanes.sync();
                        if (lanes.lane() == 0) {
                            bb[B_ACCLEN] = 0; bb[B_TOTAL] = 0; bb[B_NEED] = 0; bb[B_DROWS] = 1;
                            bb[B_DLEN] = n;
                        }
                        lanes.sync();
                        RA = b;
                        { int c = committed_len(RA); if (kv_cur > c) kv_cur = c; kv_len = kv_cur; }
                        active--; num_blocks--;                                   // Q3
                        break;
                    }
                    active--;                                                     // Q4
                }
            }
            // MB:719-721 early stop
            JF_STAMP(8);
            if (!err) {
                bool all_full = true;
                for (int b = 0; b < num_blocks; ++b) if (blk(b)[B_TOTAL] < n) all_full = false;
                if (all_full) {
                    events |= EVT_EARLY;
                    collect_ret(false, kv_len, has_lnt ? lnt : -1);
                }
            }
        }
        if (err) done = 1;
        JF_STAMP(9);
        // physical KV: everything kept from this forward came from candidate row best_row
        if (best_row != 0 && kv_len > kv_before) { kv_src_row = best_row; kv_copy_dst = kv_before; kv_copy_len = kv_len - kv_before; }
        next_iteration(d);
    }
};

}  // namespace jfmb

// ---------------------------------------------------------------------------------------------
// Kernel bodies shared by the HIP kernels (DevLanes) and tests/hostsim (single lane).
// ---------------------------------------------------------------------------------------------
namespace jfmb {

JF_HD int decode_packed(uint64_t v) { return (int)(~(uint32_t)(v & 0xFFFFFFFFull)); }

// ---- the loop around a step (jf_mb_loop_*): device-side views ----------------------------------------
// Resident driver block of one prompt (int32, caller-allocated, layout part of the ABI: include/jacobiforcing.h JF_DRV_*):
// the reference driver's per-prompt bookkeeping (JacobiForcing/jacobi_forcing_inference_MR_humaneval.py:152-273 = DRV) kept
// next to the state machine so that a finished call restarts without a host round trip.
enum : int { D_ACTIVE = JF_DRV_ACTIVE, D_STOP = JF_DRV_STOP, D_CALLS = JF_DRV_CALLS, D_ITERS = JF_DRV_ITERS, D_NEW = JF_DRV_NEW,
             D_BUDGET = JF_DRV_BUDGET, D_MAX_CALLS = JF_DRV_MAX_CALLS, D_TEXT_LEN = JF_DRV_TEXT_LEN, D_CURSOR = JF_DRV_CURSOR,
             D_FIN_RET_LEN = JF_DRV_FIN_RET_LEN, D_FIN_NEXT = JF_DRV_FIN_NEXT, D_FIN_ITERS = JF_DRV_FIN_ITERS,
             D_FIN_OFF = JF_DRV_FIN_OFF, D_HDR = JF_DRV_HDR_INTS };

struct LoopDev {                   // what a step needs beyond the state block (all nullable / zero = classic jf_mb_step)
    int32_t *kv_len;               // [P] committed length per prompt, written by every begin / step
    int32_t *mailbox;              // host-visible summary + descriptor table (JF_MB_MAILBOX_*); the pack launch stamps it
    int32_t seq;                   // sequence number the mailbox is stamped with
    int32_t t_align, t_cap, valid_align;
    int32_t *drv;                  // [P, drv_ints] resident driver blocks (nullable)
    int64_t drv_ints;
    const uint32_t *draws;         // [P, draw_len] pre-drawn 32-bit words for the restart drafts (DRV:209-215's random.choice)
    int32_t draw_len, text_cap, max_seq_len;
    int32_t publish_fence;         // jf_mb_loop.flags & JF_MB_LOOP_PUBLISH_FENCE
    jf_mb_params prm;              // parameters of the calls (restarts begin with them)
};

JF_HD int32_t align_up(int32_t v, int32_t a) { return a > 1 ? (v + a - 1) / a * a : v; }

inline LoopDev make_loop_dev(const jf_mb_loop *lp, int32_t seq, const jf_mb_params *params) {
    LoopDev d{};
    d.kv_len = lp->kv_len; d.mailbox = lp->mailbox; d.seq = seq;
    d.t_align = lp->t_align < 1 ? 1 : lp->t_align; d.t_cap = lp->t_cap; d.valid_align = lp->valid_align < 1 ? 1 : lp->valid_align;
    d.drv = lp->drv; d.drv_ints = lp->drv_ints; d.draws = lp->draws; d.draw_len = lp->draw_len;
    d.text_cap = lp->drv ? (int32_t)(lp->drv_ints - JF_DRV_HDR_INTS) : 0; d.max_seq_len = lp->max_seq_len;
    d.publish_fence = (lp->flags & JF_MB_LOOP_PUBLISH_FENCE) ? 1 : 0;
    d.prm = *params;
    return d;
}

// DRV:152-160, 232-250 at the end of a call + DRV:206-215 the next call's input, for one prompt whose machine `m` has just
// finished a call (m.done, no error).  Results of the finished call go to the driver block; the state machine is begun
// again in place unless the prompt stops.  `d` keeps this step's KV-copy triple / accepted count (the forward it describes
// has already happened) and carries the new call's B and T.
template <class M>
JF_HD void drv_call_end(M &m, const LoopDev &lp, int p, jf_mb_desc *d) {
    int32_t *D = lp.drv + (int64_t)p * lp.drv_ints;
    int32_t *text = D + D_HDR;
    int32_t *S = m.S;
    const int n = lp.prm.n;
    const int ret_len = S[H_RET_LEN], next_tok = S[H_NEXT_TOK], iters = m.iters, kv = m.kv_len;
    const int ev = m.events, acc = m.ra_accepted, ksr = m.kv_src_row, kcd = m.kv_copy_dst, kcl = m.kv_copy_len;
    const int old_len = D[D_TEXT_LEN], cursor = D[D_CURSOR];
    const int calls = D[D_CALLS] + 1, it_total = D[D_ITERS] + iters, new_total = D[D_NEW] + ret_len;
    const int budget = D[D_BUDGET], max_calls = D[D_MAX_CALLS];
    const int32_t *ret = m.ret();
    const bool fits = old_len + ret_len <= lp.text_cap;
    if (fits) m.copy(text + old_len, ret, ret_len);                              // generated_ids += ret (DRV:236)
    const bool has_eos = m.eos >= 0 && m.find_first_eq(ret, ret_len, m.eos) < ret_len;   // DRV:154-160
    int stop = 0;
    if (!fits) stop = JF_STOP_TEXT_FULL;
    else if (has_eos) stop = JF_STOP_EOS;
    else if (new_total >= budget) stop = JF_STOP_MAX_NEW_TOKENS;
    else if (calls >= max_calls) stop = JF_STOP_MAX_CALLS;
    else if (kv + n * (lp.prm.K + 1) + lp.t_cap > lp.max_seq_len) stop = JF_STOP_MAX_SEQ_LEN;   // the next call could outgrow the cache row
    const int new_len = fits ? old_len + ret_len : old_len;
    m.lanes.sync();
    if (m.lanes.lane() == 0) {
        D[D_CALLS] = calls; D[D_ITERS] = it_total; D[D_NEW] = new_total; D[D_TEXT_LEN] = new_len;
        D[D_FIN_RET_LEN] = ret_len; D[D_FIN_NEXT] = next_tok; D[D_FIN_ITERS] = iters; D[D_FIN_OFF] = old_len;
        D[D_STOP] = stop; D[D_ACTIVE] = stop ? 0 : 1;
        if (!stop) D[D_CURSOR] = (cursor + n - 1) % lp.draw_len;
    }
    int new_events = 0;
    if (!stop) {
        // next input = [first_correct_token] + n-1 tokens drawn from the text so far (DRV:209-215); draw k of this prompt
        // is text[(u_k * len) >> 32] with u_k the k-th word of its pre-drawn stream
        const uint32_t *u = lp.draws + (int64_t)p * lp.draw_len;
        const int dl = lp.draw_len;
        auto tok = [=](int i) -> int64_t {
            if (i == 0) return next_tok;
            const uint32_t w = u[(cursor + i - 1) % dl];
            const int idx = (int)(((uint64_t)w * (uint64_t)(uint32_t)new_len) >> 32);
            return idx < old_len ? text[idx] : ret[idx - old_len];
        };
        const int c_nb = S[H_NB], c_rmax = S[H_RMAX], c_tmax = S[H_TMAX], c_lpool = S[H_LPOOL], thr = S[H_LOOK_THR];
        m.begin(lp.prm, tok, kv, d, thr);
        new_events = m.events;
        m.lanes.sync();
        if (m.lanes.lane() == 0) {        // the block keeps the capacities it was allocated with (m may run on a compact image)
            S[H_NB] = c_nb; S[H_RMAX] = c_rmax; S[H_TMAX] = c_tmax; S[H_LPOOL] = c_lpool;
        }
    }
    m.lanes.sync();
    if (d && m.lanes.lane() == 0) {
        d->kv_src_row = ksr; d->kv_copy_dst = kcd; d->kv_copy_len = kcl; d->accepted = acc;
        d->events = ev | new_events | EVT_CALL_END | (stop ? EVT_STOPPED : 0);
        d->ret_len = ret_len; d->next_token = next_tok;
    }
    m.lanes.sync();
}

// Summary of the next forward, written where the host polls for it (the mailbox: mapped pinned host memory).  Run by ONE extra
// group of lanes of the pack step (index P) — the launch queued behind the convergence launch, so every prompt's descriptor is
// final — while the prompts' own lanes write the forward's inputs.  copy_tables = false: every
// prompt's stepper has put its own descriptor / driver record into the mailbox (the fused launch), only the header is
// written here.  The sequence number goes last, behind a system-scope release.
JF_HD int32_t mb_fin_value(const int32_t *D, int j) {        // word j of a prompt's driver record in the mailbox
    const int slot = j == 0 ? D_STOP : j == 1 ? D_CALLS : j == 2 ? D_ITERS : j == 3 ? D_NEW : j == 4 ? D_FIN_RET_LEN
                   : j == 5 ? D_FIN_NEXT : j == 6 ? D_FIN_ITERS : D_FIN_OFF;
    return D[slot];
}
template <class Lanes>
JF_HD void mb_publish_body(Lanes lanes, int P, const jf_mb_desc *desc, const LoopDev &lp, bool copy_tables) {
    int rtot = 0, rmain = 0, tmax = 0, nvalid = 0, ndone = 0, maxkv = 0, err_p = 0, acc = 0, nend = 0;
    for (int q = lanes.lane(); q < P; q += lanes.count()) {
        const jf_mb_desc d = desc[q];
        if (d.B > 0) { rtot += d.B; rmain += 1; nvalid += d.B * d.T; tmax = imax(tmax, d.T); maxkv = imax(maxkv, d.kv_len); }
        ndone += d.done ? 1 : 0;
        acc += d.accepted;
        nend += (d.events & EVT_CALL_END) ? 1 : 0;
        if (d.error && (err_p == 0 || q + 1 < err_p)) err_p = q + 1;
    }
    rtot = lanes.reduce_sum(rtot); rmain = lanes.reduce_sum(rmain); nvalid = lanes.reduce_sum(nvalid);
    ndone = lanes.reduce_sum(ndone); acc = lanes.reduce_sum(acc); nend = lanes.reduce_sum(nend);
    tmax = -lanes.reduce_min(-tmax); maxkv = -lanes.reduce_min(-maxkv);
    err_p = lanes.reduce_min(err_p ? err_p : INT32_MAX); if (err_p == INT32_MAX) err_p = 0;
    int32_t *mb = lp.mailbox;
    const int dints = (int)(sizeof(jf_mb_desc) / 4);
    if (copy_tables) {
        const int32_t *src = (const int32_t *)desc;
        for (int i = lanes.lane(); i < P * dints; i += lanes.count()) lanes.mail(mb + JF_MB_MAILBOX_HDR + i, src[i]);
        if (lp.drv) {
            int32_t *fin = mb + JF_MB_MAILBOX_HDR + P * dints;
            for (int i = lanes.lane(); i < P * JF_MB_FIN_INTS; i += lanes.count()) {
                const int q = i / JF_MB_FIN_INTS;
                lanes.mail(fin + i, mb_fin_value(lp.drv + (int64_t)q * lp.drv_ints, i - q * JF_MB_FIN_INTS));
            }
        }
    }
    if (lanes.lane() == 0) {
        const int tpad = rtot ? imin(align_up(tmax, lp.t_align), imax(lp.t_cap, tmax)) : 0;
        lanes.mail(mb + JF_MB_RTOT, rtot); lanes.mail(mb + JF_MB_RMAIN, rmain); lanes.mail(mb + JF_MB_TPAD, tpad);
        lanes.mail(mb + JF_MB_TMAX, tmax); lanes.mail(mb + JF_MB_NVALID, nvalid);
        lanes.mail(mb + JF_MB_NVALID_PAD, align_up(nvalid, lp.valid_align)); lanes.mail(mb + JF_MB_NDONE, ndone);
        lanes.mail(mb + JF_MB_MAXKV, maxkv); lanes.mail(mb + JF_MB_ERROR, err_p); lanes.mail(mb + JF_MB_ACCEPTED, acc);
        lanes.mail(mb + JF_MB_NCALL_END, nend);
    }
    lanes.publish(mb + JF_MB_SEQ, lp.seq, lp.publish_fence != 0);
}

template <class Lanes>
JF_HD void mb_begin_body(Lanes lanes, int p, int32_t *states, int64_t state_ints, const jf_mb_params &prm,
                         const int64_t *input_ids, const int32_t *kv_len, jf_mb_desc *desc, int32_t *kv_len_out = nullptr) {
    int32_t *S = states + (int64_t)p * state_ints;
    Layout lay = make_layout(prm.n, prm.K, prm.pool_size, prm.max_blocks);
    Machine<Lanes> m(S, lanes, lay);
    const int64_t *in = input_ids + (int64_t)p * prm.n;
    if (kv_len[p] == JF_MB_KEEP) {                            // prompt keeps running its current call
        if (desc && lanes.lane() == 0) { desc[p].events = 0; desc[p].accepted = 0; desc[p].kv_copy_len = 0; desc[p].kv_src_row = 0; }
        return;
    }
    m.begin(prm, [in](int i) { return in[i]; }, kv_len[p], desc ? desc + p : nullptr);
    if (kv_len_out && lanes.lane() == 0 && kv_len[p] >= 0) kv_len_out[p] = kv_len[p];
}

// Forward inputs of the current iteration (MB:417-436).  B and T of every prompt come from the descriptors when given (one
// contiguous table), else from the state blocks.
//   order 0  rows prompt by prompt (row 0, then the prompt's candidate rows)
//   order 1  row 0 of every prompt with rows first, in prompt order — when every prompt is running these ARE the cache rows
//            in order and attend in place — then the candidate rows prompt by prompt (the only rows whose K/V prefix has
//            to be gathered, MB:93-127's cost paid for those rows alone)
//   Tpad_in > 0: row length chosen by the caller; else max T rounded up to t_align (at most t_cap unless a row is longer).
struct PackOut {
    int64_t *input_ids; int32_t *positions, *row_prompt, *row_len, *valid_index;
    int32_t *row_cand, *row_kv;          // nullable: candidate scratch row (-1 = main cache) / committed prefix length of the row
};
template <class Lanes>
JF_HD void mb_pack_body(Lanes lanes, int p, int P, int32_t *states, int64_t state_ints, const jf_mb_desc *desc, int32_t Tpad_in,
                        int32_t t_align, int32_t t_cap, int64_t pad_fill, int order, int32_t cand_rows, const PackOut &o,
                        int32_t valid_align, const LoopDev &lp, int publish /* 0 no, 1 header only, 2 header + tables */) {
    if (p >= P) {                        // the extra lane group of the loop API's pack launch: the host's summary, nothing to pack
        if (publish) mb_publish_body(lanes, P, desc, lp, publish == 2);
        return;
    }
    int32_t *S = states + (int64_t)p * state_ints;
    // exclusive prefixes over the prompts before this one and totals over all of them, lanes in parallel
    int rows_b = 0, valid_b = 0, act_b = 0, act_t = 0, cand_b = 0, va_b = 0, va_t = 0, vb_b = 0, v_t = 0, tmax = 0;
    const int slow_p = desc ? (desc[p].events & EVT_SLOW_NEXT) : 0;
    for (int q = lanes.lane(); q < P; q += lanes.count()) {
        int Bq, Tq;
        if (desc) { Bq = desc[q].B; Tq = desc[q].T; }
        else { const int32_t *Q = states + (int64_t)q * state_ints; Bq = Q[H_B]; Tq = Q[H_T]; }
        if (Bq <= 0) continue;
        const int before = q < p ? 1 : 0;
        // order of the position list (= of the logits rows, = of the convergence launch's stream): prompts whose step
        // will be a long one first (EVT_SLOW_NEXT), the forward's own row order is untouched
        const int sq = desc ? (desc[q].events & EVT_SLOW_NEXT) : 0;
        const int vbefore = desc ? ((sq > slow_p || (sq == slow_p && q < p)) ? 1 : 0) : before;
        rows_b += before * Bq; valid_b += vbefore * Bq * Tq;
        act_b += before; act_t += 1;
        cand_b += before * (Bq - 1);
        va_b += vbefore * Tq; va_t += Tq;
        vb_b += vbefore * (Bq - 1) * Tq;
        v_t += Bq * Tq;
        tmax = imax(tmax, Tq);
    }
    rows_b = lanes.reduce_sum(rows_b); valid_b = lanes.reduce_sum(valid_b); act_b = lanes.reduce_sum(act_b);
    act_t = lanes.reduce_sum(act_t); cand_b = lanes.reduce_sum(cand_b); va_b = lanes.reduce_sum(va_b);
    va_t = lanes.reduce_sum(va_t); vb_b = lanes.reduce_sum(vb_b); v_t = lanes.reduce_sum(v_t);
    tmax = -lanes.reduce_min(-tmax);
    const int Tpad = Tpad_in > 0 ? Tpad_in : imin(align_up(tmax, t_align), imax(t_cap, tmax));
    Layout lay = layout_of(S);
    const int B = desc ? desc[p].B : S[H_B], T = desc ? desc[p].T : S[H_T], kv = S[H_KV_LEN];
    const int base0 = order ? act_b : rows_b;
    const int cbase = order ? act_t + cand_b : rows_b + 1;
    lanes.sync();
    if (lanes.lane() == 0) { S[H_ROW_BASE] = base0; S[H_CAND_BASE] = cbase; S[H_TPAD] = Tpad; }
    for (int r = 0; r < B; ++r) {
        const int32_t *src = S + lay.off_out + r * lay.TMAX;
        const int row = r == 0 ? base0 : cbase + r - 1;
        const int64_t ob = (int64_t)row * Tpad;
        for (int t = lanes.lane(); t < Tpad; t += lanes.count()) {
            o.input_ids[ob + t] = t < T ? (int64_t)src[t] : pad_fill;
            o.positions[ob + t] = kv + t;
        }
        if (lanes.lane() == 0) {
            o.row_prompt[row] = p; o.row_len[row] = T;
            if (o.row_cand) o.row_cand[row] = r == 0 ? -1 : p * imax(cand_rows, 1) + r - 1;
            if (o.row_kv) o.row_kv[row] = kv;
        }
        // flat position of every token that carries a draft, in forward-row order (lm_head / argmax run on these only)
        if (o.valid_index) {
            const int vb = order ? (r == 0 ? va_b : va_t + vb_b + (r - 1) * T) : valid_b + r * T;
            for (int t = lanes.lane(); t < T; t += lanes.count()) o.valid_index[vb + t] = (int32_t)(ob + t);
        }
    }
    if (o.valid_index && p == P - 1 && valid_align > 1) {            // round the list up with "no position" entries
        const int nvp = align_up(v_t, valid_align);
        for (int i = v_t + lanes.lane(); i < nvp; i += lanes.count()) o.valid_index[i] = -1;
    }
    lanes.sync();
}

// greedy token of (row r, position t) of a prompt whose row 0 is forward row base0 and whose candidate rows start at cbase
struct PackedRows {
    const uint64_t *pk; int64_t base0, cbase, tpad, plen;
    JF_HD int64_t index(int r, int t) const { return (r == 0 ? base0 : cbase + r - 1) * tpad + t; }
};

// What follows Machine::step in the loop API: the resident driver's call end, then the prompt's committed length.
// (the loop's view is passed by reference + a flag, never as a nullable pointer: a select between an address and null keeps
// the whole kernel-argument struct in scratch memory)
template <class M>
JF_HD void loop_after_step(M &m, const LoopDev &lp, bool has_loop, int p, bool was_done, jf_mb_desc *d) {
    if (!has_loop) return;
    if (lp.drv && !was_done && m.done && !m.err && lp.drv[(int64_t)p * lp.drv_ints + D_ACTIVE]) drv_call_end(m, lp, p, d);
    if (lp.kv_len && m.lanes.lane() == 0 && !m.err && !was_done) lp.kv_len[p] = m.kv_len;
}

template <class Lanes>
JF_HD void mb_step_body(Lanes lanes, int p, int32_t *states, int64_t state_ints, uint64_t *packed,
                        int64_t packed_len, jf_mb_desc *desc, const LoopDev &lp, bool has_loop, bool fast = true) {
    int32_t *S = states + (int64_t)p * state_ints;
    Layout lay = layout_of(S);
    Machine<Lanes> m(S, lanes, lay);
    m.allow_fast = fast;
    const PackedRows rows{packed, S[H_ROW_BASE], S[H_CAND_BASE], S[H_TPAD], packed_len};
    const int B = S[H_B];
    const bool was_done = S[H_DONE] != 0;
    auto G = [rows](int r, int t) -> int {
        const int64_t idx = rows.index(r, t);
        return (idx >= 0 && idx < rows.plen) ? decode_packed(rows.pk[idx]) : -1;
    };
    m.step(G, desc ? desc + p : nullptr);
    loop_after_step(m, lp, has_loop, p, was_done, desc ? desc + p : nullptr);
    // re-zero this prompt's slice of the argmax workspace for the next jf_argmax_partial
    lanes.sync();
    for (int r = 0; r < B; ++r) {
        const int64_t lo = rows.index(r, 0), hi = lo + rows.tpad;
        for (int64_t i = lo + lanes.lane(); i < hi && i < packed_len; i += lanes.count()) packed[i] = 0;
    }
}

template <class Lanes>
JF_HD void mb_read_ret_body(Lanes lanes, int p, const int32_t *states, int64_t state_ints, int64_t *ret, int32_t ret_cap) {
    const int32_t *S = states + (int64_t)p * state_ints;
    Layout lay = layout_of(S);
    const int len = S[H_RET_LEN];
    for (int i = lanes.lane(); i < ret_cap; i += lanes.count())
        ret[(int64_t)p * ret_cap + i] = i < len ? (int64_t)S[lay.off_ret + i] : -1;
}

// ---- compact image of one prompt's state (fused verify launch) -------------------------------------
// Copy the parts of the HBM state block G (layout LG) a step can touch into the compact image C (layout LC).  Returns false
// when the current state does not fit the compact capacities (the caller then steps on G itself).  Any number of lanes
// (lanes.sync() must cover all of them).  Two global round trips: the header, then everything else — block entries, span
// table and a fixed-size prefix of EVERY pool slot (the entry lengths are checked afterwards, from the image).
template <class Lanes>
JF_HD bool state_to_compact(Lanes lanes, const int32_t *G, const Layout &LG, int32_t *C, const Layout &LC) {
    for (int i = lanes.lane(); i < H_SPANS; i += lanes.count()) C[i] = G[i];
    lanes.sync();
    const int len_lists = C[H_LEN_LISTS], nsp = C[H_NSPANS], pool_count = C[H_POOL_COUNT], pool_head = C[H_POOL_HEAD];
    if (len_lists >= LC.NB || nsp > LC.NB || C[H_T] > LC.TMAX || C[H_B] > LC.RMAX) return false;   // a spawn needs one free entry
    const int n_sp = 3 * nsp, n_blk = len_lists * LG.blk_stride, per_slot = 1 + LC.LPOOL, n_pool = imax(LG.pool_size, 0) * per_slot;
    for (int i = lanes.lane(); i < n_sp + n_blk + n_pool; i += lanes.count()) {
        if (i < n_sp) C[H_SPANS + i] = G[H_SPANS + i];
        else if (i < n_sp + n_blk) C[LC.off_blocks + (i - n_sp)] = G[LG.off_blocks + (i - n_sp)];
        else {
            const int k = i - n_sp - n_blk, slot = k / per_slot, j = k - slot * per_slot;
            C[LC.off_pool + slot * per_slot + j] = G[LG.off_pool + slot * (1 + LG.LPOOL) + j];
        }
    }
    lanes.sync();
    bool fits = true;
    for (int i = 0; i < pool_count; ++i) {
        const int slot = wrap(pool_head + i, LG.pool_size);
        if (C[LC.off_pool + slot * per_slot] > LC.LPOOL) fits = false;
    }
    return fits;
}

// Write a stepped compact image back: header + spans, every listed block, the live pool entries, the next forward's rows,
// and ret when the call ended.  ONE flat loop over all of it: the regions are a handful of short copies, and as separate
// loops each costs a dependent LDS-load -> store round per trip (1.6 us for ~1 100 ints on 192 lanes); flat, a lane has
// five or six independent element copies in flight (profiles/verify_trace_r03.txt).
template <class Lanes>
JF_HD void compact_to_state(Lanes lanes, const int32_t *C, const Layout &LC, int32_t *G, const Layout &LG) {
    const int len_lists = C[H_LEN_LISTS], nsp = C[H_NSPANS], pool_count = C[H_POOL_COUNT], pool_head = C[H_POOL_HEAD];
    const int B = C[H_B], T = C[H_T];
    int pool_w = 0;                                              // widest live pool entry (its length word included)
    for (int k = 0; k < pool_count; ++k) pool_w = imax(pool_w, 1 + C[LC.off_pool + wrap(pool_head + k, LG.pool_size) * (1 + LC.LPOOL)]);
    const int n0 = H_SPANS + 3 * nsp;                            // header + spans: same offsets in both layouts
    const int n1 = n0 + len_lists * LG.blk_stride;               // listed blocks (blk_stride depends on n and RMAX only)
    const int n2 = n1 + pool_count * pool_w;                     // live pool entries, pool_w columns each
    const int n3 = n2 + B * T;                                   // rows of the next forward
    const int n4 = n3 + (C[H_DONE] ? C[H_RET_LEN] : 0);          // ret of a call that ended
    for (int i = lanes.lane(); i < n4; i += lanes.count()) {
        int src, dst;
        if (i < n0) { src = i; dst = i; }
        else if (i < n1) { src = LC.off_blocks + (i - n0); dst = LG.off_blocks + (i - n0); }
        else if (i < n2) {
            const int j = i - n1, k = j / pool_w, e = j - k * pool_w, slot = wrap(pool_head + k, LG.pool_size);
            src = LC.off_pool + slot * (1 + LC.LPOOL) + e;
            if (e > C[LC.off_pool + slot * (1 + LC.LPOOL)]) continue;          // beyond this entry's tokens
            dst = LG.off_pool + slot * (1 + LG.LPOOL) + e;
        } else if (i < n3) {
            const int j = i - n2, r = j / T, t = j - r * T;
            src = LC.off_out + r * LC.TMAX + t; dst = LG.off_out + r * LG.TMAX + t;
        } else { src = LC.off_ret + (i - n3); dst = LG.off_ret + (i - n3); }
        G[dst] = C[src];
    }
}

// ---- engine single-block step (JD:567-710) for one row; pads handled by the caller ----------------
struct EngineRowOut { int acc_len, n_new, eos, active_next, copy_len, seed; };

// greedy(i) for i in [0, L-1); draft row d[0..L)
template <class Lanes, class GreedyFn>
JF_HD EngineRowOut engine_row_body(Lanes lanes, const int64_t *d, int L, GreedyFn greedy, int eos_id, int remaining,
                                   int64_t *new_tokens, int64_t *next_draft) {
    EngineRowOut o;
    int m = L - 1;
    for (int i = lanes.lane(); i < L - 1; i += lanes.count())
        if (d[i + 1] != (int64_t)greedy(i)) { m = i; break; }
    m = lanes.reduce_min(m);
    int acc_len = m + 1;                                   // JD:572 / 589-590
    if (acc_len < 1) acc_len = 1;
    if (acc_len > L) acc_len = L;
    int eos = 0;
    if (eos_id >= 0 && acc_len > 1) {                      // JD:597-602
        int e = acc_len;
        for (int i = 1 + lanes.lane(); i < acc_len; i += lanes.count())
            if (d[i] == (int64_t)eos_id) { e = i; break; }
        e = lanes.reduce_min(e);
        if (e < acc_len) { acc_len = e + 1; eos = 1; }
    }
    int n_new = acc_len - 1;
    int seed;
    if (n_new > 0) {                                       // JD:609-614
        for (int i = lanes.lane(); i < n_new; i += lanes.count()) new_tokens[i] = d[1 + i];
        seed = (int)d[acc_len - 1];
    } else {                                               // JD:619-631 autoregressive fallback
        const int nt = greedy(0);
        if (lanes.lane() == 0) new_tokens[0] = nt;
        n_new = 1;
        seed = nt;
        if (eos_id >= 0 && nt == eos_id) eos = 1;
    }
    const int active_next = (!eos && n_new < remaining) ? 1 : 0;   // JD:659-669
    int copy_len = 0;
    if (active_next) {                                     // JD:680-709
        if (lanes.lane() == 0) next_draft[0] = seed;
        if (acc_len < L) {
            const int off = acc_len == 1 ? 1 : acc_len - 1;
            int rem = (L - 1) - off;
            copy_len = rem < L - 1 ? rem : L - 1;
            for (int i = lanes.lane(); i < copy_len; i += lanes.count()) next_draft[1 + i] = greedy(off + i);
        } else {
            if (lanes.lane() == 0) next_draft[1] = greedy(L - 2);
            copy_len = 1;
        }
    }
    o.acc_len = acc_len; o.n_new = n_new; o.eos = eos; o.active_next = active_next; o.copy_len = copy_len; o.seed = seed;
    return o;
}

// ---- the loop around the engine steps (jf_engine_loop_commit, include/jacobiforcing.h) ---------------------------------
// What the reference's decoders do per row on the host after a step (JD:609-654, JDN:581-639: extend_tokens, the cached
// length back to len(seq), the token budget) on the loop's device arrays, and the record the host polls for.  `lanes` spans
// the whole launch (one workgroup): phase 1 one lane per row, phase 2 one lane per (row, column).
struct EngineLoopRow { int n, eos, act, fallback, bad; };
JF_HD EngineLoopRow engine_loop_row(const jf_engine_loop &lp, int b) {
    EngineLoopRow o;
    if (lp.kind == JF_EL_KIND_GREEDY) {
        const jf_engine_row &r = ((const jf_engine_row *)lp.rows)[b];
        o.n = r.n_new; o.eos = r.eos; o.act = r.active_next; o.fallback = r.acc_len == 1; o.bad = r.rsv[0] == JF_E_LAUNCH;
    } else {
        const jf_rs_row &r = ((const jf_rs_row *)lp.rows)[b];
        o.n = r.n_committed; o.eos = r.eos; o.act = r.active_next; o.fallback = 0; o.bad = r.rsv != 0;
    }
    if (o.n < 0) o.n = 0;
    if (o.n > lp.L) o.n = lp.L;
    return o;
}
template <class Lanes>
JF_HD void engine_loop_commit_body(Lanes lanes, const jf_engine_loop &lp, int32_t seq) {
    int32_t *mb = lp.mailbox;
    if (lanes.lane() == 0) { lanes.mail(mb + JF_EL_ERROR, 0); lanes.mail(mb + JF_EL_STEP_ERROR, 0); }
    lanes.sync();
    for (int b = lanes.lane(); b < lp.B; b += lanes.count()) {
        const EngineLoopRow r = engine_loop_row(lp, b);
        const int s = lp.slot ? lp.slot[b] : b;
        const int have = lp.ring_len[s];
        if (r.bad) lanes.mail(mb + JF_EL_STEP_ERROR, b + 1);                 // (any such row: the host raises)
        if (have + r.n > lp.ring_cap) lanes.mail(mb + JF_EL_ERROR, b + 1);
        else lp.ring_len[s] = have + r.n;
        lp.remaining[b] -= r.n;
        lp.kv_start[b] += r.n;
        lanes.mail(mb + JF_EL_HDR + b, r.n | (r.eos ? 1 << 16 : 0) | (r.act ? 1 << 17 : 0) | (r.fallback ? 1 << 18 : 0));
    }
    lanes.sync();
    const int64_t cells = (int64_t)lp.B * lp.L;
    for (int64_t i = lanes.lane(); i < cells; i += lanes.count()) {
        const int b = (int)(i / lp.L), j = (int)(i - (int64_t)b * lp.L);
        const EngineLoopRow r = engine_loop_row(lp, b);
        lp.positions[i] = lp.kv_start[b] + j;
        if (j < r.n) {
            const int s = lp.slot ? lp.slot[b] : b;
            const int at = lp.ring_len[s] - r.n + j;                           // (a ring that was full kept its length: the store lands on
            if (at >= 0) lp.ring[(int64_t)s * lp.ring_cap + at] = lp.tokens[i];   //  older tokens, inside the ring; the host raises on the error word)
        }
    }
    for (int c = lanes.lane(); c < lp.n_cursors && c < 3; c += lanes.count()) {
        const int64_t v = lp.cursors[c];
        lanes.mail(mb + JF_EL_CURSORS + 2 * c, (int32_t)(uint32_t)(v & 0xFFFFFFFFll));
        lanes.mail(mb + JF_EL_CURSORS + 2 * c + 1, (int32_t)(v >> 32));
    }
    lanes.publish(mb + JF_EL_SEQ, seq, (lp.flags & JF_MB_LOOP_PUBLISH_FENCE) != 0);
}

// ---- HF single-block step (SB = modeling/cllm2_qwen2_modeling_kv_terminate_on_eos_improved.py:197-273) -------------
// Everything between two forwards of jacobi_forward_greedy for one call; greedy[i] = decode(packed[i]) verifies out[i+1].
template <class Lanes>
JF_HD void sb_step_body(Lanes lanes, int64_t *out, int L, uint64_t *packed, int eos_id, int total, int cap, int64_t *acc_buf,
                        int kv_before, jf_sb_desc *desc) {
    auto G = [packed](int i) { return decode_packed(packed[i]); };
    const int kv_after = kv_before + L;
    int m = L - 1;                                           // SB:197-200: accepted = 1 + leading matches
    for (int i0 = 0; i0 < L - 1; i0 += lanes.count()) {
        const int i = i0 + lanes.lane();
        const int f = lanes.first_true((i < L - 1) && (out[i + 1] != (int64_t)G(i)));
        if (f < lanes.count()) { m = i0 + f; break; }
    }
    const int raw = m + 1;
    int num = raw, eos_hit = 0;
    if (eos_id >= 0) {                                       // SB:204-211: EOS inside the accepted prefix caps it
        for (int i0 = 0; i0 < raw; i0 += lanes.count()) {
            const int i = i0 + lanes.lane();
            const int f = lanes.first_true(i < raw && out[i] == (int64_t)eos_id);
            if (f < lanes.count()) { num = i0 + f + 1; eos_hit = 1; break; }
        }
    }
    for (int i = lanes.lane(); i < num; i += lanes.count())   // SB:214-215 (writes past the preallocated tensor are dropped)
        if (total + i < cap) acc_buf[total + i] = out[i];
    total += num;
    int done = 0, kv = kv_after, next = -1, next_len = 0;
    if (eos_hit) {                                           // SB:219-227: desired_len = total_accepted (no prompt length, as written)
        kv = kv_after < total ? kv_after : total;
        next = eos_id; done = 1;
    } else if (raw < L) {                                    // SB:231-255
        kv = kv_after - (L - raw);
        next = G(raw - 1);
        if (eos_id >= 0 && next == eos_id) {
            if (lanes.lane() == 0 && total < cap) acc_buf[total] = next;
            total += 1;
            kv = kv < total ? kv : total;
            done = 1;
        } else {
            next_len = L - raw;                              // [next] + greedy[raw : L-1]
            lanes.sync();                                    // acc_buf reads of `out` are done
            for (int j = lanes.lane(); j < next_len; j += lanes.count()) out[j] = (j == 0) ? (int64_t)next : (int64_t)G(raw + j - 1);
        }
    } else {                                                 // SB:258-273: bonus token on a full accept
        next = G(L - 1);
        if (lanes.lane() == 0 && total < cap) acc_buf[total] = next;
        total += 1;
        if (eos_id >= 0 && next == eos_id) { kv = kv < total ? kv : total; done = 1; }
    }
    lanes.sync();
    for (int i = lanes.lane(); i < L; i += lanes.count()) packed[i] = 0;
    if (lanes.lane() == 0) {
        desc->raw = raw; desc->num = num; desc->total = total; desc->done = done; desc->next_token = next;
        desc->kv_len = kv; desc->next_len = next_len; desc->eos = eos_hit;
    }
}

}  // namespace jfmb

/*
 * jacobiforcing.h — C ABI of the MI355X-native Jacobi-decoding hot path.
 *
 * The reference (hao-ai-lab/JacobiForcing) is pure Python; it has no FFI of its own.  The entry
 * points below are what a ctypes binding on the reference side would bind for its loop body
 * (INTEGRATION.md shows the stub).  Each entry cites the reference code it replaces; citations use
 *   MB  = modeling/cllm2_qwen2_modeling_kv_terminate_on_eos_improved_multiblock_lookahead_unified.py
 *   SB  = modeling/cllm2_qwen2_modeling_kv_terminate_on_eos_improved.py
 *   JD  = inference_engine/engine/jacobi_decoding.py
 *   JDN = inference_engine/engine/jacobi_decoding_nongreedy.py
 *   MR  = inference_engine/engine/model_runner.py
 *   ATT = inference_engine/layers/attention.py
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch) unless marked "host";
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return value 0 = ok, negative = JF_E_* (message via jf_last_error()); nothing throws;
 *   - no allocation happens inside any call; workspaces are sized by the *_bytes() helpers;
 *   - thread-compatible (one caller thread per GPU/process), not thread-safe.
 */
#ifndef JACOBIFORCING_H
#define JACOBIFORCING_H

#include <stddef.h>
#include <stdint.h>

/* the only symbols the shared library exports (it is built with -fvisibility=hidden) */
#define JF_API __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

#define JF_VERSION 600

enum {
    JF_OK = 0,
    JF_E_INVALID = -1,   /* bad argument (shape / alignment / null) -> ValueError on the Python side */
    JF_E_CAPACITY = -2,  /* a fixed capacity would be exceeded       -> RuntimeError                  */
    JF_E_LAUNCH = -3,    /* HIP launch / runtime failure             -> RuntimeError                  */
    JF_E_SHAPE = -4      /* rows that torch could not broadcast (MB:482, reachable with K >= 3) -> RuntimeError */
};

enum { JF_F32 = 0, JF_BF16 = 1 };

JF_API int jf_version(void);
/* Timing of ONE call from outside (bench.py's per-kernel figures): the events armed here (nullable hipEvent_t, created with
 * timing enabled) are taken by the NEXT call of jf_rs_probs or jf_rs_step on this thread and carry the START timestamp of that
 * call's first launch and the STOP timestamp of its last one (hipExtLaunchKernel: the dispatches' own timestamps, what rocprofv3
 * reports as kernel durations) — not the time of two event packets around them.  JF_VERIFY_EVENTS=bracket, or a call that takes
 * several launches on its slow path, records them in front of and behind the launches instead.  Other entry points ignore it;
 * jf_mb_loop_iterate takes its events as arguments. */
JF_API int jf_timing_arm(void *ev_begin, void *ev_end);
JF_API const char *jf_last_error(void);
/* Which GPU the library's launches of this process go to: writes "pci=DDDD:BB:DD.F uuid=<32 hex digits> arch=<gcnArchName>
 * cus=<compute units>" for HIP device `device` (< 0: the calling thread's current device) into buf (NUL-terminated, at most cap
 * bytes).  bench.py gathers it from every rank so that the N-GPU line proves N distinct devices (one process per
 * GPU — the reference's only precedent is scripts/inference/scanning_hyperparameter_jacobi_decoding_mr.sh:30-84, which pins a
 * process per CUDA_VISIBLE_DEVICES entry and never checks). */
JF_API int jf_device_identity(int device, char *buf, size_t cap);

/* ---------------------------------------------------------------------------------------------
 * (a2) block-local logits argmax.  Replaces torch.argmax(block_logits, dim=-1) at MB:476, SB:197,
 * JD:357, JD:567, MR:917.  torch semantics: first index of the maximum, NaN is the maximum,
 * -0.0 == +0.0.
 *
 * logits      [R, V] row-major, row stride `row_stride` elements (>= V), dtype JF_F32 or JF_BF16
 * packed      [R] uint64 workspace; MUST be all-zero on entry.  On exit packed[r] =
 *             (order_key(max) << 32) | ~argmax.  Consumers (jf_argmax_decode, jf_mb_step,
 *             jf_engine_step, ...) re-zero it after reading, so one torch.zeros() at start-up
 *             is enough.
 * Algorithmic bytes: R*V*esize read (+ 8*R written).
 */
JF_API int jf_argmax_partial(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride,
                      uint64_t *packed, void *stream);

/* Same, for logits computed on a compacted list of positions: row i's result goes to
 * packed[out_index[i]]; rows with out_index[i] < 0 (list padding) are not read at all.
 * The reference runs lm_head + argmax over every position of the padded [B, T] rectangle
 * (MB:463-476); with jf_mb_pack's valid_index only positions that carry a draft token are
 * computed, and jf_mb_step still finds them at (row_base + b) * Tpad + t.
 */
JF_API int jf_argmax_scatter(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride,
                      const int32_t *out_index, uint64_t *packed, void *stream);

/* packed -> int64 token ids (and re-zero packed).  greedy [R] int64. */
JF_API int jf_argmax_decode(uint64_t *packed, int64_t R, int64_t *greedy, void *stream);

/* convenience: partial + decode.  */
JF_API int jf_argmax_rows(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride,
                   uint64_t *packed, int64_t *greedy, void *stream);

/* ---------------------------------------------------------------------------------------------
 * (a3) token equality + accepted-prefix scan.  Replaces MB:482-486 / SB:199-200 / JD:253-293:
 *   accepted[b] = 1 + #leading i with draft[b, i+1] == greedy[b, i],  i in [0, L-2].
 * draft  [draft_rows, L] int64 (draft_rows == 1 broadcasts, MB:482), greedy [B, >=L-1] int64 with
 * row stride greedy_stride.  accepted [B] int32, best_idx [1] int32 = first index of max (MB:489).
 */
JF_API int jf_accept_lengths(const int64_t *draft, int draft_rows, const int64_t *greedy, int64_t greedy_stride,
                      int B, int L, int32_t *accepted, int32_t *best_idx, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Multiblock Jacobi state machine (a1, a3-a12): one generation call of
 * jacobi_forward_greedy_multiblock (MB:227-740) per prompt, P prompts side by side.
 *
 * The per-prompt state is an opaque int32 block in device memory, sized by jf_mb_state_ints().
 * Per iteration the caller runs:   forward(out) -> jf_argmax_partial -> jf_mb_step -> read desc.
 */
typedef struct jf_mb_params {
    int32_t n;               /* n_token_seq_len (MB:149)                                  */
    int32_t K;               /* max concurrent blocks (MB:151)                            */
    int32_t spawn_threshold; /* ceil(r * n), computed by the caller in double (MB:262)    */
    int32_t pool_size;       /* n_gram_pool_size (MB:155); 0/1 disables recycling         */
    int32_t eos_id;          /* -1 = EOS handling off (MB:234)                            */
    int32_t pad_id;          /* -1 = none (spawn then fails like MB:631-632)              */
    int32_t max_iter;        /* max_iteration_count (MB:166)                              */
    int32_t max_blocks;      /* capacity of the block lists (>= K; Q3 can exceed K)       */
    double lookahead_start_ratio; /* MB:154, compared as total_acc / n >= ratio (MB:577)  */
} jf_mb_params;

/* one per prompt, written by jf_mb_begin / jf_mb_step (device memory or mapped pinned host memory) */
typedef struct jf_mb_desc {
    int32_t B;          /* rows of the next forward (0 when done)                              */
    int32_t T;          /* tokens per row of the next forward                                  */
    int32_t done;       /* 1: call finished; ret/next_token/iters valid                        */
    int32_t error;      /* 0 or JF_E_* raised inside the state machine                         */
    int32_t iters;      /* Jacobi iterations so far (MB:413-415)                               */
    int32_t kv_len;     /* committed KV length after this step (MB:617-626, 733-736)           */
    int32_t ret_len;    /* tokens in ret (valid when done)                                     */
    int32_t next_token; /* MB:547/614/739 (valid when done; -1 if never set)                   */
    int32_t kv_src_row; /* candidate row whose K/V for [kv_copy_dst, +kv_copy_len) must be     */
    int32_t kv_copy_dst;/*   copied onto row 0 (0 when nothing to copy, MB:500-502)            */
    int32_t kv_copy_len;
    int32_t events;     /* bit0 spawn, bit1 switch, bit2 early-stop (banners MB:634/660/720);
                           bit3 call ended, bit4 prompt stopped (resident driver); bit5 the step ran as the
                           straight-line fast path (diagnostic); bit6 the NEXT step will not (more than one
                           block in flight / spawn or block end within reach): jf_mb_loop_* list such prompts'
                           positions first, so their steps run under the logits stream                    */
    int32_t accepted;   /* tokens the real-active block accepted in this step                  */
    int32_t nspans;
    int32_t rsv0, rsv1;   /* on error: rsv0 = state-machine source line, rsv1 = (draft rows << 16) | candidate rows for JF_E_SHAPE */
} jf_mb_desc;

JF_API int64_t jf_mb_state_ints(const jf_mb_params *p); /* int32 elements per prompt state */
JF_API int32_t jf_mb_max_rows(const jf_mb_params *p);   /* max B (candidate rows)          */
JF_API int32_t jf_mb_max_tokens(const jf_mb_params *p); /* max T per row                   */

/* Start a generation call for P prompts (MB:230-262, then the first build_out_and_spans MB:317-377).
 * states      [P, state_ints] int32
 * input_ids   [P, n] int64 — token 0 is the correct next token, not yet cached (MB:243)
 * kv_len      [P] int32 — committed KV length == prompt_len for this call (MB:261), or
 *             JF_MB_INACTIVE: the prompt is finished and rides along with B = 0, or
 *             JF_MB_KEEP: leave this prompt's running call untouched (rolling restarts in a batch)
 * desc        [P] jf_mb_desc
 */
#define JF_MB_INACTIVE (-1)
#define JF_MB_KEEP (-2)
JF_API int jf_mb_begin(int32_t *states, int64_t state_ints, int P, const jf_mb_params *params,
                const int64_t *input_ids, const int32_t *kv_len, jf_mb_desc *desc, void *stream);

/* Emit the forward inputs for the current iteration (MB:417-436) in a row-padded layout:
 *   rows of prompt p start at row_base[p] = sum_{q<p} B_q (computed on the device),
 *   input_ids [Rtot, Tpad] int64 (padding = pad_fill), positions [Rtot, Tpad] int32 (kv_len + t),
 *   row_prompt [Rtot] int32, row_len [Rtot] int32 (T of that row).
 * Also records (row_base, Tpad) in each state so jf_mb_step can find its greedy tokens.
 * valid_index (nullable) [>= roundup(sum_p B_p*T_p, valid_align)] int32: flat index row*Tpad + t of
 *   every position that carries a draft token, prompt by prompt, row by row; the list is rounded
 *   up to a multiple of valid_align (>= 1) with -1 entries.  Its length is known on the host from
 *   the descriptors (sum of B*T).  Feed it to the lm_head gather and to jf_argmax_scatter.
 */
JF_API int jf_mb_pack(int32_t *states, int64_t state_ints, int P, int32_t Tpad, int64_t pad_fill,
               int64_t *input_ids, int32_t *positions, int32_t *row_prompt, int32_t *row_len,
               int32_t *valid_index, int32_t valid_align, void *stream);

/* One loop body after the forward (MB:467-721, and MB:723-740 when the call ends):
 * verify every span against the packed argmax results, pick the best candidate row, EOS cap,
 * accept, re-draft, pool + candidate build, KV bookkeeping, spawn, promote, early stop, then the
 * next build_out_and_spans.  packed is indexed (row_base[p] + b) * Tpad + t and is re-zeroed.
 */
JF_API int jf_mb_step(int32_t *states, int64_t state_ints, int P, uint64_t *packed, int64_t packed_len,
               jf_mb_desc *desc, void *stream);

/* jf_argmax_scatter / jf_argmax_partial + jf_mb_step as ONE launch: the argmax items of the whole forward and one
 * "stepper" workgroup per prompt that pre-loads the prompt's live state into LDS while the logits stream, waits for its
 * own rows only and runs the loop body on LDS — a prompt's state machine overlaps the other prompts' streaming and there
 * is no dependent launch.  Same results as the two calls.
 *   logits [R, V] (rows follow valid_index when out_index is given, else the Rtot x Tpad rectangle of jf_mb_pack),
 *   Tpad as passed to jf_mb_pack,
 *   packed: zero on entry, left zero.  packed_len = Rtot * Tpad (positions of the forward), packed_cap = entries the
 *   buffer holds (>= packed_len).  Inside this launch every (position, chunk of the vocabulary) item owns the slot
 *   packed[position * chunks + chunk] and its non-zero result word is its own arrival flag (no counters, nothing to
 *   order); rows are split into at most packed_cap / packed_len chunks, so give small forwards room for ~16 chunks
 *   (JF_MB_PACKED_ENTRIES) — with packed_cap == packed_len a row is one item,
 *   params: the jf_mb_params the states were begun with.
 * Rows that are not 16-byte aligned, or more prompts than half of the workgroups the device keeps resident of this kernel
 * (waiting steppers must never be able to fill the chip; 640 on an MI355X), fall back to the two launches.  A stepper that waits longer than 2 s for its rows
 * reports JF_E_LAUNCH in its descriptor instead of hanging the GPU. */
#define JF_MB_PACKED_ENTRIES(positions) ((int64_t)(positions) + 65536)   /* a capacity that never limits the chunking of small forwards much */
JF_API int jf_mb_verify(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride, const int32_t *out_index,
                 int32_t *states, int64_t state_ints, int P, uint64_t *packed, int64_t packed_len, int64_t packed_cap,
                 int32_t Tpad, jf_mb_desc *desc, const jf_mb_params *params, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The loop around the step (jf_mb_loop_*): everything between two forwards without a host round trip on the critical
 * path.  The reference's driver (JacobiForcing/jacobi_forcing_inference_MR_humaneval.py:152-273 = DRV) and its function
 * (MB:413-721) go back to Python after every iteration (6-10 host syncs) and after every call (DRV:206-250).
 * Here one launch (jf_mb_verify's, or jf_mb_step's when the fused launch does not apply)
 *   - runs the loop body of every prompt,
 *   - writes each prompt's committed KV length where the forward reads it (kv_len, replaces MB:36-59's trims),
 *   - with a resident driver block per prompt: ends the call (DRV:232-250: text += ret, stop on EOS / max_new_tokens /
 *     max_calls), and begins the next one with [first_correct_token] + n-1 tokens drawn from the text (DRV:209-215),
 *   - and mails every prompt's descriptor to a mailbox in mapped pinned host memory.
 * The pack launch queued right behind it (row length chosen on the device) first stamps the mailbox with a summary of the
 * next forward and a sequence number (system-scope release) — the host polls that word instead of copying descriptors and
 * synchronising the stream — and then writes the next forward's inputs while the host is still waking up.
 */
enum {  /* mailbox layout (int32): header, then P descriptors (16 ints each), then P driver records (JF_MB_FIN_INTS each) */
    JF_MB_SEQ = 0,        /* written last: the sequence number passed to the call                            */
    JF_MB_RTOT = 1,       /* rows of the next forward (sum of B)                                             */
    JF_MB_RMAIN = 2,      /* prompts with rows (= rows that use the main cache; they come first, order 1)    */
    JF_MB_TPAD = 3,       /* padded row length the pack step uses                                            */
    JF_MB_TMAX = 4,       /* longest row                                                                     */
    JF_MB_NVALID = 5,     /* positions that carry a draft token (sum of B*T)                                 */
    JF_MB_NVALID_PAD = 6, /* ... rounded up to valid_align: rows of the compacted logits                     */
    JF_MB_NDONE = 7,      /* prompts whose descriptor says done                                              */
    JF_MB_MAXKV = 8,      /* largest committed length among prompts with rows                                */
    JF_MB_ERROR = 9,      /* 0, or 1 + the first prompt whose descriptor carries an error                    */
    JF_MB_ACCEPTED = 10,  /* sum of desc.accepted                                                            */
    JF_MB_NCALL_END = 11, /* prompts whose call ended in this launch (resident driver)                       */
    JF_MB_MAILBOX_HDR = 16,
    JF_MB_FIN_INTS = 8    /* per prompt: stop, calls, total iterations, new tokens, last ret_len, last next_token, last iters, text offset of last ret */
};
#define JF_MB_MAILBOX_INTS(P) (JF_MB_MAILBOX_HDR + (P) * (16 + JF_MB_FIN_INTS))

enum {  /* resident driver block of one prompt (int32): header, then the text (prompt + generated tokens) */
    JF_DRV_ACTIVE = 0,    /* 1 while the prompt keeps starting calls                                         */
    JF_DRV_STOP = 1,      /* JF_STOP_* once it stopped                                                       */
    JF_DRV_CALLS = 2,     /* calls so far (the prefill counts as one, DRV:234)                               */
    JF_DRV_ITERS = 3,     /* total Jacobi iterations of the finished calls                                   */
    JF_DRV_NEW = 4,       /* generated tokens (len(generated) - prompt)                                      */
    JF_DRV_BUDGET = 5,    /* max_new_tokens                                                                  */
    JF_DRV_MAX_CALLS = 6,
    JF_DRV_TEXT_LEN = 7,  /* tokens in the text                                                              */
    JF_DRV_CURSOR = 8,    /* next word of this prompt's draw stream                                          */
    JF_DRV_FIN_RET_LEN = 9, JF_DRV_FIN_NEXT = 10, JF_DRV_FIN_ITERS = 11, JF_DRV_FIN_OFF = 12,   /* the last finished call */
    JF_DRV_HDR_INTS = 16
};
enum { JF_STOP_NONE = 0, JF_STOP_EOS = 1, JF_STOP_MAX_NEW_TOKENS = 2, JF_STOP_MAX_CALLS = 3, JF_STOP_MAX_SEQ_LEN = 4,
       JF_STOP_TEXT_FULL = 5 };

/* jf_mb_loop.flags: publish the mailbox's sequence word behind a system-scope RELEASE fence instead of the default "write the
 * tables through, drain the memory counter, store the word" order.  The default is validated on gfx950 (0 stale tables in 780 000
 * rounds) and ~7 us cheaper per launch; the Python host runs a short self-test of it when it builds its first loop on a device
 * and sets this bit if the host ever saw the word before the tables (a new driver / firmware), JF_PUBLISH_FENCE=1 forces it. */
#define JF_MB_LOOP_PUBLISH_FENCE 1
typedef struct jf_mb_loop {
    /* the state machines (same objects as jf_mb_begin / jf_mb_step / jf_mb_verify take) */
    int32_t *states; int64_t state_ints; int32_t P; int32_t order;   /* order: row order of the pack step, 0 prompt-major, 1 row 0 of every prompt first */
    uint64_t *packed; int64_t packed_cap;        /* argmax workspace, zero; capacity in entries (jf_mb_verify: room for chunk slots) */
    jf_mb_desc *desc;                             /* [P]                                                          */
    /* forward inputs, written by the pack step: capacity rows_cap x t_cap tokens                                  */
    int64_t *input_ids; int32_t *positions; int32_t *row_prompt; int32_t *row_len;
    int32_t *row_cand;                            /* [rows] -1 = the row writes the main cache, else p * cand_rows + b - 1 */
    int32_t *row_kv_len;                          /* [rows] committed prefix length of the row's prompt           */
    int32_t *valid_index;                         /* nullable: compacted position list (see jf_mb_pack)           */
    int32_t rows_cap, t_cap, t_align, valid_align, cand_rows;
    int32_t flags;                                /* JF_MB_LOOP_* bits (was rsv0 = 0)                              */
    int64_t pad_fill;
    int32_t *kv_len;                              /* nullable [P]: committed length per prompt (the cache's)      */
    int32_t *mailbox;                             /* JF_MB_MAILBOX_INTS(P) ints of MAPPED PINNED HOST memory (jf_host_alloc) */
    /* resident driver (all nullable / 0: the caller restarts calls itself with jf_mb_loop_begin)                  */
    int32_t *drv; int64_t drv_ints;               /* [P, drv_ints]: JF_DRV_HDR_INTS + text capacity               */
    const uint32_t *draws; int32_t draw_len;      /* [P, draw_len] pre-drawn 32-bit words                         */
    int32_t max_seq_len;                          /* positions a cache row holds (JF_STOP_MAX_SEQ_LEN)            */
} jf_mb_loop;

/* Mapped, coherent pinned host memory for the mailbox (the one allocation of this library; done once at start-up). */
JF_API int jf_host_alloc(size_t bytes, void **out);
JF_API int jf_host_free(void *p);
/* Host side of the hand-off: spin until mailbox[JF_MB_SEQ] == seq (acquire).  Gives up with JF_E_LAUNCH after timeout_us, or
 * when `stream` has drained and the word still has not arrived (a launch that failed). */
JF_API int jf_mailbox_wait(const int32_t *mailbox, int32_t seq, int64_t timeout_us, void *stream);

/* jf_mb_begin for the loop: begin / keep / retire the prompts (kv_len as in jf_mb_begin), write loop->kv_len, publish
 * (sequence number seq), then the pack step. */
JF_API int jf_mb_loop_begin(const jf_mb_loop *loop, int32_t seq, const jf_mb_params *params, const int64_t *input_ids,
                     const int32_t *kv_len, void *stream);
/* One iteration after the forward: the convergence check + loop body of every prompt (jf_mb_verify's launch, same logits
 * contract: rows follow valid_index when `compacted`, else the Rtot x Tpad rectangle; Rtot / Tpad are the mailbox's),
 * then the pack step of the next forward.  jf_kv_commit (when candidate rows ran) may follow on the same stream. */
JF_API int jf_mb_loop_iterate(const jf_mb_loop *loop, int32_t seq, const void *logits, int dtype, int64_t R, int64_t V,
                       int64_t row_stride, int compacted, int32_t Rtot, int32_t Tpad, const jf_mb_params *params,
                       int queue_pack, void *ev_begin, void *ev_end, void *stream);
/* ev_begin / ev_end (nullable hipEvent_t): a caller's timing of the convergence launch alone without splitting the call (the
 * pack launch still follows at once).  They carry the START and STOP timestamps of that launch's dispatch (hipExtLaunchKernel:
 * the kernel's own duration, what rocprofv3 reports); with JF_VERIFY_EVENTS=bracket in the environment, or when the check runs
 * as its two launches, they are recorded on `stream` immediately before and after it instead (launch + two event packets). */
/* The pack step alone (queue_pack = 0 above: a caller that brackets the convergence launch with its own events); it is the
 * launch that stamps the mailbox with `seq`. */
JF_API int jf_mb_loop_pack(const jf_mb_loop *loop, int32_t seq, const jf_mb_params *params, void *stream);

/* A/B knob: 0 makes every step run the general state-machine code instead of its straight-line steady-state path
 * (Machine::step_fast); results are identical (the parity suites run both).  Returns the previous setting.  Default 1, or
 * the JF_MB_FAST environment variable read once. */
JF_API int jf_mb_set_fast_path(int on);

/* Copy results of finished calls: ret [P, ret_cap] int64 (ret_len in desc). */
JF_API int jf_mb_read_ret(const int32_t *states, int64_t state_ints, int P, int64_t *ret, int32_t ret_cap,
                   void *stream);

/* ---------------------------------------------------------------------------------------------
 * KV cache (a9/a10/a18).  Layout per layer: kv [2, rows, H_kv, S_max, D] (K then V), any 2-byte
 * or 4-byte element; a "token row" is D elements.  Replaces the Triton store_kvcache_kernel
 * (ATT:10-40), DynamicCache narrow/expand/contiguous (MB:93-127, 500-502) and trims (MB:36-59).
 */
/* scatter freshly computed K/V rows into cache slots: token i goes to slot[i] = row * S_max + position
 * (-1 = skip).  Sources are [N, H_kv, D] views whose token stride is k_tok_stride / v_tok_stride ELEMENTS (heads
 * and D contiguous), so K and V can be read straight out of a fused QKV projection without a copy. */
JF_API int jf_kv_append(void *k_cache, void *v_cache, const void *k_new, const void *v_new,
                 const int64_t *slot, int64_t N, int32_t H_kv, int32_t D, int64_t S_max,
                 int64_t k_tok_stride, int64_t v_tok_stride, int32_t elem_bytes, void *stream);

/* Fused RoPE + Q re-layout + KV append for one layer (what the reference's forward does in three steps:
 * apply_rotary_pos_emb, the Triton store_kvcache_kernel ATT:10-40, and the attention input transpose).
 *   qkv        [N, (nq + 2*nkv) * D] the fused projection output, token-major (N = R*T tokens, token i = r*T + t)
 *   positions  [N] int32 absolute positions; cos/sin [max_pos, D/2] float32 tables (rotate-half convention)
 *   q_out      [R, nkv, (nq/nkv)*T, D]: rotated queries grouped by KV head (query row = g*T + t)
 *   K (rotated) and V rows are written to the main cache at slot_main[i] and, when cand caches are given, to the
 *   candidate scratch at slot_cand[i] (slot = row * S_max|T_max + position; -1 skips).
 * dtype JF_F32 or JF_BF16 for qkv / q_out / caches. */
JF_API int jf_rope_kv_append(const void *qkv, int dtype, int64_t N, int32_t T, int32_t nq, int32_t nkv, int32_t D,
                      const int32_t *positions, const float *cos_table, const float *sin_table, void *q_out,
                      void *k_cache, void *v_cache, const int64_t *slot_main, int64_t S_max,
                      void *k_cand, void *v_cand, const int64_t *slot_cand, int64_t T_max, void *stream);

/* SwiGLU gate of the MLP in one pass: out[m, i] = silu(gu[m, i]) * gu[m, I + i] for the fused gate/up projection output
 * gu [M, 2*I] (row-major), out [M, I].  fp32 arithmetic, one rounding; dtype JF_F32 or JF_BF16. */
JF_API int jf_swiglu(const void *gu, int dtype, int64_t M, int64_t I, void *out, void *stream);

/* commit accepted candidate rows: for prompt p copy desc[p].kv_copy_len token rows from the
 * candidate scratch cand[(p*cand_rows + kv_src_row-1), :, 0:len] to main[p, :, kv_copy_dst: +len],
 * for K and V of `layers` layers (pointer arrays live in device memory). */
JF_API int jf_kv_commit(void *const *main_k, void *const *main_v, void *const *cand_k, void *const *cand_v,
                 int32_t layers, const jf_mb_desc *desc, int P, int32_t cand_rows, int32_t H_kv,
                 int32_t D, int64_t S_max, int64_t T_max, int32_t elem_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Engine single-block decoder step (a15): JD:567-710 for a batch of rows with a common L.
 * draft [B, L] int64 (draft[:,0] = seed), packed argmax of logits [B, L-1, V] at
 * packed[(b*(L-1)) + i].  Per row: acc_len (JD:572,589-590), EOS cap (JD:597-602), committed tokens
 * (JD:609-631 incl. the AR fallback), next draft (JD:680-709) with pads taken from pad_stream in
 * row order starting at *pad_cursor.
 */
typedef struct jf_engine_row {
    int32_t acc_len;     /* after EOS cap                                   */
    int32_t n_new;       /* tokens committed this step (>=1)                */
    int32_t eos;         /* EOS committed                                   */
    int32_t active_next; /* row keeps decoding (not eos, below max_tokens)  */
    int32_t n_pads;      /* pads consumed for the next draft                */
    int32_t rsv[3];      /* scratch: [0] JF_E_LAUNCH when the launch gave up waiting for this row (2 s bound; else what the two-launch
                            path left: copied tokens), [1..2] the row's hand-off word inside the launch (zero the records once) */
} jf_engine_row;

JF_API int jf_engine_step(const int64_t *draft, int B, int L, uint64_t *packed, int32_t eos_id,
                   const int32_t *remaining_tokens /* [B] max_tokens - accepted so far */,
                   int64_t *new_tokens /* [B, L] */, int64_t *next_draft /* [B, L] */,
                   const int64_t *pad_stream, int64_t pad_stream_len, int64_t *pad_cursor,
                   jf_engine_row *rows, void *stream);

/* ---------------------------------------------------------------------------------------------
 * The loop around the engine steps (f3): what JD:609-710 / JDN:581-639 do on the host between two forwards of a batch —
 * append the committed tokens to the request, advance its length, its token budget and the cached length, lay out the
 * next forward — as ONE small launch behind jf_engine_step / jf_rs_step, on device arrays, with one record per
 * iteration in mapped host memory (the mailbox of jf_mb_loop: jf_host_alloc / jf_mailbox_wait).  The caller keeps the
 * draft as one [B, L] device tensor (the step's next_draft IS the next draft) and reads nothing else back.
 *
 *   rows / tokens   the records and the committed tokens [B, L] the step in front has written (kind 0: jf_engine_row +
 *                   new_tokens of jf_engine_step, n = n_new; kind 1: jf_rs_row + committed of jf_rs_step, n = n_committed)
 *   remaining [B]   in/out: -= n (the steps read it: max_tokens - accepted so far, JD:659-669)
 *   kv_start  [B]   in/out: += n — position of the row's seed = len(seq) - 1, i.e. the cached length the next forward
 *                   continues from (the reference trims the cache back to len(seq), BM:534-564, and re-forwards the seed, MR:1221)
 *   positions [B,L] out: kv_start + j of the NEXT forward (MR:1221-1232)
 *   slot [B], ring [*, ring_cap], ring_len [*]: row b appends its n tokens to ring row slot[b] (slots stay put when the
 *                   caller compacts the batch after a request finished); a ring that is full drops the tokens and reports
 *                   the row in the header's error word
 *   cursors [n_cursors <= 3]: the steps' device stream cursors (pads | uniforms, bonus draws, pads), reported in the header
 *   mailbox         JF_EL_HDR + B words: [JF_EL_SEQ] = seq (written last: poll it), [JF_EL_ERROR] = a row + 1 whose ring was full,
 *                   [JF_EL_STEP_ERROR] = a row + 1 the step in front gave up on (its 2 s in-launch wait: rsv markers of the records),
 *                   [JF_EL_CURSORS + 2 i ..] = cursor i (low, high word); then per row
 *                   n | eos << 16 | active_next << 17 | (kind 0: acc_len == 1, the autoregressive fallback JD:619-631) << 18.
 *   flags           JF_MB_LOOP_PUBLISH_FENCE: publish behind a system-scope release fence (see jf_mb_loop.flags)
 */
enum { JF_EL_SEQ = 0, JF_EL_ERROR = 1, JF_EL_STEP_ERROR = 2, JF_EL_CURSORS = 4, JF_EL_HDR = 16 };
#define JF_EL_MAILBOX_INTS(B) (JF_EL_HDR + (B))
#define JF_EL_KIND_GREEDY 0
#define JF_EL_KIND_SAMPLING 1
typedef struct jf_engine_loop {
    int32_t B, L, kind, ring_cap;
    const void *rows;
    const int64_t *tokens;
    int32_t *remaining;
    int32_t *kv_start;
    int32_t *positions;
    const int32_t *slot;
    int64_t *ring;
    int32_t *ring_len;
    const int64_t *cursors;
    int32_t n_cursors;
    int32_t flags;
    int32_t *mailbox;
} jf_engine_loop;
JF_API int jf_engine_loop_commit(const jf_engine_loop *loop, int32_t seq, void *stream);

/* ---------------------------------------------------------------------------------------------
 * HF single-block step (a14): one iteration body of jacobi_forward_greedy (SB:197-273) after the forward, one launch.
 *   out [L] int64 (in/out): the forwarded draft; on a rejection it is overwritten with the next draft
 *       [greedy@mismatch] + greedy[raw : L-1] (SB:231-255), next_len = L - raw tokens;
 *   packed [L]: jf_argmax_partial of logits [L, V] (greedy[i] verifies out[i+1]; greedy[L-1] is the bonus, SB:258-264);
 *   total / cap / acc_buf: tokens accepted so far in this call, capacity and storage of accepted_n_gram (SB:145: the
 *       reference writes into the preallocated input tensor; writes past its end are dropped);
 *   kv_before: cache length before the forward.  desc->kv_len is the length the reference trims the cache to, including
 *       its EOS quirk (desired_len = total_accepted without the prompt, SB:221-225).  packed is re-zeroed.
 */
typedef struct jf_sb_desc {
    int32_t raw;         /* accepted incl. position 0, before the EOS cap (SB:199-202)  */
    int32_t num;         /* after the EOS cap (SB:204-211)                               */
    int32_t total;       /* total_accepted after this iteration                          */
    int32_t done;        /* the call returns here (EOS)                                  */
    int32_t next_token;  /* SB:235 / 260 / the EOS id                                    */
    int32_t kv_len;      /* committed cache length after the trims                       */
    int32_t next_len;    /* length of the next draft in `out` (0 when the call ends)     */
    int32_t eos;         /* EOS was inside the accepted prefix                           */
} jf_sb_desc;
JF_API int jf_sb_step(int64_t *out, int L, uint64_t *packed, int32_t eos_id, int32_t total, int32_t cap, int64_t *acc_buf,
               int32_t kv_before, jf_sb_desc *desc, void *stream);

/* The PAGED layout's index fill.  This package's default layout keeps one contiguous cache row per request and never calls it;
 * with Config.kv_cache_layout = "paged" (engine/model_runner.py: the reference's memory model — a pool of blocks, block tables,
 * slot mappings) every Jacobi step's forward does: from host lists through ops.PagedFill.fill (the callback contract), from the
 * chunk loop's DEVICE lengths through PagedFill.fill_device (no host copy).  A reference-side caller that keeps inference_engine's
 * paged cache and varlen attention binds it the same way (route 1).
 * Caller side of the batched forward when the KV cache is PAGED as in the reference
 * (MR:1204-1265 "jacobi.buffer_fill" + _get_slot_mapping_pattern MR:965-986): for B sequences of
 * committed length S_i (seq_len, >= 1) and a draft [B, L] (column 0 = the cached seed), fill
 *   input_ids [B*L] int64, positions [B*L] int64 (S_i - 1 + j),
 *   slot_mapping [B*L] int32 = block_tables[i, (S_i-1+j) / block_size] * block_size + (S_i-1+j) % block_size,
 *   cu_seqlens_q [B+1] (i*L), cu_seqlens_k [B+1] (prefix sums of S_i - 1 + L), cache_seqlens [B] (S_i - 1)
 * in one launch (the reference: a Python loop over the batch with ~8 launches per sequence).
 * block_tables [B, max_cols] int32, -1 = no block.  err (nullable, zero it first): first row + 1
 * with S < 1 or a position without a block (the reference raises ValueError / RuntimeError there).
 */
JF_API int jf_engine_fill(const int64_t *draft, int B, int L, const int32_t *seq_len,
                   const int32_t *block_tables, int max_cols, int block_size, int64_t *input_ids,
                   int64_t *positions, int32_t *slot_mapping, int32_t *cu_seqlens_q,
                   int32_t *cu_seqlens_k, int32_t *cache_seqlens, int32_t *err, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Non-greedy verify (a19), JDN = inference_engine/engine/jacobi_decoding_nongreedy.py.
 *
 * The reference builds its target distribution IN THE DTYPE OF THE LOGITS (JDN:64-70 has no .float(); the engine's
 * logits are bf16, MR:1382), so dtype selects the arithmetic of every jf_rs_* call:
 *   JF_F32   xs = fl32(x / T), p = softmax(xs) rounded once to float32.
 *   JF_BF16  xs = bf16(float(x) / float(T)) (one rounding of the float32 quotient; T == 1 leaves x as it is),
 *            p = softmax(xs) rounded once to bf16: torch's rounding points.
 *            `u < p`, the inverse-CDF draws and the masked argmax all use the ROUNDED p.
 *   "softmax" is the EXACT quotient exp(xs - max xs) / sum exp(xs - max xs) (torch's float32 kernels approximate it to an
 *   ulp of float32, which after the bf16 rounding is one ulp on ~1 % of the entries): jf_rs_step / jf_rs_onpolicy_step
 *   evaluate every probability a decision depends on in float64 (accept tests against the float32 row sum with a proven
 *   error band, the row's float64 sum where that cannot decide; rejected rows entirely in float64), so token ids, draw counts
 *   and stream cursors equal the definition bit for bit (oracle/jacobi_oracle.py: exact_softmax_rows).  Rows holding NaN /
 *   +inf keep the plain float32 formula (NaN where torch's softmax is NaN).
 *
 * jf_rs_probs: fused softmax-gather + argmax over logits [R, V] read once.  For row r:
 *   p_draft[r] = p[draft_next[r]] (JDN:65-70, 328) as far as one pass over the logits can know it: the float64 exp of the
 *   gathered logit over the float32 row sum, whose relative error is below ~4e-5 (+ 3.5e-7 |max xs|).  JF_F32: that value;
 *   the steps test u against the band around it.  JF_BF16: a float holding a bf16 value — the LOWER of the two roundings the
 *   band allows; when they differ (the float32 sum cannot decide the rounding, ~1 % of the rows) the SIGN BIT is set and the
 *   upper candidate is the next bf16 (|p_draft| is the probability either way).  An accept test whose uniform falls between
 *   the candidates is re-decided by the step with the row's float64 sum.
 *   row_max[r] = max xs (exact), row_sumexp[r] = sum exp(xs - max) in float32 and the packed argmax of the RAW logits (next
 *   draft, JDN:446/619).  packed must be zero on entry.
 */
JF_API int jf_rs_probs(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride,
                const int64_t *draft_next, float temperature, float *p_draft, float *row_max,
                float *row_sumexp, uint64_t *packed, void *workspace, size_t workspace_bytes, void *stream);

/* (a19) top_k / top_p on the target distribution — _apply_top_k + _apply_top_p of _build_target_probs, JDN:72-123 (the reference
 * reads both with getattr(sp, ...): they exist only on request objects a caller planted them on).  Call AFTER jf_rs_probs on the
 * same rows.  The filtered, renormalised distribution of a row is a function of each element's exactly rounded probability and
 * its id, fixed by ONE small record per row — the tensor itself (the reference's `probs`, 0.6 GB at 64 x 31 rows) is never
 * written; jf_rs_step / jf_rs_onpolicy_step evaluate the function on the few rows they read again:
 *
 *     p    = softmax(xs)[id] rounded once to the dtype, with the float64 row sum `sum` (the definition above)
 *     y    = top-k on:  (bits(p) > cut1 || (bits(p) == cut1 && id <= tie1)) ? p / s1 : 0        else p
 *     out  = top-p on:  (bits(y) > cut2 || (bits(y) == cut2 && id <= tie2)) ? y / s2 : 0        else y
 *
 * (quotients in the dtype's arithmetic: float32 division, rounded again for bf16).  A cut is the smallest kept value of its stage,
 * `tie` the last kept id among the ids that hold exactly that value (they are kept in id order: V - 1 = all of them, -1 = none).
 * What torch leaves to its kernels is defined: equal probabilities are ordered by token id, sums are exact sums rounded once
 * (oracle/jacobi_oracle.py filter_probs_row; tests/golden/filter_vectors.json pins it against the reference's tensors).
 *   filt [R]       out: the records.
 *   p_draft [R]    out: the FINAL filtered probability of draft_next[r] (no rounding candidates).
 *   row_max / row_sumexp [R]  in: jf_rs_probs'; out: marked (+inf / -1) — the steps then take p_draft as it is; the records carry
 *                  the rows' statistics from here on (row_max, sum).
 *   workspace      jf_rs_filter_workspace_bytes(dtype, R, V) bytes, 16-byte aligned: none for bf16 rows of up to 524 288 ids (a
 *                  workgroup per row keeps the COUNT of every bf16 pattern of the row's scaled logits in LDS — the probability is a
 *                  monotone function of the pattern, so cuts, sums and tie groups are sums over <= 65 536 counters — and reads the
 *                  row a second time only to find the last kept id of a tie group); float32 rows use R x V x 4 bytes of scratch.
 * top_k <= 0 or >= V, top_p <= 0 or >= 1: that stage is off (JDN:75, 95-96); top_p is compared AS GIVEN (a double): a value that
 * a float would round to 0 or 1 keeps its stage on.  packed / the greedy next-draft tail of jf_rs_probs are untouched (argmax
 * of the LOGITS, JDN:446).
 * jf_rs_filter_expand: the dense tensor of the records, probs [R, V] in the dtype of the logits (row stride V) — what the
 * reference's `probs` holds; for callers that want it, and the parity tests. */
#define JF_RS_FILT_TOPK 1u
#define JF_RS_FILT_TOPP 2u
typedef struct jf_rs_filter_row {
    double   sum;        /* float64 sum of exp(xs - row_max); 0: the row filters to zeros (NaN / inf logits)                  */
    float    row_max;    /* max xs                                                                                              */
    float    x_keep;     /* no id whose scaled logit lies below this is kept (-inf: not known): consumers skip their exps          */
    uint32_t cut1;       /* top-k: bits (as a float32 value) of the smallest kept probability                                    */
    int32_t  tie1;       /*        last kept id among the ids AT cut1                                                            */
    float    s1;         /*        sum of the kept probabilities, rounded once to the dtype, >= 1e-12 in the dtype (JDN:83)      */
    uint32_t cut2;       /* top-p: the same on y                                                                                 */
    int32_t  tie2;
    float    s2;
    uint32_t flags;      /* JF_RS_FILT_TOPK | JF_RS_FILT_TOPP: the stages that are on                                            */
    uint32_t rsv;
} jf_rs_filter_row;
JF_API size_t jf_rs_filter_workspace_bytes(int dtype, int64_t R, int64_t V);
JF_API int jf_rs_filter(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride, const int64_t *draft_next,
                 float temperature, int32_t top_k, double top_p, jf_rs_filter_row *filt, float *p_draft, float *row_max,
                 float *row_sumexp, void *workspace, size_t workspace_bytes, void *stream);
JF_API int jf_rs_filter_expand(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride, float temperature,
                        const jf_rs_filter_row *filt, void *probs, void *stream);
JF_API size_t jf_rs_workspace_bytes(int64_t R, int64_t V);   /* per-chunk (max, sum-exp) partials */

typedef struct jf_rs_row {
    int32_t n_committed;   /* tokens committed (accepted drafts + bonus), >= 1          */
    int32_t eos;           /* EOS committed                                              */
    int32_t reject_pos;    /* first rejected position, -1 = whole block accepted         */
    int32_t n_bonus_draws; /* residual-sampling draws used (<= 16)                        */
    int32_t n_uniforms;    /* accept/reject uniforms consumed                             */
    int32_t n_pads;        /* random pads consumed by the next draft                      */
    int32_t active_next;   /* row keeps decoding                                          */
    int32_t rsv;
} jf_rs_row;

/* jf_rs_step: the sequential accept/reject of every row of a batch (JDN:581-639).
 *   draft [B, L]; logits [B*(L-1), V]; p_draft/row_max/row_sumexp/packed [B*(L-1)] from jf_rs_probs.
 *   Position t of row b is accepted iff u < p_draft (JDN:328-340); on the first rejection a bonus token != proposed is
 *   drawn by inverse CDF of p (float64 running sum in vocabulary order: smallest index whose sum exceeds u * total) with
 *   up to 16 draws (JDN:135-146), then argmax of the masked distribution (JDN:147-153).  Randomness is injected:
 *   u_stream / bonus_stream (floats in [0,1)) and pad_stream (token ids) are consumed cyclically from *cursor in row
 *   order, exactly where the reference calls torch.rand / torch.multinomial / torch.randint.
 *   committed [B, L], next_draft [B, L] (JDN:444-466), rows [B].  packed is re-zeroed.
 *   workspace: jf_rs_step_workspace_bytes(B) bytes (float64 sums of the rejected rows per segment and per wave-tile, a row
 *   list and the hand-off words of the one-launch step: ~9 KB per row), 16-byte aligned, ZERO before its first use (one
 *   torch.zeros at start-up); the calls themselves never need it re-zeroed (the hand-off words carry a per-call generation
 *   number).
 *   Batches of at most 128 rows (B * (L-1) <= 4096) run as ONE launch when the device keeps enough workgroups of it resident
 *   (asked of the runtime): accept walk, row sums, draw counting, bonus walks and the finish are roles of one kernel whose
 *   in-kernel waits are bounded (2 s; a timeout is reported as JF_E_LAUNCH in rows[0].rsv, which the caller clears).
 *   JF_RS_FUSED=0 selects the six launches that larger batches use.
 */
JF_API size_t jf_rs_step_workspace_bytes(int64_t rows);   /* rows = B (jf_rs_step) or R (jf_rs_onpolicy_step) */
JF_API int jf_rs_step(const void *logits, int dtype, int64_t V, int64_t row_stride, const int64_t *draft, int B, int L,
               const float *p_draft, const float *row_max, const float *row_sumexp, uint64_t *packed,
               float temperature, int32_t eos_id, const int32_t *remaining_tokens,
               const float *u_stream, int64_t u_len, int64_t *u_cursor,
               const float *bonus_stream, int64_t bonus_len, int64_t *bonus_cursor,
               const int64_t *pad_stream, int64_t pad_len, int64_t *pad_cursor,
               int64_t *committed, int64_t *next_draft, jf_rs_row *rows,
               void *workspace, size_t workspace_bytes, const jf_rs_filter_row *filt /* nullable: jf_rs_filter's records of the B*(L-1) rows */,
               void *stream);

/* ---------------------------------------------------------------------------------------------
 * On-policy rollout step, JDO = inference_engine/engine/jacobi_decoding_nongreedy_on_policy.py.
 * One sequence, R = proposed tokens of the current block that are not accepted yet; logits [R, V] and
 * p_draft / row_max / row_sumexp / packed [R] from jf_rs_probs(draft_next = proposed).
 *   verify  (JDO:270-327): accept proposed[t] iff u < p_draft[t]; first rejection -> bonus != proposed[t]
 *           (inverse CDF, <= 16 draws, then masked argmax: JDO:157-168) and stop; a committed token in
 *           stop_ids[n_stop] ends the block (stop_hit).
 *   redraft (JDO:465-477): if not stopped and n_committed < R, rows n_committed..R-1 each draw one sample
 *           from p[row] (inverse CDF as above) -> redraft[row] (the block's new guesses).
 * u_stream feeds the accept tests (torch.rand), m_stream every multinomial draw (bonus draws first, then
 * the re-draft rows in order), both consumed cyclically from *cursor, which is advanced.  packed is re-zeroed.
 */
typedef struct jf_op_row {
    int32_t n_committed;     /* accepted proposals + bonus, >= 1                     */
    int32_t stop_hit;        /* a stop token was committed                           */
    int32_t reject_pos;      /* first rejected position, -1 = all accepted           */
    int32_t n_bonus_draws;   /* multinomial draws used by the bonus (<= 16)          */
    int32_t n_uniforms;      /* accept/reject uniforms consumed                      */
    int32_t n_redraft;       /* rows re-drafted = R - n_committed, or 0              */
    int32_t redraft_base_lo; /* m_stream index of the first re-draft draw (64 bit)   */
    int32_t redraft_base_hi;
} jf_op_row;

JF_API int jf_rs_onpolicy_step(const void *logits, int dtype, int64_t V, int64_t row_stride, const int64_t *proposed, int R,
                        const float *p_draft, const float *row_max, const float *row_sumexp, uint64_t *packed,
                        float temperature, const int32_t *stop_ids, int n_stop,
                        const float *u_stream, int64_t u_len, int64_t *u_cursor,
                        const float *m_stream, int64_t m_len, int64_t *m_cursor,
                        int64_t *committed /* [R] */, int64_t *redraft /* [R] */, jf_op_row *row,
                        void *workspace /* jf_rs_step_workspace_bytes(R) */, size_t workspace_bytes,
                        const jf_rs_filter_row *filt /* nullable: jf_rs_filter's records of the R rows */, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* JACOBIFORCING_H */

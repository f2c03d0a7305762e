// jf_engine.hip — (a15) the engine decoder's single-block step and (a16) the paged-KV index fill of its caller.
#include "jf_common.h"
#include <atomic>
#include <cstdlib>

// ------------------------------------------------------------------------------------------------
// engine single-block step (JD:567-710): one wavefront per row, then one small launch hands out the pads
// ------------------------------------------------------------------------------------------------
// Rows are independent except for the pad stream, which the reference consumes in row order (JD:705-707 calls
// torch.randint per row): launch 1 runs every row on its own wavefront (accept scan = one ballot per 64 tokens), launch 2
// turns the rows' pad counts into stream offsets with a wavefront scan and fills all pads in parallel.  A dependent
// launch boundary (~1.5 us) is cheaper than an in-kernel agent-scope fence + arrival counter (~3.5 us per workgroup) — round 1's
// measurement; the one-word hand-off of engine_step_kernel below needs neither fence nor counter and replaces the pair.
__global__ __launch_bounds__(64) void engine_rows_kernel(const int64_t *draft, int L, const unsigned long long *packed, int eos_id,
                                                          const int32_t *remaining, int64_t *new_tokens, int64_t *next_draft,
                                                          jf_engine_row *rows) {
    const int b = blockIdx.x;
    const unsigned long long *pk = packed + (int64_t)b * (L - 1);
    auto G = [pk](int i) { return jfmb::decode_packed(pk[i]); };
    jfmb::EngineRowOut o = jfmb::engine_row_body(DevLanes{}, draft + (int64_t)b * L, L, G, eos_id, remaining[b],
                                                 new_tokens + (int64_t)b * L, next_draft + (int64_t)b * L);
    if (threadIdx.x == 0) {
        rows[b].acc_len = o.acc_len; rows[b].n_new = o.n_new; rows[b].eos = o.eos; rows[b].active_next = o.active_next;
        rows[b].n_pads = o.active_next ? (L - 1 - o.copy_len) : 0;
        rows[b].rsv[0] = o.copy_len; rows[b].rsv[1] = 0; rows[b].rsv[2] = 0;
    }
}

__global__ __launch_bounds__(256) void engine_pads_kernel(int B, int L, unsigned long long *packed, int64_t *next_draft,
                                                           const int64_t *pad_stream, int64_t pad_len, int64_t *pad_cursor,
                                                           jf_engine_row *rows) {
    __shared__ int s_total;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 64) {                                         // exclusive scan of n_pads in row order, 64 rows per pass
        int run = 0;
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int b = b0 + lane;
            const int np = b < B ? rows[b].n_pads : 0;
            int x = np;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(x, off, 64);
                if (lane >= off) x += o;
            }
            if (b < B) rows[b].rsv[1] = run + x - np;
            run += __shfl(x, 63, 64);
        }
        if (lane == 0) s_total = run;
    }
    __syncthreads();
    const int64_t base = *pad_cursor;
    for (int64_t idx = tid; idx < (int64_t)B * (L - 1); idx += 256) {
        const int b = (int)(idx / (L - 1)), i = (int)(idx - (int64_t)b * (L - 1));
        const int np = rows[b].n_pads, cl = rows[b].rsv[0];
        if (i < np) {
            const int64_t k = base + rows[b].rsv[1] + i;
            next_draft[(int64_t)b * L + 1 + cl + i] = pad_stream[pad_len > 0 ? (k % pad_len) : 0];
        }
        packed[idx] = 0ull;                                  // consumed by engine_rows_kernel: ready for the next argmax
    }
    __syncthreads();
    for (int b = tid; b < B; b += 256) rows[b].rsv[1] = 0;
    if (tid == 0) *pad_cursor = base + s_total;
}

// The two launches as ONE (rows up to ENGINE_ONE_LAUNCH_ROWS): workgroups [0, B) are the rows (wavefront 0 of each), workgroup B
// hands out the pads.  A row's hand-off is one 8-byte word — (generation << 32) | n_pads << 16 | copy_len — stored into its
// own record (rsv[1..2]): the word is payload and arrival flag at once, so there is nothing to order (the same hand-off as the
// convergence launch's result slots, jf_multiblock.hip).  The pad workgroup has the highest block id and only waits for lower
// ones: they are dispatched before it, no residency assumption.  It re-zeroes packed[] after every row has published, i.e.
// after every row has read its greedy tokens.
constexpr int ENGINE_ONE_LAUNCH_ROWS = 2048;
__global__ __launch_bounds__(256) void engine_step_kernel(const int64_t *draft, int B, int L, unsigned long long *packed, int eos_id,
                                                           const int32_t *remaining, int64_t *new_tokens, int64_t *next_draft,
                                                           const int64_t *pad_stream, int64_t pad_len, int64_t *pad_cursor,
                                                           jf_engine_row *rows, uint32_t gen) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b < B) {
        if (tid >= 64) return;
        const unsigned long long *pk = packed + (int64_t)b * (L - 1);
        auto G = [pk](int i) { return jfmb::decode_packed(pk[i]); };
        jfmb::EngineRowOut o = jfmb::engine_row_body(SoloWaveLanes{}, draft + (int64_t)b * L, L, G, eos_id, remaining[b],
                                                     new_tokens + (int64_t)b * L, next_draft + (int64_t)b * L);
        if (tid == 0) {
            const int np = o.active_next ? (L - 1 - o.copy_len) : 0;
            rows[b].acc_len = o.acc_len; rows[b].n_new = o.n_new; rows[b].eos = o.eos; rows[b].active_next = o.active_next;
            rows[b].n_pads = np;                              // (rsv[0] stays the pad workgroup's: the timeout marker below)
            __hip_atomic_store((unsigned long long *)__builtin_assume_aligned(&rows[b].rsv[1], 8),
                               ((unsigned long long)gen << 32) | ((unsigned long long)np << 16) | (unsigned long long)o.copy_len,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    // (dynamic LDS sized by B — 2 B ints: the launch's LDS request applies to the row workgroups too)
    extern __shared__ int s_dyn[];
    int *s_w = s_dyn, *s_off = s_dyn + B;                     // n_pads << 16 | copy_len per row; pad-stream offset per row
    __shared__ int s_total, s_bad;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = tid; r < B; r += 256) {                      // wait for every row's word (bounded: 2 s of the 100 MHz clock)
        unsigned long long w;
        unsigned spins = 0;
        bool ok = true;
        while (((w = __hip_atomic_load((const unsigned long long *)__builtin_assume_aligned(&rows[r].rsv[1], 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != gen) {
            __builtin_amdgcn_s_sleep(4);
            if ((++spins & 255u) == 0u && __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { ok = false; break; }
        }
        if (!ok) {                                            // the row never published: report it, touch nothing it may still be reading
            w = 0ull;
            s_bad = 1;
            // in a word NO row workgroup writes (the record's rsv[0]): a row that was slow, not hung, rewrites all of its own
            // fields when it does run and would erase a marker kept in one of them
            __hip_atomic_store(&rows[r].rsv[0], (int32_t)JF_E_LAUNCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_w[r] = (int)(w & 0xFFFFFFFFull);
    }
    __syncthreads();
    if (s_bad) return;                                        // no pads, no re-zeroing, no cursor: the host raises on the marker
    if (tid < 64) {                                           // exclusive scan of n_pads in row order, 64 rows per pass
        int run = 0;
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int r = b0 + tid;
            const int np = r < B ? (s_w[r] >> 16) & 0xFFFF : 0;
            int x = np;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(x, off, 64);
                if (tid >= off) x += o;
            }
            if (r < B) s_off[r] = run + x - np;
            run += __shfl(x, 63, 64);
        }
        if (tid == 0) s_total = run;
    }
    __syncthreads();
    const int64_t base = *pad_cursor;
    for (int64_t idx = tid; idx < (int64_t)B * (L - 1); idx += 256) {
        const int r = (int)(idx / (L - 1)), i = (int)(idx - (int64_t)r * (L - 1));
        const int wv = s_w[r];
        if (i < ((wv >> 16) & 0xFFFF)) {
            const int64_t k = base + s_off[r] + i;
            next_draft[(int64_t)r * L + 1 + (wv & 0xFFFF) + i] = pad_stream[pad_len > 0 ? (k % pad_len) : 0];
        }
        packed[idx] = 0ull;                                  // every row has read its slice: ready for the next argmax
    }
    if (tid == 0) *pad_cursor = base + s_total;
}

extern "C" int jf_engine_step(const int64_t *draft, int B, int L, uint64_t *packed, int32_t eos_id, const int32_t *remaining_tokens,
                              int64_t *new_tokens, int64_t *next_draft, const int64_t *pad_stream, int64_t pad_stream_len,
                              int64_t *pad_cursor, jf_engine_row *rows, void *stream) {
    if (B <= 0) return JF_OK;
    if (L < 2) return fail(JF_E_INVALID, "Draft must have at least 2 tokens (seed + 1 speculative)");   // MR:1144-1145
    if (!draft || !packed || !remaining_tokens || !new_tokens || !next_draft || !pad_cursor || !rows || (!pad_stream && pad_stream_len > 0))
        return fail(JF_E_INVALID, "jf_engine_step: null pointer");
    hipStream_t s = (hipStream_t)stream;
    static const bool one_launch = [] { const char *e = getenv("JF_ENGINE_ONE_LAUNCH"); return !(e && e[0] == '0'); }();
    if (one_launch && B <= ENGINE_ONE_LAUNCH_ROWS && L <= 0xFFFF && ((uintptr_t)rows % 8) == 0) {   // 32-byte records: rsv[1..2] is an aligned 8-byte word
        static std::atomic<uint32_t> generation{0};
        uint32_t gen = ++generation;
        if (gen == 0) gen = ++generation;                       // 0 is what a fresh record holds
        engine_step_kernel<<<B + 1, 256, (size_t)B * 2 * sizeof(int), s>>>(draft, B, L, (unsigned long long *)packed, eos_id, remaining_tokens, new_tokens, next_draft,
                                                 pad_stream, pad_stream_len, pad_cursor, rows, gen);
        return check_launch("engine_step_kernel");
    }
    engine_rows_kernel<<<B, 64, 0, s>>>(draft, L, (const unsigned long long *)packed, eos_id, remaining_tokens, new_tokens, next_draft, rows);
    engine_pads_kernel<<<1, 256, 0, s>>>(B, L, (unsigned long long *)packed, next_draft, pad_stream, pad_stream_len, pad_cursor, rows);
    return check_launch("engine_step kernels");
}

// ------------------------------------------------------------------------------------------------
// (f3) the loop around the engine steps: jf_engine_loop_commit (include/jacobiforcing.h), one workgroup behind the step
// ------------------------------------------------------------------------------------------------
struct BlockLanes {                                           // every lane of ONE workgroup (engine_loop_commit_body's policy)
    __device__ __forceinline__ int lane() const { return threadIdx.x; }
    __device__ __forceinline__ int count() const { return blockDim.x; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ void mail(int32_t *word, int32_t v) const { DevLanes{}.mail(word, v); }   // written through (jf_common.h)
    __device__ __forceinline__ void publish(int32_t *word, int32_t v, bool fence) const {
        if (fence) {                                          // JF_MB_LOOP_PUBLISH_FENCE: the formal order (DevLanes::publish)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wavefront's mailed words have left ...
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // ... before the word the host polls
    }
};

__global__ __launch_bounds__(256) void engine_loop_commit_kernel(jf_engine_loop lp, int32_t seq) {
    jfmb::engine_loop_commit_body(BlockLanes{}, lp, seq);
}

extern "C" int jf_engine_loop_commit(const jf_engine_loop *loop, int32_t seq, void *stream) {
    if (!loop) return fail(JF_E_INVALID, "jf_engine_loop_commit: null loop");
    const jf_engine_loop &l = *loop;
    if (l.B <= 0 || l.L < 2 || l.ring_cap <= 0) return fail(JF_E_INVALID, "jf_engine_loop_commit: B=%d L=%d ring_cap=%d", l.B, l.L, l.ring_cap);
    if (l.kind != JF_EL_KIND_GREEDY && l.kind != JF_EL_KIND_SAMPLING) return fail(JF_E_INVALID, "jf_engine_loop_commit: kind=%d", l.kind);
    if (!l.rows || !l.tokens || !l.remaining || !l.kv_start || !l.positions || !l.ring || !l.ring_len || !l.mailbox ||
        (l.n_cursors > 0 && !l.cursors) || l.n_cursors < 0 || l.n_cursors > 3)
        return fail(JF_E_INVALID, "jf_engine_loop_commit: null pointer / n_cursors=%d", l.n_cursors);
    engine_loop_commit_kernel<<<1, 256, 0, (hipStream_t)stream>>>(l, seq);
    return check_launch("engine_loop_commit_kernel");
}

// ------------------------------------------------------------------------------------------------
// (a14) HF single-block step (SB = modeling/cllm2_qwen2_modeling_kv_terminate_on_eos_improved.py:197-273): everything
// between two forwards of jacobi_forward_greedy in one launch, one descriptor for the host to read.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sb_step_kernel(int64_t *out, int L, unsigned long long *packed, int eos_id, int total,
                                                      int cap, int64_t *acc_buf, int kv_before, jf_sb_desc *desc) {
    jfmb::sb_step_body(DevLanes{}, out, L, (uint64_t *)packed, eos_id, total, cap, acc_buf, kv_before, desc);
}

extern "C" int jf_sb_step(int64_t *out, int L, uint64_t *packed, int32_t eos_id, int32_t total, int32_t cap, int64_t *acc_buf,
                          int32_t kv_before, jf_sb_desc *desc, void *stream) {
    if (L < 1) return fail(JF_E_INVALID, "jf_sb_step: L=%d", L);
    if (!out || !packed || !acc_buf || !desc || total < 0 || cap < 0 || kv_before < 0) return fail(JF_E_INVALID, "jf_sb_step: bad argument");
    sb_step_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out, L, (unsigned long long *)packed, eos_id, total, cap, acc_buf, kv_before, desc);
    return check_launch("sb_step_kernel");
}

// ------------------------------------------------------------------------------------------------
// (a16) paged-KV caller side: every index buffer of one batched Jacobi forward in one launch (MR:1204-1265)
// ------------------------------------------------------------------------------------------------
// One wavefront per sequence.  err (nullable) gets the first failing row + 1: S < 1 (MR:1222-1223) or a position whose
// block is beyond the table / unallocated (MR:1240-1247).
__global__ __launch_bounds__(64) void engine_fill_kernel(const int64_t *__restrict__ draft, int B, int L,
                                                          const int32_t *__restrict__ seq_len,
                                                          const int32_t *__restrict__ block_tables, int max_cols, int block_size,
                                                          int64_t *__restrict__ input_ids, int64_t *__restrict__ positions,
                                                          int32_t *__restrict__ slot_mapping, int32_t *__restrict__ cu_q,
                                                          int32_t *__restrict__ cu_k, int32_t *__restrict__ cache_seqlens,
                                                          int32_t *__restrict__ err) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const int S = seq_len[i];
    // cu_seqlens_k[i+1] = sum_{q<=i} (S_q - 1 + L)
    int part = 0;
    for (int q = lane; q <= i; q += 64) part += seq_len[q] - 1 + L;
    part = wave_sum_i32(part);
    if (lane == 0) {
        if (i == 0) { cu_q[0] = 0; cu_k[0] = 0; }
        cu_q[i + 1] = (i + 1) * L;
        cu_k[i + 1] = part;
        cache_seqlens[i] = S - 1;
    }
    bool bad = S < 1;
    const int32_t *bt = block_tables + (int64_t)i * max_cols;
    for (int j = lane; j < L; j += 64) {
        const int64_t o = (int64_t)i * L + j;
        const int pos = S - 1 + j;
        input_ids[o] = draft[o];
        positions[o] = pos;
        int slot = -1;
        if (pos >= 0) {
            const int blk = pos / block_size, off = pos - blk * block_size;
            const int id = blk < max_cols ? bt[blk] : -1;
            if (id >= 0) slot = id * block_size + off; else bad = true;
        }
        slot_mapping[o] = slot;
    }
    if (err && __ballot(bad) != 0ull && lane == 0) atomicCAS(err, 0, i + 1);
}

extern "C" int jf_engine_fill(const int64_t *draft, int B, int L, const int32_t *seq_len, const int32_t *block_tables,
                              int max_cols, int block_size, int64_t *input_ids, int64_t *positions, int32_t *slot_mapping,
                              int32_t *cu_seqlens_q, int32_t *cu_seqlens_k, int32_t *cache_seqlens, int32_t *err, void *stream) {
    if (B <= 0) return JF_OK;
    if (L < 2) return fail(JF_E_INVALID, "Draft must have at least 2 tokens (seed + 1 speculative)");   // MR:1144-1145
    if (!draft || !seq_len || !block_tables || !input_ids || !positions || !slot_mapping || !cu_seqlens_q || !cu_seqlens_k ||
        !cache_seqlens)
        return fail(JF_E_INVALID, "jf_engine_fill: null pointer");
    if (max_cols <= 0 || block_size <= 0) return fail(JF_E_INVALID, "jf_engine_fill: max_cols=%d block_size=%d", max_cols, block_size);
    engine_fill_kernel<<<B, 64, 0, (hipStream_t)stream>>>(draft, B, L, seq_len, block_tables, max_cols, block_size, input_ids,
                                                         positions, slot_mapping, cu_seqlens_q, cu_seqlens_k, cache_seqlens, err);
    return check_launch("engine_fill_kernel");
}


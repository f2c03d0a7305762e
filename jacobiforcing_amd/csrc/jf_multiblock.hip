// jf_multiblock.hip — (a1, a4-a12) kernels of the multiblock Jacobi state machine (jf_mb_core.h): one 64-lane wavefront
// per prompt.
#ifdef JF_EXP_MB_TRACE
// experiment build only (tools/mb_step_trace.py): shader-clock stamps of prompt 0's step, read back by jf_exp_read_trace
#include <hip/hip_runtime.h>
__device__ unsigned long long g_mb_trace[32];
#define JF_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_mb_trace[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#endif
#ifdef JF_EXP_VERIFY_TRACE
// experiment build only (tools/verify_trace.py): the state machine's phase stamps per stepper workgroup, 16 per prompt
#include <hip/hip_runtime.h>
__device__ unsigned long long g_mtrace[16 * 256];
#define JF_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 256) g_mtrace[16 * blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#endif
#include "jf_argmax_dev.h"
#include <hip/hip_ext.h>
#include <atomic>
#include <mutex>

// ------------------------------------------------------------------------------------------------
// multiblock state machine: one wavefront per prompt
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void mb_begin_kernel(int32_t *states, int64_t state_ints, jf_mb_params prm,
                                                       const int64_t *input_ids, const int32_t *kv_len, jf_mb_desc *desc,
                                                       int32_t *kv_len_out) {
    jfmb::mb_begin_body(DevLanes{}, blockIdx.x, states, state_ints, prm, input_ids, kv_len, desc, kv_len_out);
}
// publish: the loop API's summary for the host goes out at the START of this launch (prompt 0's wavefront), while the other
// wavefronts write the forward's inputs: 1 = header only (the steppers of the fused launch mailed their own descriptors),
// 2 = header + descriptor table + driver records
__global__ __launch_bounds__(64) void mb_pack_kernel(int32_t *states, int64_t state_ints, const jf_mb_desc *desc, int32_t Tpad,
                                                      int32_t t_align, int32_t t_cap, int64_t pad_fill, int32_t order,
                                                      int32_t cand_rows, jfmb::PackOut o, int32_t valid_align, jfmb::LoopDev lp,
                                                      int32_t publish) {
    const int P = publish ? (int)gridDim.x - 1 : (int)gridDim.x;            // a publishing launch carries one extra workgroup for it
    jfmb::mb_pack_body(DevLanes{}, blockIdx.x, P, states, state_ints, desc, Tpad, t_align, t_cap, pad_fill, order,
                       cand_rows, o, valid_align, lp, publish);
}
__global__ __launch_bounds__(64) void mb_step_kernel(int32_t *states, int64_t state_ints, unsigned long long *packed,
                                                      int64_t packed_len, jf_mb_desc *desc, jfmb::LoopDev lp, int has_loop, int fast) {
    JF_STAMP(0);
    jfmb::mb_step_body(DevLanes{}, blockIdx.x, states, state_ints, (uint64_t *)packed, packed_len, desc, lp, has_loop != 0, fast != 0);
    JF_STAMP(12);
}
#ifdef JF_EXP_MB_TRACE
extern "C" int jf_exp_read_trace(unsigned long long *out32) {
    return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_mb_trace), sizeof(unsigned long long) * 32);
}
#endif
__global__ __launch_bounds__(64) void mb_read_ret_kernel(const int32_t *states, int64_t state_ints, int64_t *ret,
                                                          int32_t ret_cap) {
    jfmb::mb_read_ret_body(DevLanes{}, blockIdx.x, states, state_ints, ret, ret_cap);
}

static int g_fast_path = -1;      // -1: not decided yet (JF_MB_FAST, default on)
static int fast_path() {
    if (g_fast_path < 0) { const char *e = getenv("JF_MB_FAST"); g_fast_path = (e && e[0] == '0') ? 0 : 1; }
    return g_fast_path;
}
extern "C" int jf_mb_set_fast_path(int on) { const int old = fast_path(); g_fast_path = on ? 1 : 0; return old; }

static int check_params(const jf_mb_params *p, const char *who) {
    if (!p) return fail(JF_E_INVALID, "%s: null params", who);
    if (p->n < 1 || p->n > 1024) return fail(JF_E_INVALID, "%s: n=%d out of range [1,1024]", who, p->n);
    if (p->K < 1) return fail(JF_E_INVALID, "%s: K=%d", who, p->K);
    if (p->pool_size < 0 || p->pool_size > 64) return fail(JF_E_INVALID, "%s: pool_size=%d out of range [0,64]", who, p->pool_size);
    if (p->max_iter < 0) return fail(JF_E_INVALID, "%s: max_iter=%d", who, p->max_iter);
    if (p->max_blocks > jfmb::MAX_NB || p->K > jfmb::MAX_NB) return fail(JF_E_INVALID, "%s: max_blocks=%d too large", who, p->max_blocks);
    return JF_OK;
}

extern "C" int64_t jf_mb_state_ints(const jf_mb_params *p) {
    if (check_params(p, "jf_mb_state_ints")) return -1;
    return jfmb::make_layout(p->n, p->K, p->pool_size, p->max_blocks).total;
}
extern "C" int32_t jf_mb_max_rows(const jf_mb_params *p) {
    if (check_params(p, "jf_mb_max_rows")) return -1;
    return jfmb::make_layout(p->n, p->K, p->pool_size, p->max_blocks).RMAX;
}
extern "C" int32_t jf_mb_max_tokens(const jf_mb_params *p) {
    if (check_params(p, "jf_mb_max_tokens")) return -1;
    return jfmb::make_layout(p->n, p->K, p->pool_size, p->max_blocks).TMAX;
}

extern "C" int jf_mb_begin(int32_t *states, int64_t state_ints, int P, const jf_mb_params *params, const int64_t *input_ids,
                           const int32_t *kv_len, jf_mb_desc *desc, void *stream) {
    if (P <= 0) return JF_OK;
    int rc = check_params(params, "jf_mb_begin");
    if (rc) return rc;
    if (!states || !input_ids || !kv_len) return fail(JF_E_INVALID, "jf_mb_begin: null pointer");
    if (state_ints < jf_mb_state_ints(params)) return fail(JF_E_INVALID, "jf_mb_begin: state block too small");
    mb_begin_kernel<<<P, 64, 0, (hipStream_t)stream>>>(states, state_ints, *params, input_ids, kv_len, desc, nullptr);
    return check_launch("mb_begin_kernel");
}

extern "C" int jf_mb_pack(int32_t *states, int64_t state_ints, int P, int32_t Tpad, int64_t pad_fill, int64_t *input_ids,
                          int32_t *positions, int32_t *row_prompt, int32_t *row_len, int32_t *valid_index,
                          int32_t valid_align, void *stream) {
    if (P <= 0) return JF_OK;
    if (!states || !input_ids || !positions || !row_prompt || !row_len) return fail(JF_E_INVALID, "jf_mb_pack: null pointer");
    if (Tpad <= 0) return fail(JF_E_INVALID, "jf_mb_pack: Tpad=%d", Tpad);
    const jfmb::PackOut o{input_ids, positions, row_prompt, row_len, valid_index, nullptr, nullptr};
    mb_pack_kernel<<<P, 64, 0, (hipStream_t)stream>>>(states, state_ints, nullptr, Tpad, 1, Tpad, pad_fill, 0, 1, o,
                                                      valid_align < 1 ? 1 : valid_align, jfmb::LoopDev{}, 0);
    return check_launch("mb_pack_kernel");
}

extern "C" int jf_mb_step(int32_t *states, int64_t state_ints, int P, uint64_t *packed, int64_t packed_len, jf_mb_desc *desc,
                          void *stream) {
    if (P <= 0) return JF_OK;
    if (!states || !packed) return fail(JF_E_INVALID, "jf_mb_step: null pointer");
    mb_step_kernel<<<P, 64, 0, (hipStream_t)stream>>>(states, state_ints, (unsigned long long *)packed, packed_len, desc,
                                                      jfmb::LoopDev{}, 0, fast_path());
    return check_launch("mb_step_kernel");
}

extern "C" int jf_mb_read_ret(const int32_t *states, int64_t state_ints, int P, int64_t *ret, int32_t ret_cap, void *stream) {
    if (P <= 0) return JF_OK;
    if (!states || !ret || ret_cap <= 0) return fail(JF_E_INVALID, "jf_mb_read_ret: bad argument");
    mb_read_ret_kernel<<<P, 64, 0, (hipStream_t)stream>>>(states, state_ints, ret, ret_cap);
    return check_launch("mb_read_ret_kernel");
}

// ------------------------------------------------------------------------------------------------
// jf_mb_verify — the whole convergence check of one iteration in ONE launch (MB:473-486 + MB:487-721):
//
//   workgroups [0, P)        one STEPPER per prompt.  While the logits stream, its 256 threads copy the live part of the
//                            prompt's state block into a compact LDS image (same Machine, smaller Layout); then wavefront 0
//                            polls the prompt's argmax result slots (8-byte agent-scope loads) into LDS until all have
//                            arrived, runs Machine::step entirely on LDS, and writes the image + descriptor back.
//                            A step that does not fit the compact capacities (runaway block lists, Q3/Q4) is redone on
//                            the HBM block: nothing was written before that, so the result is the same.
//   workgroups [P, ...)      the argmax items of jf_argmax_scatter / _partial (jf_argmax_dev.h).  Every (row, chunk) item owns
//                            one result slot, packed[position * chunks + chunk], and stores its (key, ~index) word there.
//
// Hand-off: the result word is its own arrival flag.  The slots are zero on entry, every real key is >= 0x007FFFFF, so a
// non-zero slot has arrived; payload and flag being ONE 8-byte agent-scope store there is nothing to order — the item does
// not drain its memory counter, adds to no counter and ends with the store in flight (round 2/3a: atomicMax, s_waitcnt
// vmcnt(0), add on a per-prompt counter, and a gather after the count was seen: ~3 us more behind the last row,
// profiles/verify_slots_ab_r03.txt).  The stepper's wavefront 0 polls the slots of its own positions (agent-scope loads,
// s_sleep between rounds, bounded by a wall-clock limit), keeps the maximum over a position's chunks — the poll is the
// gather — and wavefronts 1-3 re-zero the slots after the step, so the call leaves packed[] zero.
// Steppers only WAIT for item workgroups and never the other way round, and they are the lowest block ids (dispatched
// first), so an item workgroup can always be scheduled: no residency assumption, no deadlock.
// ------------------------------------------------------------------------------------------------
struct VerifyArgs {
    ArgmaxArgs am;
    int32_t *states;
    int64_t state_ints;
    int P;
    int64_t packed_len;            // Rtot * Tpad: positions of the forward = slots per chunk
    jf_mb_desc *desc;
    int32_t Tpad;
    int32_t compacted;             // 1: logits rows follow valid_index (B*T per prompt); 0: the Rtot x Tpad rectangle
    int32_t lds_ints;              // ints of dynamic LDS available for (compact image + greedy tokens); 0 = step on HBM
    int32_t fast;                  // Machine::step_fast allowed (jf_mb_set_fast_path)
    int32_t has_loop;              // jf_mb_loop_iterate: kv_len / resident driver / mailbox (lp) apply
    int64_t items;                 // (row, chunk) items of the launch (wave items in wave mode)
    jfmb::LoopDev lp;
};

#ifdef JF_EXP_VERIFY_TRACE
// experiment build only (tools/verify_trace.py): wall-clock stamps (100 MHz) of the launch — [0] first item start, [1] last
// item end, then 8 per stepper: start, image copied, rows arrived, tokens gathered, stepped, written back (wavefronts 1-3),
// descriptor out, end (summary published when this was the last prompt)
__device__ unsigned long long g_vtrace[2 + 8 * 256];
__device__ unsigned long long g_vitems[2 * 8192];       // (start, end) per item workgroup: plain stores, no atomics in the stream
#define JF_VSTAMP(p, k) do { if (threadIdx.x == ((k) == 5 ? 64 : 0) && (p) < 256) g_vtrace[2 + 8 * (p) + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" __attribute__((visibility("default"))) int jf_exp_read_vtrace(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vtrace), sizeof(unsigned long long) * (size_t)n);
}
extern "C" __attribute__((visibility("default"))) int jf_exp_read_mtrace(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mtrace), sizeof(unsigned long long) * (size_t)n);
}
extern "C" __attribute__((visibility("default"))) int jf_exp_read_vitems(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vitems), sizeof(unsigned long long) * (size_t)n);
}
extern "C" __attribute__((visibility("default"))) int jf_exp_reset_vtrace(void) {
    static unsigned long long z[2 * 8192];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_vitems), z, sizeof(z));
    static unsigned long long init[2 + 8 * 256];
    for (int i = 0; i < 2 + 8 * 256; ++i) init[i] = 0ull;
    init[0] = ~0ull;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_vtrace), init, sizeof(init));
}
#else
#define JF_VSTAMP(p, k) do { } while (0)
#endif

struct Lanes192 {                  // wavefronts 1-3 of a stepper workgroup (write-back while wavefront 0 publishes)
    __device__ __forceinline__ int lane() const { return threadIdx.x - 64; }
    __device__ __forceinline__ int count() const { return 192; }
};
struct Lanes256 {                  // all four wavefronts of a stepper workgroup (copy-in only)
    __device__ __forceinline__ int lane() const { return threadIdx.x; }
    __device__ __forceinline__ int count() const { return 256; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
};

__device__ __forceinline__ unsigned long long ld_agent_u64(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int VERIFY_LDS_HDR = 32;                               // ints in front of the compact image (descriptor + flags)
constexpr unsigned long long VERIFY_WAIT_TICKS = 200000000ull;   // 2 s of the 100 MHz constant clock: never hang the GPU

// Maximum over the chunk slots of one position (adjacent words); false while one of them is still zero.  Sixteen loads are
// issued before the first is looked at: a position of a small forward has up to 16 chunks, and one dependent round trip
// per chunk was 3 us between the last item and the step at one prompt (round 3: batches of eight, two round trips at 16 chunks).
__device__ __forceinline__ bool slots_max(const unsigned long long *q, int cpr, unsigned long long &mx) {
    mx = 0ull;
    bool zero = false;
    for (int c0 = 0; c0 < cpr; c0 += 16) {
        unsigned long long v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = c0 + u < cpr ? ld_agent_u64(q + c0 + u) : ~0ull;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (c0 + u < cpr) { zero |= v[u] == 0ull; mx = v[u] > mx ? v[u] : mx; }
    }
    return !zero;
}

__device__ __forceinline__ void verify_stepper(const VerifyArgs &a, int p, int32_t *smem) {   // one call site; as a real call the whole launch pays its register budget
    using namespace jfmb;
    int32_t *G = a.states + (int64_t)p * a.state_ints;
    const Layout LG = layout_of(G);
    const Layout LC = compact_layout(LG, G[H_K]);
    const int B = G[H_B], T = G[H_T];
    const PackedRows rows{(const uint64_t *)a.am.packed, G[H_ROW_BASE], G[H_CAND_BASE], G[H_TPAD], a.packed_len};
    const int ng = B * T;                                        // greedy tokens this prompt consumes
    const bool was_done = G[H_DONE] != 0;
    const LoopDev &lp = a.lp;
    const bool has_loop = a.has_loop != 0;
    JF_VSTAMP(p, 0);
    // dynamic LDS: [0,16) descriptor, [16,32) flags, then the compact image, then the greedy tokens
    jf_mb_desc *s_desc = (jf_mb_desc *)smem;
    int32_t *img = smem + VERIFY_LDS_HDR;
    // ---- while the logits stream: compact image of the live state ---------------------------------
    bool use_lds = a.lds_ints >= VERIFY_LDS_HDR + LC.total + ng && !G[H_DONE] && !G[H_ERR];
    if (use_lds) {
        const bool ok = state_to_compact(Lanes256{}, G, LG, img, LC);
        if (threadIdx.x == 0) smem[16] = ok ? 1 : 0;
    }
    __syncthreads();
    if (use_lds) use_lds = smem[16] != 0;
#ifdef JF_EXP_STEP_TWICE
    // experiment (tools/verify_trace.py --twice): the step runs twice through the SAME instructions, first on a second copy of
    // the image, to tell instruction-fetch misses from dependent LDS latency in the step's 4-6 us
    int32_t *img2 = img + ((LC.total + LC.RMAX * LC.TMAX + 3) & ~3);
    if (use_lds) { for (int i = threadIdx.x; i < LC.total; i += AM_TPB) img2[i] = img[i]; }
    __syncthreads();
#endif
    JF_VSTAMP(p, 1);
    // Wavefront 0 is this prompt's state machine.  The other three park at the barrier below (a parked wavefront issues
    // nothing) and come back for the write-back, which is store-issue bound: 256 lanes instead of 64.
    jf_mb_desc *dg = a.desc ? a.desc + p : nullptr;
    // (Round 5, the launch's tail, measured and NOT adopted: the wavefronts that park at the barrier below reading ahead what a
    //  call end will read — driver header, the next draft's draw words, the text — so that drv_call_end's three dependent round
    //  trips hit the L2: scripted window 83.3-83.6 us against 82.7-87.9 without, headline window 65.3 against 64.8 —
    //  profiles/verify_tail_ab_r05.txt.  The headline pays for loads it rarely needs; the item is closed.)
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int32_t *gtok = img + LC.total;                          // [B, T] greedy tokens (LDS) when use_lds
        // ---- wait for this prompt's rows: poll their result slots; the poll is the gather -------------------------
        // compacted logits: the B*T draft-carrying positions have items; the rectangle: all B*Tpad (their slots must all
        // have been written before the re-zero below, also the ones nobody reads)
        const int tw = a.compacted ? T : (int)rows.tpad, nw = B * tw, cpr = a.am.chunks_per_row;
        const unsigned long long *pk = (const unsigned long long *)rows.pk;
        bool timed_out = false;
        if (nw > 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned spins = 0;
            for (;;) {
                bool miss = false;
                for (int i = lane; i < nw; i += 64) {
                    const int r = i / tw, t = i - r * tw;
                    const int64_t idx = rows.index(r, t);
                    int tok = -1;
                    if (idx >= 0 && idx < rows.plen) {
                        unsigned long long mx;
                        if (!slots_max(pk + idx * cpr, cpr, mx)) { miss = true; continue; }
                        tok = decode_packed(mx);
                    }
                    if (use_lds && t < T) gtok[r * T + t] = tok;
                }
                if (__ballot(miss) == 0ull) break;
                __builtin_amdgcn_s_sleep(8);                     // ~0.2 us between rounds: pollers must not load the memory system
                if ((++spins & 255u) == 0u && __builtin_amdgcn_s_memrealtime() - t0 > VERIFY_WAIT_TICKS) { timed_out = true; break; }
            }
        }
        JF_VSTAMP(p, 2);
        int wb = 0;                                              // 1: stepped on the image, all four wavefronts write it back
        if (timed_out) {                                         // item workgroups never arrived: report, do not step
            if (lane == 0) { G[H_ERR] = JF_E_LAUNCH; G[H_DONE] = 1; if (dg) { dg->error = JF_E_LAUNCH; dg->done = 1; dg->B = 0; dg->T = 0; } }
            wb = -1;
        } else {
            auto Gglobal = [rows, cpr](int r, int t) -> int {
                const int64_t idx = rows.index(r, t);
                if (idx < 0 || idx >= rows.plen) return -1;
                unsigned long long mx;
                (void)slots_max((const unsigned long long *)rows.pk + idx * cpr, cpr, mx);
                return decode_packed(mx);
            };
            if (use_lds) {
                SoloWaveLanes{}.sync();
                JF_VSTAMP(p, 3);
#ifdef JF_EXP_STEP_TWICE
                const int npass = 1 + (a.fast >> 1);
                Machine<SoloWaveLanes> m(img, SoloWaveLanes{}, LC);
#pragma nounroll
                for (int pass = 0; pass < npass; ++pass) {
                    const bool last = pass == npass - 1;
                    m = Machine<SoloWaveLanes>(last ? img : img2, SoloWaveLanes{}, LC);
                    m.allow_fast = (a.fast & 1) != 0;
                    const int32_t *gt = gtok;
                    m.step([gt, T](int r, int t) -> int { return gt[r * T + t]; }, last ? s_desc : (jf_mb_desc *)(img2 + LC.total));
                    SoloWaveLanes{}.sync();
                    if (!last && threadIdx.x == 0 && blockIdx.x < 256) g_mtrace[16 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
                }
#else
                Machine<SoloWaveLanes> m(img, SoloWaveLanes{}, LC);
                m.allow_fast = a.fast != 0;
                const int32_t *gt = gtok;
                m.step([gt, T](int r, int t) -> int { return gt[r * T + t]; }, s_desc);
                SoloWaveLanes{}.sync();
#endif
                if (!s_desc->error) {
                    wb = 1;
                    loop_after_step(m, lp, has_loop, p, was_done, s_desc);
                    SoloWaveLanes{}.sync();
                }
                JF_VSTAMP(p, 4);
            }
            if (!wb) {                                           // step on the HBM block (capacities the parameters ask for)
                Machine<SoloWaveLanes> m(G, SoloWaveLanes{}, LG);
                m.allow_fast = a.fast != 0;
                m.step(Gglobal, dg);
                loop_after_step(m, lp, has_loop, p, was_done, dg);
            }
        }
        if (lane == 0) smem[17] = wb;
    }
    __syncthreads();
    const int wb = smem[17];
    const bool mail = has_loop && lp.mailbox;
    if (threadIdx.x >= 64) {
        // ---- wavefronts 1-3: the image goes back to the HBM block and the prompt's argmax slots are re-zeroed, while
        // wavefront 0 hands the descriptor over (nothing below depends on these stores; the kernel boundary orders them
        // before the pack launch)
        if (wb > 0) compact_to_state(Lanes192{}, img, LC, G, LG);
        JF_VSTAMP(p, 5);
        if (wb >= 0) {
            const int64_t cpr = a.am.chunks_per_row;
            for (int r = 0; r < B; ++r) {                         // a position's slots are adjacent: one contiguous range per row
                const int64_t lo = rows.index(r, 0), hi = lo + rows.tpad < a.packed_len ? lo + rows.tpad : a.packed_len;
                for (int64_t i = lo * cpr + (threadIdx.x - 64); i < hi * cpr; i += AM_TPB - 64) a.am.packed[i] = 0ull;
            }
        }
        return;
    }
    // ---- wavefront 0: the descriptor goes to the descriptor table and, loop API, straight into this prompt's slot of the
    // host mailbox together with its driver record (the summary over all prompts is the pack launch's first action)
    constexpr int DINTS = (int)(sizeof(jf_mb_desc) / 4);
    if (dg && threadIdx.x < DINTS) {
        const int v = wb > 0 ? ((const int32_t *)s_desc)[threadIdx.x] : ((const int32_t *)dg)[threadIdx.x];
        if (wb > 0) ((int32_t *)dg)[threadIdx.x] = v;
        if (mail) DevLanes{}.mail(lp.mailbox + JF_MB_MAILBOX_HDR + p * DINTS + threadIdx.x, v);   // written through: see DevLanes::mail
    }
    if (mail && lp.drv && threadIdx.x >= 32 && threadIdx.x < 32 + JF_MB_FIN_INTS)
        DevLanes{}.mail(lp.mailbox + JF_MB_MAILBOX_HDR + a.P * DINTS + p * JF_MB_FIN_INTS + (threadIdx.x - 32),
                        mb_fin_value(lp.drv + (int64_t)p * lp.drv_ints, threadIdx.x - 32));
    JF_VSTAMP(p, 6);
    JF_VSTAMP(p, 7);
}

template <int DT, bool WAVE, bool NT>
__global__ __launch_bounds__(AM_TPB, 5) void mb_verify_kernel(VerifyArgs a) {   // five workgroups per CU: 1 280 resident >= 64 steppers + 1 024 items
    extern __shared__ __attribute__((aligned(16))) int32_t smem[];
    if ((int)blockIdx.x < a.P) { verify_stepper(a, blockIdx.x, smem); return; }
    const int64_t blk = (int64_t)blockIdx.x - a.P;
#ifdef JF_EXP_VERIFY_TRACE
    if (threadIdx.x == 0 && blk < 8192) g_vitems[2 * blk] = __builtin_amdgcn_s_memrealtime();
#endif
    // the item workgroups walk the items in order, workgroup k taking k, k + G, k + 2G, ...: rows near the head of the list are
    // finished early in the stream instead of every row sharing the bandwidth until the end (verify_launch picks G)
    const int64_t G = (int64_t)gridDim.x - a.P;
    if constexpr (WAVE) {
        for (int64_t it = blk; it * (AM_TPB / 64) < a.items; it += G) (void)argmax_wave_item<DT, NT>(a.am, it * (AM_TPB / 64) + (threadIdx.x >> 6));
    } else {
        for (int64_t it = blk; it < a.items; it += G) {
            (void)argmax_wg_item<DT, true, NT>(a.am, it);
            __syncthreads();                                     // the item's LDS partials are reused by the next one
        }
    }
#ifdef JF_EXP_VERIFY_TRACE
    if (threadIdx.x == 0 && blk < 8192) g_vitems[2 * blk + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// Largest number of stepper workgroups a fused launch may carry: half of what the device keeps resident of this kernel
// variant at this LDS request (cached per variant; 0 if the runtime cannot tell, which selects the two-launch path).
static int verify_stepper_cap(const void *kern, int variant, size_t shm) {
    static std::mutex mu;
    static size_t seen_shm[8];
    static int seen_cap[8], seen_dev[8];
    static bool seen[8];
    std::lock_guard<std::mutex> g(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (seen[variant] && seen_shm[variant] == shm && seen_dev[variant] == dev) return seen_cap[variant];   // per variant, LDS request AND device
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, AM_TPB, shm) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        per_cu = 0;
    }
    const long long cap = (long long)per_cu * cus / 2;
    seen[variant] = true; seen_shm[variant] = shm; seen_dev[variant] = dev; seen_cap[variant] = (int)(cap > 0x7FFFFFFF ? 0x7FFFFFFF : cap);
    return seen_cap[variant];
}

static int verify_launch(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride, const int32_t *out_index,
                         int32_t *states, int64_t state_ints, int P, uint64_t *packed, int64_t packed_len, int64_t packed_cap,
                         int32_t Tpad, jf_mb_desc *desc, const jf_mb_params *params,
                         const jfmb::LoopDev *lp, void *stream, const char *who, int *fused_out = nullptr, int64_t valid_rows = -1,
                         hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr) {
    if (fused_out) *fused_out = 0;
    if (P <= 0) return JF_OK;
    int rc = check_params(params, who);
    if (rc) return rc;
    if (!logits || !states || !packed || R <= 0 || Tpad <= 0)
        return fail(JF_E_INVALID, "%s: null pointer or empty forward", who);
    if (packed_len <= 0 || packed_cap < packed_len)
        return fail(JF_E_INVALID, "%s: packed holds %lld entries, the forward has %lld positions", who, (long long)packed_cap, (long long)packed_len);
    if (dtype != JF_F32 && dtype != JF_BF16) return fail(JF_E_INVALID, "%s: dtype %d", who, dtype);
    if (V <= 0 || row_stride < V || V > 0x7FFFFFFFll) return fail(JF_E_INVALID, "%s: bad shape V=%lld stride=%lld", who, (long long)V, (long long)row_stride);
    ArgmaxPlan pl;
    rc = argmax_plan(logits, dtype, R, V, row_stride, true, &pl, packed_cap / packed_len);   // one result slot per (position, chunk)
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    // compact image + greedy tokens of one prompt in LDS; a configuration that needs more than 16 KB steps on HBM instead
    // (the LDS request applies to every workgroup of the launch and must not cut the streaming workgroups' residency)
    const jfmb::Layout LG = jfmb::make_layout(params->n, params->K, params->pool_size, params->max_blocks);
    const jfmb::Layout LC = jfmb::compact_layout(LG, params->K);
    int64_t lds_ints = VERIFY_LDS_HDR + (int64_t)LC.total + (int64_t)LC.RMAX * LC.TMAX;
    lds_ints = (lds_ints + 3) & ~3ll;
#ifdef JF_EXP_STEP_TWICE
    const int64_t lds_step_ints = lds_ints;
    lds_ints += LC.total + 16 + 4;                              // the second image + its descriptor
    lds_ints = (lds_ints + 3) & ~3ll;
    if (lds_ints * 4 > 32 * 1024) lds_ints = 0;
#else
    if (lds_ints * 4 > 16 * 1024) lds_ints = 0;
#endif
    const size_t shm = (size_t)lds_ints * 4;
    void (*kern)(VerifyArgs) = nullptr;
    int variant = 0;
#define JF_V(DT, WV, NTF) kern = mb_verify_kernel<DT, WV, NTF>
    if (dtype == JF_F32) {
        if (pl.wave_mode) { if (pl.nt) JF_V(JF_F32, true, true); else JF_V(JF_F32, true, false); }
        else { if (pl.nt) JF_V(JF_F32, false, true); else JF_V(JF_F32, false, false); }
    } else {
        if (pl.wave_mode) { if (pl.nt) JF_V(JF_BF16, true, true); else JF_V(JF_BF16, true, false); }
        else { if (pl.nt) JF_V(JF_BF16, false, true); else JF_V(JF_BF16, false, false); }
    }
#undef JF_V
    variant = (dtype == JF_BF16 ? 4 : 0) + (pl.wave_mode ? 2 : 0) + (pl.nt ? 1 : 0);
    // Steppers wait inside the launch, so they must never be able to fill the chip: they may hold at most half of the
    // workgroups this kernel can keep resident (registers, LDS request, 256 threads: asked of the runtime, not assumed).
    // More prompts than that, or unaligned logits, run the convergence check as its two launches.
    const int cap = pl.vec ? verify_stepper_cap((const void *)kern, variant, shm) : 0;
    if (!pl.vec || P > cap || pl.cpr * packed_len > packed_cap) {
        if (ev_begin) (void)hipEventRecord(ev_begin, s);         // two launches: the events bracket both
        rc = out_index ? jf_argmax_scatter(logits, dtype, R, V, row_stride, out_index, packed, stream)
                       : jf_argmax_partial(logits, dtype, R, V, row_stride, packed, stream);
        if (rc) return rc;
        mb_step_kernel<<<P, 64, 0, s>>>(states, state_ints, (unsigned long long *)packed, packed_len, desc,
                                        lp ? *lp : jfmb::LoopDev{}, lp ? 1 : 0, fast_path());
        if (ev_end) (void)hipEventRecord(ev_end, s);
        return check_launch("mb_step_kernel");
    }
    VerifyArgs a;
    a.am = ArgmaxArgs{logits, R, V, row_stride, (unsigned long long *)packed, (int)pl.cpr, pl.chunk, out_index, pl.reverse, valid_rows, 1};
    a.states = states; a.state_ints = state_ints; a.P = P; a.packed_len = packed_len;
    a.desc = desc; a.Tpad = Tpad; a.compacted = out_index ? 1 : 0; a.lds_ints = (int32_t)lds_ints;
    a.has_loop = lp ? 1 : 0;
    a.fast = fast_path();
#ifdef JF_EXP_STEP_TWICE
    { const char *e = getenv("JF_EXP_TWICE"); if (e && e[0] == '1') a.fast |= 2; a.lds_ints = lds_ints ? (int32_t)lds_step_ints : 0; }
#endif
    a.lp = lp ? *lp : jfmb::LoopDev{};
    a.items = pl.items;
    // item workgroups: 1024 (four per CU) keep the memory system as full as one per item does — 768 already lose 1-2 %,
    // 512 4 %, 256 19 % in situ — and work the list through in order, so that the prompts listed first (EVT_SLOW_NEXT)
    // see their rows ~20 us before the others (profiles/verify_item_wgs_r03.txt).  JF_VERIFY_ITEM_WGS overrides
    // (0 = one workgroup per item, the round-2 shape)
    static const long long wgs_env = [] { const char *e = getenv("JF_VERIFY_ITEM_WGS"); return e && *e ? atoll(e) : -1ll; }();
    int64_t item_wgs = wgs_env > 0 ? wgs_env : (wgs_env == 0 ? pl.blocks : 1024);
    if (item_wgs > pl.blocks) item_wgs = pl.blocks;
    const int64_t blocks = item_wgs + P;
    if (blocks > 0x7FFFFFFFll) return fail(JF_E_CAPACITY, "%s: grid too large", who);
    const dim3 grid((unsigned)blocks), block(AM_TPB);
    // Timing events: attached to THIS dispatch (its start / stop timestamps — what rocprofv3 reports as the kernel's duration)
    // unless JF_VERIFY_EVENTS=bracket asks for hipEventRecord in front of and behind the launch (the launch + two event packets:
    // ~3-4 us more, the figure of rounds 1-3 and of this round's earlier sessions)
    const bool bracket = (ev_begin || ev_end) && jf_timing_bracket();      // (the environment is read by TIMED calls only)
    bool launched = false;
    if ((ev_begin || ev_end) && !bracket) {
        void *kargs[] = {(void *)&a};
        launched = hipExtLaunchKernel((const void *)kern, grid, block, kargs, shm, s, ev_begin, ev_end, 0) == hipSuccess;
        if (!launched) (void)hipGetLastError();                  // (a runtime without it: the bracket below)
    }
    if (!launched) {
        if (ev_begin) (void)hipEventRecord(ev_begin, s);
        kern<<<grid, block, shm, s>>>(a);
        if (ev_end) (void)hipEventRecord(ev_end, s);
    }
    if (fused_out) *fused_out = 1;
    return check_launch("mb_verify_kernel");
}

extern "C" int jf_mb_verify(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride, const int32_t *out_index,
                            int32_t *states, int64_t state_ints, int P, uint64_t *packed, int64_t packed_len, int64_t packed_cap,
                            int32_t Tpad, jf_mb_desc *desc, const jf_mb_params *params, void *stream) {
    return verify_launch(logits, dtype, R, V, row_stride, out_index, states, state_ints, P, packed, packed_len, packed_cap, Tpad,
                         desc, params, nullptr, stream, "jf_mb_verify");
}

// ------------------------------------------------------------------------------------------------
// jf_mb_loop_*: the loop around the step (include/jacobiforcing.h)
// ------------------------------------------------------------------------------------------------
// (Round 6 tried a probe here — a one-lane launch on a stream of its own that mails a magic word the host must read back before a
//  fresh block is handed out — after the soak had lost the first record of a freshly mapped mailbox twice in 25 600 cases.  It never
//  fired in ~140 000 allocations, and with it the GPU suite under `pytest -n 8` stalled twice in a subprocess-spawning test (72 s,
//  then a 900 s timeout: profiles/soak_r06.txt), so it is gone again.  What fixed the lost record is that mailboxes are no longer
//  mapped and unmapped per chunk: ops._MailboxPool.)
extern "C" int jf_host_alloc(size_t bytes, void **out) {
    if (!out || bytes == 0) return fail(JF_E_INVALID, "jf_host_alloc: bad argument");
    void *p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(JF_E_LAUNCH, "jf_host_alloc: %s", hipGetErrorString(e)); }
    memset(p, 0, bytes);
    *out = p;
    return JF_OK;
}
extern "C" int jf_host_free(void *p) {
    if (p && hipHostFree(p) != hipSuccess) { (void)hipGetLastError(); return fail(JF_E_LAUNCH, "jf_host_free failed"); }
    return JF_OK;
}

extern "C" int jf_mailbox_wait(const int32_t *mailbox, int32_t seq, int64_t timeout_us, void *stream) {
    if (!mailbox) return fail(JF_E_INVALID, "jf_mailbox_wait: null mailbox");
    const volatile int32_t *w = mailbox + JF_MB_SEQ;
    struct timespec t0, t;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 1;; ++spins) {
        if (__atomic_load_n((const int32_t *)w, __ATOMIC_ACQUIRE) == seq) return JF_OK;
        __builtin_ia32_pause();
        if ((spins & 1023u) == 0u) {
            clock_gettime(CLOCK_MONOTONIC, &t);
            const int64_t us = (int64_t)(t.tv_sec - t0.tv_sec) * 1000000 + (t.tv_nsec - t0.tv_nsec) / 1000;
            // a launch that failed never publishes: once the stream has drained the word is final
            if (us > 2000 && (spins & 0xFFFFFu) == 0u && hipStreamQuery((hipStream_t)stream) == hipSuccess) {
                if (__atomic_load_n((const int32_t *)w, __ATOMIC_ACQUIRE) == seq) return JF_OK;
                return fail(JF_E_LAUNCH, "jf_mailbox_wait: the stream drained without publishing sequence %d (mailbox holds %d)", seq, (int)*w);
            }
            if (timeout_us > 0 && us > timeout_us)
                return fail(JF_E_LAUNCH, "jf_mailbox_wait: sequence %d not published within %lld us (mailbox holds %d)", seq, (long long)timeout_us, (int)*w);
        }
    }
}

static int check_loop(const jf_mb_loop *lp, const char *who) {
    if (!lp) return fail(JF_E_INVALID, "%s: null loop", who);
    if (lp->P <= 0) return fail(JF_E_INVALID, "%s: P=%d", who, lp->P);
    if (!lp->states || !lp->packed || !lp->desc || !lp->input_ids || !lp->positions || !lp->row_prompt || !lp->row_len ||
        !lp->row_cand || !lp->row_kv_len || !lp->mailbox)
        return fail(JF_E_INVALID, "%s: null pointer in jf_mb_loop", who);
    if (lp->t_cap <= 0 || lp->rows_cap <= 0) return fail(JF_E_INVALID, "%s: forward buffers have no capacity", who);
    if (lp->drv && (!lp->draws || lp->draw_len <= 0 || lp->drv_ints <= JF_DRV_HDR_INTS))
        return fail(JF_E_INVALID, "%s: resident driver without a draw stream / text capacity", who);
    return JF_OK;
}

// publish: 1 = the launch in front mailed the per-prompt tables itself (fused verify), 2 = this launch copies them
static int loop_pack(const jf_mb_loop *lp, const jfmb::LoopDev &d, int publish, hipStream_t s) {
    const jfmb::PackOut o{lp->input_ids, lp->positions, lp->row_prompt, lp->row_len, lp->valid_index, lp->row_cand, lp->row_kv_len};
    mb_pack_kernel<<<lp->P + (publish ? 1 : 0), 64, 0, s>>>(lp->states, lp->state_ints, lp->desc, 0, lp->t_align < 1 ? 1 : lp->t_align, lp->t_cap,
                                        lp->pad_fill, lp->order ? 1 : 0, lp->cand_rows, o, lp->valid_align < 1 ? 1 : lp->valid_align,
                                        d, publish);
    return check_launch("mb_pack_kernel");
}

extern "C" int jf_mb_loop_begin(const jf_mb_loop *loop, int32_t seq, const jf_mb_params *params, const int64_t *input_ids,
                                const int32_t *kv_len, void *stream) {
    int rc = check_loop(loop, "jf_mb_loop_begin");
    if (rc) return rc;
    rc = check_params(params, "jf_mb_loop_begin");
    if (rc) return rc;
    if (!input_ids || !kv_len) return fail(JF_E_INVALID, "jf_mb_loop_begin: null pointer");
    if (loop->state_ints < jf_mb_state_ints(params)) return fail(JF_E_INVALID, "jf_mb_loop_begin: state block too small");
    mb_begin_kernel<<<loop->P, 64, 0, (hipStream_t)stream>>>(loop->states, loop->state_ints, *params, input_ids, kv_len, loop->desc,
                                                             loop->kv_len);
    rc = check_launch("mb_begin_kernel");
    if (rc) return rc;
    return loop_pack(loop, jfmb::make_loop_dev(loop, seq, params), 2, (hipStream_t)stream);
}

extern "C" int jf_mb_loop_iterate(const jf_mb_loop *loop, int32_t seq, const void *logits, int dtype, int64_t R, int64_t V,
                                  int64_t row_stride, int compacted, int32_t Rtot, int32_t Tpad, const jf_mb_params *params,
                                  int queue_pack, void *ev_begin, void *ev_end, void *stream) {
    int rc = check_loop(loop, "jf_mb_loop_iterate");
    if (rc) return rc;
    if (Rtot <= 0 || Rtot > loop->rows_cap || Tpad <= 0 || (int64_t)Rtot * Tpad > loop->packed_cap)
        return fail(JF_E_INVALID, "jf_mb_loop_iterate: forward of %d x %d positions exceeds the buffers", Rtot, Tpad);
    if (compacted && !loop->valid_index) return fail(JF_E_INVALID, "jf_mb_loop_iterate: compacted logits without a position list");
    const jfmb::LoopDev d = jfmb::make_loop_dev(loop, seq, params);
    int fused = 0;
    // rows of the compacted logits beyond the position list's entries are list padding: the mailbox the caller has just waited
    // on says how many entries there are, so the items need not look at the list before they start to stream
    int64_t nvalid = -1;
    if (compacted) {
        nvalid = loop->mailbox[JF_MB_NVALID];
        if (nvalid < 0 || nvalid > R) return fail(JF_E_INVALID, "jf_mb_loop_iterate: the mailbox lists %lld positions, the logits have %lld rows", (long long)nvalid, (long long)R);
    }
    rc = verify_launch(logits, dtype, R, V, row_stride, compacted ? loop->valid_index : nullptr, loop->states, loop->state_ints,
                       loop->P, loop->packed, (int64_t)Rtot * Tpad, loop->packed_cap, Tpad, loop->desc, params, &d,
                       stream, "jf_mb_loop_iterate", &fused, nvalid, (hipEvent_t)ev_begin, (hipEvent_t)ev_end);
    if (rc || !queue_pack) return rc;
    return loop_pack(loop, d, fused ? 1 : 2, (hipStream_t)stream);     // the fused launch's steppers mailed their own descriptors
}

extern "C" int jf_mb_loop_pack(const jf_mb_loop *loop, int32_t seq, const jf_mb_params *params, void *stream) {
    int rc = check_loop(loop, "jf_mb_loop_pack");
    if (rc) return rc;
    rc = check_params(params, "jf_mb_loop_pack");
    return rc ? rc : loop_pack(loop, jfmb::make_loop_dev(loop, seq, params), 2, (hipStream_t)stream);
}

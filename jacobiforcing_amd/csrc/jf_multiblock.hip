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
#include <mutex>

// ------------------------------------------------------------------------------------------------
// multiblock state machine: one wavefront per prompt
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void mb_begin_kernel(int32_t *states, int64_t state_ints, jf_mb_params prm,
                                                       const int64_t *input_ids, const int32_t *kv_len, jf_mb_desc *desc) {
    jfmb::mb_begin_body(DevLanes{}, blockIdx.x, states, state_ints, prm, input_ids, kv_len, desc);
}
__global__ __launch_bounds__(64) void mb_pack_kernel(int32_t *states, int64_t state_ints, int32_t Tpad, int64_t pad_fill,
                                                      int64_t *input_ids, int32_t *positions, int32_t *row_prompt,
                                                      int32_t *row_len, int32_t *valid_index, int32_t valid_align) {
    jfmb::mb_pack_body(DevLanes{}, blockIdx.x, gridDim.x, states, state_ints, Tpad, pad_fill, input_ids, positions, row_prompt,
                       row_len, valid_index, valid_align);
}
__global__ __launch_bounds__(64) void mb_step_kernel(int32_t *states, int64_t state_ints, unsigned long long *packed,
                                                      int64_t packed_len, jf_mb_desc *desc) {
    JF_STAMP(0);
    jfmb::mb_step_body(DevLanes{}, blockIdx.x, states, state_ints, (uint64_t *)packed, packed_len, desc);
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
    mb_begin_kernel<<<P, 64, 0, (hipStream_t)stream>>>(states, state_ints, *params, input_ids, kv_len, desc);
    return check_launch("mb_begin_kernel");
}

extern "C" int jf_mb_pack(int32_t *states, int64_t state_ints, int P, int32_t Tpad, int64_t pad_fill, int64_t *input_ids,
                          int32_t *positions, int32_t *row_prompt, int32_t *row_len, int32_t *valid_index,
                          int32_t valid_align, void *stream) {
    if (P <= 0) return JF_OK;
    if (!states || !input_ids || !positions || !row_prompt || !row_len) return fail(JF_E_INVALID, "jf_mb_pack: null pointer");
    if (Tpad <= 0) return fail(JF_E_INVALID, "jf_mb_pack: Tpad=%d", Tpad);
    mb_pack_kernel<<<P, 64, 0, (hipStream_t)stream>>>(states, state_ints, Tpad, pad_fill, input_ids, positions, row_prompt, row_len,
                                                      valid_index, valid_align < 1 ? 1 : valid_align);
    return check_launch("mb_pack_kernel");
}

extern "C" int jf_mb_step(int32_t *states, int64_t state_ints, int P, uint64_t *packed, int64_t packed_len, jf_mb_desc *desc,
                          void *stream) {
    if (P <= 0) return JF_OK;
    if (!states || !packed) return fail(JF_E_INVALID, "jf_mb_step: null pointer");
    mb_step_kernel<<<P, 64, 0, (hipStream_t)stream>>>(states, state_ints, (unsigned long long *)packed, packed_len, desc);
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
//                            waits for the prompt's arrival count, pulls the prompt's argmax results (8-byte agent-scope
//                            loads) into LDS, runs Machine::step entirely on LDS, and writes the image + descriptor back.
//                            A step that does not fit the compact capacities (runaway block lists, Q3/Q4) is redone on
//                            the HBM block: nothing was written before that, so the result is the same.
//   workgroups [P, ...)      the argmax items of jf_argmax_scatter / _partial (jf_argmax_dev.h).  After its atomicMax a
//                            publishing lane drains its memory counter and adds 1 to arrive[prompt of the row].
//
// Hand-off (cdna_hip_programming.md, Guideline 16, "8-byte agent atomics both sides"): payload = device-scope atomicMax
// on packed[], s_waitcnt vmcnt(0), relaxed agent-scope add on the counter; the stepper polls the counter with relaxed
// agent-scope loads (s_sleep between polls, bounded by a wall-clock limit) and reads packed[] with agent-scope loads.
// arrive[] must be zero on entry; every stepper resets its word, so the call leaves it zero.
// Steppers only WAIT for item workgroups and never the other way round, and they are the lowest block ids (dispatched
// first), so an item workgroup can always be scheduled: no residency assumption, no deadlock.
// ------------------------------------------------------------------------------------------------
struct VerifyArgs {
    ArgmaxArgs am;
    int32_t *states;
    int64_t state_ints;
    int P;
    int64_t packed_len;
    const int32_t *row_prompt;     // [Rtot] prompt of every forward row (jf_mb_pack)
    int32_t *arrive;               // [P]
    jf_mb_desc *desc;
    int32_t Tpad;
    int32_t compacted;             // 1: logits rows follow valid_index (B*T per prompt); 0: the Rtot x Tpad rectangle
    int32_t lds_ints;              // ints of dynamic LDS available for (compact image + greedy tokens); 0 = step on HBM
};

#ifdef JF_EXP_VERIFY_TRACE
// experiment build only (tools/verify_trace.py): wall-clock stamps (100 MHz) of the launch — [0] first item start, [1] last
// item end, then 8 per stepper: start, image copied, rows arrived, tokens gathered, stepped, written back, end
__device__ unsigned long long g_vtrace[2 + 8 * 256];
__device__ unsigned long long g_vitems[2 * 8192];       // (start, end) per item workgroup: plain stores, no atomics in the stream
#define JF_VSTAMP(p, k) do { if (threadIdx.x == 0 && (p) < 256) g_vtrace[2 + 8 * (p) + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
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

struct Lanes256 {                  // all four wavefronts of a stepper workgroup (copy-in only)
    __device__ __forceinline__ int lane() const { return threadIdx.x; }
    __device__ __forceinline__ int count() const { return 256; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
};

__device__ __forceinline__ unsigned long long ld_agent_u64(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int VERIFY_ARRIVE_STRIDE = 64;                         // ints between two prompts' arrival words: one 256-byte line each,
                                                                 // so polls and arrivals of different prompts use different channels
constexpr int VERIFY_LDS_HDR = 32;                               // ints in front of the compact image (descriptor + flags)
constexpr unsigned long long VERIFY_WAIT_TICKS = 200000000ull;   // 2 s of the 100 MHz constant clock: never hang the GPU

__device__ __forceinline__ void verify_arrive(const VerifyArgs &a, int owner) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the atomicMax above is performed before the count moves
    __hip_atomic_fetch_add(a.arrive + (int64_t)owner * VERIFY_ARRIVE_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ void verify_stepper(const VerifyArgs &a, int p, int32_t *smem) {
    using namespace jfmb;
    int32_t *G = a.states + (int64_t)p * a.state_ints;
    const Layout LG = layout_of(G);
    const Layout LC = compact_layout(LG, G[H_K]);
    const int B = G[H_B], T = G[H_T];
    const int64_t base = G[H_ROW_BASE];
    const int64_t tpad = G[H_TPAD];
    const int ng = B * T;                                        // greedy tokens this prompt consumes
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
    JF_VSTAMP(p, 1);
    // Wavefront 0 is this prompt's state machine.  The other three park at the barrier below (a parked wavefront issues
    // nothing) and come back for the write-back, which is store-issue bound: 256 lanes instead of 64.
    jf_mb_desc *dg = a.desc ? a.desc + p : nullptr;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int32_t *gtok = img + LC.total;                          // [B, T] greedy tokens (LDS) when use_lds
        // ---- wait for this prompt's rows -----------------------------------------------------------
        const int expected = (a.compacted ? ng : B * (int)tpad) * a.am.chunks_per_row;
        bool timed_out = false;
        if (expected > 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned spins = 0;
            int32_t *word = a.arrive + (int64_t)p * VERIFY_ARRIVE_STRIDE;
            while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) {
                __builtin_amdgcn_s_sleep(20);                    // ~0.6 us between polls: pollers must not load the memory system
                if ((++spins & 255u) == 0u && __builtin_amdgcn_s_memrealtime() - t0 > VERIFY_WAIT_TICKS) { timed_out = true; break; }
            }
            if (lane == 0) __hip_atomic_store(word, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        JF_VSTAMP(p, 2);
        int wb = 0;                                              // 1: stepped on the image, all four wavefronts write it back
        if (timed_out) {                                         // item workgroups never arrived: report, do not step
            if (lane == 0) { G[H_ERR] = JF_E_LAUNCH; G[H_DONE] = 1; if (dg) { dg->error = JF_E_LAUNCH; dg->done = 1; dg->B = 0; dg->T = 0; } }
            wb = -1;
        } else {
            const unsigned long long *pk = a.am.packed;
            const int64_t plen = a.packed_len;
            auto Gglobal = [pk, base, tpad, plen](int r, int t) -> int {
                const int64_t idx = (base + r) * tpad + t;
                return (idx >= 0 && idx < plen) ? decode_packed(ld_agent_u64(pk + idx)) : -1;
            };
            if (use_lds) {
                for (int i = lane; i < ng; i += 64) gtok[i] = Gglobal(i / T, i - (i / T) * T);
                SoloWaveLanes{}.sync();
                JF_VSTAMP(p, 3);
                Machine<SoloWaveLanes> m(img, SoloWaveLanes{}, LC);
                const int32_t *gt = gtok;
                m.step([gt, T](int r, int t) -> int { return gt[r * T + t]; }, s_desc);
                SoloWaveLanes{}.sync();
                JF_VSTAMP(p, 4);
                if (!s_desc->error) wb = 1;
            }
            if (!wb) {                                           // step on the HBM block (capacities the parameters ask for)
                Machine<SoloWaveLanes> m(G, SoloWaveLanes{}, LG);
                m.step(Gglobal, dg);
            }
        }
        if (lane == 0) smem[17] = wb;
    }
    __syncthreads();
    const int wb = smem[17];
    if (wb < 0) return;
    if (wb > 0) {
        compact_to_state(Lanes256{}, img, LC, G, LG);
        if (dg && threadIdx.x < (int)(sizeof(jf_mb_desc) / 4)) ((int32_t *)dg)[threadIdx.x] = ((const int32_t *)s_desc)[threadIdx.x];
    }
    JF_VSTAMP(p, 5);
    // re-zero this prompt's slice of the argmax workspace for the next launch
    const int64_t lo = base * tpad, hi = (base + B) * tpad;
    for (int64_t i = lo + threadIdx.x; i < hi && i < a.packed_len; i += AM_TPB) a.am.packed[i] = 0ull;
    JF_VSTAMP(p, 6);
}

template <int DT, bool WAVE, bool NT>
__global__ __launch_bounds__(AM_TPB) void mb_verify_kernel(VerifyArgs a) {
    extern __shared__ __attribute__((aligned(16))) int32_t smem[];
    if ((int)blockIdx.x < a.P) { verify_stepper(a, blockIdx.x, smem); return; }
    const int64_t blk = (int64_t)blockIdx.x - a.P;
#ifdef JF_EXP_VERIFY_TRACE
    if (threadIdx.x == 0 && blk < 8192) g_vitems[2 * blk] = __builtin_amdgcn_s_memrealtime();
#endif
    int owner = -1;                                              // prompt of this item's row: looked up while the row streams
    if constexpr (WAVE) {
        const int64_t orow = argmax_wave_item<DT, NT>(a.am, blk * (AM_TPB / 64) + (threadIdx.x >> 6), a.row_prompt, a.Tpad, &owner);
        if ((threadIdx.x & 63) == 0 && orow >= 0) verify_arrive(a, owner);
    } else {
        const int64_t orow = argmax_wg_item<DT, true, NT>(a.am, blk, a.row_prompt, a.Tpad, &owner);
        if (threadIdx.x == 0 && orow >= 0) verify_arrive(a, owner);
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
    static int seen_cap[8];
    static bool seen[8];
    std::lock_guard<std::mutex> g(mu);
    if (seen[variant] && seen_shm[variant] == shm) return seen_cap[variant];
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, AM_TPB, shm) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        per_cu = 0;
    }
    const long long cap = (long long)per_cu * cus / 2;
    seen[variant] = true; seen_shm[variant] = shm; seen_cap[variant] = (int)(cap > 0x7FFFFFFF ? 0x7FFFFFFF : cap);
    return seen_cap[variant];
}

extern "C" int jf_mb_verify(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride, const int32_t *out_index,
                            int32_t *states, int64_t state_ints, int P, uint64_t *packed, int64_t packed_len, int32_t Tpad,
                            const int32_t *row_prompt, int32_t *arrive, jf_mb_desc *desc, const jf_mb_params *params,
                            void *stream) {
    if (P <= 0) return JF_OK;
    int rc = check_params(params, "jf_mb_verify");
    if (rc) return rc;
    if (!logits || !states || !packed || !row_prompt || !arrive || R <= 0 || Tpad <= 0)
        return fail(JF_E_INVALID, "jf_mb_verify: null pointer or empty forward");
    if (dtype != JF_F32 && dtype != JF_BF16) return fail(JF_E_INVALID, "jf_mb_verify: dtype %d", dtype);
    if (V <= 0 || row_stride < V || V > 0x7FFFFFFFll) return fail(JF_E_INVALID, "jf_mb_verify: bad shape V=%lld stride=%lld", (long long)V, (long long)row_stride);
    ArgmaxPlan pl;
    rc = argmax_plan(logits, dtype, R, V, row_stride, true, &pl);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    // compact image + greedy tokens of one prompt in LDS; a configuration that needs more than 16 KB steps on HBM instead
    // (the LDS request applies to every workgroup of the launch and must not cut the streaming workgroups' residency)
    const jfmb::Layout LG = jfmb::make_layout(params->n, params->K, params->pool_size, params->max_blocks);
    const jfmb::Layout LC = jfmb::compact_layout(LG, params->K);
    int64_t lds_ints = VERIFY_LDS_HDR + (int64_t)LC.total + (int64_t)LC.RMAX * LC.TMAX;
    lds_ints = (lds_ints + 3) & ~3ll;
    if (lds_ints * 4 > 16 * 1024) lds_ints = 0;
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
    if (!pl.vec || P > cap) {
        rc = out_index ? jf_argmax_scatter(logits, dtype, R, V, row_stride, out_index, packed, stream)
                       : jf_argmax_partial(logits, dtype, R, V, row_stride, packed, stream);
        if (rc) return rc;
        return jf_mb_step(states, state_ints, P, packed, packed_len, desc, stream);
    }
    VerifyArgs a;
    a.am = ArgmaxArgs{logits, R, V, row_stride, (unsigned long long *)packed, (int)pl.cpr, pl.chunk, out_index, pl.reverse};
    a.states = states; a.state_ints = state_ints; a.P = P; a.packed_len = packed_len; a.row_prompt = row_prompt;
    a.arrive = arrive; a.desc = desc; a.Tpad = Tpad; a.compacted = out_index ? 1 : 0; a.lds_ints = (int32_t)lds_ints;
    const int64_t blocks = pl.blocks + P;
    if (blocks > 0x7FFFFFFFll) return fail(JF_E_CAPACITY, "jf_mb_verify: grid too large");
    const dim3 grid((unsigned)blocks), block(AM_TPB);
    kern<<<grid, block, shm, s>>>(a);
    return check_launch("mb_verify_kernel");
}

// jf_common.h — shared by the translation units of libjacobiforcing.so (gfx950 / CDNA4 only).
//
// Everything on this path is HBM/latency-bound integer and compare work: no MFMA.  Translation units:
//   jf_api.hip        error plumbing, jf_version / jf_last_error
//   jf_argmax.hip     (a2) vocabulary argmax (jf_argmax_partial / _scatter / _decode / _rows), (a3) accept scan
//   jf_argmax_dev.h   device bodies of the argmax items, shared with the fused verify launch
//   jf_multiblock.hip (a1, a4-a12) the multiblock Jacobi state machine kernels (jf_mb_*), one wavefront per prompt, and
//                     jf_mb_verify: argmax items + per-prompt state-machine steps in ONE launch
//   jf_kv.hip         (a18) KV append, fused RoPE + append, SwiGLU; (a9/a10) KV commit
//   jf_engine.hip     (a15) engine single-block step, (a16) paged-KV index fill
//   jf_sampling.hip   (a19) non-greedy verify (jf_rs_probs, jf_rs_step) and the on-policy rollout step
#ifndef JF_COMMON_H
#define JF_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jacobiforcing.h"
#include "jf_mb_core.h"

// ---- error plumbing (jf_api.hip) ------------------------------------------------------------------
int fail(int code, const char *fmt, ...);          // records the message for jf_last_error(), returns `code`
int check_launch(const char *what);                // JF_OK or JF_E_LAUNCH after a kernel launch
// jf_timing_arm: events the NEXT timed entry point called on this thread attaches to its launch(es) (taken = cleared)
struct JfTiming { hipEvent_t begin = nullptr, end = nullptr; bool any() const { return begin || end; } };
JfTiming jf_take_timing();
bool jf_timing_bracket();                          // JF_VERIFY_EVENTS=bracket: record the events around the launch instead

// ------------------------------------------------------------------------------------------------
// wave-level helpers (wavefront = 64 lanes)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint64_t o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        int o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// one wavefront per workgroup
struct DevLanes {
    static constexpr bool WAVE64 = true;                               // one token per lane code paths (Machine::step_fast64)
    __device__ __forceinline__ int shfl(int v, int src) const { return __shfl(v, src, 64); }
    __device__ __forceinline__ int lane() const { return threadIdx.x; }
    __device__ __forceinline__ int count() const { return 64; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ int reduce_min(int v) const { return wave_min_i32(v); }
    __device__ __forceinline__ int reduce_sum(int v) const { return wave_sum_i32(v); }
    __device__ __forceinline__ int first_true(bool pred) const {      // lowest lane with pred, 64 if none
        const unsigned long long m = __ballot(pred);
        return m ? __builtin_ctzll(m) : 64;
    }
    __device__ __forceinline__ int count_true(bool pred) const { return __popcll(__ballot(pred)); }
    __device__ __forceinline__ int prefix_count(bool pred) const {
        const unsigned long long m = __ballot(pred);
        return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    }
    // Mailbox hand-off: stores the host will read (mail), then one word the host polls (publish).  The mailbox is coherent host
    // memory, but a PLAIN store to it may sit in this XCD's L2 until the kernel ends: with plain stores and a drained memory
    // counter the host saw the sequence word before the tables in front of it in 9 of 10 rounds of tools/mailbox_stress.py
    // (the tables of a launch that also stamps the mailbox: jf_mb_loop_begin and the non-fused pack).  mail() therefore writes
    // THROUGH (system-scope store); the drained counter then does order them in front of the sequence word (0 stale reads in
    // 780 000 rounds, profiles/mailbox_order_r03.txt).  A system-scope RELEASE fence instead would also write back everything
    // else this L2 holds (the state blocks the convergence launch has just stored, the forward inputs being packed):
    // +7 us on the pack launch and on the host's wake-up (profiles/verify_release_ab_r03.txt); publish(..., fence = true) is that
    // variant, chosen per loop at run time (jf_mb_loop.flags: the host's start-up self-test of the cheap order, ops.MultiblockLoop).
    __device__ __forceinline__ void mail(int32_t *word, int32_t v) const {
        __hip_atomic_store(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __device__ __forceinline__ void publish(int32_t *word, int32_t v, bool fence = false) const {
        if (fence) {                                          // jf_mb_loop.flags & JF_MB_LOOP_PUBLISH_FENCE (wave-uniform): the formal order
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            if (lane() == 0) __hip_atomic_store(word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane() == 0) __hip_atomic_store(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
};

// one wavefront that is the only live wavefront of a larger workgroup (the steppers of jf_mb_verify): a phase boundary is a
// memory fence (s_waitcnt vmcnt(0) lgkmcnt(0)), not an s_barrier — nobody else is there to wait for, and the ~80
// barriers of a step cost ~3 us
struct SoloWaveLanes : DevLanes {
    __device__ __forceinline__ void sync() const { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }
};

// ------------------------------------------------------------------------------------------------
// vocabulary-stream helpers shared by the argmax and the softmax-gather kernels
// ------------------------------------------------------------------------------------------------
// order-preserving key of an fp32 payload with torch.argmax semantics:
//   NaN -> greatest, -0.0 == +0.0, otherwise numeric order.
__device__ __forceinline__ uint32_t order_key(uint32_t u) {
    u = (u == 0x80000000u) ? 0u : u;
    const uint32_t k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((u & 0x7fffffffu) > 0x7f800000u) ? 0xFFFFFFFFu : k;
}

constexpr int AM_TPB = 256;
#define JF_LOAD(p) __builtin_nontemporal_load(p)      // streams read once (the softmax kernels; the argmax picks per launch)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int DT> struct Elem;
template <> struct Elem<JF_F32> { using T = uint32_t; static constexpr int EPV = 4; };
template <> struct Elem<JF_BF16> { using T = uint16_t; static constexpr int EPV = 8; };

template <int DT>
__device__ __forceinline__ void consume_vec(const u32x4 v, uint32_t idx0, uint32_t &best, uint32_t &bidx) {
    if constexpr (DT == JF_F32) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t k = order_key(w[j]);
            if (k > best) { best = k; bidx = idx0 + j; }
        }
    } else {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t k0 = order_key(w[j] << 16);          // low half = element 2j
            if (k0 > best) { best = k0; bidx = idx0 + 2 * j; }
            const uint32_t k1 = order_key(w[j] & 0xFFFF0000u);  // high half = element 2j+1
            if (k1 > best) { best = k1; bidx = idx0 + 2 * j + 1; }
        }
    }
}

template <int DT>
__device__ __forceinline__ uint32_t load_key(const void *row, int64_t i) {
    if constexpr (DT == JF_F32) return order_key(((const uint32_t *)row)[i]);
    else return order_key(((uint32_t)((const uint16_t *)row)[i]) << 16);
}

// ---- exact scalar-order scan of [begin, end) of one row (any alignment): used for unaligned rows, ragged
// tails and the rare chunks that contain a NaN.
template <int DT>
__device__ __forceinline__ void scan_exact(const void *p, int64_t begin, int64_t end, int tid, uint32_t &best, uint32_t &bidx) {
    for (int64_t j = begin + tid; j < end; j += AM_TPB) {
        const uint32_t k = load_key<DT>(p, j);
        if (k > best) { best = k; bidx = (uint32_t)j; }
    }
}

// ---- lean per-vector keys --------------------------------------------------------------------
// Signed "two's-complement-like" key of an IEEE payload: k = w ^ ((w >> 31) & 0x7FFFFFFF).  Numeric order as a
// signed integer; +NaN sorts above +inf, -NaN below -inf (both are detected and sent to the exact path);
// key(-0.0) == -1 and key(+0.0) == 0, so -1 is bumped to 0 to make the two zeros tie (torch semantics).
typedef int16_t i16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int32_t skey32(uint32_t w) { return (int32_t)(w ^ (((uint32_t)((int32_t)w >> 31)) & 0x7FFFFFFFu)); }
__device__ __forceinline__ uint32_t skey16x2(uint32_t w) {      // both halves at once
    const i16x2 v = __builtin_bit_cast(i16x2, w);
    const i16x2 sh = v >> (int16_t)15;                           // v_pk_ashrrev_i16
    return w ^ (__builtin_bit_cast(uint32_t, sh) & 0x7FFF7FFFu);
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
    const i16x2 x = __builtin_bit_cast(i16x2, a), y = __builtin_bit_cast(i16x2, b);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(x, y));
}
__device__ __forceinline__ uint32_t pk_min_i16(uint32_t a, uint32_t b) {
    const i16x2 x = __builtin_bit_cast(i16x2, a), y = __builtin_bit_cast(i16x2, b);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(x, y));
}
__device__ __forceinline__ int32_t hmax_i16x2(uint32_t a) {
    const int32_t lo = (int32_t)(int16_t)(a & 0xFFFFu), hi = (int32_t)a >> 16;
    return lo > hi ? lo : hi;
}
__device__ __forceinline__ int32_t hmin_i16x2(uint32_t a) {
    const int32_t lo = (int32_t)(int16_t)(a & 0xFFFFu), hi = (int32_t)a >> 16;
    return lo < hi ? lo : hi;
}

// KEEPV: the best vector stays in registers (no end-of-item reload; 4 selects per vector).  MINT: the running minimum that
// catches negative NaNs — a consumer that also sums exp() of every element sees those in its sum and passes false.
template <int DT, bool KEEPV, bool MINT = true> struct FastTrack;
template <bool KEEPV, bool MINT> struct FastTrack<JF_F32, KEEPV, MINT> {
    int32_t best = INT32_MIN, mn = INT32_MAX;
    uint32_t bvec = 0xFFFFFFFFu;
    u32x4 bv = {0u, 0u, 0u, 0u};     // KEEPV: the best vector itself (saves the end-of-item reload, costs 4 selects per vector)
    // returns the vector's largest key (the softmax kernels derive their running maximum from it)
    __device__ __forceinline__ int32_t consume_ret(const u32x4 v, uint32_t i) {
        const int32_t k0 = skey32(v.x), k1 = skey32(v.y), k2 = skey32(v.z), k3 = skey32(v.w);
        int32_t m = max(max(k0, k1), max(k2, k3));
        if constexpr (MINT) mn = min(mn, min(min(k0, k1), min(k2, k3)));
        m = (m == -1) ? 0 : m;
        if constexpr (KEEPV) { if (m > best) { best = m; bvec = i; bv = v; } }
        else { if (m > best) { best = m; bvec = i; } }
        return m;
    }
    __device__ __forceinline__ void consume(const u32x4 v, uint32_t i) { (void)consume_ret(v, i); }
    __device__ __forceinline__ bool saw_nan() const { return best > (int32_t)0x7F800000 || mn < (int32_t)0x807FFFFF; }
    // first element of the vector at bvec whose canonical key equals best
    __device__ __forceinline__ uint32_t resolve(const void *p) const {
        u32x4 v;
        if constexpr (KEEPV) v = bv; else v = *(const u32x4 *)((const uint32_t *)p + bvec);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t j = 3;
#pragma unroll
        for (int q = 3; q >= 0; --q) { int32_t k = skey32(w[q]); k = (k == -1) ? 0 : k; if (k == best) j = q; }
        return bvec + j;
    }
    __device__ __forceinline__ uint32_t ukey() const { return (uint32_t)best ^ 0x80000000u; }   // == order_key()
};
template <bool KEEPV, bool MINT> struct FastTrack<JF_BF16, KEEPV, MINT> {
    int32_t best = INT32_MIN;
    uint32_t mnp = 0x7FFF7FFFu;     // packed running min
    uint32_t bvec = 0xFFFFFFFFu;
    u32x4 bv = {0u, 0u, 0u, 0u};     // KEEPV: the best vector itself (saves the end-of-item reload, costs 4 selects per vector)
    __device__ __forceinline__ int32_t consume_ret(const u32x4 v, uint32_t i) {
        const uint32_t k0 = skey16x2(v.x), k1 = skey16x2(v.y), k2 = skey16x2(v.z), k3 = skey16x2(v.w);
        const uint32_t pm = pk_max_i16(pk_max_i16(k0, k1), pk_max_i16(k2, k3));
        if constexpr (MINT) mnp = pk_min_i16(mnp, pk_min_i16(pk_min_i16(k0, k1), pk_min_i16(k2, k3)));
        int32_t m = hmax_i16x2(pm);
        m = (m == -1) ? 0 : m;
        if constexpr (KEEPV) { if (m > best) { best = m; bvec = i; bv = v; } }
        else { if (m > best) { best = m; bvec = i; } }
        return m;
    }
    __device__ __forceinline__ void consume(const u32x4 v, uint32_t i) { (void)consume_ret(v, i); }
    __device__ __forceinline__ bool saw_nan() const { return best > 0x7F80 || hmin_i16x2(mnp) < (int32_t)(int16_t)0x807F; }
    __device__ __forceinline__ uint32_t resolve(const void *p) const {
        u32x4 v;
        if constexpr (KEEPV) v = bv; else v = *(const u32x4 *)((const uint16_t *)p + bvec);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t j = 7;
#pragma unroll
        for (int q = 7; q >= 0; --q) {
            const uint32_t h = (q & 1) ? (w[q >> 1] >> 16) : (w[q >> 1] & 0xFFFFu);
            int32_t k = (int32_t)(int16_t)(h ^ ((h & 0x8000u) ? 0x7FFFu : 0u));
            k = (k == -1) ? 0 : k;
            if (k == best) j = q;
        }
        return bvec + j;
    }
    // same value order_key(w16 << 16) gives for non-NaN payloads
    __device__ __forceinline__ uint32_t ukey() const {
        const uint32_t k16 = (uint32_t)best & 0xFFFFu;
        return best >= 0 ? ((k16 | 0x8000u) << 16) : (((k16 ^ 0x8000u) << 16) | 0xFFFFu);
    }
};

template <int DT>
__device__ __forceinline__ float load_f(const void *row, int64_t i) {
    if constexpr (DT == JF_F32) return ((const float *)row)[i];
    else return __uint_as_float(((uint32_t)((const uint16_t *)row)[i]) << 16);
}

#endif  // JF_COMMON_H

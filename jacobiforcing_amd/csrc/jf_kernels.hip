// jf_kernels.hip — gfx950 (MI355X / CDNA4) kernels and the C ABI of include/jacobiforcing.h.
//
// Everything here is HBM/latency-bound integer and compare work: no MFMA.  The kernels that move real bytes are the
// vocabulary streams — argmax (jf_argmax_partial / jf_argmax_scatter: R*V*esize per launch, 16 B per lane per load, eight
// loads in flight, one compare chain per 16-byte vector, wave shuffles, one 64-bit atomicMax per (row, chunk)) and its
// softmax-gather sibling for the non-greedy verify (jf_rs_probs).  The Jacobi state machine runs one 64-lane wavefront per
// prompt (jf_mb_core.h).  Sections: (a2) argmax, (a3) accept scan, (a1, a4-a12) multiblock state machine, (a18) KV
// append / RoPE / SwiGLU, (a9/a10) KV commit, (a15) engine step, (a16) paged index fill, (a19) non-greedy + on-policy.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jacobiforcing.h"
#ifdef JF_EXP_MB_TRACE
// experiment build only (tools/mb_step_trace.py): shader-clock stamps of prompt 0's step, read back by jf_exp_read_trace
__device__ unsigned long long g_mb_trace[32];
#define JF_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_mb_trace[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#endif
#include "jf_mb_core.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(JF_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return JF_OK;
}

extern "C" int jf_version(void) { return JF_VERSION; }
extern "C" const char *jf_last_error(void) { return g_err; }

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
};

// ------------------------------------------------------------------------------------------------
// (a2) argmax over the vocabulary — the convergence kernel's HBM stream
// ------------------------------------------------------------------------------------------------
// order-preserving key of an fp32 payload with torch.argmax semantics:
//   NaN -> greatest, -0.0 == +0.0, otherwise numeric order.
__device__ __forceinline__ uint32_t order_key(uint32_t u) {
    u = (u == 0x80000000u) ? 0u : u;
    const uint32_t k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((u & 0x7fffffffu) > 0x7f800000u) ? 0xFFFFFFFFu : k;
}

constexpr int AM_TPB = 256;
#ifdef JF_EXP_NO_NT
#define JF_LOAD(p) (*(p))
#else
#define JF_LOAD(p) __builtin_nontemporal_load(p)
#endif
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

template <int DT, bool KEEPV> struct FastTrack;
template <bool KEEPV> struct FastTrack<JF_F32, KEEPV> {
    int32_t best = INT32_MIN, mn = INT32_MAX;
    uint32_t bvec = 0xFFFFFFFFu;
    u32x4 bv = {0u, 0u, 0u, 0u};     // KEEPV: the best vector itself (saves the end-of-item reload, costs 4 selects per vector)
    __device__ __forceinline__ void consume(const u32x4 v, uint32_t i) {
        const int32_t k0 = skey32(v.x), k1 = skey32(v.y), k2 = skey32(v.z), k3 = skey32(v.w);
        int32_t m = max(max(k0, k1), max(k2, k3));
        mn = min(mn, min(min(k0, k1), min(k2, k3)));
        m = (m == -1) ? 0 : m;
        if constexpr (KEEPV) { if (m > best) { best = m; bvec = i; bv = v; } }
        else { if (m > best) { best = m; bvec = i; } }
    }
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
template <bool KEEPV> struct FastTrack<JF_BF16, KEEPV> {
    int32_t best = INT32_MIN;
    uint32_t mnp = 0x7FFF7FFFu;     // packed running min
    uint32_t bvec = 0xFFFFFFFFu;
    u32x4 bv = {0u, 0u, 0u, 0u};     // KEEPV: the best vector itself (saves the end-of-item reload, costs 4 selects per vector)
    __device__ __forceinline__ void consume(const u32x4 v, uint32_t i) {
        const uint32_t k0 = skey16x2(v.x), k1 = skey16x2(v.y), k2 = skey16x2(v.z), k3 = skey16x2(v.w);
        const uint32_t pm = pk_max_i16(pk_max_i16(k0, k1), pk_max_i16(k2, k3));
        mnp = pk_min_i16(mnp, pk_min_i16(pk_min_i16(k0, k1), pk_min_i16(k2, k3)));
        int32_t m = hmax_i16x2(pm);
        m = (m == -1) ? 0 : m;
        if constexpr (KEEPV) { if (m > best) { best = m; bvec = i; bv = v; } }
        else { if (m > best) { best = m; bvec = i; } }
    }
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

// Software-pipelined scan of `nvec` 16-byte vectors starting at q (lane-strided by STR vectors): two register sets of
// eight vectors; the loads of set B are issued before set A is consumed and vice versa, so a wavefront keeps 8-16 KB in
// flight at all times instead of draining between batches.  The last pair is peeled so every load in the loop body is
// unconditional (a conditional load would make the compiler wait for vmcnt(0)).
template <int DT, int STR, class FT>
__device__ __forceinline__ void scan_pipelined(FT &ft, const u32x4 *q, int k, int nvec, uint32_t ebase) {
    constexpr int EPV = Elem<DT>::EPV;
    constexpr int BATCH = 8 * STR;
    const int nfull = (nvec > k + 7 * STR) ? ((nvec - k - 7 * STR - 1) / BATCH + 1) : 0;
    const int npairs = nfull >> 1;
    u32x4 A[8], B[8];
    if (npairs >= 1) {
#pragma unroll
        for (int u = 0; u < 8; ++u) A[u] = JF_LOAD(q + u * STR);
        for (int p = 0; p < npairs - 1; ++p) {
#pragma unroll
            for (int u = 0; u < 8; ++u) B[u] = JF_LOAD(q + BATCH + u * STR);
            __builtin_amdgcn_sched_barrier(0);          // keep the loads of the next set ABOVE the compare chain
#pragma unroll
            for (int u = 0; u < 8; ++u) ft.consume(A[u], ebase + (uint32_t)(k + u * STR) * EPV);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) A[u] = JF_LOAD(q + 2 * BATCH + u * STR);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; ++u) ft.consume(B[u], ebase + (uint32_t)(k + BATCH + u * STR) * EPV);
            __builtin_amdgcn_sched_barrier(0);
            q += 2 * BATCH;
            k += 2 * BATCH;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) B[u] = JF_LOAD(q + BATCH + u * STR);
#pragma unroll
        for (int u = 0; u < 8; ++u) ft.consume(A[u], ebase + (uint32_t)(k + u * STR) * EPV);
#pragma unroll
        for (int u = 0; u < 8; ++u) ft.consume(B[u], ebase + (uint32_t)(k + BATCH + u * STR) * EPV);
        q += 2 * BATCH;
        k += 2 * BATCH;
    }
    if (nfull & 1) {
#pragma unroll
        for (int u = 0; u < 8; ++u) A[u] = JF_LOAD(q + u * STR);
#pragma unroll
        for (int u = 0; u < 8; ++u) ft.consume(A[u], ebase + (uint32_t)(k + u * STR) * EPV);
        q += BATCH;
        k += BATCH;
    }
    for (; k < nvec; k += STR, q += STR) {
        const u32x4 v0 = JF_LOAD(q);
        ft.consume(v0, ebase + (uint32_t)k * EPV);
    }
}

// VEC: rows are 16-byte aligned -> 16 B per lane per load, UNROLL independent loads in flight per lane
// (4 or 8 KB per wavefront), one compare chain per 16-byte vector.  All loop arithmetic is 32-bit.
template <int DT, bool VEC, int UNROLL>
__global__ __launch_bounds__(AM_TPB) void argmax_partial_kernel(const void *__restrict__ logits, int64_t R, int64_t V,
                                                                 int64_t row_stride, unsigned long long *__restrict__ packed,
                                                                 int chunks_per_row, int64_t chunk_elems,
                                                                 const int32_t *__restrict__ out_index) {
    using E = Elem<DT>;
    constexpr int EPV = E::EPV;
    const int64_t item = blockIdx.x;
    const int64_t row = item / chunks_per_row;
    // slot of this row's result (jf_argmax_scatter): read up front so its latency hides behind the stream; < 0 = padding row
    const int64_t orow = out_index ? (int64_t)out_index[row] : row;
    if (orow < 0) return;
    const int c = (int)(item - row * chunks_per_row);
    const int64_t begin = (int64_t)c * chunk_elems;
    int64_t end = begin + chunk_elems;
    if (end > V) end = V;
    const typename E::T *p = (const typename E::T *)logits + row * row_stride;
    const int tid = threadIdx.x;

    uint32_t best = 0u, bidx = 0xFFFFFFFFu;   // every real key is >= 0x007FFFFF > 0
    if constexpr (VEC) {
        FastTrack<DT, true> ft;                                       // small problems: skip the end-of-item reload
        const uint32_t ebase = (uint32_t)begin;                       // element index of the chunk start (V < 2^31)
        const int nvec = (int)((end - begin) / EPV);                  // full 16-byte vectors in this chunk
        const u32x4 *q = (const u32x4 *)p + (begin / EPV) + tid;
        int k = tid;
        if constexpr (UNROLL == 16) {
            scan_pipelined<DT, AM_TPB>(ft, q, k, nvec, ebase);
        } else {
            for (; k + (UNROLL - 1) * AM_TPB < nvec; k += UNROLL * AM_TPB, q += UNROLL * AM_TPB) {
                u32x4 v[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) v[u] = JF_LOAD(q + u * AM_TPB);
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) ft.consume(v[u], ebase + (uint32_t)(k + u * AM_TPB) * EPV);
            }
            for (; k < nvec; k += AM_TPB, q += AM_TPB) {
                const u32x4 v0 = JF_LOAD(q);
                ft.consume(v0, ebase + (uint32_t)k * EPV);
            }
        }
        const int64_t vec_end = begin + ((end - begin) / EPV) * EPV;
        if (__syncthreads_or(ft.saw_nan() ? 1 : 0)) {
            scan_exact<DT>(p, begin, vec_end, tid, best, bidx);          // NaN somewhere in this chunk: exact rescan
        } else if (ft.bvec != 0xFFFFFFFFu) {
            best = ft.ukey();
            bidx = ft.resolve(p);
        }
        scan_exact<DT>(p, vec_end, end, tid, best, bidx);                 // ragged tail (V % EPV), indices above all vectors
    } else {
        scan_exact<DT>(p, begin, end, tid, best, bidx);
    }
    // (key, first index) -> one u64 whose max is the answer: larger key wins, then smaller index
    uint64_t pk = ((uint64_t)best << 32) | (uint64_t)(~bidx);
    pk = wave_max_u64(pk);
    __shared__ uint64_t s_part[AM_TPB / 64];
    if ((tid & 63) == 0) s_part[tid >> 6] = pk;
    __syncthreads();
    if (tid == 0) {
        uint64_t m = s_part[0];
#pragma unroll
        for (int w = 1; w < AM_TPB / 64; ++w) m = s_part[w] > m ? s_part[w] : m;
        atomicMax(packed + orow, (unsigned long long)m);
    }
}

// Wave-independent variant: every wavefront owns one (row, chunk) item end to end — no LDS, no workgroup
// barrier; the NaN vote is a ballot, the reduction six shuffles, the publish one atomicMax per wavefront.
template <int DT, int UNROLL>
__global__ __launch_bounds__(AM_TPB) void argmax_wave_kernel(const void *__restrict__ logits, int64_t R, int64_t V,
                                                              int64_t row_stride, unsigned long long *__restrict__ packed,
                                                              int chunks_per_row, int64_t chunk_elems,
                                                              const int32_t *__restrict__ out_index) {
    using E = Elem<DT>;
    constexpr int EPV = E::EPV;
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * (AM_TPB / 64) + (threadIdx.x >> 6);
    if (item >= R * chunks_per_row) return;
    const int64_t row = item / chunks_per_row;
    const int64_t orow = out_index ? (int64_t)out_index[row] : row;
    if (orow < 0) return;
    const int c = (int)(item - row * chunks_per_row);
    const int64_t begin = (int64_t)c * chunk_elems;
    int64_t end = begin + chunk_elems;
    if (end > V) end = V;
    const typename E::T *p = (const typename E::T *)logits + row * row_stride;

    FastTrack<DT, false> ft;                                          // one wave per SIMD: VALU latency is exposed, keep it lean
    const uint32_t ebase = (uint32_t)begin;
    const int nvec = (int)((end - begin) / EPV);
    const u32x4 *q = (const u32x4 *)p + (begin / EPV) + lane;
    int k = lane;
    if constexpr (UNROLL == 16) {
        scan_pipelined<DT, 64>(ft, q, k, nvec, ebase);
    } else {
        for (; k + (UNROLL - 1) * 64 < nvec; k += UNROLL * 64, q += UNROLL * 64) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = JF_LOAD(q + u * 64);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) ft.consume(v[u], ebase + (uint32_t)(k + u * 64) * EPV);
        }
        for (; k < nvec; k += 64, q += 64) {
            const u32x4 v0 = JF_LOAD(q);
            ft.consume(v0, ebase + (uint32_t)k * EPV);
        }
    }
    uint32_t best = 0u, bidx = 0xFFFFFFFFu;
    const int64_t vec_end = begin + (int64_t)nvec * EPV;
    if (__ballot(ft.saw_nan()) != 0ull) {
        for (int64_t j = begin + lane; j < vec_end; j += 64) {
            const uint32_t kk = load_key<DT>(p, j);
            if (kk > best) { best = kk; bidx = (uint32_t)j; }
        }
    } else if (ft.bvec != 0xFFFFFFFFu) {
        best = ft.ukey();
        bidx = ft.resolve(p);
    }
    for (int64_t j = vec_end + lane; j < end; j += 64) {
        const uint32_t kk = load_key<DT>(p, j);
        if (kk > best) { best = kk; bidx = (uint32_t)j; }
    }
    uint64_t pk = wave_max_u64(((uint64_t)best << 32) | (uint64_t)(~bidx));
    if (lane == 0) atomicMax(packed + orow, (unsigned long long)pk);
}

__global__ void argmax_decode_kernel(unsigned long long *packed, int64_t R, int64_t *greedy) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) {
        greedy[r] = (int64_t)jfmb::decode_packed(packed[r]);
        packed[r] = 0ull;
    }
}

static int64_t env_i64(const char *name, int64_t dflt) {
    const char *e = getenv(name);
    return (e && *e) ? atoll(e) : dflt;
}

// Balanced chunking: cpr chunks per row of equal size (rounded up to `gran` elements).
static int64_t pick_chunk(int64_t gran, int64_t R, int64_t V, int64_t target_items) {
    int64_t c = env_i64("JF_ARGMAX_CHUNK", 0);
    if (c > 0) return ((c + gran - 1) / gran) * gran;
    int64_t per_row = (target_items + R - 1) / R;
    if (per_row < 1) per_row = 1;
    const int64_t max_per_row = (V + gran - 1) / gran;
    if (per_row > max_per_row) per_row = max_per_row;
    int64_t chunk = (V + per_row - 1) / per_row;
    return ((chunk + gran - 1) / gran) * gran;
}

static int argmax_launch(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride, const int32_t *out_index,
                         uint64_t *packed, void *stream) {
    if (R == 0) return JF_OK;
    if (!logits || !packed) return fail(JF_E_INVALID, "jf_argmax_partial: null pointer");
    if (dtype != JF_F32 && dtype != JF_BF16) return fail(JF_E_INVALID, "jf_argmax_partial: dtype %d", dtype);
    if (R < 0 || V <= 0 || row_stride < V || V > 0x7FFFFFFFll)
        return fail(JF_E_INVALID, "jf_argmax_partial: bad shape R=%lld V=%lld stride=%lld", (long long)R, (long long)V,
                    (long long)row_stride);
    const int esz = dtype == JF_F32 ? 4 : 2;
    const int epv = 16 / esz;
    const bool vec = (((uintptr_t)logits) % 16 == 0) && ((row_stride * esz) % 16 == 0);
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *pk = (unsigned long long *)packed;
    // Measured on MI355X (profiles/argmax_microbench_r01*.txt): big problems stream best as ~1 wavefront per SIMD
    // (1024 items, each a long contiguous range with 8 x 16 B per lane in flight: 6.8 TB/s fp32 at R>=512);
    // below ~140 MB the kernel is launch/ramp bound and 4-wave workgroups sharing a chunk (best vector kept in
    // registers, one item per ~64 KB, 256..1024 items) are 5-20 % faster.
    const int64_t bytes = R * V * esz;
    const bool wave_mode = vec && env_i64("JF_ARGMAX_WAVE", bytes >= (140ll << 20) ? 1 : 0) != 0;
    const int64_t unroll = env_i64("JF_ARGMAX_UNROLL", 8);       // 4, 8, or 16 (= two pipelined sets of 8)
    const bool deep = unroll >= 8;
    const bool pipe = unroll >= 16;
    if (wave_mode) {
        // one item per wavefront, ~one wavefront per SIMD (256 CUs x 4 SIMDs = 1024 slots).  Split each row into the
        // smallest number of chunks whose makespan ceil(items / 1024) * (V / per_row) is within 10 % of the best
        // split of up to 4 wavefronts per SIMD: e.g. R = 384 -> 5 chunks per row (1920 items, two rounds of V/5) instead
        // of 3 (1152 items: a second round for only 128 of them).
        int64_t items_target = env_i64("JF_ARGMAX_ITEMS", 0);
        if (items_target <= 0) {
            const int64_t slots = 1024, max_pr = (4 * slots + R - 1) / R;
            double best = 1e30;
            for (int64_t pr = 1; pr <= max_pr; ++pr) {
                const double ms = (double)((R * pr + slots - 1) / slots) / (double)pr;
                if (ms < best) best = ms;
            }
            int64_t pick = 1;
            for (int64_t pr = 1; pr <= max_pr; ++pr) {
                const double ms = (double)((R * pr + slots - 1) / slots) / (double)pr;
                if (ms <= best * 1.10) { pick = pr; break; }
            }
            items_target = R * pick;
        }
        const int64_t chunk = pick_chunk(64 * epv, R, V, items_target);
        const int64_t cpr = (V + chunk - 1) / chunk;
        const int64_t items = R * cpr;
        const int64_t blocks = (items + (AM_TPB / 64) - 1) / (AM_TPB / 64);
        if (blocks > 0x7FFFFFFFll) return fail(JF_E_CAPACITY, "jf_argmax_partial: grid too large");
        dim3 grid((unsigned)blocks), block(AM_TPB);
#define JF_LAUNCHW(DT, UNR) argmax_wave_kernel<DT, UNR><<<grid, block, 0, s>>>(logits, R, V, row_stride, pk, (int)cpr, chunk, out_index)
        if (dtype == JF_F32) { if (pipe) JF_LAUNCHW(JF_F32, 16); else if (deep) JF_LAUNCHW(JF_F32, 8); else JF_LAUNCHW(JF_F32, 4); }
        else { if (pipe) JF_LAUNCHW(JF_BF16, 16); else if (deep) JF_LAUNCHW(JF_BF16, 8); else JF_LAUNCHW(JF_BF16, 4); }
#undef JF_LAUNCHW
        return check_launch("argmax_wave_kernel");
    }
    int64_t wg_items = bytes >> 16;
    if (wg_items < 256) wg_items = 256;
    if (wg_items > 1024) wg_items = 1024;
    const int64_t chunk = pick_chunk((int64_t)AM_TPB * epv, R, V, env_i64("JF_ARGMAX_ITEMS", wg_items));
    const int64_t cpr = (V + chunk - 1) / chunk;
    const int64_t items = R * cpr;
    if (items > 0x7FFFFFFFll) return fail(JF_E_CAPACITY, "jf_argmax_partial: grid too large");
    dim3 grid((unsigned)items), block(AM_TPB);
#define JF_LAUNCH(DT, VECF, UNR) argmax_partial_kernel<DT, VECF, UNR><<<grid, block, 0, s>>>(logits, R, V, row_stride, pk, (int)cpr, chunk, out_index)
    if (dtype == JF_F32) {
        if (!vec) JF_LAUNCH(JF_F32, false, 4);
        else if (pipe) JF_LAUNCH(JF_F32, true, 16);
        else if (deep) JF_LAUNCH(JF_F32, true, 8);
        else JF_LAUNCH(JF_F32, true, 4);
    } else {
        if (!vec) JF_LAUNCH(JF_BF16, false, 4);
        else if (pipe) JF_LAUNCH(JF_BF16, true, 16);
        else if (deep) JF_LAUNCH(JF_BF16, true, 8);
        else JF_LAUNCH(JF_BF16, true, 4);
    }
#undef JF_LAUNCH
    return check_launch("argmax_partial_kernel");
}

extern "C" int jf_argmax_partial(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride,
                                 uint64_t *packed, void *stream) {
    return argmax_launch(logits, dtype, R, V, row_stride, nullptr, packed, stream);
}

extern "C" int jf_argmax_scatter(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride,
                                 const int32_t *out_index, uint64_t *packed, void *stream) {
    if (R > 0 && !out_index) return fail(JF_E_INVALID, "jf_argmax_scatter: null out_index");
    return argmax_launch(logits, dtype, R, V, row_stride, out_index, packed, stream);
}

extern "C" int jf_argmax_decode(uint64_t *packed, int64_t R, int64_t *greedy, void *stream) {
    if (R == 0) return JF_OK;
    if (!packed || !greedy || R < 0) return fail(JF_E_INVALID, "jf_argmax_decode: bad argument");
    argmax_decode_kernel<<<dim3((unsigned)((R + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        (unsigned long long *)packed, R, greedy);
    return check_launch("argmax_decode_kernel");
}

extern "C" int jf_argmax_rows(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride, uint64_t *packed,
                              int64_t *greedy, void *stream) {
    int rc = jf_argmax_partial(logits, dtype, R, V, row_stride, packed, stream);
    if (rc) return rc;
    return jf_argmax_decode(packed, R, greedy, stream);
}

// ------------------------------------------------------------------------------------------------
// (a3) accepted-prefix scan
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void accept_lengths_kernel(const int64_t *draft, int draft_rows, const int64_t *greedy,
                                                              int64_t greedy_stride, int B, int L, int32_t *accepted,
                                                              int32_t *best_idx) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int b = wave; b < B; b += 4) {
        const int64_t *d = draft + (int64_t)(draft_rows == 1 ? 0 : b) * L;
        const int64_t *g = greedy + (int64_t)b * greedy_stride;
        int m = L - 1;
        for (int i0 = 0; i0 < L - 1; i0 += 64) {   // ballot + first-set-bit per 64 tokens
            const int i = i0 + lane;
            const bool mis = (i < L - 1) && (d[i + 1] != g[i]);
            const unsigned long long bal = __ballot(mis);
            if (bal) { m = i0 + __ffsll((long long)bal) - 1; break; }
        }
        if (lane == 0) accepted[b] = (L == 0) ? 0 : m + 1;
    }
    __syncthreads();
    if (threadIdx.x == 0 && best_idx) {
        int best = -1, bi = 0;
        for (int b = 0; b < B; ++b)
            if (accepted[b] > best) { best = accepted[b]; bi = b; }
        *best_idx = bi;
    }
}

extern "C" int jf_accept_lengths(const int64_t *draft, int draft_rows, const int64_t *greedy, int64_t greedy_stride, int B,
                                 int L, int32_t *accepted, int32_t *best_idx, void *stream) {
    if (B <= 0) return JF_OK;
    if (!draft || !greedy || !accepted) return fail(JF_E_INVALID, "jf_accept_lengths: null pointer");
    if (draft_rows != 1 && draft_rows != B)
        return fail(JF_E_INVALID, "jf_accept_lengths: draft rows %d do not broadcast against %d", draft_rows, B);
    if (L < 0 || greedy_stride < L - 1) return fail(JF_E_INVALID, "jf_accept_lengths: bad L/stride");
    accept_lengths_kernel<<<1, 256, 0, (hipStream_t)stream>>>(draft, draft_rows, greedy, greedy_stride, B, L, accepted, best_idx);
    return check_launch("accept_lengths_kernel");
}

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
// KV cache: append (scatter) and candidate-row commit.  A token row is D elements = D*esz bytes
// (256 B for bf16 / D=128): 16 lanes x 16 B, four token rows per wavefront instruction.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kv_append_kernel(uint4 *__restrict__ k_cache, uint4 *__restrict__ v_cache,
                                                         const uint4 *__restrict__ k_new, const uint4 *__restrict__ v_new,
                                                         const int64_t *__restrict__ slot, int64_t N, int H_kv,
                                                         int vec_per_row, int64_t S_max, int64_t k_tok_vecs,
                                                         int64_t v_tok_vecs) {
    // one (token, head) row per vec_per_row lanes
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rowid = gid / vec_per_row;
    const int lane = (int)(gid - rowid * vec_per_row);
    if (rowid >= N * H_kv) return;
    const int64_t tok = rowid / H_kv;
    const int h = (int)(rowid - tok * H_kv);
    const int64_t sl = slot[tok];
    if (sl < 0) return;                                   // ATT:24 (slot == -1 skipped)
    const int64_t brow = sl / S_max, pos = sl - brow * S_max;
    const int64_t dst = ((brow * H_kv + h) * S_max + pos) * vec_per_row + lane;
    const int64_t hoff = (int64_t)h * vec_per_row + lane;
    k_cache[dst] = k_new[tok * k_tok_vecs + hoff];
    v_cache[dst] = v_new[tok * v_tok_vecs + hoff];
}

extern "C" int jf_kv_append(void *k_cache, void *v_cache, const void *k_new, const void *v_new, const int64_t *slot, int64_t N,
                            int32_t H_kv, int32_t D, int64_t S_max, int64_t k_tok_stride, int64_t v_tok_stride,
                            int32_t elem_bytes, void *stream) {
    if (N <= 0) return JF_OK;
    if (!k_cache || !v_cache || !k_new || !v_new || !slot) return fail(JF_E_INVALID, "jf_kv_append: null pointer");
    const int64_t row_bytes = (int64_t)D * elem_bytes;
    if (row_bytes % 16 != 0 || H_kv <= 0 || S_max <= 0) return fail(JF_E_INVALID, "jf_kv_append: row bytes %lld not /16", (long long)row_bytes);
    if ((k_tok_stride * elem_bytes) % 16 != 0 || (v_tok_stride * elem_bytes) % 16 != 0 || k_tok_stride < (int64_t)H_kv * D ||
        v_tok_stride < (int64_t)H_kv * D || ((uintptr_t)k_new) % 16 != 0 || ((uintptr_t)v_new) % 16 != 0)
        return fail(JF_E_INVALID, "jf_kv_append: source strides/pointers must be 16-byte aligned and >= H_kv*D");
    const int vpr = (int)(row_bytes / 16);
    const int64_t threads = N * H_kv * vpr;
    kv_append_kernel<<<dim3((unsigned)((threads + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
        (uint4 *)k_cache, (uint4 *)v_cache, (const uint4 *)k_new, (const uint4 *)v_new, slot, N, H_kv, vpr, S_max,
        k_tok_stride * elem_bytes / 16, v_tok_stride * elem_bytes / 16);
    return check_launch("kv_append_kernel");
}

// ---- fused RoPE + Q re-layout + KV append (one launch per layer instead of ~10 elementwise launches) -------------
template <typename T> __device__ __forceinline__ float ld_f(const T *p);
template <> __device__ __forceinline__ float ld_f<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ld_f<uint16_t>(const uint16_t *p) { return __uint_as_float(((uint32_t)*p) << 16); }
template <typename T> __device__ __forceinline__ void st_f(T *p, float v);
template <> __device__ __forceinline__ void st_f<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_f<uint16_t>(uint16_t *p, float v) {      // round-to-nearest-even like torch
    uint32_t u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) { *p = 0x7FC0; return; }
    u += 0x7FFFu + ((u >> 16) & 1u);
    *p = (uint16_t)(u >> 16);
}

template <typename T>
__global__ __launch_bounds__(256) void rope_kv_append_kernel(const T *__restrict__ qkv, int64_t N, int Tlen, int nq, int nkv, int D,
                                                              const int32_t *__restrict__ positions, const float *__restrict__ cos_t,
                                                              const float *__restrict__ sin_t, T *__restrict__ q_out,
                                                              T *__restrict__ k_cache, T *__restrict__ v_cache,
                                                              const int64_t *__restrict__ slot_main, int64_t S_max,
                                                              T *__restrict__ k_cand, T *__restrict__ v_cand,
                                                              const int64_t *__restrict__ slot_cand, int64_t T_max) {
    const int half = D >> 1;
    const int heads = nq + 2 * nkv;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = N * heads * half;
    if (gid >= total) return;
    const int i = (int)(gid % half);
    const int64_t th = gid / half;
    const int h = (int)(th % heads);
    const int64_t tok = th / heads;
    const T *src = qkv + (tok * heads + h) * D;
    const float x1 = ld_f(src + i), x2 = ld_f(src + half + i);
    float o1 = x1, o2 = x2;
    if (h < nq + nkv) {                                   // rotate q and k heads, v passes through
        const int64_t pos = positions[tok];
        const float c = cos_t[pos * half + i], sn = sin_t[pos * half + i];
        o1 = x1 * c - x2 * sn;
        o2 = x2 * c + x1 * sn;
    }
    if (h < nq) {
        const int G = nq / nkv;
        const int kvh = h / G, g = h - kvh * G;
        const int64_t r = tok / Tlen, t = tok - r * Tlen;
        T *dst = q_out + (((r * nkv + kvh) * (int64_t)G * Tlen) + (int64_t)g * Tlen + t) * D;
        st_f(dst + i, o1);
        st_f(dst + half + i, o2);
        return;
    }
    const bool is_v = h >= nq + nkv;
    const int kvh = is_v ? h - nq - nkv : h - nq;
    const int64_t sm = slot_main[tok];
    if (sm >= 0) {
        const int64_t brow = sm / S_max, pos = sm - brow * S_max;
        T *dst = (is_v ? v_cache : k_cache) + ((brow * nkv + kvh) * S_max + pos) * D;
        st_f(dst + i, o1);
        st_f(dst + half + i, o2);
    }
    if (slot_cand) {
        const int64_t sc = slot_cand[tok];
        if (sc >= 0) {
            const int64_t brow = sc / T_max, pos = sc - brow * T_max;
            T *dst = (is_v ? v_cand : k_cand) + ((brow * nkv + kvh) * T_max + pos) * D;
            st_f(dst + i, o1);
            st_f(dst + half + i, o2);
        }
    }
}

extern "C" int jf_rope_kv_append(const void *qkv, int dtype, int64_t N, int32_t T, int32_t nq, int32_t nkv, int32_t D,
                                 const int32_t *positions, const float *cos_table, const float *sin_table, void *q_out,
                                 void *k_cache, void *v_cache, const int64_t *slot_main, int64_t S_max, void *k_cand,
                                 void *v_cand, const int64_t *slot_cand, int64_t T_max, void *stream) {
    if (N <= 0) return JF_OK;
    if (!qkv || !positions || !cos_table || !sin_table || !q_out || !k_cache || !v_cache || !slot_main)
        return fail(JF_E_INVALID, "jf_rope_kv_append: null pointer");
    if (T <= 0 || N % T != 0 || nq <= 0 || nkv <= 0 || nq % nkv != 0 || D <= 0 || (D & 1) || S_max <= 0)
        return fail(JF_E_INVALID, "jf_rope_kv_append: bad shape N=%lld T=%d nq=%d nkv=%d D=%d", (long long)N, T, nq, nkv, D);
    if (slot_cand && (!k_cand || !v_cand || T_max <= 0)) return fail(JF_E_INVALID, "jf_rope_kv_append: candidate cache missing");
    const int64_t total = N * (nq + 2 * nkv) * (D / 2);
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == JF_F32)
        rope_kv_append_kernel<float><<<grid, block, 0, s>>>((const float *)qkv, N, T, nq, nkv, D, positions, cos_table, sin_table,
                                                         (float *)q_out, (float *)k_cache, (float *)v_cache, slot_main, S_max,
                                                         (float *)k_cand, (float *)v_cand, slot_cand, T_max);
    else if (dtype == JF_BF16)
        rope_kv_append_kernel<uint16_t><<<grid, block, 0, s>>>((const uint16_t *)qkv, N, T, nq, nkv, D, positions, cos_table, sin_table,
                                                            (uint16_t *)q_out, (uint16_t *)k_cache, (uint16_t *)v_cache, slot_main,
                                                            S_max, (uint16_t *)k_cand, (uint16_t *)v_cand, slot_cand, T_max);
    else return fail(JF_E_INVALID, "jf_rope_kv_append: dtype %d", dtype);
    return check_launch("rope_kv_append_kernel");
}

// ---- SwiGLU gate: 8 bf16 (or 4 fp32) per lane per load on both halves -------------------------------------------
template <typename T, int EPV>
__global__ __launch_bounds__(256) void swiglu_kernel(const T *__restrict__ gu, int64_t M, int64_t I, T *__restrict__ out) {
    const int64_t vec_per_row = I / EPV;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= M * vec_per_row) return;
    const int64_t m = gid / vec_per_row, v = gid - m * vec_per_row;
    const T *g = gu + m * 2 * I + v * EPV;
    const T *u = g + I;
    T *o = out + m * I + v * EPV;
    T gv[EPV], uv[EPV], ov[EPV];
    *reinterpret_cast<uint4 *>(gv) = *reinterpret_cast<const uint4 *>(g);
    *reinterpret_cast<uint4 *>(uv) = *reinterpret_cast<const uint4 *>(u);
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
        const float x = ld_f(gv + j), y = ld_f(uv + j);
        st_f(ov + j, (x / (1.f + expf(-x))) * y);
    }
    *reinterpret_cast<uint4 *>(o) = *reinterpret_cast<const uint4 *>(ov);
}

extern "C" int jf_swiglu(const void *gu, int dtype, int64_t M, int64_t I, void *out, void *stream) {
    if (M <= 0 || I <= 0) return JF_OK;
    if (!gu || !out) return fail(JF_E_INVALID, "jf_swiglu: null pointer");
    const int epv = dtype == JF_F32 ? 4 : 8;
    if ((dtype != JF_F32 && dtype != JF_BF16) || I % epv != 0 || ((uintptr_t)gu) % 16 || ((uintptr_t)out) % 16)
        return fail(JF_E_INVALID, "jf_swiglu: dtype/alignment (I must be a multiple of %d)", epv);
    const int64_t total = M * (I / epv);
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (dtype == JF_F32) swiglu_kernel<float, 4><<<grid, block, 0, (hipStream_t)stream>>>((const float *)gu, M, I, (float *)out);
    else swiglu_kernel<uint16_t, 8><<<grid, block, 0, (hipStream_t)stream>>>((const uint16_t *)gu, M, I, (uint16_t *)out);
    return check_launch("swiglu_kernel");
}

__global__ __launch_bounds__(256) void kv_commit_kernel(void *const *main_k, void *const *main_v, void *const *cand_k,
                                                         void *const *cand_v, const jf_mb_desc *desc, int cand_rows, int H_kv,
                                                         int vec_per_row, int64_t S_max, int64_t T_max) {
    const int p = blockIdx.x;
    const int layer = blockIdx.y >> 1, which = blockIdx.y & 1;
    const jf_mb_desc d = desc[p];
    if (d.kv_copy_len <= 0 || d.kv_src_row <= 0) return;
    const uint4 *src = (const uint4 *)(which ? cand_v[layer] : cand_k[layer]);
    uint4 *dst = (uint4 *)(which ? main_v[layer] : main_k[layer]);
    const int64_t crow = (int64_t)p * cand_rows + (d.kv_src_row - 1);
    const int64_t total = (int64_t)H_kv * d.kv_copy_len * vec_per_row;
    for (int64_t i = threadIdx.x; i < total; i += blockDim.x) {
        const int lane = (int)(i % vec_per_row);
        const int64_t r = i / vec_per_row;
        const int t = (int)(r % d.kv_copy_len);
        const int h = (int)(r / d.kv_copy_len);
        const int64_t s_off = ((crow * H_kv + h) * T_max + t) * vec_per_row + lane;
        const int64_t d_off = (((int64_t)p * H_kv + h) * S_max + d.kv_copy_dst + t) * vec_per_row + lane;
        dst[d_off] = src[s_off];
    }
}

extern "C" int jf_kv_commit(void *const *main_k, void *const *main_v, void *const *cand_k, void *const *cand_v, int32_t layers,
                            const jf_mb_desc *desc, int P, int32_t cand_rows, int32_t H_kv, int32_t D, int64_t S_max,
                            int64_t T_max, int32_t elem_bytes, void *stream) {
    if (P <= 0 || layers <= 0 || cand_rows <= 0) return JF_OK;
    if (!main_k || !main_v || !cand_k || !cand_v || !desc) return fail(JF_E_INVALID, "jf_kv_commit: null pointer");
    const int64_t row_bytes = (int64_t)D * elem_bytes;
    if (row_bytes % 16 != 0) return fail(JF_E_INVALID, "jf_kv_commit: row bytes %lld not /16", (long long)row_bytes);
    kv_commit_kernel<<<dim3(P, layers * 2), 256, 0, (hipStream_t)stream>>>(main_k, main_v, cand_k, cand_v, desc, cand_rows, H_kv,
                                                                         (int)(row_bytes / 16), S_max, T_max);
    return check_launch("kv_commit_kernel");
}

// ------------------------------------------------------------------------------------------------
// engine single-block step: 4 wavefronts take the rows round-robin, then one pass hands out pads
// ------------------------------------------------------------------------------------------------
struct WaveLanes {   // one wavefront inside a 256-thread workgroup; no workgroup barrier inside a row
    __device__ __forceinline__ int lane() const { return threadIdx.x & 63; }
    __device__ __forceinline__ int count() const { return 64; }
    __device__ __forceinline__ void sync() const {}
    __device__ __forceinline__ int reduce_min(int v) const { return wave_min_i32(v); }
    __device__ __forceinline__ int reduce_sum(int v) const { return wave_sum_i32(v); }
};

__global__ __launch_bounds__(256) void engine_step_kernel(const int64_t *draft, int B, int L, unsigned long long *packed,
                                                           int eos_id, const int32_t *remaining, int64_t *new_tokens,
                                                           int64_t *next_draft, const int64_t *pad_stream, int64_t pad_len,
                                                           int64_t *pad_cursor, jf_engine_row *rows) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int b = wave; b < B; b += 4) {
        const unsigned long long *pk = packed + (int64_t)b * (L - 1);
        auto G = [pk](int i) { return jfmb::decode_packed(pk[i]); };
        jfmb::EngineRowOut o = jfmb::engine_row_body(WaveLanes{}, draft + (int64_t)b * L, L, G, eos_id, remaining[b],
                                                     new_tokens + (int64_t)b * L, next_draft + (int64_t)b * L);
        if (lane == 0) {
            rows[b].acc_len = o.acc_len; rows[b].n_new = o.n_new; rows[b].eos = o.eos; rows[b].active_next = o.active_next;
            rows[b].n_pads = o.active_next ? (L - 1 - o.copy_len) : 0;
            rows[b].rsv[0] = o.copy_len;
        }
    }
    __syncthreads();
    // pads in row order (JD:705-707 consumes torch.randint sequentially): exclusive scan over rows
    __shared__ int64_t s_base;
    if (threadIdx.x == 0) s_base = *pad_cursor;
    __syncthreads();
    int64_t run = s_base;
    for (int b = 0; b < B; ++b) {
        const int np = rows[b].n_pads, cl = rows[b].rsv[0];
        for (int i = threadIdx.x; i < np; i += blockDim.x) {
            const int64_t k = run + i;
            next_draft[(int64_t)b * L + 1 + cl + i] = pad_stream[pad_len > 0 ? (k % pad_len) : 0];
        }
        run += np;
    }
    __syncthreads();
    if (threadIdx.x == 0) *pad_cursor = run;
    for (int64_t i = threadIdx.x; i < (int64_t)B * (L - 1); i += blockDim.x) packed[i] = 0ull;
}

extern "C" int jf_engine_step(const int64_t *draft, int B, int L, uint64_t *packed, int32_t eos_id, const int32_t *remaining_tokens,
                              int64_t *new_tokens, int64_t *next_draft, const int64_t *pad_stream, int64_t pad_stream_len,
                              int64_t *pad_cursor, jf_engine_row *rows, void *stream) {
    if (B <= 0) return JF_OK;
    if (L < 2) return fail(JF_E_INVALID, "Draft must have at least 2 tokens (seed + 1 speculative)");   // MR:1144-1145
    if (!draft || !packed || !remaining_tokens || !new_tokens || !next_draft || !pad_cursor || !rows || (!pad_stream && pad_stream_len > 0))
        return fail(JF_E_INVALID, "jf_engine_step: null pointer");
    engine_step_kernel<<<1, 256, 0, (hipStream_t)stream>>>(draft, B, L, (unsigned long long *)packed, eos_id, remaining_tokens,
                                                        new_tokens, next_draft, pad_stream, pad_stream_len, pad_cursor, rows);
    return check_launch("engine_step_kernel");
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

// ------------------------------------------------------------------------------------------------
// (a19) non-greedy verify: fused online-softmax gather + argmax, logits read once
// ------------------------------------------------------------------------------------------------
template <int DT>
__device__ __forceinline__ float load_f(const void *row, int64_t i) {
    if constexpr (DT == JF_F32) return ((const float *)row)[i];
    else return __uint_as_float(((uint32_t)((const uint16_t *)row)[i]) << 16);
}

// Stage 1 — one workgroup per (row, chunk): 16 B per lane per load, four vectors (16/32 elements) per lane per round.
// (hot loop: hardware v_exp_f32 via __expf, ~1e-6 relative; the verify tolerance is 2e-5)
// Per round the lane first raises its running max over the whole round (register-resident values), rescales its sum once,
// then adds exp(x - m) for every element: one exp per element plus one per round, branch-free.  The argmax tracker runs on
// the same registers.  Partials (m, s) go to the workspace, the argmax to `packed` by atomicMax.
template <int DT, int NV>
__device__ __forceinline__ void rs_round(const u32x4 (&vv)[NV], float inv_t, float &m, float &s) {
    constexpr int EPV = Elem<DT>::EPV;
    constexpr int NE = NV * EPV;
    float x[NE];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const uint32_t w[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (DT == JF_F32) {
                x[u * 4 + j] = __uint_as_float(w[j]) * inv_t;
            } else {
                x[u * 8 + 2 * j] = __uint_as_float(w[j] << 16) * inv_t;
                x[u * 8 + 2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u) * inv_t;
            }
        }
    }
    float mx = x[0];
#pragma unroll
    for (int j = 1; j < NE; ++j) mx = fmaxf(mx, x[j]);
    const float mn = fmaxf(m, mx);
    if (mn == -INFINITY) return;                        // nothing finite yet: keep (m, s) = (-inf, 0), never form inf - inf
    float acc = (m == -INFINITY) ? 0.f : s * __expf(m - mn);
#pragma unroll
    for (int j = 0; j < NE; ++j) acc += __expf(x[j] - mn);
    s = acc;
    m = mn;
}

template <int DT, bool VEC>
__global__ __launch_bounds__(256) void rs_probs_partial_kernel(const void *logits, int64_t R, int64_t V, int64_t row_stride,
                                                                float inv_t, float2 *__restrict__ partial,
                                                                unsigned long long *packed, int cpr, int64_t chunk_elems) {
    using E = Elem<DT>;
    constexpr int EPV = E::EPV;
    const int64_t item = blockIdx.x;
    const int64_t row = item / cpr;
    const int c = (int)(item - row * cpr);
    const int64_t begin = (int64_t)c * chunk_elems;
    int64_t end = begin + chunk_elems;
    if (end > V) end = V;
    const typename E::T *p = (const typename E::T *)logits + row * row_stride;
    const int tid = threadIdx.x;
    float m = -INFINITY, s = 0.f;
    uint32_t best = 0u, bidx = 0xFFFFFFFFu;
    int64_t done = begin;
    if constexpr (VEC) {
        const int nvec = (int)((end - begin) / EPV);
        const u32x4 *q = (const u32x4 *)p + (begin / EPV) + tid;
        const uint32_t ebase = (uint32_t)begin;
        FastTrack<DT, true> ft;                          // same vector-granular argmax tracker as the greedy kernel
        int k = tid;
        for (; k + 3 * 256 < nvec; k += 4 * 256, q += 4 * 256) {
            const u32x4 vv[4] = {JF_LOAD(q), JF_LOAD(q + 256), JF_LOAD(q + 512), JF_LOAD(q + 768)};
#pragma unroll
            for (int u = 0; u < 4; ++u) ft.consume(vv[u], ebase + (uint32_t)(k + u * 256) * EPV);
            rs_round<DT, 4>(vv, inv_t, m, s);
        }
        for (; k < nvec; k += 256, q += 256) {          // this lane's remaining vectors, one at a time
            const u32x4 vv[1] = {JF_LOAD(q)};
            ft.consume(vv[0], ebase + (uint32_t)k * EPV);
            rs_round<DT, 1>(vv, inv_t, m, s);
        }
        done = begin + (int64_t)nvec * EPV;
        if (__syncthreads_or(ft.saw_nan() ? 1 : 0)) {
            scan_exact<DT>(p, begin, done, tid, best, bidx);               // NaN in the chunk: exact key rescan
        } else if (ft.bvec != 0xFFFFFFFFu) {
            best = ft.ukey();
            bidx = ft.resolve(p);
        }
    }
    for (int64_t i = done + tid; i < end; i += 256) {    // unaligned rows / ragged tail (V % EPV)
        const uint32_t kk = load_key<DT>(p, i);
        if (kk > best) { best = kk; bidx = (uint32_t)i; }
        const float xv = load_f<DT>(p, i) * inv_t;
        if (xv > m) { s = (m == -INFINITY ? 0.f : s * __expf(m - xv)) + 1.f; m = xv; }
        else if (xv != -INFINITY) s += __expf(xv - m);
    }
    // merge (m, s) pairs: six shuffle steps inside the wavefront, then one LDS hop across the four wavefronts
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float m2 = __shfl_xor(m, off, 64), s2 = __shfl_xor(s, off, 64);
        const float M = fmaxf(m, m2);
        s = (M == -INFINITY) ? 0.f : ((m == -INFINITY ? 0.f : s * expf(m - M)) + (m2 == -INFINITY ? 0.f : s2 * expf(m2 - M)));
        m = M;
    }
    __shared__ float sm[4], ss[4];
    __shared__ uint64_t sp[4];
    uint64_t pk = wave_max_u64(((uint64_t)best << 32) | (uint64_t)(~bidx));
    if ((tid & 63) == 0) { sp[tid >> 6] = pk; sm[tid >> 6] = m; ss[tid >> 6] = s; }
    __syncthreads();
    if (tid == 0) {
        float M = -INFINITY;
        for (int i = 0; i < 4; ++i) M = sm[i] > M ? sm[i] : M;
        float Ssum = 0.f;
        for (int i = 0; i < 4; ++i) Ssum += (sm[i] == -INFINITY) ? 0.f : ss[i] * expf(sm[i] - M);
        partial[item] = make_float2(M, Ssum);
        uint64_t mm = sp[0];
        for (int w = 1; w < 4; ++w) mm = sp[w] > mm ? sp[w] : mm;
        atomicMax(packed + row, (unsigned long long)mm);
    }
}

// Stage 2 — one thread per row: merge the chunk partials, then the gathered probability of the drafted id.
template <int DT>
__global__ __launch_bounds__(256) void rs_probs_finish_kernel(const void *logits, int64_t R, int64_t V, int64_t row_stride,
                                                               const int64_t *draft_next, float inv_t, const float2 *partial,
                                                               int cpr, float *p_draft, float *row_max, float *row_sumexp) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= R) return;
    float M = -INFINITY;
    for (int c = 0; c < cpr; ++c) M = fmaxf(M, partial[row * cpr + c].x);
    float S = 0.f;
    for (int c = 0; c < cpr; ++c) {
        const float2 ps = partial[row * cpr + c];
        S += (ps.x == -INFINITY) ? 0.f : ps.y * expf(ps.x - M);
    }
    row_max[row] = M;
    row_sumexp[row] = S;
    const int64_t tok = draft_next[row];
    const void *p = (const char *)logits + row * row_stride * (DT == JF_F32 ? 4 : 2);
    p_draft[row] = (tok >= 0 && tok < V) ? expf(load_f<DT>(p, tok) * inv_t - M) / S : 0.f;
}

static int64_t rs_chunk(int dtype, int64_t R, int64_t V, int64_t *cpr_out) {
    const int64_t gran = 4 * (int64_t)256 * (dtype == JF_F32 ? 4 : 8);     // one full round per workgroup
    int64_t per_row = (2048 + R - 1) / R;
    if (per_row < 1) per_row = 1;
    if (per_row > 64) per_row = 64;
    int64_t chunk = (V + per_row - 1) / per_row;
    chunk = ((chunk + gran - 1) / gran) * gran;
    *cpr_out = (V + chunk - 1) / chunk;
    return chunk;
}

extern "C" size_t jf_rs_workspace_bytes(int64_t R, int64_t V) {
    (void)V;
    return (size_t)(R > 0 ? R : 0) * 64 * sizeof(float2);
}

extern "C" int jf_rs_probs(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride, const int64_t *draft_next,
                           float temperature, float *p_draft, float *row_max, float *row_sumexp, uint64_t *packed,
                           void *workspace, size_t workspace_bytes, void *stream) {
    if (R <= 0) return JF_OK;
    if (!logits || !draft_next || !p_draft || !row_max || !row_sumexp || !packed || !workspace)
        return fail(JF_E_INVALID, "jf_rs_probs: null pointer");
    if (dtype != JF_F32 && dtype != JF_BF16) return fail(JF_E_INVALID, "jf_rs_probs: dtype %d", dtype);
    if (workspace_bytes < jf_rs_workspace_bytes(R, V)) return fail(JF_E_INVALID, "jf_rs_probs: workspace too small");
    const float t = (temperature <= 0.f) ? 1.f : temperature;    // JDN:66-67
    const float inv_t = 1.f / t;
    const int esz = dtype == JF_F32 ? 4 : 2;
    const bool vec = (((uintptr_t)logits) % 16 == 0) && ((row_stride * esz) % 16 == 0);
    int64_t cpr = 1;
    const int64_t chunk = rs_chunk(dtype, R, V, &cpr);
    const dim3 grid((unsigned)(R * cpr)), block(256);
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *pk = (unsigned long long *)packed;
    float2 *part = (float2 *)workspace;
    if (dtype == JF_F32) {
        if (vec) rs_probs_partial_kernel<JF_F32, true><<<grid, block, 0, s>>>(logits, R, V, row_stride, inv_t, part, pk, (int)cpr, chunk);
        else rs_probs_partial_kernel<JF_F32, false><<<grid, block, 0, s>>>(logits, R, V, row_stride, inv_t, part, pk, (int)cpr, chunk);
        rs_probs_finish_kernel<JF_F32><<<dim3((unsigned)((R + 255) / 256)), 256, 0, s>>>(logits, R, V, row_stride, draft_next, inv_t, part, (int)cpr, p_draft, row_max, row_sumexp);
    } else {
        if (vec) rs_probs_partial_kernel<JF_BF16, true><<<grid, block, 0, s>>>(logits, R, V, row_stride, inv_t, part, pk, (int)cpr, chunk);
        else rs_probs_partial_kernel<JF_BF16, false><<<grid, block, 0, s>>>(logits, R, V, row_stride, inv_t, part, pk, (int)cpr, chunk);
        rs_probs_finish_kernel<JF_BF16><<<dim3((unsigned)((R + 255) / 256)), 256, 0, s>>>(logits, R, V, row_stride, draft_next, inv_t, part, (int)cpr, p_draft, row_max, row_sumexp);
    }
    return check_launch("rs_probs kernels");
}

// ------------------------------------------------------------------------------------------------
// Accept/reject of every row of a batch (JDN:581-639).  The reference visits the rows in order and draws torch.rand /
// torch.multinomial / torch.randint as it goes, so the position of every draw in the injected streams depends on the rows
// before it.  Three launches keep that order exact while the only wide work — the inverse-CDF draw of the bonus token on a
// rejected position, two passes over V — runs one workgroup per row in parallel:
//   rs_accept_kernel  (1 workgroup)  sequential accept scans from LDS-staged p_draft / uniforms: per row accepted count,
//                                    rejected position, uniforms used; bonus draws are ASSUMED to take one draw per row
//   rs_bonus_kernel   (B workgroups) the bonus draw of each rejected row at its assumed stream position
//   rs_finish_kernel  (1 workgroup)  if some row needed more than one draw (its sample hit the proposed token) the rows
//                                    after it are redone in order with the true positions (rare); EOS, next drafts, pads,
//                                    cursors, packed re-zeroed
// ------------------------------------------------------------------------------------------------
template <int DT>
__device__ __forceinline__ double rs_slice_sum(const void *row, int64_t lo, int64_t hi, float inv_t, float M, float Sx) {
    double acc = 0.0;
    for (int64_t i = lo; i < hi; ++i) acc += (double)(expf(load_f<DT>(row, i) * inv_t - M) / Sx);
    return acc;
}
template <int DT>
__device__ __forceinline__ int64_t rs_slice_pick(const void *row, int64_t lo, int64_t hi, float inv_t, float M, float Sx,
                                                 double pre, double thr) {
    double run = pre;
    for (int64_t i = lo; i < hi; ++i) {
        run += (double)(expf(load_f<DT>(row, i) * inv_t - M) / Sx);
        if (run > thr) return i;
    }
    return hi - 1;
}

struct RsShared {
    double sum[256], pre[256];
    double total;
    uint64_t best[4];
    int pick;
};

// Residual sampling of one row by the whole workgroup (JDN:135-153 / JDO:157-168): inverse CDF over p = exp(x/T - M)/S in
// vocabulary order with a float64 running sum (two-level: each thread owns a contiguous slice), up to 16 draws from
// stream[(base + tr) % len] until the sample differs from `proposed`, then the argmax of the masked distribution.
// Uniform control flow; returns the token, *draws = stream entries consumed.
template <int DT>
__device__ int rs_bonus_row(const void *row, int64_t V, float inv_t, float M, float Sx, int64_t proposed, const float *stream,
                            int64_t stream_len, int64_t base, RsShared &sh, int *draws_out) {
    const int tid = threadIdx.x;
    const int64_t per = (V + 255) / 256;
    const int64_t lo = (int64_t)tid * per < V ? (int64_t)tid * per : V;
    const int64_t hi = (lo + per < V) ? lo + per : V;
    const double acc = rs_slice_sum<DT>(row, lo, hi, inv_t, M, Sx);
    sh.sum[tid] = acc;
    __syncthreads();
    if (tid == 0) {
        double run = 0.0;
        for (int i = 0; i < 256; ++i) { sh.pre[i] = run; run += sh.sum[i]; }
        sh.total = run;
    }
    __syncthreads();
    int bonus = -1, draws = 0;
    for (int tr = 0; tr < 16 && bonus < 0; ++tr) {
        const double thr = (double)stream[(base + tr) % stream_len] * sh.total;
        if (tid == 0) sh.pick = (int)(V - 1);               // clamp when thr >= total
        __syncthreads();
        const double pre = sh.pre[tid];
        if (hi > lo && thr >= pre && thr < pre + acc) sh.pick = (int)rs_slice_pick<DT>(row, lo, hi, inv_t, M, Sx, pre, thr);
        __syncthreads();
        draws++;
        if ((int64_t)sh.pick != proposed) bonus = sh.pick;
        __syncthreads();
    }
    if (bonus < 0) {
        // 16 collisions: argmax of p with the proposed id masked (JDN:147-153); all mass on it -> keep it
        uint32_t best = 0u, bidx = 0xFFFFFFFFu;
        for (int64_t i = tid; i < V; i += 256) {
            if (i == proposed) continue;
            const uint32_t k = load_key<DT>(row, i);
            if (k > best) { best = k; bidx = (uint32_t)i; }
        }
        const uint64_t pk = wave_max_u64(((uint64_t)best << 32) | (uint64_t)(~bidx));
        if ((tid & 63) == 0) sh.best[tid >> 6] = pk;
        __syncthreads();
        uint64_t mm = sh.best[0];
        for (int w = 1; w < 4; ++w) mm = sh.best[w] > mm ? sh.best[w] : mm;
        const int alt = jfmb::decode_packed(mm);
        const float palt = (alt >= 0 && alt < V) ? expf(load_f<DT>(row, alt) * inv_t - M) / Sx : 0.f;
        bonus = (palt > 0.f) ? alt : (int)proposed;
        __syncthreads();
    }
    *draws_out = draws;
    return bonus;
}

constexpr int RS_STAGE = 8192;      // floats of p_draft / uniforms staged in LDS by the accept scan (B * (L-1) <= this, else global)

__global__ __launch_bounds__(256) void rs_accept_kernel(const int64_t *draft, int B, int L, const float *p_draft, int eos_id,
                                                         const float *u_stream, int64_t u_len, const int64_t *u_cursor,
                                                         int64_t *committed, jf_rs_row *rows) {
    __shared__ float s_p[RS_STAGE], s_u[RS_STAGE];
    const int tid = threadIdx.x;
    const int n = B * (L - 1);
    const int64_t uc0 = *u_cursor;
    const bool staged = n <= RS_STAGE;
    if (staged) {
        for (int i = tid; i < n; i += 256) { s_p[i] = p_draft[i]; s_u[i] = u_stream[(uc0 + i) % u_len]; }   // at most n uniforms are used
    }
    __syncthreads();
    if (tid != 0) return;
    int used_total = 0, n_rej = 0;
    for (int b = 0; b < B; ++b) {                                   // JDN:326-348, rows in order
        const int64_t *d = draft + (int64_t)b * L;
        int64_t *cm = committed + (int64_t)b * L;
        const int r0 = b * (L - 1);
        int nacc = 0, eos = 0, rej = -1, used = 0;
        for (int t = 0; t < L - 1; ++t) {
            const int64_t proposed = d[t + 1];
            const float u = staged ? s_u[used_total + used] : u_stream[(uc0 + used_total + used) % u_len];
            const float pd = staged ? s_p[r0 + t] : p_draft[r0 + t];
            used++;
            if (u < pd) {
                cm[nacc++] = proposed;
                if (eos_id >= 0 && proposed == eos_id) { eos = 1; break; }
                continue;
            }
            rej = t;
            break;
        }
        rows[b].n_committed = nacc; rows[b].eos = eos; rows[b].reject_pos = rej; rows[b].n_uniforms = used;
        rows[b].n_bonus_draws = 0; rows[b].n_pads = 0; rows[b].active_next = 0;
        rows[b].rsv = n_rej;                                        // bonus draws before this row if every draw is a single one
        used_total += used;
        if (rej >= 0) n_rej++;
    }
}

template <int DT>
__global__ __launch_bounds__(256) void rs_bonus_kernel(const void *logits, int64_t V, int64_t row_stride, const int64_t *draft, int L,
                                                        const float *row_max, const float *row_sumexp, float temp,
                                                        const float *b_stream, int64_t b_len, const int64_t *b_cursor,
                                                        int64_t *committed, jf_rs_row *rows) {
    __shared__ RsShared sh;
    const int b = blockIdx.x;
    const int rej = rows[b].reject_pos;
    if (rej < 0) return;
    const int64_t r = (int64_t)b * (L - 1) + rej;
    const void *row = (const char *)logits + r * row_stride * (DT == JF_F32 ? 4 : 2);
    int draws = 0;
    const int bonus = rs_bonus_row<DT>(row, V, 1.f / temp, row_max[r], row_sumexp[r], draft[(int64_t)b * L + rej + 1], b_stream, b_len,
                                       *b_cursor + rows[b].rsv, sh, &draws);
    if (threadIdx.x == 0) {
        committed[(int64_t)b * L + rows[b].n_committed] = bonus;
        rows[b].n_bonus_draws = draws;
    }
}

template <int DT>
__global__ __launch_bounds__(256) void rs_finish_kernel(const void *logits, int64_t V, int64_t row_stride, const int64_t *draft,
                                                         int B, int L, const float *row_max, const float *row_sumexp,
                                                         unsigned long long *packed, float temp, int eos_id,
                                                         const int32_t *remaining, int64_t *u_cursor, const float *b_stream,
                                                         int64_t b_len, int64_t *b_cursor, const int64_t *pad_stream,
                                                         int64_t pad_len, int64_t *pad_cursor, int64_t *committed,
                                                         int64_t *next_draft, jf_rs_row *rows) {
    __shared__ RsShared sh;
    __shared__ int s_first_bad;
    __shared__ int64_t s_bc, s_pc;
    const int tid = threadIdx.x;
    const int64_t bc0 = *b_cursor;
    if (tid == 0) {
        int fb = -1;
        for (int b = 0; b < B && fb < 0; ++b)
            if (rows[b].reject_pos >= 0 && rows[b].n_bonus_draws != 1) fb = b;
        s_first_bad = fb;
    }
    __syncthreads();
    // rows after the first one that needed more than one draw sampled at the wrong stream positions: redo them in order
    if (s_first_bad >= 0) {
        int64_t base = bc0 + rows[s_first_bad].rsv + rows[s_first_bad].n_bonus_draws;
        for (int b = s_first_bad + 1; b < B; ++b) {
            const int rej = rows[b].reject_pos;
            if (rej < 0) continue;
            const int64_t r = (int64_t)b * (L - 1) + rej;
            const void *row = (const char *)logits + r * row_stride * (DT == JF_F32 ? 4 : 2);
            int draws = 0;
            const int bonus = rs_bonus_row<DT>(row, V, 1.f / temp, row_max[r], row_sumexp[r], draft[(int64_t)b * L + rej + 1], b_stream,
                                               b_len, base, sh, &draws);
            if (tid == 0) { committed[(int64_t)b * L + rows[b].n_committed] = bonus; rows[b].n_bonus_draws = draws; }
            base += draws;
            __syncthreads();
        }
    }
    __syncthreads();
    // finalize: bonus joins the committed tokens, EOS, next draft (JDN:444-466 / 619-638), stream cursors
    if (tid == 0) {
        int64_t uc = *u_cursor, bc = bc0, pc = *pad_cursor;
        for (int b = 0; b < B; ++b) {
            jf_rs_row &rw = rows[b];
            uc += rw.n_uniforms;
            int n = rw.n_committed;
            if (rw.reject_pos >= 0) {
                bc += rw.n_bonus_draws;
                if (eos_id >= 0 && committed[(int64_t)b * L + n] == eos_id) rw.eos = 1;
                n += 1;
            }
            rw.n_committed = n;
            rw.active_next = (!rw.eos && n < remaining[b]) ? 1 : 0;
            int n_pads = 0;
            if (rw.active_next) {
                const int acc_len = 1 + n;
                int copy_len = 1;
                if (acc_len < L) {
                    const int off = acc_len > 1 ? acc_len - 1 : 1;
                    const int rem = (L - 1) - off;
                    copy_len = rem < L - 1 ? rem : L - 1;
                }
                n_pads = L - 1 - copy_len;
            }
            rw.n_pads = n_pads;
            rw.rsv = (int32_t)(pc - *pad_cursor);                    // this row's offset into the pad stream
            pc += n_pads;
        }
        s_bc = bc; s_pc = pc;
        *u_cursor = uc; *b_cursor = bc;
    }
    __syncthreads();
    const int64_t pc0 = *pad_cursor;
    for (int b = 0; b < B; ++b) {
        const jf_rs_row rw = rows[b];
        if (!rw.active_next) continue;
        const int64_t r0 = (int64_t)b * (L - 1);
        int64_t *nd = next_draft + (int64_t)b * L;
        const int n = rw.n_committed, acc_len = 1 + n;
        int copy_len;
        if (tid == 0) nd[0] = committed[(int64_t)b * L + n - 1];
        if (acc_len < L) {
            const int off = acc_len > 1 ? acc_len - 1 : 1;
            const int rem = (L - 1) - off;
            copy_len = rem < L - 1 ? rem : L - 1;
            for (int i = tid; i < copy_len; i += 256) nd[1 + i] = jfmb::decode_packed(packed[r0 + off + i]);
        } else {
            if (tid == 0) nd[1] = jfmb::decode_packed(packed[r0 + L - 2]);
            copy_len = 1;
        }
        for (int i = tid; i < rw.n_pads; i += 256) nd[1 + copy_len + i] = pad_stream[(pc0 + rw.rsv + i) % pad_len];
    }
    __syncthreads();
    for (int64_t i = tid; i < (int64_t)B * (L - 1); i += 256) packed[i] = 0ull;
    if (tid == 0) {
        *pad_cursor = s_pc;
        for (int b = 0; b < B; ++b) rows[b].rsv = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// On-policy rollout step (JDO = inference_engine/engine/jacobi_decoding_nongreedy_on_policy.py): sequential accept /
// reject of ONE sequence's proposed tokens with a stop-token SET (JDO:270-327), then a fresh sample of every not yet
// accepted position from this forward's distribution (JDO:465-477) — one workgroup per re-drafted row.
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void rs_onpolicy_verify_kernel(const void *logits, int64_t V, int64_t row_stride,
                                                                  const int64_t *proposed, int R, const float *p_draft,
                                                                  const float *row_max, const float *row_sumexp, float temp,
                                                                  const int32_t *stop_ids, int n_stop, const float *u_stream,
                                                                  int64_t u_len, int64_t *u_cursor, const float *m_stream,
                                                                  int64_t m_len, int64_t *m_cursor, int64_t *committed,
                                                                  jf_op_row *out) {
    __shared__ RsShared sh;
    __shared__ int s_n, s_stop, s_rej, s_used;
    const int tid = threadIdx.x;
    const int64_t uc = *u_cursor, mc = *m_cursor;
    auto is_stop = [&](int64_t tok) { for (int k = 0; k < n_stop; ++k) if (tok == (int64_t)stop_ids[k]) return true; return false; };
    if (tid == 0) {
        int n = 0, stop = 0, rej = -1, used = 0;
        for (int t = 0; t < R; ++t) {                                  // JDO:293-320
            const int64_t x = proposed[t];
            const float u = u_stream[(uc + used) % u_len];
            used++;
            if (u < p_draft[t]) {
                committed[n++] = x;
                if (is_stop(x)) { stop = 1; break; }
                continue;
            }
            rej = t;
            break;
        }
        s_n = n; s_stop = stop; s_rej = rej; s_used = used;
    }
    __syncthreads();
    const int rej = s_rej;
    int draws = 0;
    if (rej >= 0) {                                                    // JDO:157-168 (bonus != proposed)
        const void *row = (const char *)logits + (int64_t)rej * row_stride * (DT == JF_F32 ? 4 : 2);
        const int bonus = rs_bonus_row<DT>(row, V, 1.f / temp, row_max[rej], row_sumexp[rej], proposed[rej], m_stream, m_len, mc, sh,
                                           &draws);
        if (tid == 0) {
            committed[s_n] = bonus;
            s_n = s_n + 1;
            if (is_stop(bonus)) s_stop = 1;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int n = s_n;
        const int n_redraft = (!s_stop && n < R) ? R - n : 0;          // JDO:465: not stopped and accepted < gen_len
        const int64_t base = mc + draws;
        out->n_committed = n; out->stop_hit = s_stop; out->reject_pos = rej; out->n_bonus_draws = draws;
        out->n_uniforms = s_used; out->n_redraft = n_redraft;
        out->redraft_base_lo = (int32_t)(base & 0xFFFFFFFFll); out->redraft_base_hi = (int32_t)(base >> 32);
        *u_cursor = uc + s_used;
        *m_cursor = base + n_redraft;
    }
}

// one workgroup per logits row: rows >= n_committed draw one sample each (inverse CDF, float64 running sum in vocabulary
// order, the same arithmetic as the bonus draw); every row's argmax slot is re-zeroed.
template <int DT>
__global__ __launch_bounds__(256) void rs_sample_rows_kernel(const void *logits, int64_t V, int64_t row_stride, int R,
                                                              const float *row_max, const float *row_sumexp, float temp,
                                                              const float *m_stream, int64_t m_len, const jf_op_row *res,
                                                              int64_t *redraft, unsigned long long *packed) {
    __shared__ double s_sum[256], s_pre[256];
    __shared__ double s_total;
    __shared__ int s_pick;
    const int li = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) packed[li] = 0ull;
    const int n = res->n_committed;
    if (res->n_redraft <= 0 || li < n) return;
    const int64_t base = ((int64_t)res->redraft_base_hi << 32) | (int64_t)(uint32_t)res->redraft_base_lo;
    const float inv_t = 1.f / temp;
    const void *row = (const char *)logits + (int64_t)li * row_stride * (DT == JF_F32 ? 4 : 2);
    const float M = row_max[li], Sx = row_sumexp[li];
    const int64_t per = (V + 255) / 256;
    const int64_t lo = (int64_t)tid * per < V ? (int64_t)tid * per : V;
    const int64_t hi = (lo + per < V) ? lo + per : V;
    const double acc = rs_slice_sum<DT>(row, lo, hi, inv_t, M, Sx);
    s_sum[tid] = acc;
    __syncthreads();
    if (tid == 0) {
        double run = 0.0;
        for (int i = 0; i < 256; ++i) { s_pre[i] = run; run += s_sum[i]; }
        s_total = run;
        s_pick = (int)(V - 1);
    }
    __syncthreads();
    const double thr = (double)m_stream[(base + (li - n)) % m_len] * s_total;
    const double pre = s_pre[tid];
    if (hi > lo && thr >= pre && thr < pre + acc) s_pick = (int)rs_slice_pick<DT>(row, lo, hi, inv_t, M, Sx, pre, thr);
    __syncthreads();
    if (tid == 0) redraft[li] = s_pick;
}

extern "C" int jf_rs_onpolicy_step(const void *logits, int dtype, int64_t V, int64_t row_stride, const int64_t *proposed, int R,
                                   const float *p_draft, const float *row_max, const float *row_sumexp, uint64_t *packed,
                                   float temperature, const int32_t *stop_ids, int n_stop, const float *u_stream, int64_t u_len,
                                   int64_t *u_cursor, const float *m_stream, int64_t m_len, int64_t *m_cursor,
                                   int64_t *committed, int64_t *redraft, jf_op_row *row, void *stream) {
    if (R <= 0) return JF_OK;
    if (!logits || !proposed || !p_draft || !row_max || !row_sumexp || !packed || !u_stream || !u_cursor || !m_stream ||
        !m_cursor || !committed || !redraft || !row || (n_stop > 0 && !stop_ids))
        return fail(JF_E_INVALID, "jf_rs_onpolicy_step: null pointer");
    if (u_len <= 0 || m_len <= 0 || n_stop < 0) return fail(JF_E_INVALID, "jf_rs_onpolicy_step: empty random stream");
    const float t = (temperature <= 0.f) ? 1.f : temperature;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == JF_F32) {
        rs_onpolicy_verify_kernel<JF_F32><<<1, 256, 0, s>>>(logits, V, row_stride, proposed, R, p_draft, row_max, row_sumexp, t, stop_ids, n_stop, u_stream, u_len, u_cursor, m_stream, m_len, m_cursor, committed, row);
        rs_sample_rows_kernel<JF_F32><<<R, 256, 0, s>>>(logits, V, row_stride, R, row_max, row_sumexp, t, m_stream, m_len, row, redraft, (unsigned long long *)packed);
    } else if (dtype == JF_BF16) {
        rs_onpolicy_verify_kernel<JF_BF16><<<1, 256, 0, s>>>(logits, V, row_stride, proposed, R, p_draft, row_max, row_sumexp, t, stop_ids, n_stop, u_stream, u_len, u_cursor, m_stream, m_len, m_cursor, committed, row);
        rs_sample_rows_kernel<JF_BF16><<<R, 256, 0, s>>>(logits, V, row_stride, R, row_max, row_sumexp, t, m_stream, m_len, row, redraft, (unsigned long long *)packed);
    } else return fail(JF_E_INVALID, "jf_rs_onpolicy_step: dtype %d", dtype);
    return check_launch("rs_onpolicy kernels");
}

extern "C" int jf_rs_step(const void *logits, int dtype, int64_t V, int64_t row_stride, const int64_t *draft, int B, int L,
                          const float *p_draft, const float *row_max, const float *row_sumexp, uint64_t *packed,
                          float temperature, int32_t eos_id, const int32_t *remaining, const float *u_stream, int64_t u_len,
                          int64_t *u_cursor, const float *bonus_stream, int64_t bonus_len, int64_t *bonus_cursor,
                          const int64_t *pad_stream, int64_t pad_len, int64_t *pad_cursor, int64_t *committed,
                          int64_t *next_draft, jf_rs_row *rows, void *stream) {
    if (B <= 0) return JF_OK;
    if (L < 2) return fail(JF_E_INVALID, "Draft must have at least 2 tokens (seed + 1 speculative)");
    if (!logits || !draft || !p_draft || !row_max || !row_sumexp || !packed || !remaining || !u_stream || !u_cursor ||
        !bonus_stream || !bonus_cursor || !pad_stream || !pad_cursor || !committed || !next_draft || !rows)
        return fail(JF_E_INVALID, "jf_rs_step: null pointer");
    if (u_len <= 0 || bonus_len <= 0 || pad_len <= 0) return fail(JF_E_INVALID, "jf_rs_step: empty random stream");
    const float t = (temperature <= 0.f) ? 1.f : temperature;
    if (dtype != JF_F32 && dtype != JF_BF16) return fail(JF_E_INVALID, "jf_rs_step: dtype %d", dtype);
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *pk = (unsigned long long *)packed;
    rs_accept_kernel<<<1, 256, 0, s>>>(draft, B, L, p_draft, eos_id, u_stream, u_len, u_cursor, committed, rows);
    if (dtype == JF_F32) {
        rs_bonus_kernel<JF_F32><<<B, 256, 0, s>>>(logits, V, row_stride, draft, L, row_max, row_sumexp, t, bonus_stream, bonus_len, bonus_cursor, committed, rows);
        rs_finish_kernel<JF_F32><<<1, 256, 0, s>>>(logits, V, row_stride, draft, B, L, row_max, row_sumexp, pk, t, eos_id, remaining, u_cursor, bonus_stream, bonus_len, bonus_cursor, pad_stream, pad_len, pad_cursor, committed, next_draft, rows);
    } else {
        rs_bonus_kernel<JF_BF16><<<B, 256, 0, s>>>(logits, V, row_stride, draft, L, row_max, row_sumexp, t, bonus_stream, bonus_len, bonus_cursor, committed, rows);
        rs_finish_kernel<JF_BF16><<<1, 256, 0, s>>>(logits, V, row_stride, draft, B, L, row_max, row_sumexp, pk, t, eos_id, remaining, u_cursor, bonus_stream, bonus_len, bonus_cursor, pad_stream, pad_len, pad_cursor, committed, next_draft, rows);
    }
    return check_launch("rs_step kernels");
}

// jf_argmax_dev.h — device bodies of the vocabulary argmax (a2), shared by the stand-alone launches (jf_argmax.hip) and
// the fused verify launch (jf_multiblock.hip: the same items, each storing its result into a slot of its own).
//
// 16 B per lane per load, eight independent loads in flight per lane, one compare chain per 16-byte vector (FastTrack,
// jf_common.h), wave shuffles, one 64-bit atomicMax (stand-alone launches) or store (fused launch) per (row, chunk).  NT selects non-temporal loads: better for streams
// that do not fit the Infinity Cache, 3 % worse below ~60 MB (profiles/argmax_nt_keepv_ab_r01.txt) — chosen per launch.
#ifndef JF_ARGMAX_DEV_H
#define JF_ARGMAX_DEV_H
#include "jf_common.h"

template <bool NT>
__device__ __forceinline__ u32x4 am_load(const u32x4 *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

struct ArgmaxArgs {
    const void *logits;
    int64_t R, V, row_stride;
    unsigned long long *packed;
    int chunks_per_row;
    int64_t chunk_elems;
    const int32_t *out_index;      // nullable: row i -> packed[out_index[i]], negative = skip the row unread
    int32_t reverse;               // item order: 0 rows first to last, 1 last to first, 2 chunk-major (see argmax_wg_item)
    int64_t valid_rows;            // < 0: a negative out_index entry is looked at BEFORE the row is read (skip it unread).
                                   // >= 0: rows [valid_rows, R) are list padding and skipped without a look; the others start
                                   // streaming at once and out_index is only consulted when the result is published (one
                                   // dependent load less in front of every item)
    int32_t slots;                 // 0: chunks of a row meet in packed[orow] (atomicMax).  1 (fused verify launch): every
                                   // (row, chunk) item owns packed[orow * chunks_per_row + chunk] and stores its result there —
                                   // a non-zero slot IS the arrival (every real key is >= 0x007FFFFF), nothing to wait for
};

__device__ __forceinline__ void am_publish(const ArgmaxArgs &a, int64_t orow, int c, unsigned long long m) {
    if (a.slots) __hip_atomic_store(a.packed + orow * a.chunks_per_row + c, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else atomicMax(a.packed + orow, m);
}

// One (row, chunk) item by a 256-thread workgroup sharing the chunk.  Returns the result slot (orow, -1 = skipped row) to
// every thread; thread 0 has published the item (am_publish) when the function returns.
template <int DT, bool VEC, bool NT>
__device__ __forceinline__ int64_t argmax_wg_item(const ArgmaxArgs &a, int64_t item) {
    using E = Elem<DT>;
    constexpr int EPV = E::EPV;
    constexpr int UNROLL = 8;
    // item -> (row, chunk): rows first to last (0), last to first (1), or chunk-major = the same column range of every row
    // before the next range (2: the order a column-tiled producer wrote the logits in)
    const int64_t row = a.reverse == 2 ? item % a.R : (a.reverse ? a.R - 1 - item / a.chunks_per_row : item / a.chunks_per_row);
    // slot of this row's result (jf_argmax_scatter): read up front so its latency hides behind the stream; < 0 = padding row
    if (a.valid_rows >= 0 && row >= a.valid_rows) return -1;
    const int64_t orow = a.out_index ? (int64_t)a.out_index[row] : row;
    if (a.valid_rows < 0 && orow < 0) return -1;
    const int c = (int)(a.reverse == 2 ? item / a.R : item % a.chunks_per_row);
    const int64_t begin = (int64_t)c * a.chunk_elems;
    int64_t end = begin + a.chunk_elems;
    if (end > a.V) end = a.V;
    const typename E::T *p = (const typename E::T *)a.logits + row * a.row_stride;
    const int tid = threadIdx.x;

    uint32_t best = 0u, bidx = 0xFFFFFFFFu;   // every real key is >= 0x007FFFFF > 0
    if constexpr (VEC) {
        FastTrack<DT, true> ft;                                       // the best vector stays in registers: no end-of-item reload
        const uint32_t ebase = (uint32_t)begin;                       // element index of the chunk start (V < 2^31)
        const int nvec = (int)((end - begin) / EPV);                  // full 16-byte vectors in this chunk
        const u32x4 *q = (const u32x4 *)p + (begin / EPV) + tid;
        int k = tid;
        for (; k + (UNROLL - 1) * AM_TPB < nvec; k += UNROLL * AM_TPB, q += UNROLL * AM_TPB) {
            u32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = am_load<NT>(q + u * AM_TPB);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) ft.consume(v[u], ebase + (uint32_t)(k + u * AM_TPB) * EPV);
        }
        for (; k < nvec; k += AM_TPB, q += AM_TPB) {                  // (issuing these < 8 vectors as one round of loads was
            const u32x4 v0 = am_load<NT>(q);                           //  measured: no gain at any shape, +12 VGPRs)
            ft.consume(v0, ebase + (uint32_t)k * EPV);
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
    if (tid == 0 && orow >= 0) {
        uint64_t m = s_part[0];
#pragma unroll
        for (int w = 1; w < AM_TPB / 64; ++w) m = s_part[w] > m ? s_part[w] : m;
        am_publish(a, orow, c, (unsigned long long)m);
    }
    return orow;
}

// Wave-independent variant: every wavefront owns one (row, chunk) item end to end — no LDS, no workgroup barrier; the NaN
// vote is a ballot, the reduction six shuffles, the publish one atomicMax per wavefront (lane 0).  Returns orow or -1.
template <int DT, bool NT>
__device__ __forceinline__ int64_t argmax_wave_item(const ArgmaxArgs &a, int64_t item) {
    using E = Elem<DT>;
    constexpr int EPV = E::EPV;
    constexpr int UNROLL = 8;
    const int lane = threadIdx.x & 63;
    if (item >= a.R * a.chunks_per_row) return -1;
    const int64_t row = a.reverse == 2 ? item % a.R : (a.reverse ? a.R - 1 - item / a.chunks_per_row : item / a.chunks_per_row);
    if (a.valid_rows >= 0 && row >= a.valid_rows) return -1;
    const int64_t orow = a.out_index ? (int64_t)a.out_index[row] : row;
    if (a.valid_rows < 0 && orow < 0) return -1;
    const int c = (int)(a.reverse == 2 ? item / a.R : item % a.chunks_per_row);
    const int64_t begin = (int64_t)c * a.chunk_elems;
    int64_t end = begin + a.chunk_elems;
    if (end > a.V) end = a.V;
    const typename E::T *p = (const typename E::T *)a.logits + row * a.row_stride;

    FastTrack<DT, false> ft;                                          // one wave per SIMD: VALU latency is exposed, keep it lean
    const uint32_t ebase = (uint32_t)begin;
    const int nvec = (int)((end - begin) / EPV);
    const u32x4 *q = (const u32x4 *)p + (begin / EPV) + lane;
    int k = lane;
    for (; k + (UNROLL - 1) * 64 < nvec; k += UNROLL * 64, q += UNROLL * 64) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = am_load<NT>(q + u * 64);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) ft.consume(v[u], ebase + (uint32_t)(k + u * 64) * EPV);
    }
    for (; k < nvec; k += 64, q += 64) {
        const u32x4 v0 = am_load<NT>(q);
        ft.consume(v0, ebase + (uint32_t)k * EPV);
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
    if (lane == 0 && orow >= 0) am_publish(a, orow, c, (unsigned long long)pk);
    return orow;
}

// ---- launch shape (host) ------------------------------------------------------------------------
struct ArgmaxPlan {
    bool vec, wave_mode, nt;
    int reverse;                   // item order: 0 rows first to last, 1 last to first, 2 chunk-major (JF_ARGMAX_REVERSE)
    int64_t chunk, cpr, items, blocks;
};
int argmax_plan(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride, bool fused, ArgmaxPlan *plan,
                int64_t max_cpr = 0);   // jf_argmax.hip; max_cpr > 0: at most that many chunks per row

#endif  // JF_ARGMAX_DEV_H

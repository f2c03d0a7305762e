// jf_argmax.hip — (a2) block-local argmax over the vocabulary, the convergence kernel's HBM stream, and (a3) the
// accepted-prefix scan.  16 B per lane per load, eight loads in flight, one compare chain per 16-byte vector, wave shuffles,
// one 64-bit atomicMax per (row, chunk); helpers (order keys, FastTrack) live in jf_common.h.
#include "jf_common.h"

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
__global__ __launch_bounds__(1024) void accept_lengths_kernel(const int64_t *draft, int draft_rows, const int64_t *greedy,
                                                               int64_t greedy_stride, int B, int L, int32_t *accepted,
                                                               int32_t *best_idx) {
    // one wavefront per row (16 rows in flight), first mismatch = ballot + first set bit per 64 tokens; the best row
    // (largest accepted, then lowest index: torch.argmax, MB:489) is one packed-u64 max over the rows
    __shared__ unsigned long long s_best[16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    unsigned long long best = 0ull;
    for (int b = wave; b < B; b += nw) {
        const int64_t *d = draft + (int64_t)(draft_rows == 1 ? 0 : b) * L;
        const int64_t *g = greedy + (int64_t)b * greedy_stride;
        int m = L - 1;
        for (int i0 = 0; i0 < L - 1; i0 += 64) {
            const int i = i0 + lane;
            const bool mis = (i < L - 1) && (d[i + 1] != g[i]);
            const unsigned long long bal = __ballot(mis);
            if (bal) { m = i0 + __ffsll((long long)bal) - 1; break; }
        }
        const int acc = (L == 0) ? 0 : m + 1;
        if (lane == 0) accepted[b] = acc;
        const unsigned long long k = ((unsigned long long)(uint32_t)(acc + 1) << 32) | (unsigned long long)(~(uint32_t)b);
        best = k > best ? k : best;
    }
    if (lane == 0) s_best[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0 && best_idx) {
        unsigned long long mm = 0ull;
        for (int w = 0; w < nw; ++w) mm = s_best[w] > mm ? s_best[w] : mm;
        *best_idx = mm ? (int32_t)(~(uint32_t)(mm & 0xFFFFFFFFull)) : 0;
    }
}

extern "C" int jf_accept_lengths(const int64_t *draft, int draft_rows, const int64_t *greedy, int64_t greedy_stride, int B,
                                 int L, int32_t *accepted, int32_t *best_idx, void *stream) {
    if (B <= 0) return JF_OK;
    if (!draft || !greedy || !accepted) return fail(JF_E_INVALID, "jf_accept_lengths: null pointer");
    if (draft_rows != 1 && draft_rows != B)
        return fail(JF_E_INVALID, "jf_accept_lengths: draft rows %d do not broadcast against %d", draft_rows, B);
    if (L < 0 || greedy_stride < L - 1) return fail(JF_E_INVALID, "jf_accept_lengths: bad L/stride");
    const int threads = B >= 16 ? 1024 : 64 * (B < 1 ? 1 : B);
    accept_lengths_kernel<<<1, threads, 0, (hipStream_t)stream>>>(draft, draft_rows, greedy, greedy_stride, B, L, accepted, best_idx);
    return check_launch("accept_lengths_kernel");
}


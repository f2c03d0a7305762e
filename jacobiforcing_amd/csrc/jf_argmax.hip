// jf_argmax.hip — (a2) block-local argmax over the vocabulary, the convergence kernel's HBM stream, and (a3) the
// accepted-prefix scan.  Device bodies live in jf_argmax_dev.h (shared with the fused verify launch of jf_multiblock.hip).
#include "jf_argmax_dev.h"

template <int DT, bool VEC, bool NT>
__global__ __launch_bounds__(AM_TPB) void argmax_partial_kernel(ArgmaxArgs a) {
    (void)argmax_wg_item<DT, VEC, NT>(a, blockIdx.x);
}

template <int DT, bool NT>
__global__ __launch_bounds__(AM_TPB) void argmax_wave_kernel(ArgmaxArgs a) {
    (void)argmax_wave_item<DT, NT>(a, (int64_t)blockIdx.x * (AM_TPB / 64) + (threadIdx.x >> 6));
}

__global__ void argmax_decode_kernel(unsigned long long *packed, int64_t R, int64_t *greedy) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) {
        greedy[r] = (int64_t)jfmb::decode_packed(packed[r]);
        packed[r] = 0ull;
    }
}

// Tuning overrides for the sweeps in tools/: read and validated ONCE (no environment scans on the hot path).
struct ArgmaxTune { int64_t chunk, items; int wave, nt, reverse; };
static const ArgmaxTune &argmax_tune() {
    static const ArgmaxTune t = [] {
        auto rd = [](const char *name, int64_t lo, int64_t hi, int64_t dflt) {
            const char *e = getenv(name);
            if (!e || !*e) return dflt;
            const long long v = atoll(e);
            return (v >= lo && v <= hi) ? (int64_t)v : dflt;
        };
        ArgmaxTune r;
        r.chunk = rd("JF_ARGMAX_CHUNK", 1, 1ll << 31, 0);
        r.items = rd("JF_ARGMAX_ITEMS", 1, 1ll << 24, 0);
        r.wave = (int)rd("JF_ARGMAX_WAVE", 0, 1, -1);
        r.nt = (int)rd("JF_ARGMAX_NT", 0, 1, -1);
        r.reverse = (int)rd("JF_ARGMAX_REVERSE", 0, 2, -1);
        return r;
    }();
    return t;
}

// Balanced chunking: cpr chunks per row of equal size (rounded up to `gran` elements).
static int64_t pick_chunk(int64_t gran, int64_t R, int64_t V, int64_t target_items) {
    const int64_t c = argmax_tune().chunk;
    if (c > 0) return ((c + gran - 1) / gran) * gran;
    int64_t per_row = (target_items + R - 1) / R;
    if (per_row < 1) per_row = 1;
    const int64_t max_per_row = (V + gran - 1) / gran;
    if (per_row > max_per_row) per_row = max_per_row;
    int64_t chunk = (V + per_row - 1) / per_row;
    return ((chunk + gran - 1) / gran) * gran;
}

// Launch shape, measured on MI355X (profiles/argmax_microbench_r01*.txt): big problems stream best as ~1 wavefront per SIMD
// (1024 items, each a long contiguous range with 8 x 16 B per lane in flight: 6.8 TB/s fp32 at R>=512); below ~140 MB the
// kernel is launch/ramp bound and 4-wave workgroups sharing a chunk (best vector kept in registers, one item per ~64 KB,
// 256..1024 items) are 5-20 % faster.  Non-temporal loads from 60 MB up, plain loads below.
// Inside the fused verify launch (jf_mb_verify) the 4-wave workgroups win at every size (340 MB in the bench: 69 us against
// 72.5 us per-wavefront, flat from 512 to 3072 items, profiles/verify_knobs_r02.txt): one result per workgroup, not four.
int argmax_plan(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride, bool fused, ArgmaxPlan *pl, int64_t max_cpr) {
    const int esz = dtype == JF_F32 ? 4 : 2;
    const int epv = 16 / esz;
    const ArgmaxTune &tn = argmax_tune();
    pl->vec = (((uintptr_t)logits) % 16 == 0) && ((row_stride * esz) % 16 == 0);
    const int64_t bytes = R * V * esz;
    pl->wave_mode = pl->vec && (tn.wave >= 0 ? tn.wave != 0 : (!fused && bytes >= (140ll << 20)));
    pl->nt = tn.nt >= 0 ? tn.nt != 0 : bytes >= (60ll << 20);
    pl->reverse = tn.reverse > 0 ? tn.reverse : 0;
    if (pl->wave_mode) {
        // one item per wavefront, ~one wavefront per SIMD (256 CUs x 4 SIMDs = 1024 slots).  Split each row into the
        // smallest number of chunks whose makespan ceil(items / 1024) * (V / per_row) is within 10 % of the best
        // split of up to 4 wavefronts per SIMD: e.g. R = 384 -> 5 chunks per row (1920 items, two rounds of V/5) instead
        // of 3 (1152 items: a second round for only 128 of them).
        int64_t items_target = tn.items;
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
        pl->chunk = pick_chunk(64 * epv, R, V, items_target);
        pl->cpr = (V + pl->chunk - 1) / pl->chunk;
        if (max_cpr > 0 && pl->cpr > max_cpr) { pl->chunk = pick_chunk(64 * epv, R, V, R * max_cpr); pl->cpr = (V + pl->chunk - 1) / pl->chunk; }
        pl->items = R * pl->cpr;
        pl->blocks = (pl->items + (AM_TPB / 64) - 1) / (AM_TPB / 64);
    } else {
        int64_t wg_items = bytes >> 16;
        if (wg_items < 256) wg_items = 256;
        if (wg_items > 1024) wg_items = 1024;
        // inside the convergence launch small forwards do better with twice the items (one prompt: 20.8 us at 512 items against
        // 22.8 at 256: every item is one result slot and the stepper polls them in batches of eight)
        if (fused && wg_items < 512) wg_items = 512;
        pl->chunk = pick_chunk((int64_t)AM_TPB * epv, R, V, tn.items > 0 ? tn.items : wg_items);
        pl->cpr = (V + pl->chunk - 1) / pl->chunk;
        if (max_cpr > 0 && pl->cpr > max_cpr) { pl->chunk = pick_chunk((int64_t)AM_TPB * epv, R, V, R * max_cpr); pl->cpr = (V + pl->chunk - 1) / pl->chunk; }
        pl->items = R * pl->cpr;
        pl->blocks = pl->items;
    }
    if (pl->blocks > 0x7FFFFFFFll) return fail(JF_E_CAPACITY, "jf_argmax: grid too large");
    return JF_OK;
}

static int argmax_launch(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride, const int32_t *out_index,
                         uint64_t *packed, void *stream) {
    if (R == 0) return JF_OK;
    if (!logits || !packed) return fail(JF_E_INVALID, "jf_argmax_partial: null pointer");
    if (dtype != JF_F32 && dtype != JF_BF16) return fail(JF_E_INVALID, "jf_argmax_partial: dtype %d", dtype);
    if (R < 0 || V <= 0 || row_stride < V || V > 0x7FFFFFFFll)
        return fail(JF_E_INVALID, "jf_argmax_partial: bad shape R=%lld V=%lld stride=%lld", (long long)R, (long long)V,
                    (long long)row_stride);
    ArgmaxPlan pl;
    const int rc = argmax_plan(logits, dtype, R, V, row_stride, false, &pl);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const ArgmaxArgs a{logits, R, V, row_stride, (unsigned long long *)packed, (int)pl.cpr, pl.chunk, out_index, pl.reverse, -1, 0};
    const dim3 grid((unsigned)pl.blocks), block(AM_TPB);
    if (pl.wave_mode) {
        if (dtype == JF_F32) { if (pl.nt) argmax_wave_kernel<JF_F32, true><<<grid, block, 0, s>>>(a); else argmax_wave_kernel<JF_F32, false><<<grid, block, 0, s>>>(a); }
        else { if (pl.nt) argmax_wave_kernel<JF_BF16, true><<<grid, block, 0, s>>>(a); else argmax_wave_kernel<JF_BF16, false><<<grid, block, 0, s>>>(a); }
        return check_launch("argmax_wave_kernel");
    }
#define JF_LAUNCH(DT, VECF, NTF) argmax_partial_kernel<DT, VECF, NTF><<<grid, block, 0, s>>>(a)
    if (dtype == JF_F32) {
        if (!pl.vec) JF_LAUNCH(JF_F32, false, false);
        else if (pl.nt) JF_LAUNCH(JF_F32, true, true);
        else JF_LAUNCH(JF_F32, true, false);
    } else {
        if (!pl.vec) JF_LAUNCH(JF_BF16, false, false);
        else if (pl.nt) JF_LAUNCH(JF_BF16, true, true);
        else JF_LAUNCH(JF_BF16, true, false);
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


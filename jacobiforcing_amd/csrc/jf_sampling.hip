// jf_sampling.hip — (a19) non-greedy verify: fused softmax-gather + argmax (jf_rs_probs), the batch accept / bonus / finish
// step (jf_rs_step) and the on-policy rollout step (jf_rs_onpolicy_step).
//
// dtype is part of the semantics (JDN = inference_engine/engine/jacobi_decoding_nongreedy.py):
//   JF_F32   probabilities are a float32 softmax of logits * (1/T): p = expf(x/T - M) / S.
//   JF_BF16  the reference never widens the engine's bf16 logits (MR:1382, JDN:64-70), so torch rounds after every op:
//            xs = bf16(float(x) / float(T))  (skipped when T == 1, JDN:68; ATen div_true_kernel: correctly rounded
//            float32 quotient, then one rounding to bf16), p = bf16(expf(xs - M) / S) with M = max xs, S = sum expf(xs - M)
//            in float32.  `u < p`, the inverse-CDF walk of the bonus / re-draft draws and the masked argmax all work on the
//            ROUNDED probabilities, exactly like the reference's `probs` tensor.
#include "jf_common.h"

// ------------------------------------------------------------------------------------------------
// numeric helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_rne(float x) {         // nearest bfloat16 (ties to even), as a float
    uint32_t u = __float_as_uint(x);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return x;          // NaN stays NaN
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    return __uint_as_float(u);
}

// temperature-scaled logit, exactly as torch forms it for this dtype
template <int DT>
__device__ __forceinline__ float rs_scaled(float x, float t, float inv_t, bool unit_t) {
    if constexpr (DT == JF_F32) return x * inv_t;
    else return unit_t ? x : bf16_rne(__fdiv_rn(x, t));
}
// probability of one element given the row statistics
template <int DT>
__device__ __forceinline__ float rs_prob(float xs, float M, float S) {
    const float p = expf(xs - M) / S;
    if constexpr (DT == JF_BF16) return bf16_rne(p);
    else return p;
}

struct RsRow {                 // one logits row + what is needed to turn an element into its probability
    const void *p;
    int64_t V;
    float t, inv_t, M, S;
    bool unit_t, vec;          // vec: 16-byte aligned row -> vector loads
};
template <int DT>
__device__ __forceinline__ float rs_prob_at(const RsRow &r, int64_t i) {
    return rs_prob<DT>(rs_scaled<DT>(load_f<DT>(r.p, i), r.t, r.inv_t, r.unit_t), r.M, r.S);
}
template <int DT>
__device__ __forceinline__ void rs_unpack(const u32x4 v, float (&x)[Elem<DT>::EPV]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if constexpr (DT == JF_F32) x[j] = __uint_as_float(w[j]);
        else { x[2 * j] = __uint_as_float(w[j] << 16); x[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u); }
    }
}
// the EPV raw values starting at element e0 (a multiple of EPV) as one 16-byte vector; slots at or beyond V hold -inf
// (probability 0).  Split from the arithmetic so that callers can put several independent loads in flight first.
template <int DT>
__device__ __forceinline__ u32x4 rs_load_vec(const RsRow &r, int64_t e0) {
    constexpr int EPV = Elem<DT>::EPV;
    if (r.vec && e0 + EPV <= r.V) return *((const u32x4 *)r.p + e0 / EPV);
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if constexpr (DT == JF_F32) {
            w[j] = (e0 + j < r.V) ? ((const uint32_t *)r.p)[e0 + j] : 0xFF800000u;
        } else {
            const uint32_t a = (e0 + 2 * j < r.V) ? ((const uint16_t *)r.p)[e0 + 2 * j] : 0xFF80u;
            const uint32_t b = (e0 + 2 * j + 1 < r.V) ? ((const uint16_t *)r.p)[e0 + 2 * j + 1] : 0xFF80u;
            w[j] = a | (b << 16);
        }
    }
    u32x4 v = {w[0], w[1], w[2], w[3]};
    return v;
}
template <int DT>
__device__ __forceinline__ void rs_probs_from_vec(const RsRow &r, const u32x4 v, float (&p)[Elem<DT>::EPV]) {
    constexpr int EPV = Elem<DT>::EPV;
    float x[EPV];
    rs_unpack<DT>(v, x);
#pragma unroll
    for (int j = 0; j < EPV; ++j) p[j] = rs_prob<DT>(rs_scaled<DT>(x[j], r.t, r.inv_t, r.unit_t), r.M, r.S);
}
// probabilities of the EPV elements starting at element e0 (a multiple of EPV); elements >= V give 0
template <int DT>
__device__ __forceinline__ void rs_probs_of_vec(const RsRow &r, int64_t e0, float (&p)[Elem<DT>::EPV]) {
    rs_probs_from_vec<DT>(r, rs_load_vec<DT>(r, e0), p);
}

// ------------------------------------------------------------------------------------------------
// (a19) stage 1 — fused online-softmax + argmax, logits read once
// ------------------------------------------------------------------------------------------------
// One workgroup per (row, chunk): 16 B per lane per load, eight loads in flight, processed as two rounds of four vectors
// (32 bf16 / 16 fp32 elements per lane per round).  Per round the lane's maximum comes out of the argmax tracker's packed
// integer keys (no float compare per element), the running sum is rescaled once, and every element costs an unpack, one
// fma and one v_exp_f32:  2^(x*cs - m)  with cs = log2(e) [/ T].  (m, s) live in that scaled log2 domain; partials
// (raw chunk max, s) go to the workspace, the argmax to `packed` by atomicMax.
//   SCALE 0: T == 1 (either dtype)           cs = log2 e
//   SCALE 1: JF_F32, T != 1                  cs = log2 e / T  (x * (1/T), today's fp32 behaviour)
//   SCALE 2: JF_BF16, T != 1                 xs = bf16(x * (1/T)): four integer/float ops.  A bf16 x is +-m * 2^e with an
//                                            8-bit m, and both roundings commute with the power of two, so
//                                            bf16(fl32(x / T)) is a function of m alone: the HOST checks the 128
//                                            mantissas for this T (rs_scale_is_exact) and picks this variant only when
//                                            the product reproduces torch's quotient for every one of them ...
//   SCALE 3: JF_BF16, T != 1                 ... otherwise (a bf16 midpoint lies between product and quotient for some
//                                            mantissa, ~0.2 % of temperatures) the correctly rounded quotient per element.
__device__ __forceinline__ float scale_bf16_fast(float x, float inv_t) {
    const uint32_t q = __float_as_uint(x * inv_t);
    return __uint_as_float((q + 0x7FFFu + ((q >> 16) & 1u)) & 0xFFFF0000u);   // RNE (inf stays inf; NaN stays NaN-ish)
}

// two at a time: gfx950's v_cvt_pk_bf16_f32 is the same RNE for every non-NaN fp32 pattern (walked exhaustively,
// tools/experiments/cvt_bf16_exhaustive.hip) — one multiply (v_pk_mul_f32), one conversion and two unpacks for two elements
typedef __bf16 rs_bf16x2 __attribute__((ext_vector_type(2)));
typedef float rs_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void scale_bf16_fast2(float &a, float &b, float inv_t) {
    const rs_f32x2 v = {a * inv_t, b * inv_t};
    const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, rs_bf16x2));
    a = __uint_as_float(h << 16);
    b = __uint_as_float(h & 0xFFFF0000u);
}

template <int DT> __device__ __forceinline__ float key_max_to_float(int32_t k);
template <> __device__ __forceinline__ float key_max_to_float<JF_F32>(int32_t k) {
    return __uint_as_float((uint32_t)k ^ (((uint32_t)(k >> 31)) & 0x7FFFFFFFu));
}
template <> __device__ __forceinline__ float key_max_to_float<JF_BF16>(int32_t k) {
    const uint32_t h = ((uint32_t)k ^ (((uint32_t)(k >> 15)) & 0x7FFFu)) & 0xFFFFu;
    return __uint_as_float(h << 16);
}

template <int DT, int SCALE, int NV, class FT>
__device__ __forceinline__ void rs_round(const u32x4 (&vv)[NV], FT &ft, uint32_t idx0, uint32_t idx_step, float cs, float t,
                                         float inv_t, float &m, float &s) {
    constexpr int EPV = Elem<DT>::EPV;
    constexpr int NE = NV * EPV;
    int32_t kmax = INT32_MIN;
#pragma unroll
    for (int u = 0; u < NV; ++u) { const int32_t k = ft.consume_ret(vv[u], idx0 + u * idx_step); kmax = k > kmax ? k : kmax; }
    float x[NE];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        float xv[EPV];
        rs_unpack<DT>(vv[u], xv);
#pragma unroll
        for (int j = 0; j < EPV; ++j) x[u * EPV + j] = xv[j];
    }
    float xmax = key_max_to_float<DT>(kmax);                 // the round's largest raw value (NaN if the round holds one)
    if constexpr (SCALE == 2) {
#pragma unroll
        for (int j = 0; j < NE; j += 2) scale_bf16_fast2(x[j], x[j + 1], inv_t);
        xmax = scale_bf16_fast(xmax, inv_t);
    } else if constexpr (SCALE == 3) {
#pragma unroll
        for (int j = 0; j < NE; ++j) x[j] = bf16_rne(__fdiv_rn(x[j], t));
        xmax = bf16_rne(__fdiv_rn(xmax, t));
    }
    const float mr = xmax * cs;                              // monotone: the round's largest scaled value
    const float mn = fmaxf(m, mr);
    if (mn == -INFINITY) return;                             // nothing finite yet: keep (m, s) = (-inf, 0), never form inf - inf
    // two elements per v_pk_fma_f32 / v_pk_add_f32 (packed fp32 runs at full rate per element pair): two partial sums
    // (four independent chains: a dependent v_pk_add_f32 per pair would leave the adder waiting on itself)
    constexpr int NACC = NE >= 8 ? 4 : (NE >= 4 ? 2 : 1);
    rs_f32x2 acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = rs_f32x2{0.f, 0.f};
    acc[0].x = (m == -INFINITY) ? 0.f : s * __builtin_amdgcn_exp2f(m - mn);
    const rs_f32x2 cs2 = {cs, cs}, nm2 = {-mn, -mn};
#pragma unroll
    for (int j = 0; j < NE; j += 2) {
        const rs_f32x2 xx = {x[j], x[j + 1]};
        const rs_f32x2 a = __builtin_elementwise_fma(xx, cs2, nm2);
        const rs_f32x2 e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
        acc[(j >> 1) % NACC] += e;
    }
#pragma unroll
    for (int q = NACC >> 1; q > 0; q >>= 1)
#pragma unroll
        for (int r = 0; r < q; ++r) acc[r] += acc[r + q];
    s = acc[0].x + acc[0].y;
    m = mn;
}

// raw fp32 value behind an order key (inverse of order_key for non-NaN values)
__device__ __forceinline__ float key_to_float(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

template <int DT, bool VEC, int SCALE>
__global__ __launch_bounds__(256) void rs_probs_partial_kernel(const void *logits, int64_t R, int64_t V, int64_t row_stride,
                                                                float t, float inv_t, float2 *__restrict__ partial,
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
    const float cs = (SCALE == 1 ? inv_t : 1.f) * 1.44269504088896340736f;
    float m = -INFINITY, s = 0.f;
    uint32_t best = 0u, bidx = 0xFFFFFFFFu;
    int64_t done = begin;
    if constexpr (VEC) {
        const int nvec = (int)((end - begin) / EPV);
        const u32x4 *q = (const u32x4 *)p + (begin / EPV) + tid;
        const uint32_t ebase = (uint32_t)begin;
        // the greedy kernel's vector-granular argmax tracker, trimmed for a VALU-bound loop: the best vector is re-read at
        // the end instead of carried, and negative NaNs are seen by the sum (s turns NaN) instead of a running minimum
        FastTrack<DT, false, false> ft;
        int k = tid;
        for (; k + 7 * 256 < nvec; k += 8 * 256, q += 8 * 256) {
            const u32x4 va[4] = {JF_LOAD(q), JF_LOAD(q + 256), JF_LOAD(q + 512), JF_LOAD(q + 768)};
            const u32x4 vb[4] = {JF_LOAD(q + 1024), JF_LOAD(q + 1280), JF_LOAD(q + 1536), JF_LOAD(q + 1792)};
            rs_round<DT, SCALE, 4>(va, ft, ebase + (uint32_t)k * EPV, 256u * EPV, cs, t, inv_t, m, s);
            rs_round<DT, SCALE, 4>(vb, ft, ebase + (uint32_t)(k + 1024) * EPV, 256u * EPV, cs, t, inv_t, m, s);
        }
        if (k + 3 * 256 < nvec) {                        // a remaining half batch
            const u32x4 va[4] = {JF_LOAD(q), JF_LOAD(q + 256), JF_LOAD(q + 512), JF_LOAD(q + 768)};
            rs_round<DT, SCALE, 4>(va, ft, ebase + (uint32_t)k * EPV, 256u * EPV, cs, t, inv_t, m, s);
            k += 4 * 256; q += 4 * 256;
        }
        for (; k < nvec; k += 256, q += 256) {          // this lane's remaining vectors, one at a time
            const u32x4 vv[1] = {JF_LOAD(q)};
            rs_round<DT, SCALE, 1>(vv, ft, ebase + (uint32_t)k * EPV, 0u, cs, t, inv_t, m, s);
        }
        done = begin + (int64_t)nvec * EPV;
        if (__syncthreads_or((ft.saw_nan() || s != s) ? 1 : 0)) {
            scan_exact<DT>(p, begin, done, tid, best, bidx);               // NaN (or inf - inf) in the chunk: exact key rescan
        } else if (ft.bvec != 0xFFFFFFFFu) {
            best = ft.ukey();
            bidx = ft.resolve(p);
        }
    }
    for (int64_t i = done + tid; i < end; i += 256) {    // unaligned rows / ragged tail (V % EPV)
        const uint32_t kk = load_key<DT>(p, i);
        if (kk > best) { best = kk; bidx = (uint32_t)i; }
        float xv = load_f<DT>(p, i);
        if constexpr (SCALE >= 2) xv = bf16_rne(__fdiv_rn(xv, t));
        xv *= cs;
        if (xv > m) { s = (m == -INFINITY ? 0.f : s * __builtin_amdgcn_exp2f(m - xv)) + 1.f; m = xv; }
        else if (xv != -INFINITY) s += __builtin_amdgcn_exp2f(xv - m);
    }
    // merge (m, s) pairs: six shuffle steps inside the wavefront, then one LDS hop across the four wavefronts
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float m2 = __shfl_xor(m, off, 64), s2 = __shfl_xor(s, off, 64);
        const float M = fmaxf(m, m2);
        s = (M == -INFINITY) ? 0.f : ((m == -INFINITY ? 0.f : s * exp2f(m - M)) + (m2 == -INFINITY ? 0.f : s2 * exp2f(m2 - M)));
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
        for (int i = 0; i < 4; ++i) Ssum += (sm[i] == -INFINITY) ? 0.f : ss[i] * exp2f(sm[i] - M);
        uint64_t mm = sp[0];
        for (int w = 1; w < 4; ++w) mm = sp[w] > mm ? sp[w] : mm;
        // the chunk's RAW maximum (from the argmax key) and its sum relative to M == fl(scaled(raw max) * cs): scaling and
        // the multiply are monotone, so the largest scaled value belongs to the largest raw value
        partial[item] = make_float2(M == -INFINITY ? -INFINITY : key_to_float((uint32_t)(mm >> 32)), Ssum);
        atomicMax(packed + row, (unsigned long long)mm);
    }
}

// Stage 2 — one thread per row: merge the chunk partials, then the gathered probability of the drafted id.
template <int DT>
__global__ __launch_bounds__(256) void rs_probs_finish_kernel(const void *logits, int64_t R, int64_t V, int64_t row_stride,
                                                               const int64_t *draft_next, float t, float inv_t, const float2 *partial,
                                                               int cpr, float *p_draft, float *row_max, float *row_sumexp) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= R) return;
    const bool unit_t = (t == 1.f);
    const float cs = ((DT == JF_F32 && !unit_t) ? inv_t : 1.f) * 1.44269504088896340736f;
    // the same (scale, * cs) composition stage 1 applied to its running maxima
    auto mdom = [&](float raw) { return ((DT == JF_BF16 && !unit_t) ? bf16_rne(__fdiv_rn(raw, t)) : raw) * cs; };
    float Mraw = -INFINITY;
    for (int c = 0; c < cpr; ++c) Mraw = fmaxf(Mraw, partial[row * cpr + c].x);
    const float mM = mdom(Mraw);
    float S = 0.f;
    for (int c = 0; c < cpr; ++c) {
        const float2 ps = partial[row * cpr + c];
        S += (ps.x == -INFINITY) ? 0.f : ps.y * exp2f(mdom(ps.x) - mM);
    }
    const float M = rs_scaled<DT>(Mraw, t, inv_t, unit_t);    // consumers form expf(xs - M): exactly 1 at the maximum
    row_max[row] = M;
    row_sumexp[row] = S;
    const int64_t tok = draft_next[row];
    const void *p = (const char *)logits + row * row_stride * (DT == JF_F32 ? 4 : 2);
    p_draft[row] = (tok >= 0 && tok < V) ? rs_prob<DT>(rs_scaled<DT>(load_f<DT>(p, tok), t, inv_t, unit_t), M, S) : 0.f;
}

struct RsTune { int64_t items; };
static const RsTune &rs_tune() {                            // read once: sweeps in tools/ set it before the first call
    static const RsTune t = [] {
        RsTune r{1024};                                     // resident workgroups per round: 256 CUs x 4 (122 VGPRs per lane)
        const char *e = getenv("JF_RS_ITEMS");
        if (e && *e) { const long long v = atoll(e); if (v >= 1 && v <= (1 << 20)) r.items = v; }
        return r;
    }();
    return t;
}

// Does bf16(x * fl(1/T)) equal torch's bf16(fl32(x / T)) for EVERY normal bf16 x?  Both roundings commute with powers of
// two, so the 128 mantissas decide (host float division is correctly rounded, like ATen's div_true_kernel).
static bool rs_scale_is_exact(float t) {
    const float inv_t = 1.f / t;
    auto rne = [](float v) { uint32_t u; memcpy(&u, &v, 4); u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u; return u; };
    for (int m = 128; m < 256; ++m) {
        const volatile float x = (float)m;
        const volatile float q = x * inv_t, r = x / t;
        if (rne(q) != rne(r)) return false;
    }
    return true;
}

static int64_t rs_chunk(int dtype, int64_t R, int64_t V, int64_t *cpr_out) {
    const int64_t gran = (int64_t)256 * (dtype == JF_F32 ? 4 : 8);         // one vector per lane: equal chunks, balanced items
    // Chunks per row: the workgroups of one launch run in rounds of ~`slots` (4 resident workgroups per CU at this kernel's
    // register count), so pick the split whose makespan  ceil(R * pr / slots) / pr  is smallest — fewest chunks on ties
    // (fewer, longer items win: profiles/rs_probs_microbench_*.txt) — with at least one eight-vector batch per lane.
    const int64_t slots = rs_tune().items;
    int64_t max_pr = V / (8 * gran);
    if (max_pr < 1) max_pr = 1;
    if (max_pr > 64) max_pr = 64;
    int64_t per_row = 1;
    double best = 1e30;
    for (int64_t pr = 1; pr <= max_pr; ++pr) {
        const double ms = (double)((R * pr + slots - 1) / slots) / (double)pr;
        if (ms < best * 0.90) { best = ms; per_row = pr; }     // a finer split must buy at least 10 %: short items are all overhead
    }
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
    if (V <= 0 || V > 0x7FFFFFFFll || row_stride < V) return fail(JF_E_INVALID, "jf_rs_probs: bad shape V=%lld stride=%lld", (long long)V, (long long)row_stride);
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
    const bool unit = (t == 1.f);
#define JF_RS_P(DT, VECF, SC) rs_probs_partial_kernel<DT, VECF, SC><<<grid, block, 0, s>>>(logits, R, V, row_stride, t, inv_t, part, pk, (int)cpr, chunk)
    if (dtype == JF_F32) {
        if (vec) { if (unit) JF_RS_P(JF_F32, true, 0); else JF_RS_P(JF_F32, true, 1); }
        else { if (unit) JF_RS_P(JF_F32, false, 0); else JF_RS_P(JF_F32, false, 1); }
        rs_probs_finish_kernel<JF_F32><<<dim3((unsigned)((R + 255) / 256)), 256, 0, s>>>(logits, R, V, row_stride, draft_next, t, inv_t, part, (int)cpr, p_draft, row_max, row_sumexp);
    } else {
        const bool fast = !unit && rs_scale_is_exact(t);
        if (vec) { if (unit) JF_RS_P(JF_BF16, true, 0); else if (fast) JF_RS_P(JF_BF16, true, 2); else JF_RS_P(JF_BF16, true, 3); }
        else { if (unit) JF_RS_P(JF_BF16, false, 0); else JF_RS_P(JF_BF16, false, 3); }
        rs_probs_finish_kernel<JF_BF16><<<dim3((unsigned)((R + 255) / 256)), 256, 0, s>>>(logits, R, V, row_stride, draft_next, t, inv_t, part, (int)cpr, p_draft, row_max, row_sumexp);
    }
#undef JF_RS_P
    return check_launch("rs_probs kernels");
}

// ------------------------------------------------------------------------------------------------
// Inverse-CDF draws (the injected stand-in for torch.multinomial, JDN:126-153 / JDO:150-168): the smallest index whose
// float64 running sum of probabilities, in vocabulary order, exceeds u * total.
//
// Two levels, both streaming the row with lane-contiguous 16-byte loads:
//   rs_rowsum_kernel   RS_SEG workgroups per selected row; each sums one contiguous vocabulary segment in float64
//                      (per-lane partials, one tree) -> segsum[row][seg].  The row is read ONCE however many draws follow.
//                      The workgroup whose segment holds the row's PROPOSED token also records the mass in front of it
//                      inside the segment and its own probability: with the segment sums that is the token's interval
//                      [c_lo, c_hi) of the CDF, so "this draw returns the proposed token again" (JDN:140-146: draw until
//                      the sample differs, at most 16 times) is a comparison of u * total with two numbers — the serial
//                      part of the reference's draw order costs a few cycles per draw and every row needs ONE real pick.
//   rs_pick (device)   one workgroup per draw: prefix over the RS_SEG segment sums -> the segment the threshold falls
//                      into -> re-read that one segment (V / RS_SEG elements, L2-resident) with a wavefront scan per
//                      2048/1024-element tile -> the crossing lane resolves inside its 8/4 elements.
// ------------------------------------------------------------------------------------------------
constexpr int RS_SEG = 16;
constexpr int RS_TILES = 8;                       // tiles of a segment whose per-lane sums are kept in registers
constexpr int RS_MAX_TRIES = 16;                  // JDN:135 max_tries

__host__ __device__ inline int64_t rs_seg_elems(int64_t V, int epv) {
    const int64_t tile = 256 * (int64_t)epv;
    const int64_t per = (V + RS_SEG - 1) / RS_SEG;
    return ((per + tile - 1) / tile) * tile;
}

constexpr int RS_FLAG_STRIDE = 16;                // 8-byte words between two rows' accept flags (one 128-byte line each)
struct RsWs {                                     // carve-up of the step workspace for `rows` items
    double *segsum;                               // [rows, RS_SEG] float64 mass of each vocabulary segment
    double *lo_part;                              // [rows] mass in front of the avoided token inside its segment
    double *p_avoid;                              // [rows] probability of the avoided token
    int32_t *sel_row;                             // [rows] logits row to sum for item i, -1 = none
    int32_t *avoid;                               // [rows] token a draw must not return (the rejected proposal), -1 = none
    float *pick_u;                                // [rows] the uniform of the draw that counts; < 0: masked argmax instead
    // hand-off words of the one-launch step (rs_step_fused_kernel), all tagged with the call's generation number, so nothing
    // has to be re-zeroed and a late poller can never see a recycled word
    unsigned long long *flag;                     // [rows * RS_FLAG_STRIDE] (gen << 32) | (reject_pos + 2): the accept walk has decided the
                                                  // row; one 128-byte line per row (its 17 pollers must not share lines with other rows)
    unsigned long long *pick;                     // [rows] (gen << 32) | bits of the uniform that counts (chain workgroup -> bonus workgroup)
    uint32_t *segdone;                            // [rows, RS_SEG] gen: this segment's sum is stored
    uint32_t *bonusdone;                          // [rows] gen: the row's bonus token is stored
    uint32_t *acceptdone;                         // [4]   gen: [0] the accept workgroup has written every row record, [1] the chain workgroup every draw count
};
static inline size_t rs_ws_bytes(int64_t rows) {
    const size_t r = (size_t)((rows + 3) / 4 * 4);
    return r * RS_SEG * sizeof(double) + 2 * r * sizeof(double) + 3 * r * sizeof(int32_t) +
           r * (RS_FLAG_STRIDE + 1) * sizeof(unsigned long long) + r * RS_SEG * sizeof(uint32_t) + r * sizeof(uint32_t) + 4 * sizeof(uint32_t);
}
__host__ __device__ inline RsWs rs_ws(void *ws, int64_t rows) {
    const size_t r = (size_t)((rows + 3) / 4 * 4);
    RsWs w;
    w.segsum = (double *)ws;
    w.lo_part = w.segsum + r * RS_SEG;
    w.p_avoid = w.lo_part + r;
    w.sel_row = (int32_t *)(w.p_avoid + r);
    w.avoid = w.sel_row + r;
    w.pick_u = (float *)(w.avoid + r);
    w.flag = (unsigned long long *)(w.pick_u + r);
    w.pick = w.flag + r * RS_FLAG_STRIDE;
    w.segdone = (uint32_t *)(w.pick + r);
    w.bonusdone = w.segdone + r * RS_SEG;
    w.acceptdone = w.bonusdone + r;
    return w;
}
extern "C" size_t jf_rs_step_workspace_bytes(int64_t rows) { return rows > 0 ? rs_ws_bytes(rows) : 0; }

#ifdef JF_EXP_RS_TRACE
// experiment build (tools/microbench_rs_step.py --trace): wall-clock stamps (100 MHz) of the one-launch step, min / max per role
__device__ unsigned long long g_rstrace[32];
__device__ unsigned long long g_rsrow[8 * 128];   // per row: flag stored, first segment sum saw it, its sums ready in the chain, uniform handed out
#define RS_ROWSTAMP(k, b) do { if ((b) < 128) g_rsrow[(k) * 128 + (b)] = (unsigned long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#define RS_STAMP_MIN(k) do { if (threadIdx.x == 0) atomicMin(&g_rstrace[k], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
#define RS_STAMP_MAX(k) do { if (threadIdx.x == 0) atomicMax(&g_rstrace[k], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
extern "C" __attribute__((visibility("default"))) int jf_exp_rs_trace(unsigned long long *out, int reset) {
    if (reset) {
        unsigned long long init[32];
        for (int i = 0; i < 32; ++i) init[i] = (i & 1) ? 0ull : ~0ull;      // even slots take minima, odd slots maxima
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_rstrace), init, sizeof(init));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rstrace), sizeof(unsigned long long) * 32);
}
extern "C" __attribute__((visibility("default"))) int jf_exp_rs_rows(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rsrow), sizeof(unsigned long long) * 8 * 128);
}
#else
#define RS_STAMP_MIN(k) do { } while (0)
#define RS_STAMP_MAX(k) do { } while (0)
#define RS_ROWSTAMP(k, b) do { } while (0)
#endif
#ifdef JF_EXP_RS_TRACE
#define RS_PICKSTAMP(k) do { if (blockIdx.x == 66 && threadIdx.x == 0) g_rstrace[k] = (unsigned long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RS_PICKSTAMP(k) do { } while (0)
#endif

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_incl_scan_f64(double v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

template <int DT>
__device__ __forceinline__ RsRow rs_make_row(const void *logits, int64_t r, int64_t V, int64_t row_stride, float t, float M, float S) {
    RsRow rr;
    const int esz = DT == JF_F32 ? 4 : 2;
    rr.p = (const char *)logits + r * row_stride * esz;
    rr.V = V; rr.t = t; rr.inv_t = 1.f / t; rr.M = M; rr.S = S;
    rr.unit_t = (t == 1.f);
    rr.vec = (((uintptr_t)rr.p) % 16) == 0;
    return rr;
}

__device__ __forceinline__ void st_agent_f64(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent_f64(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One vocabulary segment of one selected row (r = logits row, av = the token its draws must avoid).  SIG: results go out as
// agent-scope atomic stores followed by the segment's generation word (the consumers are other workgroups of the SAME launch).
template <int DT, bool SIG>
__device__ __forceinline__ void rs_rowsum_body(const void *logits, int64_t V, int64_t row_stride, const float *row_max,
                                               const float *row_sumexp, float t, const RsWs &w, int item, int seg, int r,
                                               int64_t av, uint32_t gen) {
    constexpr int EPV = Elem<DT>::EPV;
    const RsRow row = rs_make_row<DT>(logits, r, V, row_stride, t, row_max[r], row_sumexp[r]);
    const int64_t segE = rs_seg_elems(V, EPV);
    const int64_t lo = (int64_t)seg * segE;
    int64_t hi = lo + segE;
    if (hi > V) hi = V;
    const bool mine = av >= lo && av < hi;                   // workgroup-uniform: this segment holds the avoided token
    double acc = 0.0, front = 0.0, pav = 0.0;
    for (int64_t b0 = lo + (int64_t)threadIdx.x * EPV; b0 < hi; b0 += (int64_t)RS_TILES * 256 * EPV) {
        u32x4 v[RS_TILES];                                   // up to RS_TILES independent 16-byte loads in flight per lane
#pragma unroll
        for (int k = 0; k < RS_TILES; ++k) {
            const int64_t e0 = b0 + (int64_t)k * 256 * EPV;
            if (e0 < hi) v[k] = rs_load_vec<DT>(row, e0);
        }
#pragma unroll
        for (int k = 0; k < RS_TILES; ++k) {
            const int64_t e0 = b0 + (int64_t)k * 256 * EPV;
            if (e0 >= hi) continue;
            float p[EPV];
            rs_probs_from_vec<DT>(row, v[k], p);
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < EPV; ++j) a += (double)p[j];
            acc += a;
            if (mine) {
                if (e0 + EPV <= av) front += a;
                else if (e0 <= av) {
#pragma unroll
                    for (int j = 0; j < EPV; ++j) {
                        if (e0 + j < av) front += (double)p[j];
                        if (e0 + j == av) pav = (double)p[j];
                    }
                }
            }
        }
    }
    acc = wave_sum_f64(acc);
    __shared__ double sw[4], sf[4], sp[4];
    if (mine) { front = wave_sum_f64(front); pav = wave_sum_f64(pav); }
    if ((threadIdx.x & 63) == 0) { sw[threadIdx.x >> 6] = acc; sf[threadIdx.x >> 6] = front; sp[threadIdx.x >> 6] = pav; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double sum = (sw[0] + sw[1]) + (sw[2] + sw[3]);
        if constexpr (SIG) {
            st_agent_f64(w.segsum + (int64_t)item * RS_SEG + seg, sum);
            if (mine) { st_agent_f64(w.lo_part + item, (sf[0] + sf[1]) + (sf[2] + sf[3])); st_agent_f64(w.p_avoid + item, (sp[0] + sp[1]) + (sp[2] + sp[3])); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the sums are performed before the word that announces them
            __hip_atomic_store(w.segdone + (int64_t)item * RS_SEG + seg, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            w.segsum[(int64_t)item * RS_SEG + seg] = sum;
            if (mine) { w.lo_part[item] = (sf[0] + sf[1]) + (sf[2] + sf[3]); w.p_avoid[item] = (sp[0] + sp[1]) + (sp[2] + sp[3]); }
        }
    }
}

template <int DT>
__global__ __launch_bounds__(256) void rs_rowsum_kernel(const void *logits, int64_t V, int64_t row_stride, const float *row_max,
                                                         const float *row_sumexp, float t, RsWs w) {
    const int item = blockIdx.x / RS_SEG, seg = blockIdx.x % RS_SEG;
    const int r = w.sel_row[item];
    if (r < 0) return;
    rs_rowsum_body<DT, false>(logits, V, row_stride, row_max, row_sumexp, t, w, item, seg, r, (int64_t)w.avoid[item], 0u);
}

// total mass of a row and the CDF interval [c_lo, c_hi) of its avoided token, from the segment sums.  `total` is formed
// exactly as rs_pick forms it (segments in order), so both compare u * total with the same number.
// AGENT: the sums were stored by other workgroups of the same launch (agent-scope atomics on both sides); av then comes from
// the caller (the accept workgroup's w.avoid may not be visible yet)
template <bool AGENT = false>
__device__ __forceinline__ void rs_interval(const RsWs &w, int item, int64_t V, int epv, double &total, double &c_lo, double &c_hi,
                                            int64_t av_in = -2) {
    double sg[RS_SEG];
#pragma unroll
    for (int s = 0; s < RS_SEG; ++s) sg[s] = AGENT ? ld_agent_f64(w.segsum + (int64_t)item * RS_SEG + s) : w.segsum[(int64_t)item * RS_SEG + s];
    const int64_t av = AGENT ? av_in : (int64_t)w.avoid[item];
    const int sstar = (av >= 0 && av < V) ? (int)(av / rs_seg_elems(V, epv)) : -1;   // an id outside the vocabulary is never drawn
    double run = 0.0, before = 0.0;
#pragma unroll
    for (int s = 0; s < RS_SEG; ++s) {
        if (s == sstar) before = run;
        run += sg[s];
    }
    total = run;
    const double lo_p = sstar >= 0 ? (AGENT ? ld_agent_f64(w.lo_part + item) : w.lo_part[item]) : 0.0;
    const double p_av = sstar >= 0 ? (AGENT ? ld_agent_f64(w.p_avoid + item) : w.p_avoid[item]) : 0.0;
    c_lo = sstar >= 0 ? before + lo_p : 0.0;
    c_hi = sstar >= 0 ? c_lo + p_av : 0.0;
}

// Up to RS_MAX_TRIES draws from stream[(pos + tr) % len] by lanes 0..15 of one wavefront: the first one that does not
// fall into [c_lo, c_hi) counts (JDN:136-146).  Returns the number of stream entries consumed; *u_final = the uniform that
// counts, or -1 when all RS_MAX_TRIES samples were the avoided token (then JDN:147-153's masked argmax decides).
template <class UFn>
__device__ __forceinline__ int rs_count_draws(UFn u_at, double total, double c_lo, double c_hi, int lane, float *u_final) {
    const float u = lane < RS_MAX_TRIES ? u_at(lane) : 0.f;
    const double thr = (double)u * total;
    const bool coll = lane < RS_MAX_TRIES && thr >= c_lo && thr < c_hi;
    const unsigned free_mask = (unsigned)(~__ballot(coll)) & ((1u << RS_MAX_TRIES) - 1u);
    if (free_mask == 0u) { *u_final = -1.f; return RS_MAX_TRIES; }
    const int f = __builtin_ctz(free_mask);
    *u_final = __shfl(u, f, 64);
    return f + 1;
}

struct RsPickShared {
    double seg[RS_SEG];
    double wt[RS_TILES][4];
    double wtx[4];
    unsigned long long best[4];
    int pick;
};

// One inverse-CDF draw by the whole workgroup (uniform control flow).  Returns the index for every thread.
template <int DT>
__device__ int rs_pick(const RsRow &row, const double *segsum /* global, RS_SEG */, float u, RsPickShared &sh) {
    constexpr int EPV = Elem<DT>::EPV;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    RS_PICKSTAMP(16);                                       // 16: pick starts
    __syncthreads();                                        // sh may still be read by a previous draw
    if (tid < RS_SEG) sh.seg[tid] = ld_agent_f64(segsum + tid);   // may have been stored by another workgroup of this launch
    if (tid == 0) sh.pick = 0x7FFFFFFF;
    __syncthreads();
    double total = 0.0;
#pragma unroll
    for (int s = 0; s < RS_SEG; ++s) total += sh.seg[s];
    const double thr = (double)u * total;
    int sstar = -1;
    double excl = 0.0, run = 0.0;
#pragma unroll
    for (int s = 0; s < RS_SEG; ++s) {
        const double nx = run + sh.seg[s];
        if (sstar < 0 && nx > thr) { sstar = s; excl = run; }
        run = nx;
    }
    if (sstar < 0) return (int)(row.V - 1);                 // thr >= total: clamp like min(idx, V - 1)
    const double thr_s = thr - excl;                        // threshold inside the segment
    const int64_t segE = rs_seg_elems(row.V, EPV);
    const int64_t lo = (int64_t)sstar * segE;
    int64_t hi = lo + segE;
    if (hi > row.V) hi = row.V;
    const int ntiles = (int)((hi - lo + 256 * EPV - 1) / (256 * EPV));
    RS_PICKSTAMP(18);                                       // 18: segment known
    // per-lane sums of the first RS_TILES tiles (independent loads), wavefront inclusive scans, wave totals to LDS
    double incl[RS_TILES], lex[RS_TILES];                    // inclusive / exclusive running sums inside the wavefront
    u32x4 tv[RS_TILES];
#pragma unroll
    for (int k = 0; k < RS_TILES; ++k)
        if (k < ntiles) tv[k] = rs_load_vec<DT>(row, lo + ((int64_t)k * 256 + tid) * EPV);   // independent loads first
#pragma unroll
    for (int k = 0; k < RS_TILES; ++k) {
        double a = 0.0;
        if (k < ntiles) {
            float p[EPV];
            rs_probs_from_vec<DT>(row, tv[k], p);
#pragma unroll
            for (int j = 0; j < EPV; ++j) a += (double)p[j];
        }
        incl[k] = wave_incl_scan_f64(a, lane);
        const double up = __shfl_up(incl[k], 1, 64);        // all lanes active here: never shuffle under the crossing test
        lex[k] = lane == 0 ? 0.0 : up;
        if (lane == 63) sh.wt[k][wave] = incl[k];
    }
    RS_PICKSTAMP(20);                                       // 20: tiles loaded, probabilities and wave scans done
    __syncthreads();
    double base = 0.0;                                      // running sum of everything before tile k
    bool found = false;
    int cand = 0x7FFFFFFF;
    auto resolve = [&](int k, double ex) {                  // first element of this lane's vector in tile k whose running sum crosses
        float p[EPV];
        const int64_t e0 = lo + ((int64_t)k * 256 + tid) * EPV;
        rs_probs_of_vec<DT>(row, e0, p);
        double rr = ex;
        int lastpos = -1, hit = -1;
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
            rr += (double)p[j];
            if (p[j] > 0.f) lastpos = j;
            if (hit < 0 && rr > thr_s) hit = j;
        }
        if (hit < 0) hit = lastpos >= 0 ? lastpos : EPV - 1;   // rounding only: the scan said this lane crosses
        int64_t e = e0 + hit;
        if (e >= row.V) e = row.V - 1;
        return (int)e;
    };
#pragma unroll
    for (int k = 0; k < RS_TILES; ++k) {
        if (k < ntiles) {
            double wb = base;
            for (int w = 0; w < wave; ++w) wb += sh.wt[k][w];
            if (!found && wb + incl[k] > thr_s) { cand = resolve(k, wb + lex[k]); found = true; }
            base += (sh.wt[k][0] + sh.wt[k][1]) + (sh.wt[k][2] + sh.wt[k][3]);
        }
    }
    // segments longer than RS_TILES tiles (V > 16 * 8 * 2048 bf16 elements): remaining tiles one at a time
    for (int k = RS_TILES; k < ntiles; ++k) {
        float p[EPV];
        rs_probs_of_vec<DT>(row, lo + ((int64_t)k * 256 + tid) * EPV, p);
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < EPV; ++j) a += (double)p[j];
        const double in = wave_incl_scan_f64(a, lane);
        const double up = __shfl_up(in, 1, 64);
        __syncthreads();
        if (lane == 63) sh.wtx[wave] = in;
        __syncthreads();
        double wb = base;
        for (int w = 0; w < wave; ++w) wb += sh.wtx[w];
        if (!found && wb + in > thr_s) { cand = resolve(k, wb + (lane == 0 ? 0.0 : up)); found = true; }
        base += (sh.wtx[0] + sh.wtx[1]) + (sh.wtx[2] + sh.wtx[3]);
    }
    RS_PICKSTAMP(22);                                       // 22: resolved
    if (found) atomicMin(&sh.pick, cand);
    __syncthreads();
    int pick = sh.pick;
    if (pick == 0x7FFFFFFF) pick = (int)(hi - 1);           // rounding only: the segment sum said it crosses here
    return pick;
}

// argmax of the distribution with `proposed` masked (JDN:147-153 / JDO:164-168): first index of the largest probability —
// for bf16 logits that is a tie among every id whose ROUNDED probability equals the maximum; all mass on it -> keep it.
template <int DT>
__device__ int rs_masked_argmax(const RsRow &row, int64_t proposed, RsPickShared &sh) {
    constexpr int EPV = Elem<DT>::EPV;
    const int tid = threadIdx.x;
    unsigned long long best = 0ull;
    for (int64_t e0 = (int64_t)tid * EPV; e0 < row.V; e0 += 256 * EPV) {
        float p[EPV];
        rs_probs_of_vec<DT>(row, e0, p);
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
            const int64_t i = e0 + j;
            if (i >= row.V || i == proposed || !(p[j] > 0.f)) continue;
            const unsigned long long k = ((unsigned long long)__float_as_uint(p[j]) << 32) | (unsigned long long)(~(uint32_t)i);
            best = k > best ? k : best;
        }
    }
    best = wave_max_u64(best);
    __syncthreads();
    if ((tid & 63) == 0) sh.best[tid >> 6] = best;
    __syncthreads();
    unsigned long long mm = sh.best[0];
    for (int w = 1; w < 4; ++w) mm = sh.best[w] > mm ? sh.best[w] : mm;
    __syncthreads();
    return mm ? jfmb::decode_packed(mm) : (int)proposed;
}

// the draw that counts for one row: the pick at u (u >= 0), or the masked argmax after RS_MAX_TRIES collisions (u < 0).
// The interval test and the walk agree except when u * total lies within float64 rounding of the proposed token's CDF
// boundary; should the walk return the proposed token after all, the masked argmax is the answer (never the proposal).
template <int DT>
__device__ int rs_final_pick(const RsRow &row, const double *segsum, int64_t proposed, float u, RsPickShared &sh) {
    int y = -1;
    if (u >= 0.f) y = rs_pick<DT>(row, segsum, u, sh);
    if (y < 0 || (int64_t)y == proposed) y = rs_masked_argmax<DT>(row, proposed, sh);
    return y;
}

// ------------------------------------------------------------------------------------------------
// Accept/reject of every row of a batch (JDN:581-639).  The reference visits the rows in order and draws torch.rand /
// torch.multinomial / torch.randint as it goes, so the position of every draw in the injected streams depends on the rows
// before it.  Four launches keep that order exact while everything wide runs in parallel:
//   rs_accept_kernel  (1 workgroup)   p_draft / uniforms staged in LDS by 256 threads, then ONE wavefront walks the rows:
//                                     a row's L-1 accept tests are one ballot, so the serial chain is B steps of LDS
//                                     latency, not B*(L-1); results are written out by all threads afterwards
//   rs_rowsum_kernel  (B * RS_SEG)    float64 segment sums of every rejected row + the proposed token's CDF interval
//   rs_bonus_kernel   (B workgroups)  bonus-stream positions in row order — per rejected row one ballot over <= 16 staged
//                                     uniforms against the interval gives its number of draws and the uniform that counts;
//                                     every workgroup walks the rows up to its own (LDS only) — then ONE inverse-CDF walk
//                                     for its row (or the masked argmax).  Batches over 512 rows: rs_chain_kernel does the
//                                     walk once, in its own launch.
//   rs_finish_kernel  (1 workgroup)   EOS, next drafts, pads, cursors by parallel scans; packed re-zeroed
// ------------------------------------------------------------------------------------------------
// strided loop whose loads are issued UNR at a time before the first value is used: the small single-workgroup kernels of
// the step are chains of global round trips, and a plain `for` pays one round trip per iteration
template <int UNR, class T, class LoadFn, class StoreFn>
__device__ __forceinline__ void batched_for(int64_t n, int tid, int nthreads, LoadFn ld, StoreFn st) {
    for (int64_t i0 = tid; i0 < n; i0 += (int64_t)UNR * nthreads) {
        T v[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) { const int64_t i = i0 + (int64_t)k * nthreads; if (i < n) v[k] = ld(i); }
#pragma unroll
        for (int k = 0; k < UNR; ++k) { const int64_t i = i0 + (int64_t)k * nthreads; if (i < n) st(i, v[k]); }
    }
}


constexpr int RS_STAGE = 6144;      // floats of p_draft / uniforms staged in LDS by the accept scan (B * (L-1) <= this, else global)
constexpr int RS_ROWS_LDS = 2048;   // rows whose scan results are kept in LDS (half of it in the chain kernel)

// STAGED: the batch fits the LDS tables (B * (L-1) <= RS_STAGE, B <= RS_ROWS_LDS) — the serial part touches LDS only.
// SIG (one-launch step): the walker announces every row the moment it is decided — flag[b] = (gen << 32) | (reject_pos + 2),
// one self-contained 8-byte agent-scope store — and the workgroup ends with a release + the accept-done word.
template <bool STAGED, bool SIG, int STAGE_N = RS_STAGE, int ROWS_N = RS_ROWS_LDS>
__device__ __forceinline__ void rs_accept_body(const int64_t *draft, int B, int L, const float *p_draft, int eos_id,
                                               const float *u_stream, int64_t u_len, const int64_t *u_cursor,
                                               int64_t *committed, jf_rs_row *rows, const RsWs &w, uint32_t gen) {
    __shared__ float s_p[STAGED ? STAGE_N : 1], s_u[STAGED ? STAGE_N : 1];     // s_p carries "proposed == EOS" in its sign bit (p >= 0)
    __shared__ int s_res[STAGED ? ROWS_N : 1];                                 // nacc | eos << 15 | (rej + 1) << 16 per row
    const int tid = threadIdx.x;
    const int W = L - 1;
    const int n = B * W;
    const int64_t uc0 = *u_cursor;
    auto tok_at = [&](int i) { const int b = i / W; return draft[(int64_t)b * L + (i - b * W) + 1]; };
    auto pack_pe = [&](float p, int64_t tok) {
        return __uint_as_float((__float_as_uint(p) & 0x7FFFFFFFu) | ((eos_id >= 0 && tok == (int64_t)eos_id) ? 0x80000000u : 0u));
    };
    if constexpr (STAGED) {
        const int ul = (int)u_len, ub = (int)(uc0 % u_len);
        // loads first (eight per lane in flight), arithmetic afterwards; at most n uniforms can be used
        batched_for<8, float2>(n, tid, 256, [&](int64_t i) { return make_float2(p_draft[i], u_stream[(ub + (int)i) % ul]); },
                               [&](int64_t i, float2 v) { s_p[i] = v.x; s_u[i] = v.y; });
        if (eos_id >= 0) {
            __syncthreads();
            batched_for<8, int64_t>(n, tid, 256, [&](int64_t i) { return tok_at((int)i); },
                                    [&](int64_t i, int64_t tk) { s_p[i] = pack_pe(s_p[i], tk); });
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int lane = tid;
        int used_total = 0;
        for (int b = 0; b < B; ++b) {                               // JDN:326-348, rows in order
            int nacc = 0, eos = 0, rej = -1, used = 0;
            for (int t0 = 0; t0 < W; t0 += 64) {
                const int tt = t0 + lane;
                bool stop = false, rejb = false;
                if (tt < W) {
                    const int i = b * W + tt;
                    float pe, uu;
                    if constexpr (STAGED) { pe = s_p[i]; uu = s_u[used_total + tt]; }
                    else { pe = pack_pe(p_draft[i], tok_at(i)); uu = u_stream[(uc0 + used_total + tt) % u_len]; }
                    rejb = !(uu < __uint_as_float(__float_as_uint(pe) & 0x7FFFFFFFu));
                    stop = rejb || (__float_as_uint(pe) >> 31);     // rejected, or accepted EOS
                }
                const unsigned long long bal = __ballot(stop);
                if (bal) {
                    const int f = __builtin_ctzll(bal);
                    const bool is_rej = (__ballot(rejb) >> f) & 1ull;
                    if (is_rej) { rej = t0 + f; nacc = t0 + f; } else { eos = 1; nacc = t0 + f + 1; }
                    used = t0 + f + 1;
                    break;
                }
                const int wd = (W - t0) < 64 ? (W - t0) : 64;
                nacc = t0 + wd; used = t0 + wd;
            }
            if (lane == 0) {
                if constexpr (STAGED) s_res[b] = nacc | (eos << 15) | ((rej + 1) << 16);
                else { rows[b].n_committed = nacc; rows[b].eos = eos; rows[b].reject_pos = rej; }
                if constexpr (SIG) {
                    __hip_atomic_store(w.flag + (int64_t)b * RS_FLAG_STRIDE, ((unsigned long long)gen << 32) | (unsigned long long)(rej + 2), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                    RS_ROWSTAMP(0, b);
                }
            }
            used_total += used;
        }
        if constexpr (SIG) RS_STAMP_MAX(15);                        // 15: the accept walk has decided the last row
    }
    __syncthreads();
    __threadfence_block();
    // everything else about a row in parallel: row record, the rejected row's work item, the accepted tokens
    auto res_of = [&](int b, int &nacc, int &eos, int &rej) {
        if constexpr (STAGED) { const int r = s_res[b]; nacc = r & 0x7FFF; eos = (r >> 15) & 1; rej = (r >> 16) - 1; }
        else { nacc = rows[b].n_committed; eos = rows[b].eos; rej = rows[b].reject_pos; }
    };
    if (tid < 64) {                                                 // rejected rows in front of each row: ballot prefix, 64 rows a pass
        int before = 0;
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int b = b0 + tid;
            int nacc = 0, eos = 0, rej = -1;
            if (b < B) res_of(b, nacc, eos, rej);
            const unsigned long long bal = __ballot(b < B && rej >= 0);
            if (b < B) rows[b].rsv = before + __builtin_popcountll(bal & ((1ull << tid) - 1ull));
            before += __builtin_popcountll(bal);
        }
    }
    for (int b = tid; b < B; b += 256) {
        int nacc, eos, rej;
        res_of(b, nacc, eos, rej);
        const int64_t avoid = rej >= 0 ? draft[(int64_t)b * L + rej + 1] : -1;
        jf_rs_row &rw = rows[b];
        rw.n_committed = nacc; rw.eos = eos; rw.reject_pos = rej;
        rw.n_uniforms = rej >= 0 ? rej + 1 : nacc;                  // one uniform per tested position (JDN:329)
        if (!SIG || rej < 0) rw.n_bonus_draws = 0;                  // one-launch step: the chain workgroup owns a rejected row's count
        rw.n_pads = 0; rw.active_next = 0;
        w.sel_row[b] = rej >= 0 ? b * W + rej : -1;
        w.avoid[b] = (int32_t)avoid;
        w.pick_u[b] = -1.f;
    }
    batched_for<8, int64_t>((int64_t)B * W, tid, 256, [&](int64_t idx) { return tok_at((int)idx); },
                            [&](int64_t idx, int64_t v) {
                                const int b = (int)(idx / W), i = (int)(idx - (int64_t)b * W);
                                int nacc, eos, rej;
                                res_of(b, nacc, eos, rej);
                                if (i < nacc) committed[(int64_t)b * L + i] = v;
                            });
    if constexpr (SIG) {                                            // row records + accepted tokens, for the finishing workgroup
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(w.acceptdone, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <bool STAGED>
__global__ __launch_bounds__(256) void rs_accept_kernel(const int64_t *draft, int B, int L, const float *p_draft, int eos_id,
                                                         const float *u_stream, int64_t u_len, const int64_t *u_cursor,
                                                         int64_t *committed, jf_rs_row *rows, RsWs w) {
    rs_accept_body<STAGED, false>(draft, B, L, p_draft, eos_id, u_stream, u_len, u_cursor, committed, rows, w, 0u);
}

// bonus-stream bookkeeping in row order (one workgroup): intervals in parallel, then one wavefront walks the rejected rows
constexpr int RS_CHAIN_STAGE = 4096;
__global__ __launch_bounds__(256) void rs_chain_kernel(int B, int64_t V, int epv, const float *b_stream, int64_t b_len,
                                                        const int64_t *b_cursor, jf_rs_row *rows, RsWs w) {
    constexpr int CAP = RS_ROWS_LDS / 2;
    __shared__ double s_tot[CAP], s_lo[CAP], s_hi[CAP];      // s_tot < 0: the row was not rejected
    __shared__ float s_u[RS_CHAIN_STAGE], s_uf[CAP];
    __shared__ int s_draws[CAP];
    __shared__ int s_nrej;
    const int tid = threadIdx.x, lane = tid & 63;
    const bool in_lds = B <= CAP;
    const int64_t bc0 = *b_cursor;
    if (tid == 0) s_nrej = 0;
    __syncthreads();
    int mine = 0;
    for (int b = tid; b < B; b += 256) {
        const bool rej = rows[b].reject_pos >= 0;
        mine += rej ? 1 : 0;
        if (in_lds) {
            double t_ = -1.0, lo_ = 0.0, hi_ = 0.0;
            if (rej) rs_interval(w, b, V, epv, t_, lo_, hi_);
            s_tot[b] = t_; s_lo[b] = lo_; s_hi[b] = hi_; s_draws[b] = 0; s_uf[b] = -1.f;
        }
    }
    if (mine) atomicAdd(&s_nrej, mine);
    __syncthreads();
    const int nrej = s_nrej;
    if (nrej == 0) return;
    const int win = (RS_MAX_TRIES * nrej < RS_CHAIN_STAGE && b_len < 0x7FFFFFFFll) ? RS_MAX_TRIES * nrej : 0;   // uniforms that can be touched
    if (win) {
        const int bl = (int)b_len, bb = (int)(bc0 % b_len);
        batched_for<8, float>(win, tid, 256, [&](int64_t i) { return b_stream[(bb + (int)i) % bl]; }, [&](int64_t i, float v) { s_u[i] = v; });
    }
    __syncthreads();
    if (tid < 64) {                                          // the serial part: LDS only when the batch fits the tables
        int off = 0;                                         // stream entries consumed by the rows before
        for (int b = 0; b < B; ++b) {
            double t_, lo_, hi_;
            if (in_lds) { t_ = s_tot[b]; if (t_ < 0.0) continue; lo_ = s_lo[b]; hi_ = s_hi[b]; }
            else { if (rows[b].reject_pos < 0) continue; rs_interval(w, b, V, epv, t_, lo_, hi_); }
            float uf;
            const int o = off;
            const int draws = rs_count_draws([&](int tr) { return win ? s_u[o + tr] : b_stream[(bc0 + o + tr) % b_len]; }, t_, lo_, hi_, lane, &uf);
            if (lane == 0) {
                if (in_lds) { s_draws[b] = draws; s_uf[b] = uf; }
                else { rows[b].n_bonus_draws = draws; w.pick_u[b] = uf; }
            }
            off += draws;
        }
    }
    __syncthreads();
    if (in_lds)
        for (int b = tid; b < B; b += 256)
            if (s_tot[b] >= 0.0) { rows[b].n_bonus_draws = s_draws[b]; w.pick_u[b] = s_uf[b]; }
}

// CHAIN: the workgroup first walks the bonus stream itself, over the rejected rows up to and including its own (intervals
// in parallel, then one wavefront, LDS only) — every workgroup repeats the rows before it, which costs less than a separate
// single-workgroup launch between rs_rowsum and this one (batches up to RS_BONUS_CHAIN_ROWS rows with a staged window;
// larger ones run rs_chain_kernel first and come here with CHAIN = false).
constexpr int RS_BONUS_CHAIN_ROWS = 512;
template <int DT, bool CHAIN>
__global__ __launch_bounds__(256) void rs_bonus_kernel(const void *logits, int64_t V, int64_t row_stride, const int64_t *draft, int L,
                                                        const float *row_max, const float *row_sumexp, float temp,
                                                        const float *b_stream, int64_t b_len, const int64_t *b_cursor,
                                                        int64_t *committed, jf_rs_row *rows, RsWs w) {
    __shared__ RsPickShared sh;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int rej = rows[b].reject_pos;
    if (rej < 0) return;
    float u_final;
    if constexpr (CHAIN) {
        __shared__ double s_tot[RS_BONUS_CHAIN_ROWS], s_lo[RS_BONUS_CHAIN_ROWS], s_hi[RS_BONUS_CHAIN_ROWS];   // s_tot < 0: not rejected
        __shared__ float s_u[RS_MAX_TRIES * RS_BONUS_CHAIN_ROWS / 2];
        __shared__ float s_uf;
        __shared__ int s_dr;
        const int64_t bc0 = *b_cursor;
        const int win = RS_MAX_TRIES * (rows[b].rsv + 1);                  // rsv = rejected rows in front of this one (rs_accept)
        const bool staged = win <= RS_MAX_TRIES * RS_BONUS_CHAIN_ROWS / 2 && b_len < 0x7FFFFFFFll;
        for (int i = tid; i <= b; i += 256) {
            double t_ = -1.0, lo_ = 0.0, hi_ = 0.0;
            if (rows[i].reject_pos >= 0) rs_interval(w, i, V, Elem<DT>::EPV, t_, lo_, hi_);
            s_tot[i] = t_; s_lo[i] = lo_; s_hi[i] = hi_;
        }
        if (staged) {
            const int bl = (int)b_len, bb = (int)(bc0 % b_len);
            batched_for<8, float>(win, tid, 256, [&](int64_t i) { return b_stream[(bb + (int)i) % bl]; }, [&](int64_t i, float v) { s_u[i] = v; });
        }
        __syncthreads();
        if (tid < 64) {
            int off = 0, draws = 0;
            float uf = -1.f;
            for (int i = 0; i <= b; ++i) {
                const double t_ = s_tot[i];
                if (t_ < 0.0) continue;
                const int o = off;
                draws = rs_count_draws([&](int tr) { return staged ? s_u[o + tr] : b_stream[(bc0 + o + tr) % b_len]; }, t_, s_lo[i], s_hi[i], tid, &uf);
                off += draws;
            }
            if (tid == 0) { s_uf = uf; s_dr = draws; rows[b].n_bonus_draws = draws; }
        }
        __syncthreads();
        u_final = s_uf;
    } else {
        u_final = w.pick_u[b];
    }
    const int64_t r = (int64_t)b * (L - 1) + rej;
    const RsRow row = rs_make_row<DT>(logits, r, V, row_stride, temp, row_max[r], row_sumexp[r]);
    const int bonus = rs_final_pick<DT>(row, w.segsum + (int64_t)b * RS_SEG, draft[(int64_t)b * L + rej + 1], u_final, sh);
    if (tid == 0) committed[(int64_t)b * L + rows[b].n_committed] = bonus;
}

// exclusive prefix sum over consecutive lanes of one wavefront; *total = sum over all 64 lanes
__device__ __forceinline__ int wave_excl_scan_i32(int v, int lane, int *total) {
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(x, off, 64);
        if (lane >= off) x += o;
    }
    *total = __shfl(x, 63, 64);
    return x - v;
}

__device__ __forceinline__ void rs_finish_body(int B, int L, unsigned long long *packed, int eos_id,
                                               const int32_t *remaining, int64_t *u_cursor, int64_t *b_cursor,
                                               const int64_t *pad_stream, int64_t pad_len, int64_t *pad_cursor,
                                               int64_t *committed, int64_t *next_draft, jf_rs_row *rows) {
    __shared__ int s_total_pads;
    const int tid = threadIdx.x, lane = tid & 63;
    // rows in parallel: bonus joins the committed tokens, EOS, next-draft shape (JDN:444-466 / 619-638)
    for (int b = tid; b < B; b += 256) {
        jf_rs_row &rw = rows[b];
        int n = rw.n_committed;
        if (rw.reject_pos >= 0) {
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
    }
    __threadfence_block();
    __syncthreads();
    RS_STAMP_MAX(23);                                                 // 23: finish: rows done
    // stream cursors and every row's offset into the pad stream: exclusive scans in row order by one wavefront
    if (tid < 64) {
        int uc = 0, bc = 0, pc = 0;
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int b = b0 + lane;
            const int nu = b < B ? rows[b].n_uniforms : 0;
            const int nb = (b < B && rows[b].reject_pos >= 0) ? rows[b].n_bonus_draws : 0;
            const int np = b < B ? rows[b].n_pads : 0;
            int tu, tb, tp;
            (void)wave_excl_scan_i32(nu, lane, &tu);
            (void)wave_excl_scan_i32(nb, lane, &tb);
            const int ep = wave_excl_scan_i32(np, lane, &tp);
            if (b < B) rows[b].rsv = pc + ep;                          // this row's offset into the pad stream
            uc += tu; bc += tb; pc += tp;
        }
        if (lane == 0) { *u_cursor += uc; *b_cursor += bc; s_total_pads = pc; }
    }
    __threadfence_block();
    __syncthreads();
    RS_STAMP_MAX(25);                                                 // 25: finish: scans done
    const int64_t pc0 = *pad_cursor;
    // (this loop takes ~10 of the workgroup's 14 us at 64 rows x 32 tokens; batching its loads eight per thread, one load per
    //  element from a selected address and 32-bit index arithmetic were each measured and changed nothing: profiles/rs_step_r03.txt)
    for (int64_t idx = tid; idx < (int64_t)B * L; idx += 256) {        // every next-draft element independently
        const int b = (int)(idx / L), i = (int)(idx - (int64_t)b * L);
        const jf_rs_row rw = rows[b];
        if (!rw.active_next) continue;
        const int64_t r0 = (int64_t)b * (L - 1);
        const int n = rw.n_committed, acc_len = 1 + n;
        int off = 0, copy_len = 1;
        if (acc_len < L) {
            off = acc_len > 1 ? acc_len - 1 : 1;
            const int rem = (L - 1) - off;
            copy_len = rem < L - 1 ? rem : L - 1;
        } else {
            off = L - 2;
        }
        int64_t v;
        if (i == 0) v = committed[(int64_t)b * L + n - 1];
        else if (i - 1 < copy_len) v = jfmb::decode_packed(packed[r0 + off + (i - 1)]);
        else v = pad_stream[(pc0 + rw.rsv + (i - 1 - copy_len)) % pad_len];
        next_draft[idx] = v;
    }
    __syncthreads();
    RS_STAMP_MAX(27);                                                 // 27: finish: next drafts written
    for (int64_t i = tid; i < (int64_t)B * (L - 1); i += 256) packed[i] = 0ull;
    for (int b = tid; b < B; b += 256) rows[b].rsv = 0;
    if (tid == 0) *pad_cursor = pc0 + s_total_pads;
}

__global__ __launch_bounds__(256) void rs_finish_kernel(int B, int L, unsigned long long *packed, int eos_id,
                                                         const int32_t *remaining, int64_t *u_cursor, int64_t *b_cursor,
                                                         const int64_t *pad_stream, int64_t pad_len, int64_t *pad_cursor,
                                                         int64_t *committed, int64_t *next_draft, jf_rs_row *rows) {
    rs_finish_body(B, L, packed, eos_id, remaining, u_cursor, b_cursor, pad_stream, pad_len, pad_cursor, committed, next_draft, rows);
}

// ------------------------------------------------------------------------------------------------
// The whole step as ONE launch (batches of at most RS_FUSED_ROWS rows whose accept tests fit RS_FUSED_STAGE):
//   block 0                    the accept walk (rs_accept_body): announces every row the moment it is decided
//   block 1                    the chain: rows become CDF intervals as their flags and sums arrive (wavefronts 1-3, a thread per
//                              row), wavefront 0 counts the draws in stream order on LDS and hands every rejected row the
//                              uniform that counts as soon as the rows before it are counted
//   block 2                    waits for the accept workgroup, the chain and every bonus word, then rs_finish_body
//   blocks 3 .. B+2            the bonus draw of row b: ONE inverse-CDF walk (or the masked argmax) for the uniform it was handed
//   blocks B+3 ..              the segment sums of row (blk-B-3)/RS_SEG: wait for that row's flag, sum if it was rejected
// The B + 3 workgroups that wait for others come FIRST, the 16 B short ones last: with the roles in pipeline order the device
// filled up with segment-sum workgroups spinning on the flags of late rows, and the chain / bonus workgroups were not even
// dispatched before those had left (in-kernel stamps: first uniform handed out at 56 us of a launch whose accept walk ends at
// 17 us).  The waiting workgroups number at most RS_FUSED_ROWS + 3 — far fewer than the device keeps resident (at least one
// per CU) — and the segment sums only wait for block 0, so they always find a slot and everything they are waited for by
// comes to pass; all hand-off words carry the call's generation number (nothing to re-zero, no stale reads); payloads cross
// workgroups as agent-scope atomics (a release fence per producer would write back an L2 full of freshly written logits —
// profiles/verify_release_ab_r03.txt).  Replaces four dependent launches (accept 19 + rowsum 21 + bonus 34 + finish 14 us
// at 64 rejected rows, profiles/rs_step_r03.txt).
// ------------------------------------------------------------------------------------------------
constexpr int RS_FUSED_ROWS = 128;      // rows of a one-launch step (<= 192: the chain workgroup gives every row a thread of wavefronts 1-3)
constexpr int RS_FUSED_STAGE = 4096;    // B * (L-1) accept tests staged in LDS by its accept workgroup
struct RsFusedArgs {
    const void *logits; int64_t V, row_stride; const int64_t *draft; int B, L;
    const float *p_draft, *row_max, *row_sumexp; unsigned long long *packed; float t; int eos_id; const int32_t *remaining;
    const float *u_stream; int64_t u_len; int64_t *u_cursor;
    const float *b_stream; int64_t b_len; int64_t *b_cursor;
    const int64_t *pad_stream; int64_t pad_len; int64_t *pad_cursor;
    int64_t *committed, *next_draft; jf_rs_row *rows; RsWs w; uint32_t gen;
};

__device__ __forceinline__ unsigned long long rs_wait_flag(const unsigned long long *word, uint32_t gen) {
    unsigned long long v;
    while ((uint32_t)((v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != gen) __builtin_amdgcn_s_sleep(16);
    return v;
}
__device__ __forceinline__ void rs_wait_word(const uint32_t *word, uint32_t gen) {
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) __builtin_amdgcn_s_sleep(16);
}

template <int DT>
__global__ __launch_bounds__(256) void rs_step_fused_kernel(RsFusedArgs a) {
    const int blk = blockIdx.x, tid = threadIdx.x;
    const int B = a.B, L = a.L, W = a.L - 1;
    const RsWs &w = a.w;
    if (blk == 0) {
        RS_STAMP_MIN(0);                                             // 0: launch start (accept workgroup)
        rs_accept_body<true, true, RS_FUSED_STAGE, RS_FUSED_ROWS>(a.draft, B, L, a.p_draft, a.eos_id, a.u_stream, a.u_len, a.u_cursor,
                                                                  a.committed, a.rows, w, a.gen);
        RS_STAMP_MAX(1);                                             // 1: accept workgroup done (records + accept-done word)
        return;
    }
    if (blk >= B + 3) {                                             // ---- segment sums (the many short workgroups come last)
        const int item = (blk - B - 3) / RS_SEG, seg = (blk - B - 3) % RS_SEG;
        __shared__ int s_rej;
        if (tid == 0) s_rej = (int)(uint32_t)rs_wait_flag(w.flag + (int64_t)item * RS_FLAG_STRIDE, a.gen) - 2;
        __syncthreads();
        const int rej = s_rej;
        if (rej < 0) return;
        if (tid == 0 && seg == 0) RS_ROWSTAMP(1, item);
        rs_rowsum_body<DT, true>(a.logits, a.V, a.row_stride, a.row_max, a.row_sumexp, a.t, w, item, seg, item * W + rej,
                                 a.draft[(int64_t)item * L + rej + 1], a.gen);
        if (tid == 0 && seg == 0) RS_ROWSTAMP(4, item);
        return;
    }
    if (blk == 1) {                                                 // ---- the chain: draws of all rows in stream order
        // Wavefronts 1-3 turn rows into CDF intervals as their flags and segment sums come in (a thread per row); wavefront 0
        // counts the draws in row order on LDS and hands every rejected row its uniform the moment the rows before it are
        // counted — the walk of row i starts ~7 us after row i was decided, not after the last row's sums.
        __shared__ double s_tot[RS_FUSED_ROWS], s_lo[RS_FUSED_ROWS], s_hi[RS_FUSED_ROWS];   // s_tot < 0: not rejected
        __shared__ int s_ready[RS_FUSED_ROWS];
        __shared__ float s_u[RS_MAX_TRIES * RS_FUSED_ROWS];
        const int64_t bc0 = *a.b_cursor;
        const bool staged = a.b_len < 0x7FFFFFFFll;
        for (int i = tid; i < B; i += 256) s_ready[i] = 0;
        if (staged) {                                                // every entry the walk can touch: RS_MAX_TRIES per row
            const int bl = (int)a.b_len, bb = (int)(bc0 % a.b_len);
            batched_for<8, float>(RS_MAX_TRIES * B, tid, 256, [&](int64_t i) { return a.b_stream[(bb + (int)i) % bl]; }, [&](int64_t i, float v) { s_u[i] = v; });
        }
        __syncthreads();
        if (tid >= 64) {
            // a thread per row (B <= RS_FUSED_ROWS <= 192), every lane polling for ITS row without blocking the others of its
            // wavefront: a lane that waited in a loop of its own would hold all 64 rows back until the last of them is in
            const int i = tid - 64;
            bool fin = i >= B;
            int rp = -3;                                             // -3: the row's flag has not been seen yet
            for (;;) {                                               // wave-uniform exit (the ballot below): with a per-lane `while (!fin)` the
                if (!fin) {                                          // compiler sinks a lane's publishing behind the loop, i.e. behind ALL 64 rows
                if (rp == -3) {
                    const unsigned long long v = __hip_atomic_load(w.flag + (int64_t)i * RS_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)(v >> 32) == a.gen) rp = (int)(uint32_t)v - 2;
                }
                if (rp != -3) {
                    bool sums_in = true;
                    if (rp >= 0)
                        for (int sg = 0; sg < RS_SEG; ++sg)
                            sums_in &= __hip_atomic_load(w.segdone + (int64_t)i * RS_SEG + sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.gen;
                    if (sums_in) {
                        double t_ = -1.0, lo_ = 0.0, hi_ = 0.0;
                        if (rp >= 0) rs_interval<true>(w, i, a.V, Elem<DT>::EPV, t_, lo_, hi_, a.draft[(int64_t)i * L + rp + 1]);
                        s_tot[i] = t_; s_lo[i] = lo_; s_hi[i] = hi_;
                        __hip_atomic_store(&s_ready[i], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        RS_ROWSTAMP(2, i);
                        fin = true;
                    }
                }
                }
                if (__ballot(!fin) == 0ull) break;
                __builtin_amdgcn_s_sleep(4);
            }
            return;
        }
        // wavefront 0: one look at the ready words of the next 64 rows (a lane each), then every row of the leading run of ready
        // ones without polling again — a poll per row (an acquire on LDS each) was 0.7 us per row, 46 us for 64 rows
        int off = 0, i = 0;
        while (i < B) {
            const int r = i + tid;
            const bool rdy = r < B && __hip_atomic_load(&s_ready[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
            const unsigned long long nr = ~__ballot(rdy);
            const int run = nr ? __builtin_ctzll(nr) : 64;           // rows i .. i + run - 1 are ready
            if (run == 0) { __builtin_amdgcn_s_sleep(1); continue; }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // their intervals were stored before their ready words
            for (const int end = i + run; i < end; ++i) {
                const double t_ = s_tot[i];
                if (t_ < 0.0) continue;
                float uf;
                const int o = off;
                const int draws = rs_count_draws([&](int tr) { return staged ? s_u[o + tr] : a.b_stream[(bc0 + o + tr) % a.b_len]; }, t_, s_lo[i], s_hi[i], tid, &uf);
                if (tid == 0) {                                      // the count for the finishing workgroup (ordered by chain-done below),
                    __hip_atomic_store(&a.rows[i].n_bonus_draws, draws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(w.pick + i, ((unsigned long long)a.gen << 32) | (unsigned long long)__float_as_uint(uf), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);   // the uniform for the row's bonus workgroup: a self-contained word
                    RS_ROWSTAMP(3, i);
                }
                off += draws;
            }
        }
        RS_STAMP_MAX(7);                                             // 7: last row handed its uniform
        if (tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every count is performed before the word that says so
            __hip_atomic_store(w.acceptdone + 1, a.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if (blk >= 3) {                                                 // ---- bonus draw of row b (3 <= blk < B + 3 here)
        const int b = blk - 3;
        __shared__ RsPickShared sh;
        __shared__ int s_rej;
        __shared__ float s_uf;
        if (tid == 0) {
            const int rp = (int)(uint32_t)rs_wait_flag(w.flag + (int64_t)b * RS_FLAG_STRIDE, a.gen) - 2;
            s_rej = rp;
            if (rp >= 0) s_uf = __uint_as_float((uint32_t)rs_wait_flag(w.pick + b, a.gen));
        }
        __syncthreads();
        const int rej = s_rej;
        if (tid == 0) RS_ROWSTAMP(5, b);
        if (rej >= 0) {
            const int64_t r = (int64_t)b * W + rej;
            if (b == B - 1) RS_STAMP_MAX(17);                        // 17: last row's bonus workgroup has its uniform
            const RsRow row = rs_make_row<DT>(a.logits, r, a.V, a.row_stride, a.t, a.row_max[r], a.row_sumexp[r]);
            const int bonus = rs_final_pick<DT>(row, w.segsum + (int64_t)b * RS_SEG, a.draft[(int64_t)b * L + rej + 1], s_uf, sh);
            if (b == B - 1) RS_STAMP_MAX(19);                        // 19: ... has walked
            // n_committed of a rejected row == reject_pos (rs_accept_body)
            if (tid == 0) __hip_atomic_store((long long *)a.committed + (int64_t)b * L + rej, (long long)bonus, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(w.bonusdone + b, a.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) RS_ROWSTAMP(6, b);
        if (b == B - 1) RS_STAMP_MAX(21);                            // 21: ... has stored its token and its done word
        return;
    }
    // ---- block 2: waits until everything is in, then finishes (JDN:444-466 / 619-638)
    if (tid == 0) { rs_wait_word(w.acceptdone, a.gen); rs_wait_word(w.acceptdone + 1, a.gen); }   // accept records, the chain's draw counts
    for (int i = tid; i < B; i += 256) rs_wait_word(w.bonusdone + i, a.gen);
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    RS_STAMP_MIN(12);                                                // 12: finishing workgroup starts
    rs_finish_body(B, L, a.packed, a.eos_id, a.remaining, a.u_cursor, a.b_cursor, a.pad_stream, a.pad_len, a.pad_cursor, a.committed,
                   a.next_draft, a.rows);
    RS_STAMP_MAX(13);                                                // 13: finished
}

// ------------------------------------------------------------------------------------------------
// On-policy rollout step (JDO = inference_engine/engine/jacobi_decoding_nongreedy_on_policy.py): sequential accept /
// reject of ONE sequence's proposed tokens with a stop-token SET (JDO:270-327), then a fresh sample of every not yet
// accepted position from this forward's distribution (JDO:465-477).
//   rs_op_accept_kernel   one wavefront: the R accept tests are one ballot
//   rs_rowsum_kernel      float64 segment sums of the rejected row (+ its proposed token's CDF interval) and of every
//                         row after it (the re-draft candidates)
//   rs_op_bonus_kernel    one workgroup: draws counted against the interval, ONE walk, stop flag, stream cursors
//   rs_op_redraft_kernel  one workgroup per re-drafted row: one draw each; re-zeroes packed
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool op_is_stop(const int32_t *stop_ids, int n_stop, int64_t tok) {
    for (int k = 0; k < n_stop; ++k) if (tok == (int64_t)stop_ids[k]) return true;
    return false;
}

__global__ __launch_bounds__(64) void rs_op_accept_kernel(const int64_t *proposed, int R, const float *p_draft,
                                                           const int32_t *stop_ids, int n_stop, const float *u_stream,
                                                           int64_t u_len, const int64_t *u_cursor, int64_t *committed,
                                                           jf_op_row *out, RsWs w) {
    const int lane = threadIdx.x;
    const int64_t uc = *u_cursor;
    int n = 0, stop = 0, rej = -1, used = 0;
    for (int t0 = 0; t0 < R; t0 += 64) {                              // JDO:293-320
        const int t = t0 + lane;
        bool st = false, rejb = false;
        if (t < R) {
            const bool acc = u_stream[(uc + t) % u_len] < p_draft[t];
            rejb = !acc;
            st = rejb || op_is_stop(stop_ids, n_stop, proposed[t]);
        }
        const unsigned long long bal = __ballot(st);
        if (bal) {
            const int f = __builtin_ctzll(bal);
            const bool is_rej = (__ballot(rejb) >> f) & 1ull;
            if (is_rej) { rej = t0 + f; n = t0 + f; } else { stop = 1; n = t0 + f + 1; }
            used = t0 + f + 1;
            break;
        }
        const int wd = (R - t0) < 64 ? (R - t0) : 64;
        n = t0 + wd; used = t0 + wd;
    }
    for (int i = lane; i < n; i += 64) committed[i] = proposed[i];
    for (int i = lane; i < R; i += 64) {
        w.sel_row[i] = (rej >= 0 && i >= rej) ? i : -1;
        w.avoid[i] = (i == rej) ? (int32_t)proposed[i] : -1;
    }
    if (lane == 0) {
        out->n_committed = n; out->stop_hit = stop; out->reject_pos = rej; out->n_bonus_draws = 0;
        out->n_uniforms = used; out->n_redraft = 0; out->redraft_base_lo = 0; out->redraft_base_hi = 0;
    }
}

template <int DT>
__global__ __launch_bounds__(256) void rs_op_bonus_kernel(const void *logits, int64_t V, int64_t row_stride, const int64_t *proposed,
                                                           int R, const float *row_max, const float *row_sumexp, float temp,
                                                           const int32_t *stop_ids, int n_stop, int64_t *u_cursor,
                                                           const float *m_stream, int64_t m_len, int64_t *m_cursor,
                                                           int64_t *committed, jf_op_row *out, RsWs w) {
    __shared__ RsPickShared sh;
    __shared__ float s_uf;
    __shared__ int s_draws;
    const int tid = threadIdx.x;
    const int rej = out->reject_pos;
    const int64_t mc = *m_cursor;
    int n = out->n_committed, stop = out->stop_hit, draws = 0;
    if (rej >= 0) {                                                    // JDO:157-168 (bonus != proposed)
        if (tid < 64) {
            double t_, lo_, hi_;
            rs_interval(w, rej, V, Elem<DT>::EPV, t_, lo_, hi_);
            float uf;
            const int d = rs_count_draws([&](int tr) { return m_stream[(mc + tr) % m_len]; }, t_, lo_, hi_, tid, &uf);
            if (tid == 0) { s_uf = uf; s_draws = d; }
        }
        __syncthreads();
        draws = s_draws;
        const RsRow row = rs_make_row<DT>(logits, rej, V, row_stride, temp, row_max[rej], row_sumexp[rej]);
        const int bonus = rs_final_pick<DT>(row, w.segsum + (int64_t)rej * RS_SEG, proposed[rej], s_uf, sh);
        if (tid == 0) committed[n] = bonus;
        n += 1;
        if (op_is_stop(stop_ids, n_stop, bonus)) stop = 1;
    }
    if (tid == 0) {
        const int n_redraft = (!stop && n < R) ? R - n : 0;            // JDO:465: not stopped and accepted < gen_len
        const int64_t base = mc + draws;
        out->n_committed = n; out->stop_hit = stop; out->n_bonus_draws = draws; out->n_redraft = n_redraft;
        out->redraft_base_lo = (int32_t)(base & 0xFFFFFFFFll); out->redraft_base_hi = (int32_t)(base >> 32);
        *u_cursor += out->n_uniforms;
        *m_cursor = base + n_redraft;
    }
}

// one workgroup per logits row: rows >= n_committed draw one sample each; every row's argmax slot is re-zeroed.
template <int DT>
__global__ __launch_bounds__(256) void rs_op_redraft_kernel(const void *logits, int64_t V, int64_t row_stride, int R,
                                                             const float *row_max, const float *row_sumexp, float temp,
                                                             const float *m_stream, int64_t m_len, const jf_op_row *res,
                                                             const double *segsum, int64_t *redraft, unsigned long long *packed) {
    __shared__ RsPickShared sh;
    const int li = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) packed[li] = 0ull;
    const int n = res->n_committed;
    if (res->n_redraft <= 0 || li < n) return;
    const int64_t base = ((int64_t)res->redraft_base_hi << 32) | (int64_t)(uint32_t)res->redraft_base_lo;
    const RsRow row = rs_make_row<DT>(logits, li, V, row_stride, temp, row_max[li], row_sumexp[li]);
    const int y = rs_pick<DT>(row, segsum + (int64_t)li * RS_SEG, m_stream[(base + (li - n)) % m_len], sh);
    if (tid == 0) redraft[li] = y;
}

extern "C" int jf_rs_onpolicy_step(const void *logits, int dtype, int64_t V, int64_t row_stride, const int64_t *proposed, int R,
                                   const float *p_draft, const float *row_max, const float *row_sumexp, uint64_t *packed,
                                   float temperature, const int32_t *stop_ids, int n_stop, const float *u_stream, int64_t u_len,
                                   int64_t *u_cursor, const float *m_stream, int64_t m_len, int64_t *m_cursor,
                                   int64_t *committed, int64_t *redraft, jf_op_row *row, void *workspace,
                                   size_t workspace_bytes, void *stream) {
    if (R <= 0) return JF_OK;
    if (!logits || !proposed || !p_draft || !row_max || !row_sumexp || !packed || !u_stream || !u_cursor || !m_stream ||
        !m_cursor || !committed || !redraft || !row || !workspace || (n_stop > 0 && !stop_ids))
        return fail(JF_E_INVALID, "jf_rs_onpolicy_step: null pointer");
    if (u_len <= 0 || m_len <= 0 || n_stop < 0) return fail(JF_E_INVALID, "jf_rs_onpolicy_step: empty random stream");
    if (workspace_bytes < rs_ws_bytes(R) || ((uintptr_t)workspace % 16) != 0) return fail(JF_E_INVALID, "jf_rs_onpolicy_step: workspace too small or not 16-byte aligned");
    if (dtype != JF_F32 && dtype != JF_BF16) return fail(JF_E_INVALID, "jf_rs_onpolicy_step: dtype %d", dtype);
    if (V <= 0 || V > 0x7FFFFFFFll) return fail(JF_E_INVALID, "jf_rs_onpolicy_step: V=%lld", (long long)V);
    const float t = (temperature <= 0.f) ? 1.f : temperature;
    hipStream_t s = (hipStream_t)stream;
    const RsWs w = rs_ws(workspace, R);
    unsigned long long *pk = (unsigned long long *)packed;
    rs_op_accept_kernel<<<1, 64, 0, s>>>(proposed, R, p_draft, stop_ids, n_stop, u_stream, u_len, u_cursor, committed, row, w);
    if (dtype == JF_F32) {
        rs_rowsum_kernel<JF_F32><<<R * RS_SEG, 256, 0, s>>>(logits, V, row_stride, row_max, row_sumexp, t, w);
        rs_op_bonus_kernel<JF_F32><<<1, 256, 0, s>>>(logits, V, row_stride, proposed, R, row_max, row_sumexp, t, stop_ids, n_stop, u_cursor, m_stream, m_len, m_cursor, committed, row, w);
        rs_op_redraft_kernel<JF_F32><<<R, 256, 0, s>>>(logits, V, row_stride, R, row_max, row_sumexp, t, m_stream, m_len, row, w.segsum, redraft, pk);
    } else {
        rs_rowsum_kernel<JF_BF16><<<R * RS_SEG, 256, 0, s>>>(logits, V, row_stride, row_max, row_sumexp, t, w);
        rs_op_bonus_kernel<JF_BF16><<<1, 256, 0, s>>>(logits, V, row_stride, proposed, R, row_max, row_sumexp, t, stop_ids, n_stop, u_cursor, m_stream, m_len, m_cursor, committed, row, w);
        rs_op_redraft_kernel<JF_BF16><<<R, 256, 0, s>>>(logits, V, row_stride, R, row_max, row_sumexp, t, m_stream, m_len, row, w.segsum, redraft, pk);
    }
    return check_launch("rs_onpolicy kernels");
}

extern "C" int jf_rs_step(const void *logits, int dtype, int64_t V, int64_t row_stride, const int64_t *draft, int B, int L,
                          const float *p_draft, const float *row_max, const float *row_sumexp, uint64_t *packed,
                          float temperature, int32_t eos_id, const int32_t *remaining, const float *u_stream, int64_t u_len,
                          int64_t *u_cursor, const float *bonus_stream, int64_t bonus_len, int64_t *bonus_cursor,
                          const int64_t *pad_stream, int64_t pad_len, int64_t *pad_cursor, int64_t *committed,
                          int64_t *next_draft, jf_rs_row *rows, void *workspace, size_t workspace_bytes, void *stream) {
    if (B <= 0) return JF_OK;
    if (L < 2) return fail(JF_E_INVALID, "Draft must have at least 2 tokens (seed + 1 speculative)");
    if (L > 0x7FFF) return fail(JF_E_INVALID, "jf_rs_step: L=%d too large", L);
    if (!logits || !draft || !p_draft || !row_max || !row_sumexp || !packed || !remaining || !u_stream || !u_cursor ||
        !bonus_stream || !bonus_cursor || !pad_stream || !pad_cursor || !committed || !next_draft || !rows || !workspace)
        return fail(JF_E_INVALID, "jf_rs_step: null pointer");
    if (u_len <= 0 || bonus_len <= 0 || pad_len <= 0) return fail(JF_E_INVALID, "jf_rs_step: empty random stream");
    if (workspace_bytes < rs_ws_bytes(B) || ((uintptr_t)workspace % 16) != 0) return fail(JF_E_INVALID, "jf_rs_step: workspace too small or not 16-byte aligned");
    if (V <= 0 || V > 0x7FFFFFFFll) return fail(JF_E_INVALID, "jf_rs_step: V=%lld", (long long)V);
    const float t = (temperature <= 0.f) ? 1.f : temperature;
    if (dtype != JF_F32 && dtype != JF_BF16) return fail(JF_E_INVALID, "jf_rs_step: dtype %d", dtype);
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *pk = (unsigned long long *)packed;
    const RsWs w = rs_ws(workspace, B);
    static const bool fused_ok = !(getenv("JF_RS_FUSED") && getenv("JF_RS_FUSED")[0] == '0');   // A/B knob, read once
    if (fused_ok && (int64_t)B * (L - 1) <= RS_FUSED_STAGE && B <= RS_FUSED_ROWS && u_len < 0x7FFFFFFFll) {
        static uint32_t g_gen = 0;                                  // generation of the hand-off words (0 never used: a zeroed workspace)
        if (++g_gen == 0) ++g_gen;
        RsFusedArgs a{logits, V, row_stride, draft, B, L, p_draft, row_max, row_sumexp, pk, t, eos_id, remaining, u_stream, u_len, u_cursor,
                      bonus_stream, bonus_len, bonus_cursor, pad_stream, pad_len, pad_cursor, committed, next_draft, rows, w, g_gen};
        const unsigned grid = (unsigned)(1 + B * RS_SEG + 1 + B + 1);
        if (dtype == JF_F32) rs_step_fused_kernel<JF_F32><<<grid, 256, 0, s>>>(a);
        else rs_step_fused_kernel<JF_BF16><<<grid, 256, 0, s>>>(a);
        return check_launch("rs_step_fused_kernel");
    }
    if ((int64_t)B * (L - 1) <= RS_STAGE && B <= RS_ROWS_LDS && u_len < 0x7FFFFFFFll)
        rs_accept_kernel<true><<<1, 256, 0, s>>>(draft, B, L, p_draft, eos_id, u_stream, u_len, u_cursor, committed, rows, w);
    else
        rs_accept_kernel<false><<<1, 256, 0, s>>>(draft, B, L, p_draft, eos_id, u_stream, u_len, u_cursor, committed, rows, w);
    const bool chain_in_bonus = B <= RS_BONUS_CHAIN_ROWS;
#define JF_RS_BONUS(DT, CH) rs_bonus_kernel<DT, CH><<<B, 256, 0, s>>>(logits, V, row_stride, draft, L, row_max, row_sumexp, t, bonus_stream, bonus_len, bonus_cursor, committed, rows, w)
    if (dtype == JF_F32) {
        rs_rowsum_kernel<JF_F32><<<B * RS_SEG, 256, 0, s>>>(logits, V, row_stride, row_max, row_sumexp, t, w);
        if (chain_in_bonus) JF_RS_BONUS(JF_F32, true);
        else { rs_chain_kernel<<<1, 256, 0, s>>>(B, V, 4, bonus_stream, bonus_len, bonus_cursor, rows, w); JF_RS_BONUS(JF_F32, false); }
    } else {
        rs_rowsum_kernel<JF_BF16><<<B * RS_SEG, 256, 0, s>>>(logits, V, row_stride, row_max, row_sumexp, t, w);
        if (chain_in_bonus) JF_RS_BONUS(JF_BF16, true);
        else { rs_chain_kernel<<<1, 256, 0, s>>>(B, V, 8, bonus_stream, bonus_len, bonus_cursor, rows, w); JF_RS_BONUS(JF_BF16, false); }
    }
#undef JF_RS_BONUS
    rs_finish_kernel<<<1, 256, 0, s>>>(B, L, pk, eos_id, remaining, u_cursor, bonus_cursor, pad_stream, pad_len, pad_cursor, committed, next_draft, rows);
    return check_launch("rs_step kernels");
}

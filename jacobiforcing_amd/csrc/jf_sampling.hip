// jf_sampling.hip — (a19) non-greedy verify: fused softmax-gather + argmax (jf_rs_probs), the batch accept / bonus / finish
// step (jf_rs_step) and the on-policy rollout step (jf_rs_onpolicy_step).
//
// dtype is part of the semantics (JDN = inference_engine/engine/jacobi_decoding_nongreedy.py):
//   JF_F32   xs = fl32(x / T), probabilities are the softmax of xs rounded ONCE to float32.
//   JF_BF16  the reference never widens the engine's bf16 logits (MR:1382, JDN:64-70), so torch rounds after every op:
//            xs = bf16(float(x) / float(T))  (skipped when T == 1, JDN:68; ATen div_true_kernel: correctly rounded
//            float32 quotient, then one rounding to bf16), p = the softmax of xs rounded ONCE to bf16.  `u < p`, the
//            inverse-CDF walk of the bonus / re-draft draws and the masked argmax all work on the ROUNDED probabilities,
//            exactly like the reference's `probs` tensor.
//   "The softmax" is the exact quotient exp(xs - M) / sum exp(xs - M) (see "Exact probabilities" below): torch's float32
//   kernel approximates it to an ulp; jf_rs_probs' p_draft / row_sumexp are such float32 approximations too (one pass over
//   the logits), and every DECISION of the steps is made on the exact value.
#include "jf_common.h"

#include <hip/hip_ext.h>
#include <atomic>
#include <mutex>

// ------------------------------------------------------------------------------------------------
// numeric helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_rne(float x) {         // nearest bfloat16 (ties to even), as a float
    uint32_t u = __float_as_uint(x);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return x;          // NaN stays NaN
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    return __uint_as_float(u);
}

__device__ __forceinline__ float scale_bf16_fast(float x, float inv_t);
__device__ __forceinline__ void scale_bf16_fast2(float &a, float &b, float inv_t);
// temperature-scaled logit, exactly as torch forms it for this dtype.  fast (bf16 only): the host has checked that
// bf16(x * fl(1/T)) reproduces torch's bf16(fl32(x / T)) for every bf16 x (rs_scale_is_exact: all but ~0.2 % of temperatures)
template <int DT>
__device__ __forceinline__ float rs_scaled(float x, float t, float inv_t, bool unit_t, bool fast = false) {
    if constexpr (DT == JF_F32) return unit_t ? x : __fdiv_rn(x, t);      // torch: logits / T is a true division (round 4; x * (1/T) before)
    else return unit_t ? x : (fast ? scale_bf16_fast(x, inv_t) : bf16_rne(__fdiv_rn(x, t)));
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
    bool fast;                 // bf16 temperature scaling as a product (see rs_scaled)
    bool prob;                 // the row's statistics are marked (row_sumexp == RS_PROB_ROW, row_max = +inf: jf_rs_filter has been here): its
                               // p_draft is final, and the steps attach the row's filter record (flt) before they read the row again
    const jf_rs_filter_row *flt;   // top-k / top-p record of the row (null: the plain distribution): see rs_filter_apply
};
template <int DT> __device__ __forceinline__ float flt_div(float a, float b) {                     // torch: tensor / tensor in the dtype
    const float q = __fdiv_rn(a, b);
    if constexpr (DT == JF_BF16) return bf16_rne(q); else return q;
}
// the filtered, renormalised probability of an id from its exactly rounded probability p (include/jacobiforcing.h: jf_rs_filter)
template <int DT>
__device__ __forceinline__ float rs_filter_apply(const jf_rs_filter_row &f, float p, int64_t id) {
    float y = p;
    if (f.flags & JF_RS_FILT_TOPK) {
        const uint32_t b = __float_as_uint(p);
        y = (b > f.cut1 || (b == f.cut1 && id <= (int64_t)f.tie1)) ? flt_div<DT>(p, f.s1) : 0.f;
    }
    if (f.flags & JF_RS_FILT_TOPP) {
        const uint32_t b = __float_as_uint(y);
        y = (b > f.cut2 || (b == f.cut2 && id <= (int64_t)f.tie2)) ? flt_div<DT>(y, f.s2) : 0.f;
    }
    return y;
}
constexpr float RS_PROB_ROW = -1.f;     // row_sumexp of a probability row (row_max = +inf)
template <int DT>
__device__ __forceinline__ float rs_prob_at(const RsRow &r, int64_t i) {
    if (r.prob) return fmaxf(load_f<DT>(r.p, i), 0.f);
    return rs_prob<DT>(rs_scaled<DT>(load_f<DT>(r.p, i), r.t, r.inv_t, r.unit_t, r.fast), r.M, r.S);
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
    if (r.prob) {                                              // (slots at or beyond V were loaded as -inf)
#pragma unroll
        for (int j = 0; j < EPV; ++j) p[j] = fmaxf(x[j], 0.f);
        return;
    }
#pragma unroll
    for (int j = 0; j < EPV; ++j) p[j] = rs_prob<DT>(rs_scaled<DT>(x[j], r.t, r.inv_t, r.unit_t, r.fast), r.M, r.S);
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

// Argmax tracker of the softmax stream, in the FLOAT domain and per ROUND (round 5).  The stream unpacks every element to a
// float for its exp anyway, so the round's maximum is one v_maximum3_f32 per two elements (IEEE-754-2019 maximum: a NaN
// anywhere in the round comes out as NaN) and the bookkeeping one compare + three selects per ROUND of up to four vectors —
// the integer-key tracker of the greedy kernel (FastTrack) cost ~15 instructions per vector here and made the bf16 kernel
// VALU-bound (profiles/pmc_rs_probs_r05.txt).  Strict "greater" in stream order keeps the first of equal maxima, and the
// float compare makes -0.0 == +0.0 tie (torch.argmax); the lane re-reads the winning round once at the end (resolve).
__device__ __forceinline__ float rs_max3(float a, float b, float c) {
    return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c);
}
struct RoundTrack {
    float best = __builtin_nanf("");          // NaN: "nothing yet" — the first round always takes it over
    float all = -INFINITY;                     // maximum of every round maximum: NaN from the first NaN on (sticky)
    uint32_t bidx0 = 0xFFFFFFFFu;              // element index of the winning round's first vector
    static constexpr uint32_t bstep_vec = 256u; // vectors between a lane's vectors of one round (the workgroup's width)
    __device__ __forceinline__ void note(float rmax, bool live, uint32_t idx0) {   // live: the lane holds the round's first vector
        const bool upd = live && !(rmax <= best);
        best = upd ? rmax : best;
        bidx0 = upd ? idx0 : bidx0;
        all = __builtin_elementwise_maximum(all, rmax);
    }
    __device__ __forceinline__ bool saw_nan() const { return all != all; }
    __device__ __forceinline__ uint32_t ukey() const { return order_key(__float_as_uint(best)); }
    // first element (in index order) of the winning round that equals best; the round's vectors beyond the chunk's `nvec`
    // whole vectors (a partial last round) are not looked at
    template <int DT>
    __device__ __forceinline__ uint32_t resolve(const void *p, uint32_t ebase, int nvec) const {
        constexpr int EPV = Elem<DT>::EPV;
        constexpr uint32_t bstep = bstep_vec * EPV;
        u32x4 v[4];
        bool in[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            in[u] = (bidx0 - ebase + (uint32_t)u * bstep) / EPV < (uint32_t)nvec;
            v[u] = in[u] ? *((const u32x4 *)p + (bidx0 + (uint32_t)u * bstep) / EPV) : u32x4{0u, 0u, 0u, 0u};
        }
        uint32_t idx = bidx0;
#pragma unroll
        for (int u = 3; u >= 0; --u) {
            if (!in[u]) continue;
            const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int j = EPV - 1; j >= 0; --j) {
                float x;
                if constexpr (DT == JF_F32) x = __uint_as_float(w[j]);
                else x = __uint_as_float((j & 1) ? (w[j >> 1] & 0xFFFF0000u) : (w[j >> 1] << 16));
                if (x == best) idx = bidx0 + (uint32_t)u * bstep + (uint32_t)j;
            }
        }
        return idx;
    }
};

template <int DT, int NV>
__device__ __forceinline__ void rs_unpack_round(const u32x4 (&vv)[NV], float (&x)[NV * Elem<DT>::EPV]) {
    constexpr int EPV = Elem<DT>::EPV;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        float xv[EPV];
        rs_unpack<DT>(vv[u], xv);
#pragma unroll
        for (int j = 0; j < EPV; ++j) x[u * EPV + j] = xv[j];
    }
}
// one round on its unpacked values (x is consumed: scaled in place)
template <int DT, int SCALE, int NE>
__device__ __forceinline__ void rs_round_x(float (&x)[NE], RoundTrack &ft, bool live, uint32_t idx0, float cs, float t, float inv_t,
                                           float &m, float &s) {
    float xmax = rs_max3(x[0], x[1], x[2]);                  // the round's largest raw value (NaN if the round holds one)
#pragma unroll
    for (int j = 3; j + 1 < NE; j += 2) xmax = rs_max3(xmax, x[j], x[j + 1]);
    xmax = __builtin_elementwise_maximum(xmax, x[NE - 1]);
    ft.note(xmax, live, idx0);
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
template <int DT, int SCALE, int NV>
__device__ __forceinline__ void rs_round(const u32x4 (&vv)[NV], RoundTrack &ft, bool live, uint32_t idx0, float cs, float t, float inv_t,
                                         float &m, float &s) {
    float x[NV * Elem<DT>::EPV];
    rs_unpack_round<DT, NV>(vv, x);
    rs_round_x<DT, SCALE, NV * Elem<DT>::EPV>(x, ft, live, idx0, cs, t, inv_t, m, s);
}

// raw fp32 value behind an order key (inverse of order_key for non-NaN values)
__device__ __forceinline__ float key_to_float(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// ---- float64 helpers of the exact probability definition (see "Exact probabilities" below) ----
__device__ const double RS_EXP2_TAB[64] = {   // 2^(j/64), correctly rounded
    1.0, 1.0108892860517005, 1.0218971486541166, 1.0330248790212284, 1.0442737824274138, 1.0556451783605572, 1.0671404006768237,
    1.0787607977571199, 1.0905077326652577, 1.102382583307841, 1.1143867425958924, 1.1265216186082418, 1.1387886347566916,
    1.1511892299529827, 1.1637248587775775, 1.1763969916502812, 1.189207115002721, 1.202156731452703, 1.215247359980469,
    1.22848053610687, 1.241857812073484, 1.255380757024691, 1.2690509571917332, 1.2828700160787783, 1.2968395546510096,
    1.3109612115247644, 1.3252366431597413, 1.339667524053303, 1.3542555469368927, 1.3690024229745905, 1.383909881963832,
    1.3989796725383112, 1.4142135623730951, 1.42961333839197, 1.4451808069770467, 1.460917794180647, 1.4768261459394993,
    1.4929077282912648, 1.5091644275934228, 1.5255981507445384, 1.5422108254079407, 1.559004400237837, 1.5759808451078865,
    1.593142151342267, 1.6104903319492543, 1.6280274218573478, 1.645755478153965, 1.6636765803267364, 1.681792830507429,
    1.7001063537185235, 1.718619298122478, 1.7373338352737062, 1.7562521603732995, 1.7753764925265212, 1.7947090750031072,
    1.8142521755003989, 1.8340080864093424, 1.8539791250833855, 1.8741676341103, 1.8945759815869656, 1.9152065613971474,
    1.9360617934922943, 1.9571441241754002, 1.978456026387951};
constexpr double RS_EXP_CUT = -104.0;             // exp(d) < 2^-150: the quotient (S >= 1) rounds to 0 in bf16 and in float32
__device__ __forceinline__ void rs_load_tab(double *tab) {   // 64-entry table into LDS (callers follow with a barrier)
    if (threadIdx.x < 64) tab[threadIdx.x] = RS_EXP2_TAB[threadIdx.x];
}
// exp(d) for RS_EXP_CUT <= d <= 0, relative error <= 2.3e-16 (checked against 50-digit decimals over 20 000 arguments):
// d = k ln2/64 + r, |r| <= ln2/128, exp(r) by a degree-5 polynomial, 2^(k/64) = 2^(k >> 6) * tab[k & 63].  14 float64 ops.
__device__ __forceinline__ double rs_exp64(double d, const double *tab) {
    // k = rint(d * 64 / ln 2) by the 1.5 * 2^52 shift (the integer lands in the low mantissa bits: no v_rndne / v_cvt), and
    // 2^(k >> 6) by an integer add on the exponent field (the result stays far inside the normal range: no v_ldexp)
    const double t = __builtin_fma(d, 92.33248261689366, 6755399441055744.0);
    const int k = __double2loint(t);
    const double kf = t - 6755399441055744.0;
    const double r = __builtin_fma(-kf, 2.9815858269852933e-12, __builtin_fma(-kf, 0.01083042469326756, d));   // ln2/64 = hi (33 bits) + lo
    double p = 8.333333333333333e-3;
    p = __builtin_fma(p, r, 4.1666666666666664e-2);
    p = __builtin_fma(p, r, 1.6666666666666666e-1);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = p * r;                                                                              // exp(r) - 1
    const double tj = tab[k & 63];
    const double v = __builtin_fma(tj, p, tj);                                              // in [1, 2.02)
    return __hiloint2double(__double2hiint(v) + ((k >> 6) << 20), __double2loint(v));
}
// nearest-even bf16 of a non-negative float64 (subnormals included), as a float: float32 by round-to-odd, then RNE
__device__ __forceinline__ float rs_bf16_of_f64(double q) {
    const float f = (float)q;
    const double back = (double)f;
    uint32_t u = __float_as_uint(f);
    if (back > q) u -= 1u;                          // toward zero
    if (back != q) u |= 1u;                         // sticky
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    return __uint_as_float(u);
}
template <int DT>
__device__ __forceinline__ float rs_round_prob(double q) {
    if constexpr (DT == JF_BF16) return rs_bf16_of_f64(q);
    else return (float)q;
}
// a row statistic pair the exact path can work with (else: NaN / inf rows keep the plain float32 formula, like torch)
__device__ __forceinline__ bool rs_row_is_exact(float M, float S) {
    return (__float_as_uint(M) & 0x7F800000u) != 0x7F800000u && S > 0.f && (__float_as_uint(S) & 0x7F800000u) != 0x7F800000u;
}
// relative error bound of the streaming kernel's float32 row sum against the exact sum relative to M: float32 accumulation
// and v_exp_f32 (2^-15, three times what the sweeps show), plus the scaled maximum's rounding residual and the float32
// x * (1/T) of its exponent arguments, both proportional to |M|
__device__ __forceinline__ double rs_eps_row(float M) {
    return 3.0517578125e-5 + 2.384185791015625e-7 * (1.45 * (double)fabsf(M) + 32.0);
}
// exp(xs - M) in float64 for a scaled logit; 0 for anything below the cut (and for -inf)
__device__ __forceinline__ double rs_e64(float xs, double M, const double *tab) {
    const double d = (double)xs - M;
    const double e = rs_exp64(fmax(d, RS_EXP_CUT), tab);      // branch-free: a select, not 40 divergent branches per lane
    return d >= RS_EXP_CUT ? e : 0.0;
}

// Stage 2 — one thread per row: merge the chunk partials, then the gathered probability of the drafted id.
// p_draft is what the steps' accept tests start from: the float64 exp of the gathered logit over the float32 row sum S, whose
// relative error against the exact sum is below rs_eps_row(M).
//   JF_BF16  the LOWER of the two candidate roundings bf16(p (1 - eps)) <= bf16(p (1 + eps)); when they differ (the float32
//            sum cannot decide the rounding: ~1 % of the rows) the SIGN BIT is set — the upper candidate is the next bf16.
//   JF_F32   the centre value fl32(p); the steps test against [p (1 - eps), p (1 + eps)].
// Rows without finite statistics (NaN / +inf logits) keep the plain float32 formula (NaN where torch's softmax is NaN).
template <int DT>
__device__ __forceinline__ void rs_finish_row(const void *logits, int64_t row, int64_t V, int64_t row_stride, int64_t tok, bool have_x, float x_tok,
                                              float t, float inv_t, const float2 *partial, int cpr, float *p_draft, float *row_max,
                                              float *row_sumexp, const double *s_tab) {
    const bool unit_t = (t == 1.f);
    const float cs = ((DT == JF_F32 && !unit_t) ? inv_t : 1.f) * 1.44269504088896340736f;
    // the same (scale, * cs) composition stage 1 applied to its running maxima
    auto mdom = [&](float raw) { return ((DT == JF_BF16 && !unit_t) ? bf16_rne(__fdiv_rn(raw, t)) : raw) * cs; };
    float Mraw = -INFINITY;
    for (int c = 0; c < cpr; ++c) Mraw = fmaxf(Mraw, partial[c].x);
    const float mM = mdom(Mraw);
    float S = 0.f;
    for (int c = 0; c < cpr; ++c) {
        const float2 ps = partial[c];
        S += (ps.x == -INFINITY) ? 0.f : ps.y * exp2f(mdom(ps.x) - mM);
    }
    const float M = rs_scaled<DT>(Mraw, t, inv_t, unit_t);    // consumers form exp(xs - M): exactly 1 at the maximum
    row_max[row] = M;
    row_sumexp[row] = S;
    const void *p = (const char *)logits + row * row_stride * (DT == JF_F32 ? 4 : 2);
    float pd = 0.f;
    if (tok >= 0 && tok < V) {
        const float xs = rs_scaled<DT>(have_x ? x_tok : load_f<DT>(p, tok), t, inv_t, unit_t);
        if (!rs_row_is_exact(M, S)) pd = rs_prob<DT>(xs, M, S);
        else {
            const double ph = rs_e64(xs, (double)M, s_tab) / (double)S;
            if constexpr (DT == JF_F32) pd = (float)ph;
            else {
                const double eps = rs_eps_row(M);
                const float lo = rs_bf16_of_f64(ph * (1.0 - eps)), hi = rs_bf16_of_f64(ph * (1.0 + eps));
                pd = lo != hi ? -lo : lo;                         // (lo > 0 whenever the candidates differ)
            }
        }
    }
    p_draft[row] = pd;
}
template <int DT>
__global__ __launch_bounds__(256) void rs_probs_finish_kernel(const void *logits, int64_t R, int64_t V, int64_t row_stride,
                                                               const int64_t *draft_next, float t, float inv_t, const float2 *partial,
                                                               int cpr, float *p_draft, float *row_max, float *row_sumexp) {
    __shared__ double s_tab[64];
    rs_load_tab(s_tab);
    __syncthreads();
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= R) return;
    rs_finish_row<DT>(logits, row, V, row_stride, draft_next[row], false, 0.f, t, inv_t, partial + row * cpr, cpr, p_draft, row_max, row_sumexp, s_tab);
}

// `fin` (non-null only when a row is ONE chunk, cpr == 1): the workgroup finishes its own row — stage 2 below — instead of a
// second launch (round 5: the second launch and the gap in front of it were ~6 us of a 115 us call at 64 x 31 rows)
struct RsFinishArgs { const int64_t *draft_next; float *p_draft, *row_max, *row_sumexp; };
#ifndef JF_RS_WAVES
#define JF_RS_WAVES 1                      // no occupancy bound: forcing 64 VGPRs (8 waves per SIMD) spills inside the loop and loses 5-20 %
#endif
#ifndef JF_RS_PIPE
#define JF_RS_PIPE 1
#endif
template <int DT, bool VEC, int SCALE>
__global__ __launch_bounds__(256, JF_RS_WAVES) void rs_probs_partial_kernel(const void *logits, int64_t R, int64_t V, int64_t row_stride,
                                                                   float t, float inv_t, float2 *__restrict__ partial,
                                                                   unsigned long long *packed, int cpr, int64_t chunk_elems, RsFinishArgs fin) {
    using E = Elem<DT>;
    constexpr int EPV = E::EPV;
    const int64_t item = blockIdx.x;
    const int64_t row = item / cpr;
    const int c = (int)(item - row * cpr);
    __shared__ double s_tab[64];
    const bool fuse = fin.draft_next != nullptr;
    if (fuse) rs_load_tab(s_tab);                            // (the barriers of the reductions below come before its use)
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
        const uint32_t ebase = (uint32_t)begin;
        // buffer loads: the chunk is a raw buffer (uniform descriptor), the lane offset one constant VGPR and the position in
        // the chunk the instruction's scalar offset — no VALU address arithmetic (per-lane 64-bit pointers cost two VALU adds
        // per load), and every trip count below is wave-uniform (scalar branches, no exec-mask loops)
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((const u32x4 *)p + (begin / EPV)), 0, nvec * 16, 0x00020000);
        const uint32_t voff = (uint32_t)tid * 16u;
#define RS_LD(kb_, u_) __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, ((kb_) + (u_) * 256) * 16, 2 /* nt */))
        RoundTrack ft;
        const int nfull = nvec / (8 * 256);               // iterations in which every lane has all eight vectors
        int kb = 0;                                       // vectors of the chunk in front of this iteration
        const u32x4 ninf = Elem<DT>::EPV == 4 ? u32x4{0xFF800000u, 0xFF800000u, 0xFF800000u, 0xFF800000u}
                                               : u32x4{0xFF80FF80u, 0xFF80FF80u, 0xFF80FF80u, 0xFF80FF80u};
        if constexpr (DT == JF_BF16 && JF_RS_PIPE) {
            // bf16: the stream is half compute (27 VALU cycles per element against ~2.3 per byte of load), so a wavefront that
            // only asks for its next eight vectors after finishing the current ones has nothing in flight for two thirds of an
            // iteration.  Unpacking frees a round's 16 raw registers: the NEXT iteration's loads of that round are issued right
            // there, in front of the round's arithmetic — eight vectors per lane in flight at all times.  The loads are
            // unconditional (a branch around them would make the wait counters conservative): beyond the chunk a raw buffer
            // load returns zeros without touching memory, and what the last iteration prefetches IS the chunk's ragged rest.
            u32x4 va[4] = {RS_LD(0, 0), RS_LD(0, 1), RS_LD(0, 2), RS_LD(0, 3)};
            u32x4 vb[4] = {RS_LD(0, 4), RS_LD(0, 5), RS_LD(0, 6), RS_LD(0, 7)};
            for (int it = 0; it < nfull; ++it, kb += 8 * 256) {
                float xa[4 * EPV];
                rs_unpack_round<DT, 4>(va, xa);
                __builtin_amdgcn_sched_barrier(0);
                va[0] = RS_LD(kb + 2048, 0); va[1] = RS_LD(kb + 2048, 1); va[2] = RS_LD(kb + 2048, 2); va[3] = RS_LD(kb + 2048, 3);
                __builtin_amdgcn_sched_barrier(0);
                rs_round_x<DT, SCALE, 4 * EPV>(xa, ft, true, ebase + (uint32_t)(kb + tid) * EPV, cs, t, inv_t, m, s);
                __builtin_amdgcn_sched_barrier(0);
                float xb[4 * EPV];
                rs_unpack_round<DT, 4>(vb, xb);
                __builtin_amdgcn_sched_barrier(0);
                vb[0] = RS_LD(kb + 2048, 4); vb[1] = RS_LD(kb + 2048, 5); vb[2] = RS_LD(kb + 2048, 6); vb[3] = RS_LD(kb + 2048, 7);
                __builtin_amdgcn_sched_barrier(0);
                rs_round_x<DT, SCALE, 4 * EPV>(xb, ft, true, ebase + (uint32_t)(kb + tid + 1024) * EPV, cs, t, inv_t, m, s);
            }
            // the rest (< 2048 vectors) is in the registers already: a lane's missing vectors become -inf (no mass, never the
            // maximum), a lane without any vector of a round does not touch the tracker
            if (kb < nvec) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    va[u] = kb + tid + u * 256 < nvec ? va[u] : ninf;
                    vb[u] = kb + tid + (u + 4) * 256 < nvec ? vb[u] : ninf;
                }
                rs_round<DT, SCALE, 4>(va, ft, kb + tid < nvec, ebase + (uint32_t)(kb + tid) * EPV, cs, t, inv_t, m, s);
                if (kb + 4 * 256 < nvec) rs_round<DT, SCALE, 4>(vb, ft, kb + tid + 1024 < nvec, ebase + (uint32_t)(kb + tid + 1024) * EPV, cs, t, inv_t, m, s);
            }
        } else {
            for (int it = 0; it < nfull; ++it, kb += 8 * 256) {
                const u32x4 va[4] = {RS_LD(kb, 0), RS_LD(kb, 1), RS_LD(kb, 2), RS_LD(kb, 3)};
                const u32x4 vb[4] = {RS_LD(kb, 4), RS_LD(kb, 5), RS_LD(kb, 6), RS_LD(kb, 7)};
                rs_round<DT, SCALE, 4>(va, ft, true, ebase + (uint32_t)(kb + tid) * EPV, cs, t, inv_t, m, s);
                __builtin_amdgcn_sched_barrier(0);       // keep the second round's unpacked values out of the first round's live range
                rs_round<DT, SCALE, 4>(vb, ft, true, ebase + (uint32_t)(kb + tid + 1024) * EPV, cs, t, inv_t, m, s);
            }
            // the rest (< 2048 vectors) as one or two rounds with all of their loads in flight at once; a lane's missing vectors
            // are -inf (no mass, never the maximum), a lane without any vector of the round does not touch the tracker
#define RS_LDP(kb_, u_) ((kb_) + tid + (u_) * 256 < nvec ? RS_LD(kb_, u_) : ninf)
            if (kb < nvec) {
                const bool two = kb + 4 * 256 < nvec;
                const u32x4 va[4] = {RS_LDP(kb, 0), RS_LDP(kb, 1), RS_LDP(kb, 2), RS_LDP(kb, 3)};
                u32x4 vb[4] = {ninf, ninf, ninf, ninf};
                if (two) { vb[0] = RS_LDP(kb, 4); vb[1] = RS_LDP(kb, 5); vb[2] = RS_LDP(kb, 6); vb[3] = RS_LDP(kb, 7); }
                rs_round<DT, SCALE, 4>(va, ft, kb + tid < nvec, ebase + (uint32_t)(kb + tid) * EPV, cs, t, inv_t, m, s);
                if (two) rs_round<DT, SCALE, 4>(vb, ft, kb + tid + 1024 < nvec, ebase + (uint32_t)(kb + tid + 1024) * EPV, cs, t, inv_t, m, s);
            }
        }
#undef RS_LDP
#undef RS_LD
        done = begin + (int64_t)nvec * EPV;
        __shared__ float s_bw[4];
        // the workgroup's maximum first (six shuffles, one LDS hop): only the lanes that hold it re-read their winning round —
        // one scattered read per item instead of one per lane (256 scattered 16-byte reads were ~10 % of an item's traffic)
        float bw = ft.bidx0 != 0xFFFFFFFFu ? ft.best : -INFINITY;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bw = __builtin_elementwise_maximum(bw, __shfl_xor(bw, off, 64));
        if ((tid & 63) == 0) s_bw[tid >> 6] = bw;
        if (__syncthreads_or((ft.saw_nan() || s != s) ? 1 : 0)) {
            scan_exact<DT>(p, begin, done, tid, best, bidx);               // NaN (or inf - inf) in the chunk: exact key rescan
        } else {
            bw = rs_max3(s_bw[0], s_bw[1], __builtin_elementwise_maximum(s_bw[2], s_bw[3]));
            if (ft.bidx0 != 0xFFFFFFFFu && ft.best == bw) {
                best = ft.ukey();
                bidx = ft.template resolve<DT>(p, ebase, nvec);
            }
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
    // the drafted id's logit for the fused finish: requested now, consumed behind the reductions
    int64_t tok = -1;
    float x_tok = 0.f;
    if (fuse && tid == 0) {
        tok = fin.draft_next[row];
        if (tok >= 0 && tok < V) x_tok = load_f<DT>(p, tok);
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
        const float2 part = make_float2(M == -INFINITY ? -INFINITY : key_to_float((uint32_t)(mm >> 32)), Ssum);
        partial[item] = part;
        atomicMax(packed + row, (unsigned long long)mm);
        if (fuse) rs_finish_row<DT>(logits, row, V, row_stride, tok, true, x_tok, t, inv_t, &part, 1, fin.p_draft, fin.row_max, fin.row_sumexp, s_tab);
    }
}

struct RsTune { int64_t items; int dyn_lds; bool fuse_finish; };
static const RsTune &rs_tune() {                            // read once: sweeps in tools/ set it before the first call
    static const RsTune t = [] {
        RsTune r{1024, 0, true};                                  // workgroups per scheduling round of the chunk split; dynamic LDS per workgroup (occupancy knob of the sweeps)
        const char *e = getenv("JF_RS_ITEMS");
        if (e && *e) { const long long v = atoll(e); if (v >= 1 && v <= (1 << 20)) r.items = v; }
        e = getenv("JF_RS_DYN_LDS");
        if (e && *e) { const long long v = atoll(e); if (v >= 0 && v <= 65536) r.dyn_lds = (int)v; }
        e = getenv("JF_RS_FUSE_FINISH");
        if (e && *e == '0') r.fuse_finish = false;
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
    const JfTiming tm = jf_take_timing();                            // events armed for this call (jf_timing_arm): taken whatever happens below
    if (R <= 0) return JF_OK;
    if (!logits || !draft_next || !p_draft || !row_max || !row_sumexp || !packed || !workspace)
        return fail(JF_E_INVALID, "jf_rs_probs: null pointer");
    if (dtype != JF_F32 && dtype != JF_BF16) return fail(JF_E_INVALID, "jf_rs_probs: dtype %d", dtype);
    if (V <= 0 || V > 0x7FFFFFFFll || row_stride < V) return fail(JF_E_INVALID, "jf_rs_probs: bad shape V=%lld stride=%lld", (long long)V, (long long)row_stride);
    if (workspace_bytes < jf_rs_workspace_bytes(R, V)) return fail(JF_E_INVALID, "jf_rs_probs: workspace too small");
    const float t = (temperature <= 0.f) ? 1.f : temperature;    // JDN:66-67
    const float inv_t = 1.f / t;
    const int esz = dtype == JF_F32 ? 4 : 2;
    int64_t cpr = 1;
    const int64_t chunk = rs_chunk(dtype, R, V, &cpr);
    // vector path: 16-byte aligned rows, and a chunk the 32-bit offsets of its buffer loads can address
    const bool vec = (((uintptr_t)logits) % 16 == 0) && ((row_stride * esz) % 16 == 0) && chunk * esz < (1ll << 31) - 65536;
    const dim3 grid((unsigned)(R * cpr)), block(256);
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *pk = (unsigned long long *)packed;
    float2 *part = (float2 *)workspace;
    const bool unit = (t == 1.f);
    // timing events armed for this call (jf_timing_arm): the first launch's start and the second launch's stop timestamps
    const bool attach = tm.any() && !jf_timing_bracket();
    if (tm.begin && !attach) (void)hipEventRecord(tm.begin, s);
    const dim3 fgrid((unsigned)((R + 255) / 256));
    // one chunk per row: the stream's workgroups finish their rows themselves (JF_RS_FUSE_FINISH=0: always two launches)
    const bool fused = cpr == 1 && rs_tune().fuse_finish;
    const RsFinishArgs fin = fused ? RsFinishArgs{draft_next, p_draft, row_max, row_sumexp} : RsFinishArgs{nullptr, nullptr, nullptr, nullptr};
#define JF_RS_P(DT, VECF, SC)                                                                                                             \
    do {                                                                                                                                  \
        if (attach) hipExtLaunchKernelGGL((rs_probs_partial_kernel<DT, VECF, SC>), grid, block, rs_tune().dyn_lds, s, tm.begin, fused ? tm.end : nullptr, 0, \
                                          logits, R, V, row_stride, t, inv_t, part, pk, (int)cpr, chunk, fin);                            \
        else rs_probs_partial_kernel<DT, VECF, SC><<<grid, block, rs_tune().dyn_lds, s>>>(logits, R, V, row_stride, t, inv_t, part, pk, (int)cpr, chunk, fin); \
    } while (0)
#define JF_RS_F(DT)                                                                                                                       \
    do {                                                                                                                                  \
        if (fused) break;                                                                                                                 \
        if (attach) hipExtLaunchKernelGGL((rs_probs_finish_kernel<DT>), fgrid, block, 0, s, nullptr, tm.end, 0, logits, R, V, row_stride, \
                                          draft_next, t, inv_t, (const float2 *)part, (int)cpr, p_draft, row_max, row_sumexp);            \
        else rs_probs_finish_kernel<DT><<<fgrid, block, 0, s>>>(logits, R, V, row_stride, draft_next, t, inv_t, part, (int)cpr, p_draft,  \
                                                                  row_max, row_sumexp);                                                    \
    } while (0)
    if (dtype == JF_F32) {
        if (vec) { if (unit) JF_RS_P(JF_F32, true, 0); else JF_RS_P(JF_F32, true, 1); }
        else { if (unit) JF_RS_P(JF_F32, false, 0); else JF_RS_P(JF_F32, false, 1); }
        JF_RS_F(JF_F32);
    } else {
        const bool fast = !unit && rs_scale_is_exact(t);
        if (vec) { if (unit) JF_RS_P(JF_BF16, true, 0); else if (fast) JF_RS_P(JF_BF16, true, 2); else JF_RS_P(JF_BF16, true, 3); }
        else { if (unit) JF_RS_P(JF_BF16, false, 0); else JF_RS_P(JF_BF16, false, 3); }
        JF_RS_F(JF_BF16);
    }
#undef JF_RS_P
#undef JF_RS_F
    if (tm.end && !attach) (void)hipEventRecord(tm.end, s);
    return check_launch("rs_probs kernels");
}

// ------------------------------------------------------------------------------------------------
// Exact probabilities (round 4).
//
// Rounds 1-3 formed  p = bf16(expf(xs - M) / S)  with a float32 row sum S: two float32 softmax implementations differ in the
// last place, after the bf16 rounding that is one ulp on ~1 % of the entries, and one inverse-CDF draw in ~10^6 then lands
// on the other side of a token boundary (profiles/soak_r02.txt, seed 4555).  The probability tensor is now DEFINED as the
// exact quotient rounded once (oracle/jacobi_oracle.py, exact_softmax_rows):
//
//     p_i = RN_dtype( exp(xs_i - M) / sum_j exp(xs_j - M) )          dtype = bf16 for bf16 logits, float32 for float32 logits
//
// and everything a decision depends on is evaluated to that definition:
//   * accept tests  u < p[proposed]  (every row): p from a float64 exp of the gathered logit over the streaming kernel's
//     float32 row sum, whose relative error is bounded by RS_EPS-ish (rs_eps_row); the test is decided when u lies outside
//     the two candidate roundings — else (~1e-4 p of the tests) the accept workgroup forms the row's float64 sum itself;
//   * the CDF of a rejected row: its 16 segment workgroups first sum float64 exps (partials exchanged inside the launch:
//     every one of them needs the whole row's S), then round every element exactly: bf16 logits take a float32 product
//     whose error band (2^-21) is tested against the rounding boundaries (1 element in ~8 000 falls back to the float64
//     quotient), float32 logits always the float64 quotient;
//   * the inverse-CDF walk and the masked argmax re-derive the few probabilities they touch with the same float64 S.
// float64 evaluation errs by ~5e-16; the oracle re-decides elements that close to a rounding boundary in 60-digit decimal
// arithmetic, the kernels do not (1 element in ~10^13).
// ------------------------------------------------------------------------------------------------
constexpr int RS_SEG = 16;
constexpr int RS_TILES = 8;                       // tiles of a segment whose vectors are kept in registers by the whole-workgroup walk
constexpr int RS_MAX_TRIES = 16;                  // JDN:135 max_tries
constexpr int RS_WT = 64;                         // wave-tile sums kept per segment (16 tiles x 4 wavefronts): V <= 16 * 16 * 256 * EPV
template <int DT> struct RsKeep { static constexpr int NV = DT == JF_BF16 ? 5 : 10; };   // vectors per lane of a Qwen-sized segment

__host__ __device__ inline int64_t rs_seg_elems(int64_t V, int epv) {
    const int64_t tile = 256 * (int64_t)epv;
    const int64_t per = (V + RS_SEG - 1) / RS_SEG;
    return ((per + tile - 1) / tile) * tile;
}
// segments that hold any element: Qwen's V = 152 064 fills 14 segments of 10 240 and a part of the 15th; the 16th is empty (the
// one-launch step gives the empty ones no workgroup: the last active segment's workgroup stores their zeros)
__host__ __device__ inline int rs_active_segs(int64_t V, int epv) {
    const int64_t segE = rs_seg_elems(V, epv);
    const int64_t n = (V + segE - 1) / segE;
    return (int)(n < 1 ? 1 : (n > RS_SEG ? RS_SEG : n));
}
__host__ __device__ inline bool rs_hier_ok(int64_t V, int epv) {       // the wave-tile sums of a segment fit RS_WT entries
    return rs_seg_elems(V, epv) / (256 * (int64_t)epv) * 4 <= RS_WT;
}

constexpr int RS_FLAG_STRIDE = 16;                // 8-byte words between two rows' accept flags (one 128-byte line each)
struct RsWs {                                     // carve-up of the step workspace for `rows` items
    double *segsum;                               // [rows, RS_SEG] float64 mass of each vocabulary segment (rounded probabilities)
    double *wtsum;                                // [rows, RS_SEG, RS_WT] the same per (tile, wavefront) of a segment, tile-major
    double *s64part;                              // [rows, RS_SEG] float64 sum of exp(xs - M) over the segment (phase A)
    double *s64;                                  // [rows] the row's float64 sum (all segments, in order)
    double *lo_part;                              // [rows] mass in front of the avoided token inside its segment
    double *p_avoid;                              // [rows] probability of the avoided token
    double *iv;                                   // [rows, 3] one-launch step: the row's (total, c_lo, c_hi) — its proposed token's CDF interval
    int32_t *sel_row;                             // [rows] logits row to sum for item i, -1 = none
    int32_t *avoid;                               // [rows] token a draw must not return (the rejected proposal), -1 = none
    float *pick_u;                                // [rows] the uniform of the draw that counts; < 0: masked argmax instead
    // hand-off words of the one-launch step (rs_step_fused_kernel), all tagged with the call's generation number, so nothing
    // has to be re-zeroed and a late poller can never see a recycled word
    unsigned long long *flag;                     // [rows * RS_FLAG_STRIDE] (gen << 32) | eos << 30 | n_accepted << 16 | (reject_pos + 2):
                                                  // the accept walk has decided the row; one 128-byte line per row
    unsigned long long *pick;                     // [rows] (gen << 32) | bits of the uniform that counts (chain workgroup -> bonus workgroup)
    unsigned long long *fin;                      // [rows] (gen << 32) | n_pads: the row's bonus workgroup has finished the row
    uint32_t *s64done;                            // [rows, RS_SEG] gen: this segment's float64 partial is stored
    uint32_t *segdone;                            // [rows, RS_SEG] gen: this segment's sums are stored
    uint32_t *ivdone;                             // [rows] gen: iv is stored (by the workgroup of the row's segment 0, for the chain)
    float *segmax;                                // [rows, RS_SEG] largest scaled logit of each segment (phase A; -inf: empty) — the masked argmax starts from these
    uint32_t *acceptdone;                         // [4]   gen: [0] the accept workgroup has written every row record, [1] the chain workgroup every draw count
    const jf_rs_filter_row *filt;                 // (not in the workspace) jf_rs_filter's records of the rows, null: plain distributions
};
static inline size_t rs_ws_bytes(int64_t rows) {
    const size_t r = (size_t)((rows + 3) / 4 * 4);
    return r * RS_SEG * sizeof(double) * 2 + r * RS_SEG * RS_WT * sizeof(double) + 6 * r * sizeof(double) + 3 * r * sizeof(int32_t) +
           r * (RS_FLAG_STRIDE + 2) * sizeof(unsigned long long) + 2 * r * RS_SEG * sizeof(uint32_t) + r * sizeof(uint32_t) + r * RS_SEG * sizeof(float) + 4 * sizeof(uint32_t);
}
__host__ __device__ inline RsWs rs_ws(void *ws, int64_t rows) {
    const size_t r = (size_t)((rows + 3) / 4 * 4);
    RsWs w;
    w.segsum = (double *)ws;
    w.wtsum = w.segsum + r * RS_SEG;
    w.s64part = w.wtsum + r * RS_SEG * RS_WT;
    w.s64 = w.s64part + r * RS_SEG;
    w.lo_part = w.s64 + r;
    w.p_avoid = w.lo_part + r;
    w.iv = w.p_avoid + r;
    w.flag = (unsigned long long *)(w.iv + 3 * r);
    w.pick = w.flag + r * RS_FLAG_STRIDE;
    w.fin = w.pick + r;
    w.sel_row = (int32_t *)(w.fin + r);
    w.avoid = w.sel_row + r;
    w.pick_u = (float *)(w.avoid + r);
    w.s64done = (uint32_t *)(w.pick_u + r);
    w.segdone = w.s64done + r * RS_SEG;
    w.ivdone = w.segdone + r * RS_SEG;
    w.segmax = (float *)(w.ivdone + r);
    w.acceptdone = (uint32_t *)(w.segmax + r * RS_SEG);
    w.filt = nullptr;
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
__device__ unsigned long long g_rsphase[16];      // phases of ONE segment workgroup (row 0, segment 0)
#define RS_PHASE(item, seg, k) do { if ((item) == 0 && (seg) == 0 && threadIdx.x == 0) g_rsphase[k] = (unsigned long long)__builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" __attribute__((visibility("default"))) int jf_exp_rs_phases(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rsphase), sizeof(unsigned long long) * 16);
}
#else
#define RS_PHASE(item, seg, k) do { } while (0)
#define RS_STAMP_MIN(k) do { } while (0)
#define RS_STAMP_MAX(k) do { } while (0)
#define RS_ROWSTAMP(k, b) do { } while (0)
#endif

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// Inclusive scan over the 64 lanes with DPP moves (the data-parallel-primitive lanes of the VALU: ~8 cycles a step; the
// ds_bpermute shuffles of rounds 1-3 cost ~60 cycles each, and a segment workgroup scans every tile): Hillis-Steele inside the
// rows of 16 lanes (row_shr 1, 2, 4, 8), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3.  Lanes without
// a source add 0.  The order of the additions is fixed, and every running sum of the CDF is formed by this one function
// (rs_seg_prob_sums, rs_pick_wave, rs_pick_wg), so the sums agree wherever they are re-formed.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double rs_dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_incl_scan_f64(double v, int lane) {
#ifdef JF_RS_SHFL_SCAN
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
#else
    (void)lane;
    v += rs_dpp_f64<0x111, 0xf>(v);      // row_shr:1
    v += rs_dpp_f64<0x112, 0xf>(v);      // row_shr:2
    v += rs_dpp_f64<0x114, 0xf>(v);      // row_shr:4
    v += rs_dpp_f64<0x118, 0xf>(v);      // row_shr:8
    v += rs_dpp_f64<0x142, 0xa>(v);      // row_bcast:15 -> rows 1, 3
    v += rs_dpp_f64<0x143, 0xc>(v);      // row_bcast:31 -> rows 2, 3
#endif
    return v;
}
// the value of the lane below (0 for lane 0): the exclusive prefix from an inclusive scan
__device__ __forceinline__ double wave_shift_up_f64(double v, int lane) {
#ifdef JF_RS_SHFL_SCAN
    const double up = __shfl_up(v, 1, 64);
    return lane == 0 ? 0.0 : up;
#else
    (void)lane;
    return rs_dpp_f64<0x138, 0xf>(v);    // wave_shr:1
#endif
}

template <int DT>
__device__ __forceinline__ RsRow rs_make_row(const void *logits, int64_t r, int64_t V, int64_t row_stride, float t, float M, float S) {
    RsRow rr;
    const int esz = DT == JF_F32 ? 4 : 2;
    rr.p = (const char *)logits + r * row_stride * esz;
    // the steps pass -T when the host found the product form of the bf16 scaling exact for this T (jf_rs_step / _onpolicy_step)
    rr.fast = t < 0.f;
    t = fabsf(t);
    rr.V = V; rr.t = t; rr.inv_t = 1.f / t; rr.M = M; rr.S = S;
    rr.prob = (S == RS_PROB_ROW);
    rr.flt = nullptr;
    rr.unit_t = (t == 1.f);
    rr.vec = (((uintptr_t)rr.p) % 16) == 0;
    return rr;
}

__device__ __forceinline__ void st_agent_f64(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agent_f64(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent_f32(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent_f32(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// bounded in-kernel waits: a wait that lasts 2 s of the 100 MHz constant clock gives up; the caller reports
// JF_E_LAUNCH through rows[0].rsv instead of hanging the GPU
constexpr unsigned long long RS_WAIT_TICKS = 200000000ull;
__device__ __forceinline__ bool rs_wait_flag(const unsigned long long *word, uint32_t gen, unsigned long long *out, int sleep = 16) {
    unsigned long long v;
    unsigned spins = 0;
    unsigned long long t0 = 0;
    while ((uint32_t)((v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != gen) {
        __builtin_amdgcn_s_sleep(16);
        if ((++spins & 1023u) == 0u) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t0 == 0) t0 = now;
            else if (now - t0 > RS_WAIT_TICKS) { *out = 0ull; return false; }
        }
    }
    (void)sleep;
    *out = v;
    return true;
}
__device__ __forceinline__ bool rs_wait_word(const uint32_t *word, uint32_t gen) {
    unsigned spins = 0;
    unsigned long long t0 = 0;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) {
        __builtin_amdgcn_s_sleep(8);
        if ((++spins & 1023u) == 0u) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t0 == 0) t0 = now;
            else if (now - t0 > RS_WAIT_TICKS) return false;
        }
    }
    return true;
}

// scaled logits of one 16-byte vector, as torch forms them for this dtype
template <int DT>
__device__ __forceinline__ void rs_scaled_from_vec(const RsRow &r, const u32x4 v, float (&xs)[Elem<DT>::EPV]) {
    rs_unpack<DT>(v, xs);
    if (r.unit_t) return;                                      // wave-uniform branches OUTSIDE the element loop: the IEEE division
    if (DT == JF_BF16 && r.fast) {                             // (12 instructions per element) must not be if-converted into the product's path
#pragma unroll
        for (int j = 0; j + 1 < Elem<DT>::EPV; j += 2) scale_bf16_fast2(xs[j], xs[j + 1], r.inv_t);   // v_pk_mul_f32 + v_cvt_pk_bf16_f32: two elements at once
    } else {
#pragma unroll
        for (int j = 0; j < Elem<DT>::EPV; ++j) xs[j] = rs_scaled<DT>(xs[j], r.t, r.inv_t, false, false);
    }
}
// exact probabilities of one vector (elements >= V hold -inf: probability 0), given the row's float64 1 / S
template <int DT>
__device__ __forceinline__ void rs_exact_probs_from_vec(const RsRow &r, const u32x4 v, double invS, const double *tab, float (&p)[Elem<DT>::EPV]) {
    float xs[Elem<DT>::EPV];
    rs_scaled_from_vec<DT>(r, v, xs);
#pragma unroll
    for (int j = 0; j < Elem<DT>::EPV; ++j) p[j] = rs_round_prob<DT>(rs_e64(xs[j], (double)r.M, tab) * invS);
}
// probabilities of one vector by whichever definition the row takes: exact (invS > 0) or the plain float32 formula
template <int DT>
__device__ __forceinline__ void rs_any_probs_from_vec(const RsRow &r, const u32x4 v, double invS, const double *tab, float (&p)[Elem<DT>::EPV]) {
    if (invS > 0.0) rs_exact_probs_from_vec<DT>(r, v, invS, tab, p);
    else rs_probs_from_vec<DT>(r, v, p);
}

// the same for a row that may carry a filter record (e0 = id of the vector's first element): the exact probability of every id
// that can be kept (scaled logit >= x_keep: the others are 0 whatever their probability), then the record's map
template <int DT>
__device__ __forceinline__ void rs_row_probs_from_vec(const RsRow &r, const u32x4 v, int64_t e0, double invS, const double *tab, float (&p)[Elem<DT>::EPV]) {
    if (!r.flt) { rs_any_probs_from_vec<DT>(r, v, invS, tab, p); return; }
    const jf_rs_filter_row &f = *r.flt;
    float xs[Elem<DT>::EPV];
    rs_scaled_from_vec<DT>(r, v, xs);
    const double inv = f.sum > 0.0 ? invS : 0.0;
#pragma unroll
    for (int j = 0; j < Elem<DT>::EPV; ++j) {
        float q = 0.f;
        if (xs[j] >= f.x_keep && inv > 0.0) q = rs_filter_apply<DT>(f, rs_round_prob<DT>(rs_e64(xs[j], (double)r.M, tab) * inv), e0 + j);
        p[j] = q;
    }
}
// a row of a step: jf_rs_probs' statistics, or — filt given — the record of jf_rs_filter (which marked the statistics)
template <int DT>
__device__ __forceinline__ RsRow rs_step_row(const void *logits, int64_t r, int64_t V, int64_t row_stride, float t, const float *row_max,
                                             const float *row_sumexp, const jf_rs_filter_row *filt) {
    if (!filt) return rs_make_row<DT>(logits, r, V, row_stride, t, row_max[r], row_sumexp[r]);
    RsRow rr = rs_make_row<DT>(logits, r, V, row_stride, t, filt[r].row_max, 1.f);
    rr.flt = filt + r;
    return rr;
}
// float64 sum of a filtered row's softmax (the record's), as the consumers' S
__device__ __forceinline__ double rs_flt_sum(const RsRow &row) { return row.flt->sum; }

// float64 sum of exp(xs - M) over a whole row by one workgroup (every thread gets the result).  Used where ONE row's exact
// sum is needed on the spot: an accept test that the float32 sum cannot decide, and the rows of the on-policy accept.
template <int DT, int NB = 4 /* loads in flight per thread: the one-launch step calls this with its register budget in mind; jf_rs_filter takes 8 */>
__device__ double rs_row_s64_wg(const RsRow &row, const double *tab, double *s_red /* LDS, 4 */) {
    constexpr int EPV = Elem<DT>::EPV;
    double acc = 0.0;
    for (int64_t b0 = (int64_t)threadIdx.x * EPV; b0 < row.V; b0 += (int64_t)NB * 256 * EPV) {
        u32x4 v[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) { const int64_t e0 = b0 + (int64_t)k * 256 * EPV; if (e0 < row.V) v[k] = rs_load_vec<DT>(row, e0); }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int64_t e0 = b0 + (int64_t)k * 256 * EPV;
            if (e0 >= row.V) continue;
            float xs[EPV];
            rs_scaled_from_vec<DT>(row, v[k], xs);
#pragma unroll
            for (int j = 0; j < EPV; ++j) acc += rs_e64(xs[j], (double)row.M, tab);
        }
    }
    acc = wave_sum_f64(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}
// exact probability of one element, by the whole workgroup (uniform control flow)
template <int DT>
__device__ float rs_exact_prob_wg(const void *logits, int64_t r, int64_t V, int64_t row_stride, float t, float M, int64_t tok,
                                  const double *tab, double *s_red) {
    const RsRow row = rs_make_row<DT>(logits, r, V, row_stride, t, M, 1.f);
    const double S = rs_row_s64_wg<DT>(row, tab, s_red);
    if (tok < 0 || tok >= V) return 0.f;
    const float xs = rs_scaled<DT>(load_f<DT>(row.p, tok), row.t, row.inv_t, row.unit_t, row.fast);
    return rs_round_prob<DT>(rs_e64(xs, (double)M, tab) / S);
}

// ------------------------------------------------------------------------------------------------
// (a19, round 6) top-k / top-p of a bf16 row WITHOUT its tensor: one workgroup of 1 024 threads per row, the COUNT of every bf16
// pattern of the row's scaled logits in LDS.  The exactly rounded probability is a non-decreasing function of the scaled logit,
// and a bf16 scaled logit is one of 65 536 patterns: S = sum count x exp(x - M), the top-k cut, the nucleus, their sums and the
// groups of equal values are all sums over <= 65 536 counters, formed once the row has been streamed ONCE (integer work per
// element: scale, key, one LDS atomic; float64 exps only per OCCUPIED pattern).  A cut that falls inside a group of equal values
// keeps that group's first ids in id order: the id of the last one kept needs the row a second time (matches per tile of 8 192
// ids, then the tile that holds it).  Out: one jf_rs_filter_row per row (include/jacobiforcing.h) — 48 bytes instead of
// V x 2 — and the final probability of the drafted id.
//   counters  two 16-bit counts per LDS word (128 KB): a pattern that occurs more than 65 535 times in a row (at most two can)
//             wraps; the add returns the old value, a wrap is noticed, and the row is counted again with such patterns in 32-bit
//             side counters (rows of equal logits; never seen with a model).
//   ownership thread t owns the 64 patterns [65536 - 64 (t + 1), 65536 - 64 t): thread 0 the largest values.  Prefix sums over
//             threads are "everything above"; a thread reads its 32 words rotated by t (conflict-free), walks them in order only
//             where a cut is located.
//   order     every float64 sum is formed in a fixed order (a thread's words in its rotated order, the DPP wave scan, the 16
//             wavefronts in order): the record does not depend on scheduling.
// ------------------------------------------------------------------------------------------------
#ifdef JF_EXP_FLT_TRACE
__device__ unsigned long long g_fhtrace[16];
extern "C" __attribute__((visibility("default"))) int jf_exp_fh_trace(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fhtrace), sizeof(g_fhtrace)) == hipSuccess ? 0 : -1;
}
#define FH_STAMP(k) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) g_fhtrace[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FH_STAMP(k) do { } while (0)
#endif
constexpr int FH_TPB = 1024, FH_NW = FH_TPB / 64, FH_WORDS = 32768, FH_OVF = 8, FH_TILE = FH_TPB * 8, FH_MAX_TILES = 64, FH_ITEMS = 4096;
constexpr unsigned FH_LDS = FH_WORDS * 4 + FH_ITEMS * 4 + FH_ITEMS * 2;      // counters + the list of occupied patterns + their probabilities
struct FhShared {
    double tab[64];
    double dred[FH_NW];
    long long lred[FH_NW];
    uint32_t ovf_key[FH_OVF];
    uint32_t ovf_cnt[FH_OVF];
    int n_ovf, ovf_seen;
    int tileA[FH_MAX_TILES], tileB[FH_MAX_TILES];
    int scan[FH_TPB];
    double bd[4];                 // broadcast slots
    long long bl[4];
    int bi[8];
};
__device__ __forceinline__ uint32_t fh_key(uint32_t h16) { return (h16 & 0x8000u) ? (~h16 & 0xFFFFu) : (h16 | 0x8000u); }     // ascending with the value
__device__ __forceinline__ float fh_value(uint32_t key) { return __uint_as_float(((key & 0x8000u) ? (key & 0x7FFFu) : (~key & 0xFFFFu)) << 16); }
__device__ __forceinline__ uint32_t fh_count(const FhShared &sh, const uint32_t *hist, uint32_t key) {
    const uint32_t w = hist[key >> 1];
    uint32_t c = (key & 1u) ? (w >> 16) : (w & 0xFFFFu);
    if (sh.n_ovf) for (int i = 0; i < sh.n_ovf; ++i) c += sh.ovf_key[i] == key ? sh.ovf_cnt[i] : 0u;
    return c;
}
// workgroup sums in a fixed order (every thread gets them); `incl` = this thread's inclusive prefix over the thread ids
__device__ __forceinline__ void fh_scan(FhShared &sh, long long &cnt, double &sum, long long &cnt_incl, double &sum_incl, long long &cnt_excl, double &sum_excl) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double si = wave_incl_scan_f64(sum, lane);
    const double su = wave_shift_up_f64(si, lane);                          // (the exclusive prefix is the neighbour's inclusive one: the two agree bit for bit)
    long long ci = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const long long o = __shfl_up(ci, off, 64); if (lane >= off) ci += o; }
    const long long cu = ci - cnt;
    __syncthreads();
    if (lane == 63) { sh.dred[wave] = si; sh.lred[wave] = ci; }
    __syncthreads();
    double sb = 0.0, st = 0.0;
    long long cb = 0, ct = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { if (w == wave) { sb = st; cb = ct; } st += sh.dred[w]; ct += sh.lred[w]; }
    cnt_incl = cb + ci; sum_incl = sb + si;
    cnt_excl = cb + cu; sum_excl = sb + su;
    cnt = ct; sum = st;
}
__device__ __forceinline__ void fh_reduce(FhShared &sh, long long &cnt, double &sum) {
    long long ci, ce; double si, se;
    fh_scan(sh, cnt, sum, ci, si, ce, se);
}
__device__ __forceinline__ void fh_min_max2(FhShared &sh, int &lo, int &hi) {             // workgroup minimum of lo and maximum of hi (every thread gets them)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int a = __shfl_xor(lo, off, 64), b = __shfl_xor(hi, off, 64); lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh.scan[threadIdx.x >> 6] = lo; sh.scan[FH_NW + (threadIdx.x >> 6)] = hi; }
    __syncthreads();
    lo = sh.scan[0]; hi = sh.scan[FH_NW];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { const int a = sh.scan[w], b = sh.scan[FH_NW + w]; lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
    __syncthreads();
}
__device__ __forceinline__ int fh_min_max(FhShared &sh, int v, bool want_max) {          // workgroup min / max of an int (every thread gets it)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(v, off, 64); v = want_max ? (o > v ? o : v) : (o < v ? o : v); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh.scan[threadIdx.x >> 6] = v;
    __syncthreads();
    int r = sh.scan[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { const int o = sh.scan[w]; r = want_max ? (o > r ? o : r) : (o < r ? o : r); }
    __syncthreads();
    return r;
}

// the row streamed once: every scaled logit that can carry mass (>= M + RS_EXP_CUT) counted under its key.  side: count the listed
// patterns (sh.ovf_key) in 32-bit side counters instead (second attempt of a row whose 16-bit counters wrapped)
template <bool SIDE>
__device__ __forceinline__ void fh_count_row(FhShared &sh, uint32_t *hist, const RsRow &row) {
    constexpr int EPV = 8, NB = 8;
    const float mcut = row.M + (float)RS_EXP_CUT;
    for (int64_t b0 = (int64_t)threadIdx.x * EPV; b0 < row.V; b0 += (int64_t)NB * blockDim.x * EPV) {
        u32x4 v[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) { const int64_t e0 = b0 + (int64_t)k * blockDim.x * EPV; if (e0 < row.V) v[k] = rs_load_vec<JF_BF16>(row, e0); }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int64_t e0 = b0 + (int64_t)k * blockDim.x * EPV;
            if (e0 >= row.V) continue;
            float xs[EPV];
            rs_scaled_from_vec<JF_BF16>(row, v[k], xs);
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                if (!(xs[j] >= mcut)) continue;                            // (slots beyond V hold -inf)
                const uint32_t key = fh_key(__float_as_uint(xs[j]) >> 16);
                if constexpr (SIDE) {
                    bool listed = false;
                    for (int i = 0; i < sh.n_ovf; ++i) if (sh.ovf_key[i] == key) { atomicAdd(&sh.ovf_cnt[i], 1u); listed = true; }
                    if (listed) continue;
                }
                const uint32_t hi = key & 1u;
                const uint32_t old = atomicAdd(&hist[key >> 1], hi ? 0x10000u : 1u);
                if (__builtin_expect(((hi ? (old >> 16) : old) & 0xFFFFu) == 0xFFFFu, 0)) {        // this counter wrapped (or, rarely, its word was seen mid-carry)
                    const int slot = atomicAdd(&sh.ovf_seen, 1);
                    if (slot < FH_OVF) sh.ovf_key[slot] = key;
                }
            }
        }
    }
    __syncthreads();
}

// the row streamed once, fast: plain (non-returning) LDS adds; the thread's number of counted elements comes back for the checksum
// that notices a wrapped counter afterwards (sum of the counters != elements counted)
__device__ __forceinline__ uint32_t fh_count_row_fast(uint32_t *hist, const RsRow &row) {
    constexpr int EPV = 8, NB = 8;
    const float mcut = row.M + (float)RS_EXP_CUT;
    uint32_t mine = 0u;
    for (int64_t b0 = (int64_t)threadIdx.x * EPV; b0 < row.V; b0 += (int64_t)NB * blockDim.x * EPV) {
        u32x4 v[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) { const int64_t e0 = b0 + (int64_t)k * blockDim.x * EPV; if (e0 < row.V) v[k] = rs_load_vec<JF_BF16>(row, e0); }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int64_t e0 = b0 + (int64_t)k * blockDim.x * EPV;
            if (e0 >= row.V) continue;
            float xs[EPV];
            rs_scaled_from_vec<JF_BF16>(row, v[k], xs);
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                if (xs[j] >= mcut) {                                       // (slots beyond V hold -inf)
                    const uint32_t key = fh_key(__float_as_uint(xs[j]) >> 16);
                    atomicAdd(&hist[key >> 1], (key & 1u) ? 0x10000u : 1u);
                    ++mine;
                }
            }
        }
    }
    __syncthreads();
    return mine;
}

// packed bf16 patterns, two ids per register (v_pk_*_u16): what the zone kernel's count pass and the tie pass stream with
typedef unsigned short zn_u16x2 __attribute__((ext_vector_type(2)));
typedef short zn_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t zn_pk_sub(uint32_t a, uint32_t b) {      // v_pk_sub_u16 (wraps per half)
    return __builtin_bit_cast(uint32_t, (zn_u16x2)(__builtin_bit_cast(zn_u16x2, a) - __builtin_bit_cast(zn_u16x2, b)));
}
__device__ __forceinline__ uint32_t zn_pk_min(uint32_t a, uint32_t b) {      // v_pk_min_u16
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(zn_u16x2, a), __builtin_bit_cast(zn_u16x2, b)));
}
__device__ __forceinline__ uint32_t zn_pk_rotl1(uint32_t a) {                // each half rotated left by one: magnitude << 1 | sign
    const zn_u16x2 v = __builtin_bit_cast(zn_u16x2, a);
    return __builtin_bit_cast(uint32_t, (zn_u16x2)((v << (zn_u16x2){1, 1}) | (v >> (zn_u16x2){15, 15})));
}
__device__ __forceinline__ uint32_t zn_pk_key(uint32_t a) {                  // fh_key of both halves (ascending with the value)
    const zn_s16x2 m = __builtin_bit_cast(zn_s16x2, a) >> (zn_s16x2){15, 15};
    return a ^ (__builtin_bit_cast(uint32_t, m) | 0x80008000u);
}
__device__ __forceinline__ uint32_t zn_pk_has_zero(uint32_t e) { return (e - 0x00010001u) & ~e & 0x80008000u; }   // non-zero iff a half of e is zero
// the scaled logits of one 16-byte vector as bf16 patterns, two per word (the even id in the low half), as torch forms them
__device__ __forceinline__ void rs_scaled_pk_from_vec(const RsRow &r, const u32x4 v, uint32_t (&pk)[4]) {
    pk[0] = v.x; pk[1] = v.y; pk[2] = v.z; pk[3] = v.w;
    if (r.unit_t) return;                                                    // (wave-uniform branches outside the element loop, as rs_scaled_from_vec)
    if (r.fast) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const rs_f32x2 q = {__uint_as_float(pk[j] << 16) * r.inv_t, __uint_as_float(pk[j] & 0xFFFF0000u) * r.inv_t};
            pk[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(q, rs_bf16x2));
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = bf16_rne(__fdiv_rn(__uint_as_float(pk[j] << 16), r.t)), b = bf16_rne(__fdiv_rn(__uint_as_float(pk[j] & 0xFFFF0000u), r.t));
            pk[j] = (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xFFFF0000u);
        }
    }
}

// What top-k / top-p make of the row, from its pattern counts.  FAST: the occupied patterns as a compact list in LDS, largest value
// first — item i = occ[i] = key << 16 | count, its probability's bf16 bits in pc[i]; a thread owns `ipt` consecutive items.  !FAST:
// the counters themselves, a thread owns 64 consecutive patterns (rows with more than FH_ITEMS occupied patterns, or a pattern that
// occurs more than 65 535 times).  The stages are written once over "this thread's items" (any order) and "this thread's items,
// largest value first".
template <bool FAST>
__device__ __forceinline__ float fh_solve(FhShared &sh, const uint32_t *hist, const uint32_t *occ, uint16_t *pc, int n_occ, const RsRow &row, int64_t r,
                                          int64_t V, const int64_t *draft_next, int top_k, double top_p, bool k_on, bool p_on, jf_rs_filter_row &rec) {
    constexpr int EPV = 8;
    const int tid = threadIdx.x;
    const float M = row.M;
    const uint32_t kbase = 65536u - 64u * (uint32_t)(tid + 1);
    const int TPB = (int)blockDim.x, TILE = TPB * 8;                       // (1 024 threads over the counters, 512 over the zone table)
    const int ipt = (n_occ + TPB - 1) / TPB, i0 = tid * ipt, i1 = i0 + ipt < n_occ ? i0 + ipt : n_occ;
    // f(key, count, item) over this thread's occupied patterns (the counters: in the thread's rotated word order, conflict-free)
    auto for_mine = [&](auto f) {
        if constexpr (FAST) {
            for (int i = i0; i < i1; ++i) { const uint32_t w = occ[i]; f(w >> 16, w & 0xFFFFu, i); }
        } else {
#pragma unroll 4
            for (int i = 0; i < 32; ++i) {
                const uint32_t wi = (uint32_t)(i + tid) & 31u;
                const uint32_t w = hist[(kbase >> 1) + wi];
                if (w == 0u && sh.n_ovf == 0) continue;
                const uint32_t k0 = kbase + 2u * wi;
                const uint32_t c0 = fh_count(sh, hist, k0), c1 = fh_count(sh, hist, k0 + 1u);
                if (c0) f(k0, c0, 0);
                if (c1) f(k0 + 1u, c1, 0);
            }
        }
    };
    // the same, largest value first, until f returns true
    auto walk_mine = [&](auto f) {
        if constexpr (FAST) {
            for (int i = i0; i < i1; ++i) { const uint32_t w = occ[i]; if (f(w >> 16, w & 0xFFFFu, i)) break; }
        } else {
            for (int key = (int)kbase + 63; key >= (int)kbase; --key) {
                const uint32_t n = fh_count(sh, hist, (uint32_t)key);
                if (n && f((uint32_t)key, n, 0)) break;
            }
        }
    };
    // ---- the exact sum, then every pattern's probability (FAST: kept per item)
    long long cnt = 0;
    double sum = 0.0;
    for_mine([&](uint32_t key, uint32_t c, int) { sum += (double)c * rs_e64(fh_value(key), (double)M, sh.tab); });
    fh_reduce(sh, cnt, sum);
    const double S = sum, invS = 1.0 / S;                                  // (S >= 1: the maximum itself contributes exp(0))
    rec.sum = S;
    if constexpr (FAST) {
        for (int i = i0; i < i1; ++i) pc[i] = (uint16_t)(__float_as_uint(rs_bf16_of_f64(rs_e64(fh_value(occ[i] >> 16), (double)M, sh.tab) * invS)) >> 16);
        __syncthreads();
    }
    FH_STAMP(3);
    auto P = [&](uint32_t key, int i) {
        if constexpr (FAST) return __uint_as_float((uint32_t)pc[i] << 16);
        else return rs_bf16_of_f64(rs_e64(fh_value(key), (double)M, sh.tab) * invS);
    };
    const float floor_d = rs_round_prob<JF_BF16>(1e-12);                    // sum.clamp_min(1e-12) in the dtype
    // counts and sums of the patterns whose value val(key, item) lies above / at a bit pattern: (n above, their sum, n at, key range at)
    auto query = [&](auto val, uint32_t at_bits, long long &n_gt, double &s_gt, long long &n_eq, int &k_lo, int &k_hi) {
        long long ng = 0, ne = 0;
        double sg = 0.0;
        int lo = 0x7FFFFFFF, hi = -1;
        for_mine([&](uint32_t key, uint32_t c, int i) {
            const float v = val(key, i);
            const uint32_t b = __float_as_uint(v);
            if (b > at_bits) { ng += c; sg += (double)c * (double)v; }
            else if (b == at_bits && b != 0u) { ne += c; lo = (int)key < lo ? (int)key : lo; hi = (int)key > hi ? (int)key : hi; }
        });
        long long both = ng | (ne << 32);                                    // (each below 2^20: one reduction carries the two counts)
        fh_reduce(sh, both, sg);
        ng = both & 0xFFFFFFFFll; ne = both >> 32;
        fh_min_max2(sh, lo, hi);
        k_lo = lo; k_hi = hi;
        n_gt = ng; s_gt = sg; n_eq = ne;
    };
    // the first thread (largest values first) at which pred(inclusive count, inclusive sum) holds walks its patterns downwards and
    // returns the first key at which it holds (and that item's value in sh.bd[0]); -1: nowhere.  pred is monotone along the walk.
    auto locate = [&](auto val, auto pred) -> int {
        long long c = 0, ci, ce;
        double s = 0.0, si, se;
        for_mine([&](uint32_t key, uint32_t n, int i) { const float v = val(key, i); if (v > 0.f) { c += n; s += (double)n * (double)v; } });
        fh_scan(sh, c, s, ci, si, ce, se);
        if (tid == 0) sh.bi[0] = -1;
        __syncthreads();
        if (pred(ci, si) && !pred(ce, se)) {                                // the crossing lies in this thread's patterns
            long long cc = ce;
            double ss = se;                                                 // (the prefix above this thread; the walk re-forms the thread's own sum in key order)
            int last = -1;
            float vlast = 0.f;
            walk_mine([&](uint32_t key, uint32_t n, int i) {
                const float v = val(key, i);
                if (!(v > 0.f)) return false;
                cc += n; ss += (double)n * (double)v; last = (int)key; vlast = v;
                return pred(cc, ss);
            });
            sh.bi[0] = last; sh.bd[0] = (double)vlast;                      // (pred(ci, si) held: the walk ends on a crossing, or — rounding of the re-formed sum only — on its last item)
        }
        __syncthreads();
        const int k = sh.bi[0];
        __syncthreads();
        return k;
    };
    // ---- top-k (JDN:73-84): the k-th largest probability, the ids above it, `need1` ids at it
    uint32_t v1 = 0u;                                                       // bits of the cut value (0: every id is kept)
    long long need1 = 0, n_eq1 = 0;
    int ka1 = 0x7FFFFFFF, kb1 = -1;                                         // keys of the cut group
    float s1 = 1.f;
    if (k_on) {
        const int kk = locate(P, [&](long long c, double) { return c >= (long long)top_k; });
        if (kk >= 0) {
            v1 = __float_as_uint((float)sh.bd[0]);
            long long n_gt; double s_gt;
            query(P, v1, n_gt, s_gt, n_eq1, ka1, kb1);
            need1 = (long long)top_k - n_gt;
            if (need1 > n_eq1) need1 = n_eq1;
            s1 = rs_round_prob<JF_BF16>(s_gt + (double)need1 * (double)__uint_as_float(v1));
        } else {                                                             // fewer than k ids with mass: all of them (and the zeros) are kept
            long long n_gt, ne; double s_gt; int a, b;
            query(P, 0u, n_gt, s_gt, ne, a, b);
            s1 = rs_round_prob<JF_BF16>(s_gt);
        }
        s1 = s1 > floor_d ? s1 : floor_d;
    }
    rec.cut1 = v1; rec.s1 = s1;
    FH_STAMP(4);
    const float yv = v1 ? flt_div<JF_BF16>(__uint_as_float(v1), s1) : 0.f;   // what a kept id of the cut group becomes
    // y of a pattern: 0 at and below the cut (the cut group's kept ids enter the sums below as need1 x yv)
    auto Y = [&](uint32_t key, int i) {
        const float p = P(key, i);
        if (!k_on) return p;
        return __float_as_uint(p) > v1 ? flt_div<JF_BF16>(p, s1) : 0.f;
    };
    // ---- top-p (JDN:91-107) on y
    uint32_t v2 = 0u;
    long long c2 = 0, n_eq2 = 0;                                            // ids kept of the group at the cut / ids in that group
    int ka2 = 0x7FFFFFFF, kb2 = -1;
    bool grp1_in2 = false;                                                  // the top-k cut group's kept ids belong to the top-p cut group (yv == cut2)
    float s2 = 1.f;
    bool all2 = true;
    if (p_on) {
        const float tp = rs_round_prob<JF_BF16>((double)(float)top_p);      // `cdf <= tp`: the Python float is cast to the tensor's dtype
        long long n_all, ne; double s_all; int a, b;
        query(Y, 0u, n_all, s_all, ne, a, b);
        const double total = s_all + (double)need1 * (double)yv;
        all2 = rs_round_prob<JF_BF16>(total) <= tp;
        int kk = -1;
        // the group the cut falls into: the largest value whose cumulative sum (everything >= it), rounded, exceeds tp
        if (!all2) kk = locate(Y, [&](long long, double s) { return !(rs_round_prob<JF_BF16>(s) <= tp); });
        if (!all2 && kk < 0 && need1 == 0) all2 = true;                     // (rounding between the two orders of the same sum only)
        if (all2) s2 = rs_round_prob<JF_BF16>(total);
        else {
            const float y2 = kk >= 0 ? (float)sh.bd[0] : yv;                // nowhere above the top-k cut: the cut group itself
            v2 = __float_as_uint(y2);
            long long n_whole; double c_whole;
            query(Y, v2, n_whole, c_whole, n_eq2, ka2, kb2);
            grp1_in2 = need1 > 0 && __float_as_uint(yv) == v2;
            if (grp1_in2) n_eq2 += need1;
            long long lo = 0, hi = n_eq2;                                   // ids of the group whose own cumulative sum passes: pass(lo) holds, pass(hi) fails
            while (hi - lo > 1) { const long long m2 = lo + (hi - lo) / 2; if (rs_round_prob<JF_BF16>(c_whole + (double)m2 * (double)y2) <= tp) lo = m2; else hi = m2; }
            c2 = lo;
            if (n_whole == 0 && c2 == 0) c2 = 1;                            // keep[..., 0] = True
            s2 = rs_round_prob<JF_BF16>(c_whole + (double)c2 * (double)y2);
        }
        s2 = s2 > floor_d ? s2 : floor_d;
    }
    rec.cut2 = v2; rec.s2 = s2;
    FH_STAMP(5);
    // ---- the last kept id of a cut group that is kept in part: the row a second time
    const bool tie1_needed = k_on && v1 && need1 < n_eq1;                   // else every id at the cut is kept (tie1 = V - 1)
    const bool tie2_needed = p_on && !all2 && c2 > 0 && c2 < n_eq2;
    if (p_on && !all2 && c2 == 0) rec.tie2 = -1;
    if (tie1_needed && need1 == 0) rec.tie1 = -1;
#if defined(JF_EXP_ZN_STOP) && JF_EXP_ZN_STOP == 2                              // (experiment: no tie pass)
    const bool pass2 = false;
#else
    const bool pass2 = (tie1_needed && need1 > 0) || tie2_needed;
#endif
    if (pass2) {
        const int64_t ntiles = (V + TILE - 1) / TILE;
        __syncthreads();
        if (tid < FH_MAX_TILES) { sh.tileA[tid] = 0; sh.tileB[tid] = 0; }
        __syncthreads();
        // class A: keys of the top-k cut group; class B: keys above the top-k cut whose y equals the top-p cut (a key range: y is monotone)
        const int a_lo = ka1, a_hi = kb1, b_lo = ka2, b_hi = kb2;
        auto classes = [&](const u32x4 vv, int64_t e0, uint32_t &ma, uint32_t &mb) {
            float xs[EPV];
            rs_scaled_from_vec<JF_BF16>(row, vv, xs);
            ma = mb = 0u;
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                const int key = (int)fh_key(__float_as_uint(xs[j]) >> 16);
                const bool in = e0 + j < V;
                ma |= (in && key >= a_lo && key <= a_hi) ? 1u << j : 0u;
                mb |= (in && key >= b_lo && key <= b_hi) ? 1u << j : 0u;
            }
        };
        {
            // The pass is a stream with a handful of hits: a vector is first asked, on its packed patterns (two ids per instruction, no
            // branch), whether ANY of its eight ids belongs to a class; only such vectors are taken apart.  And it ends early: the ids
            // are wanted in id order, so once the tiles read so far hold the ranks asked for, the rest of the row is not read.
            constexpr int NB = FAST ? 4 : 8;
            const bool wantA = (tie1_needed && need1 > 0) || (tie2_needed && grp1_in2), wantB = tie2_needed;
            const bool early = !grp1_in2;                                    // (with the top-k group inside the top-p group the ranks are not per class)
            const uint32_t aLo2 = (uint32_t)(a_lo & 0xFFFF) * 0x00010001u, aSp2 = (uint32_t)((a_hi - a_lo) & 0xFFFF) * 0x00010001u;
            const uint32_t bLo2 = (uint32_t)(b_lo & 0xFFFF) * 0x00010001u, bSp2 = (uint32_t)((b_hi - b_lo) & 0xFFFF) * 0x00010001u;
            const bool rangeA = wantA && a_hi >= a_lo, rangeB = wantB && b_hi >= b_lo;
            long long cumA = 0, cumB = 0;
            constexpr int NPRE = FAST ? NB : 1;                             // (the zone kernel prefetches; the 1 024-thread kernel of the rare rows has no registers for it)
            u32x4 v[NB], vn[NPRE];
            auto load_batch = [&](int64_t tb, u32x4 (&dst)[NB]) {
#pragma unroll
                for (int k = 0; k < NB; ++k) { const int64_t e0 = (tb + k) * TILE + (int64_t)tid * EPV; if (tb + k < ntiles && e0 < V) dst[k] = rs_load_vec<JF_BF16>(row, e0); }
            };
            if constexpr (FAST) load_batch(0, v);
            for (int64_t t0 = 0; t0 < ntiles; t0 += NB) {
                if constexpr (FAST) {                                        // the next batch's loads, in front of this batch's arithmetic and its barrier
#pragma unroll
                    for (int k = 0; k < NB; ++k) {
                        const int64_t e0 = (t0 + NB + k) * TILE + (int64_t)tid * EPV;
                        if (t0 + NB + k < ntiles && e0 < V) vn[k] = rs_load_vec<JF_BF16>(row, e0);
                    }
                } else load_batch(t0, v);
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    if (t0 + k >= ntiles) continue;                          // (workgroup-uniform)
                    const int64_t e0 = (t0 + k) * TILE + (int64_t)tid * EPV;
                    if (e0 >= V) continue;
                    uint32_t pk[4], hit = 0u;
                    rs_scaled_pk_from_vec(row, v[k], pk);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t key2 = zn_pk_key(pk[j]);
                        if (rangeA) { const uint32_t d = zn_pk_sub(key2, aLo2); hit |= zn_pk_has_zero(zn_pk_min(d, aSp2) ^ d); }
                        if (rangeB) { const uint32_t d = zn_pk_sub(key2, bLo2); hit |= zn_pk_has_zero(zn_pk_min(d, bSp2) ^ d); }
                    }
                    if (__builtin_expect(hit != 0u, 0)) {
                        uint32_t ma = 0u, mb = 0u;
                        classes(v[k], e0, ma, mb);
                        if (ma && wantA) atomicAdd(&sh.tileA[t0 + k], __popc(ma));     // (members of a cut group are few: an add per lane that holds one)
                        if (mb && wantB) atomicAdd(&sh.tileB[t0 + k], __popc(mb));
                    }
                }
                if (early) {
                    __syncthreads();
#pragma unroll
                    for (int k = 0; k < NB; ++k) if (t0 + k < ntiles) { cumA += sh.tileA[t0 + k]; cumB += sh.tileB[t0 + k]; }
                    if ((!wantA || cumA >= need1) && (!wantB || cumB >= c2)) break;   // (workgroup-uniform)
                }
                if constexpr (FAST) {
#pragma unroll
                    for (int k = 0; k < NB; ++k) v[k] = vn[k];
                }
            }
        }
        __syncthreads();
        FH_STAMP(6);
        // the id of the c-th (1-based) member, in id order, of: class A ids (use_a, up to id a_max) and class B ids (use_b)
        auto nth = [&](long long c, bool use_a, int64_t a_max, bool use_b) -> int64_t {
            if (tid == 0) {
                long long before = 0;
                sh.bi[1] = -1;
                for (int tl = 0; tl < (int)ntiles; ++tl) {
                    long long n = use_b ? sh.tileB[tl] : 0;
                    if (use_a) {
                        const int64_t t_lo = (int64_t)tl * TILE;
                        int64_t t_hi = t_lo + TILE - 1;
                        if (t_hi > V - 1) t_hi = V - 1;
                        n += a_max >= t_hi ? sh.tileA[tl] : a_max >= t_lo ? sh.bl[2] : 0;
                    }
                    if (before < c && c <= before + n) { sh.bi[1] = tl; sh.bl[0] = c - before; break; }
                    before += n;
                }
            }
            __syncthreads();
            const int tl = sh.bi[1];
            if (tl < 0) { __syncthreads(); return V - 1; }
            const long long rank = sh.bl[0];
            const int64_t e0 = (int64_t)tl * TILE + (int64_t)tid * EPV;
            uint32_t ma = 0u, mb = 0u;
            if (e0 < V) classes(rs_load_vec<JF_BF16>(row, e0), e0, ma, mb);
            uint32_t mm = use_b ? mb : 0u;
            if (use_a) for (int j = 0; j < EPV; ++j) if ((ma >> j & 1u) && e0 + j <= a_max) mm |= 1u << j;
            long long mine = __popc(mm), ci, ce;
            double z = 0.0, zi, ze;
            fh_scan(sh, mine, z, ci, zi, ce, ze);
            if (tid == 0) sh.bl[1] = V - 1;
            __syncthreads();
            if (ce < rank && rank <= ci) {
                long long seen = ce;
                for (int j = 0; j < EPV; ++j) if ((mm >> j & 1u) && ++seen == rank) { sh.bl[1] = e0 + j; break; }
            }
            __syncthreads();
            const int64_t hit = sh.bl[1];
            __syncthreads();
            return hit;
        };
        int64_t tie1 = V - 1;
        if (tie1_needed && need1 > 0) { tie1 = nth(need1, true, V - 1, false); rec.tie1 = (int32_t)tie1; }
        if (tie2_needed) {
            if (grp1_in2) {
                // class A ids up to tie1 are members too: how many of them the tile that holds tie1 has (ids <= tie1), for nth()
                const int64_t tl = tie1 / TILE, e0 = tl * TILE + (int64_t)tid * EPV;
                uint32_t ma = 0u, mb = 0u;
                if (e0 < V) classes(rs_load_vec<JF_BF16>(row, e0), e0, ma, mb);
                long long mine = 0, ci, ce; double z = 0.0, zi, ze;
                for (int j = 0; j < EPV; ++j) mine += ((ma >> j & 1u) && e0 + j <= tie1) ? 1 : 0;
                fh_scan(sh, mine, z, ci, zi, ce, ze);
                if (tid == 0) sh.bl[2] = mine;
                __syncthreads();
            }
            rec.tie2 = (int32_t)nth(c2, grp1_in2, tie1, true);
        }
    }
    FH_STAMP(7);
    // ---- nothing below this scaled logit is kept (the steps skip the exps below it), and the drafted id's final probability
    {
        int klow = k_on && v1 ? ka1 : 0;
        if (p_on && !all2 && kb2 >= 0 && !grp1_in2) klow = ka2 > klow ? ka2 : klow;
        rec.x_keep = klow > 0 && klow < 0x7FFFFFFF ? fh_value((uint32_t)klow) : -INFINITY;
    }
    float pd = 0.f;
    if (tid == 0) {
        const int64_t tok = draft_next[r];
        if (tok >= 0 && tok < V) {
            const float xs = rs_scaled<JF_BF16>(load_f<JF_BF16>(row.p, tok), row.t, row.inv_t, row.unit_t, row.fast);
            const float p = xs >= M + (float)RS_EXP_CUT ? rs_bf16_of_f64(rs_e64(xs, (double)M, sh.tab) * invS) : 0.f;
            pd = rs_filter_apply<JF_BF16>(rec, p, tok);
        }
    }
    return pd;
}

constexpr uint32_t FH_RETRY = 0x80000000u;        // (internal, in jf_rs_filter_row::flags between the two launches) the row is left to rs_filter_hist_kernel
__device__ __forceinline__ void fh_hist_row(FhShared &sh, unsigned char *fh_dyn, int64_t r, const void *logits, int64_t V, int64_t row_stride,
                                            const int64_t *draft_next, float t, int top_k, double top_p, jf_rs_filter_row *filt, float *p_draft,
                                            float *row_max, float *row_sumexp) {
    uint32_t *hist = (uint32_t *)fh_dyn;
    uint32_t *occ = hist + FH_WORDS;                                        // [FH_ITEMS] the occupied patterns, largest value first
    uint16_t *pc = (uint16_t *)(occ + FH_ITEMS);                            // [FH_ITEMS] their probabilities (bf16 bits)
    const int tid = threadIdx.x;
    const float M = row_max[r];
    jf_rs_filter_row rec;
    rec.sum = 0.0; rec.row_max = M; rec.x_keep = -INFINITY; rec.cut1 = 0u; rec.tie1 = (int32_t)V - 1; rec.s1 = 1.f; rec.cut2 = 0u; rec.tie2 = (int32_t)V - 1; rec.s2 = 1.f;
    const bool k_on = top_k > 0 && (int64_t)top_k < V, p_on = top_p > 0.0 && top_p < 1.0;
    rec.flags = (k_on ? JF_RS_FILT_TOPK : 0u) | (p_on ? JF_RS_FILT_TOPP : 0u); rec.rsv = 0u;
    const bool finite = (__float_as_uint(M) & 0x7F800000u) != 0x7F800000u;
    auto finish = [&](float pd) {
        if (tid == 0) { filt[r] = rec; p_draft[r] = pd; row_max[r] = INFINITY; row_sumexp[r] = RS_PROB_ROW; }
    };
    if (!finite) { finish(0.f); return; }                                    // NaN / inf logits: the row filters to zeros (sum = 0)
    const RsRow row = rs_make_row<JF_BF16>(logits, r, V, row_stride, t, M, 1.f);
    FH_STAMP(0);
    rs_load_tab(sh.tab);
    // ---- 1. the counts (plain adds)
    for (int w = tid; w < FH_WORDS; w += FH_TPB) hist[w] = 0u;
    if (tid == 0) { sh.n_ovf = 0; sh.ovf_seen = 0; }
    if (tid < FH_OVF) sh.ovf_cnt[tid] = 0u;
    __syncthreads();
    FH_STAMP(1);
    const uint32_t counted = fh_count_row_fast(hist, row);
    FH_STAMP(2);
    // ---- 2. the occupied patterns as a list, largest value first; the checksum that notices a wrapped 16-bit counter
    const uint32_t kbase = 65536u - 64u * (uint32_t)(tid + 1);
    long long n_mine = 0, in_hist = 0;
    unsigned long long occ_mask = 0ull;                                      // bit b: this thread's pattern kbase + b is occupied
#pragma unroll 4
    for (int i = 0; i < 32; ++i) {
        const uint32_t wi = (uint32_t)(i + tid) & 31u;                       // (rotated: conflict-free)
        const uint32_t w = hist[(kbase >> 1) + wi];
        n_mine += ((w & 0xFFFFu) ? 1 : 0) + ((w >> 16) ? 1 : 0);
        in_hist += (w & 0xFFFFu) + (w >> 16);
        occ_mask |= ((w & 0xFFFFu) ? 1ull : 0ull) << (2u * wi) | ((w >> 16) ? 2ull : 0ull) << (2u * wi);
    }
    long long tot_counted = counted, ci, ce;
    double d0 = 0.0, d1 = 0.0, di, de;
    fh_reduce(sh, tot_counted, d0);
    fh_reduce(sh, in_hist, d1);
    long long n_occ = n_mine;
    fh_scan(sh, n_occ, d0, ci, di, ce, de);
    const bool wrapped = in_hist != tot_counted;                             // (workgroup-uniform)
    float pd;
    if (!wrapped && n_occ <= FH_ITEMS) {
        // this thread's patterns behind everything above them, largest first: a pattern's place is the number of occupied ones above
        // it in the thread's block (the mask), so the words can be read in the rotated order again
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {
            const uint32_t wi = (uint32_t)(i + tid) & 31u;
            const uint32_t w = hist[(kbase >> 1) + wi];
            if (w == 0u) continue;
            const uint32_t b0 = 2u * wi;
            if (w & 0xFFFFu) occ[(int)ce + __popcll(b0 + 1u < 64u ? occ_mask >> (b0 + 1u) : 0ull)] = ((kbase + b0) << 16) | (w & 0xFFFFu);
            if (w >> 16) occ[(int)ce + __popcll(b0 + 2u < 64u ? occ_mask >> (b0 + 2u) : 0ull)] = ((kbase + b0 + 1u) << 16) | (w >> 16);
        }
        __syncthreads();
        pd = fh_solve<true>(sh, hist, occ, pc, (int)n_occ, row, r, V, draft_next, top_k, top_p, k_on, p_on, rec);
    } else {
        if (wrapped) {                                                       // count again, noticing the wraps; then once more with side counters for those patterns
            for (int w = tid; w < FH_WORDS; w += FH_TPB) hist[w] = 0u;
            __syncthreads();
            fh_count_row<false>(sh, hist, row);
            if (tid == 0) {                                                  // the distinct keys seen wrapping (<= 8 real ones + neighbours seen mid-carry)
                int n = sh.ovf_seen < FH_OVF ? sh.ovf_seen : FH_OVF, m = 0;
                for (int i = 0; i < n; ++i) { bool dup = false; for (int q = 0; q < m; ++q) dup |= sh.ovf_key[q] == sh.ovf_key[i]; if (!dup) sh.ovf_key[m++] = sh.ovf_key[i]; }
                sh.n_ovf = m; sh.ovf_seen = 0;
            }
            for (int w = tid; w < FH_WORDS; w += FH_TPB) hist[w] = 0u;
            __syncthreads();
            fh_count_row<true>(sh, hist, row);
        }
        pd = fh_solve<false>(sh, hist, occ, pc, 0, row, r, V, draft_next, top_k, top_p, k_on, p_on, rec);
    }
    finish(pd);
}

// retry_only: the rows rs_filter_zone_kernel left (FH_RETRY), a workgroup walks the rows blockIdx.x, + gridDim.x, ...; else every row
__global__ __launch_bounds__(FH_TPB, 1) void rs_filter_hist_kernel(const void *logits, int64_t R, int64_t V, int64_t row_stride, const int64_t *draft_next,
                                                                    float t, int top_k, double top_p, jf_rs_filter_row *filt, float *p_draft,
                                                                    float *row_max, float *row_sumexp, int retry_only) {
    __shared__ FhShared sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char fh_dyn[];
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
        if (retry_only && !(filt[r].flags & FH_RETRY)) continue;            // (workgroup-uniform)
        __syncthreads();
        fh_hist_row(sh, fh_dyn, r, logits, V, row_stride, draft_next, t, top_k, top_p, filt, p_draft, row_max, row_sumexp);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// The same record from 32 KB of counters instead of 128 KB, so that a CU holds THREE rows at once and one row's solving overlaps
// the others' streaming.  Between M - 104 and M the scaled logits of a row span ~34 000 bf16 patterns, but nearly all of those are
// the patterns of magnitudes below 2^-16 on either side of zero, which a row of 152 064 logits visits a handful of times: the table
// holds the patterns with 2^-16 <= |x| < 2^16 (32 binades x 128 x 2 signs = 8 192 32-bit counters, the two signs of a magnitude
// side by side), a row's smaller magnitudes go to a list of 64 keys (sorted and counted by 64 threads), and a row that does not
// fit — more than 64 such ids, a magnitude of 2^16 or more that carries mass, a pattern more than 65 535 times, more than 4 096
// occupied patterns — is left to rs_filter_hist_kernel (FH_RETRY), launched behind this one over the flagged rows only.  From the
// list of occupied patterns on, the two kernels share fh_solve.
//
// The count pass is what a row costs (the first version spent ~30 instructions and five branches per element on it: 43 us per row
// with three rows per CU, all of it VALU issue).  Now it works on the PACKED bf16 patterns, two ids per register, without a branch
// per element: rotate each half left by one (magnitude << 1 | sign), subtract the table's first magnitude — that IS the counter's
// index when it is below 8 192 — OR the four words of a 16-byte vector together and look at the top bits once per vector: a vector
// with an id outside the table (one in ~10^4) takes the slow path per element, every other one is eight plain LDS adds.  Nothing is
// compared with M - 104 per element: patterns below it are dropped when the list is built (the slot's value says so), and since the
// counters are 32 bits wide there is no checksum for wrapped counters either.
// ------------------------------------------------------------------------------------------------
#ifndef JF_ZN_WAVES
#define JF_ZN_WAVES 6                       // waves per SIMD asked of the compiler: 6 = three rows per CU (<= 85 VGPRs), 4 = two
#endif
constexpr int ZN_TPB = 512, ZN_SLOTS = 8192, ZN_TINY = 64, ZN_EXP_LO = 111, ZN_EXP_HI = 142;
constexpr uint32_t ZN_LO16 = (uint32_t)ZN_EXP_LO << 7;                       // first magnitude pattern of the table
constexpr unsigned ZN_LDS = (ZN_SLOTS + 4) * 4;                               // the counters; the list of occupied patterns and their probabilities re-use them
static_assert(FH_ITEMS * 4 + FH_ITEMS * 2 <= ZN_SLOTS * 4, "occ + pc must fit the counters they replace");
__global__ __launch_bounds__(ZN_TPB, JF_ZN_WAVES) void rs_filter_zone_kernel(const void *logits, int64_t R, int64_t V, int64_t row_stride, const int64_t *draft_next,
                                                                    float t, int top_k, double top_p, jf_rs_filter_row *filt, float *p_draft,
                                                                    float *row_max, float *row_sumexp) {
    constexpr int EPV = 8, NB = 4;                                          // (four loads in flight per thread x three workgroups per CU)
    __shared__ FhShared sh;
    __shared__ uint32_t s_tiny[ZN_TINY];
    __shared__ int s_ntiny, s_retry, s_ndist;
    extern __shared__ __attribute__((aligned(16))) unsigned char zn_dyn[];
    uint32_t *table = (uint32_t *)zn_dyn;                                   // [ZN_SLOTS] counter of magnitude LO16 + (i >> 1), sign i & 1
    uint32_t *occ = table;                                                   // [FH_ITEMS] the list, once every counter has been read
    uint16_t *pc = (uint16_t *)(occ + FH_ITEMS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r = blockIdx.x;
    const float M = row_max[r];
    jf_rs_filter_row rec;
    rec.sum = 0.0; rec.row_max = M; rec.x_keep = -INFINITY; rec.cut1 = 0u; rec.tie1 = (int32_t)V - 1; rec.s1 = 1.f; rec.cut2 = 0u; rec.tie2 = (int32_t)V - 1; rec.s2 = 1.f;
    const bool k_on = top_k > 0 && (int64_t)top_k < V, p_on = top_p > 0.0 && top_p < 1.0;
    rec.flags = (k_on ? JF_RS_FILT_TOPK : 0u) | (p_on ? JF_RS_FILT_TOPP : 0u); rec.rsv = 0u;
    if ((__float_as_uint(M) & 0x7F800000u) == 0x7F800000u) {               // NaN / inf logits: the row filters to zeros (sum = 0)
        if (tid == 0) { filt[r] = rec; p_draft[r] = 0.f; row_max[r] = INFINITY; row_sumexp[r] = RS_PROB_ROW; }
        return;
    }
    const RsRow row = rs_make_row<JF_BF16>(logits, r, V, row_stride, t, M, 1.f);
    FH_STAMP(8);
    rs_load_tab(sh.tab);
    for (int w = tid; w < ZN_SLOTS; w += ZN_TPB) table[w] = 0u;
    if (tid == 0) { sh.n_ovf = 0; s_ntiny = 0; s_retry = 0; s_ndist = 0; }
    __syncthreads();
    FH_STAMP(9);
    // ---- 1. the counts
    const float mcut = M + (float)RS_EXP_CUT;
    {
        const int64_t step = (int64_t)NB * ZN_TPB * EPV;
        u32x4 v[NB], vn[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) { const int64_t e0 = (int64_t)tid * EPV + (int64_t)k * ZN_TPB * EPV; if (e0 < V) v[k] = rs_load_vec<JF_BF16>(row, e0); }
        for (int64_t b0 = (int64_t)tid * EPV; b0 < V; b0 += step) {
#pragma unroll
            for (int k = 0; k < NB; ++k) { const int64_t e0 = b0 + step + (int64_t)k * ZN_TPB * EPV; if (e0 < V) vn[k] = rs_load_vec<JF_BF16>(row, e0); }   // the next round's loads, in front of this round's arithmetic
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int64_t e0 = b0 + (int64_t)k * ZN_TPB * EPV;
                if (e0 >= V) continue;
                uint32_t pk[4], d[4];
                rs_scaled_pk_from_vec(row, v[k], pk);
#pragma unroll
                for (int j = 0; j < 4; ++j) d[j] = zn_pk_sub(zn_pk_rotl1(pk[j]), (2u * ZN_LO16) * 0x00010001u);
                if (__builtin_expect(((d[0] | d[1] | d[2] | d[3]) & 0xE000E000u) == 0u, 1)) {      // all eight ids inside the table
#pragma unroll
                    for (int j = 0; j < 4; ++j) { atomicAdd(&table[d[j] & 0xFFFFu], 1u); atomicAdd(&table[d[j] >> 16], 1u); }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t h = (j & 1) ? (pk[j >> 1] >> 16) : (pk[j >> 1] & 0xFFFFu), idx = (j & 1) ? (d[j >> 1] >> 16) : (d[j >> 1] & 0xFFFFu);
                        if (idx < (uint32_t)ZN_SLOTS) { atomicAdd(&table[idx], 1u); continue; }
                        if (!(__uint_as_float(h << 16) >= mcut)) continue;     // no mass (and the slots beyond V, which hold -inf)
                        if ((h & 0x7FFFu) < ZN_LO16) {                         // a magnitude below 2^-16: the list
                            const int at = atomicAdd(&s_ntiny, 1);
                            if (at < ZN_TINY) s_tiny[at] = fh_key(h);
                        } else s_retry = 1;                                      // a magnitude of 2^16 or more
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) v[k] = vn[k];
        }
    }
    __syncthreads();
    FH_STAMP(10);
#if defined(JF_EXP_ZN_STOP) && JF_EXP_ZN_STOP == 1                              // (experiment: the count pass alone)
    if (tid == 0) { filt[r] = rec; p_draft[r] = 0.f; row_max[r] = INFINITY; row_sumexp[r] = RS_PROB_ROW; }
    return;
#endif
    // ---- 2. the list of occupied patterns, largest value first: positive magnitudes downwards, the small magnitudes, negative magnitudes upwards.
    //         List place L < 4096 is the positive magnitude LO16 + 4095 - L, L >= 4096 the negative magnitude LO16 + L - 4096; wavefront w takes
    //         the places [1024 w, 1024 w + 1024), a lane the places 64 j + lane of them (16 reads, adjacent lanes two words apart).
    const int n_tiny = s_ntiny < ZN_TINY ? s_ntiny : ZN_TINY;
    uint32_t my_tiny = 0u, my_cnt = 0u;
    int my_pos = -1;
    if (tid < n_tiny) {                                                      // 64 threads: distinct keys, their counts, their places (largest first)
        my_tiny = s_tiny[tid];
        bool first = true;
        for (int j = 0; j < n_tiny; ++j) { const uint32_t kj = s_tiny[j]; my_cnt += kj == my_tiny ? 1u : 0u; first &= !(kj == my_tiny && j < tid); }
        if (first) {
            my_pos = 0;
            for (int j = 0; j < n_tiny; ++j) {
                const uint32_t kj = s_tiny[j];
                bool jfirst = kj > my_tiny;
                for (int q = 0; q < j && jfirst; ++q) jfirst = s_tiny[q] != kj;
                my_pos += jfirst ? 1 : 0;
            }
            atomicAdd(&s_ndist, 1);
        }
    }
    uint32_t cnt16[16];
    int before[16], mine_n = 0;
    bool wide = false;
    const bool neg = wave >= 4;                                              // (wavefront-uniform)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t L = (uint32_t)wave * 1024u + 64u * j + lane;
        const uint32_t off = neg ? L - 4096u : 4095u - L, h = (neg ? 0x8000u : 0u) | (ZN_LO16 + off);
        uint32_t c = table[2u * off + (neg ? 1u : 0u)];
        if (!(__uint_as_float(h << 16) >= mcut)) c = 0u;                     // patterns without mass were counted, and are dropped here
        wide |= c > 0xFFFFu;
        cnt16[j] = c;
        const unsigned long long m = __ballot(c != 0u);
        before[j] = mine_n + __popcll(m & ((1ull << lane) - 1ull));          // (mine_n: the wavefront's occupied places of the rounds in front — uniform)
        mine_n += __popcll(m);
    }
    if (wide) s_retry = 1;                                                   // a pattern more than 65 535 times: the list's items hold 16-bit counts
    __syncthreads();                                                         // (every counter has been read: the list may overwrite them)
    if (lane == 0) sh.scan[wave] = mine_n;
    __syncthreads();
    int base = 0, n_pos = 0, n_table = 0;
#pragma unroll
    for (int w = 0; w < ZN_TPB / 64; ++w) { const int n = sh.scan[w]; if (w < wave) base += n; if (w < 4) n_pos += n; n_table += n; }
    const int n_dist = s_ndist;
    const int n_occ = n_table + n_dist;
    if (s_retry || s_ntiny > ZN_TINY || n_occ > FH_ITEMS) {                 // (workgroup-uniform) left to the other kernel
        if (tid == 0) filt[r].flags = FH_RETRY;
        return;
    }
    if (neg) base += n_dist;                                                 // (the negative values: behind the small magnitudes)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (!cnt16[j]) continue;
        const uint32_t L = (uint32_t)wave * 1024u + 64u * j + lane;
        const uint32_t off = neg ? L - 4096u : 4095u - L, h = (neg ? 0x8000u : 0u) | (ZN_LO16 + off);
        occ[base + before[j]] = (fh_key(h) << 16) | cnt16[j];
    }
    if (my_pos >= 0) occ[n_pos + my_pos] = (my_tiny << 16) | my_cnt;
    __syncthreads();
    FH_STAMP(11);
    const float pd = fh_solve<true>(sh, nullptr, occ, pc, n_occ, row, r, V, draft_next, top_k, top_p, k_on, p_on, rec);
    if (tid == 0) { filt[r] = rec; p_draft[r] = pd; row_max[r] = INFINITY; row_sumexp[r] = RS_PROB_ROW; }
}

// ------------------------------------------------------------------------------------------------
// (a19, round 5) top-k / top-p filtering of the target distribution — _apply_top_k + _apply_top_p of _build_target_probs
// (JDN:72-123; the reference reads both with getattr: they exist only on request objects a caller planted them on).
//
// jf_rs_filter turns R logits rows into R rows of the FILTERED, RENORMALISED probabilities in the dtype of the logits — the
// tensor the reference's `probs` is — and marks the rows as probability rows (row_max = +inf, row_sumexp = RS_PROB_ROW): the
// steps then read an element as its own probability (RsRow::prob) through the code path rows without finite statistics
// already take, accept tests included (p_draft is the final value, no rounding candidates).  One workgroup per row:
//   1. the row's float64 sum and every exactly rounded probability (the definition of "Exact probabilities" above) -> out row;
//   2. top-k: the k-th largest value by bisection over the value's bit pattern (a count pass per step: the out row is
//      L2-resident), the ids above it, the first (k - that many) ids AT it in index order; their exact sum rounded once;
//      q = p / max(sum, 1e-12) in torch's dtype arithmetic (float32 quotient, rounded again for bf16);
//   3. top-p on the result: values in descending order, equal values by index; the cumulative sum — exact, rounded once to the
//      dtype per position — is compared with top_p cast to the dtype; bisection finds the lowest value whose whole group is
//      kept, the group below it keeps the ids whose own cumulative sum still passes (binary search over the count), at least
//      one id is kept; renormalised like 2.
// What torch leaves to its kernels is DEFINED (oracle/jacobi_oracle.py filter_probs_row): equal values are
// ordered by index (torch.topk / torch.sort pick by kernel), sums are exact sums rounded once (torch accumulates in float32
// in kernel order).  Reproduces the reference's recorded tensors bit for bit in bf16 wherever a cut does not fall inside a
// group of equal values, and its kept sets (values within a few ulps) in float32 (tests/golden/filter_vectors.json).
// Every reduction is formed in a fixed order (a thread's elements in index order, 64-lane scans, four wavefront partials
// in order): the result does not depend on scheduling.
// ------------------------------------------------------------------------------------------------
template <int DT> __device__ __forceinline__ uint32_t flt_key(const void *row, int64_t i) {      // non-negative value -> its bits (order-preserving)
    if constexpr (DT == JF_F32) return ((const uint32_t *)row)[i];
    else return (uint32_t)((const uint16_t *)row)[i] << 16;
}
template <int DT> __device__ __forceinline__ void flt_store(void *row, int64_t i, float v) {
    if constexpr (DT == JF_F32) ((float *)row)[i] = v;
    else ((uint16_t *)row)[i] = (uint16_t)(__float_as_uint(v) >> 16);
}
constexpr int FLT_TPB = 512;                                             // threads of a row's workgroup: 8 wavefronts (with 256 the row's float64 exps
                                                                         // and load round trips had one wavefront per SIMD to hide behind: 350-450 us per row)
constexpr int FLT_NW = FLT_TPB / 64;
struct FltShared {
    double tab[64], dsum[FLT_NW];
    unsigned long long cnt[FLT_NW];
    uint32_t umax[FLT_NW];
    int scan[FLT_TPB];
};
// workgroup totals in a fixed order; every thread gets them
__device__ __forceinline__ void flt_reduce(FltShared &sh, unsigned long long &cnt, double &sum) {
    const int tid = threadIdx.x;
    sum = wave_sum_f64(sum);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) { sh.cnt[tid >> 6] = cnt; sh.dsum[tid >> 6] = sum; }
    __syncthreads();
    cnt = 0ull; sum = 0.0;
#pragma unroll
    for (int w = 0; w < FLT_NW; ++w) { cnt += sh.cnt[w]; sum += sh.dsum[w]; }      // wavefront order: fixed
}
// float64 sum of exp(xs - M) over the row by the FLT_TPB threads (rs_row_s64_wg is the 256-thread one of the steps)
template <int DT>
__device__ __forceinline__ double flt_row_s64(FltShared &sh, const RsRow &row) {
    constexpr int EPV = Elem<DT>::EPV, NB = 4;
    double acc = 0.0;
    for (int64_t b0 = (int64_t)threadIdx.x * EPV; b0 < row.V; b0 += (int64_t)NB * FLT_TPB * EPV) {
        u32x4 v[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) { const int64_t e0 = b0 + (int64_t)k * FLT_TPB * EPV; if (e0 < row.V) v[k] = rs_load_vec<DT>(row, e0); }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int64_t e0 = b0 + (int64_t)k * FLT_TPB * EPV;
            if (e0 >= row.V) continue;
            float xs[EPV];
            rs_scaled_from_vec<DT>(row, v[k], xs);
#pragma unroll
            for (int j = 0; j < EPV; ++j) acc += rs_e64(xs[j], (double)row.M, sh.tab);
        }
    }
    unsigned long long none = 0ull;
    flt_reduce(sh, none, acc);
    return acc;
}
// every element of a probability row as (index, key), a thread's elements in index order: 16-byte vectors, eight loads in
// flight per thread (a pass over the L2-resident row is a chain of load round trips: with four 2-byte loads in flight a pass of
// 152 064 ids took ~120 us, this way ~15)
template <int DT, class F>
__device__ __forceinline__ void flt_for_each(const void *row, int64_t V, F f) {
    constexpr int EPV = Elem<DT>::EPV, NB = 8;
    RsRow rr;
    rr.p = row; rr.V = V; rr.vec = (((uintptr_t)row) % 16) == 0;
    for (int64_t b0 = (int64_t)threadIdx.x * EPV; b0 < V; b0 += (int64_t)NB * FLT_TPB * EPV) {
        u32x4 v[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) { const int64_t e0 = b0 + (int64_t)k * FLT_TPB * EPV; if (e0 < V) v[k] = rs_load_vec<DT>(rr, e0); }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int64_t e0 = b0 + (int64_t)k * FLT_TPB * EPV;
            if (e0 >= V) continue;
            const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                uint32_t key;
                if constexpr (DT == JF_F32) key = w[j]; else key = (j & 1) ? (w[j >> 1] & 0xFFFF0000u) : (w[j >> 1] << 16);
                if (e0 + j < V) f(e0 + j, key);
            }
        }
    }
}
// count and exact sum of the row's elements with key >= lo (and, hi_excl != 0, key < hi_excl)
template <int DT>
__device__ __forceinline__ void flt_count_sum(FltShared &sh, const void *row, int64_t V, uint32_t lo, uint32_t hi_excl, unsigned long long &cnt, double &sum) {
    unsigned long long c = 0ull;
    double a = 0.0;
    flt_for_each<DT>(row, V, [&](int64_t, uint32_t k) {
        const bool in = k >= lo && (hi_excl == 0u || k < hi_excl);
        c += in ? 1ull : 0ull;
        a += in ? (double)__uint_as_float(k) : 0.0;
    });
    cnt = c; sum = a;
    flt_reduce(sh, cnt, sum);
}
// ---- bf16 rows: a COUNT histogram of the row's values in LDS (a bf16 probability is one of 16 257 bit patterns; 64 KB of
// counters) stands in for the row in every count / sum / maximum below: count x value is exact in float64, so the sums are the
// sums of the elements, formed from <= 16 384 terms instead of 152 064 loads — a bisection step costs ~1 us on LDS instead of a pass
// over the row (a 0.6 GB tensor at 64 x 31 rows: ~17 passes from HBM took 3.3 ms per call).  The row itself is read four to six
// times in all: float64 sum, probabilities (+ histogram), one renormalising pass per active stage (+ histogram of its result), and
// a tie pass where a cut falls inside a group of equal values.
constexpr int FLT_BINS = 16384;                                            // bins of the 16-bit patterns 0 .. 0x3FFF (1.0 = 0x3F80)
__device__ __forceinline__ void flt_hist_zero(uint32_t *hist) {
    for (int b = threadIdx.x; b < FLT_BINS; b += FLT_TPB) hist[b] = 0u;
    __syncthreads();
}
__device__ __forceinline__ void flt_hist_count_sum(FltShared &sh, const uint32_t *hist, uint32_t lo, uint32_t hi_excl, unsigned long long &cnt, double &sum) {
    const uint32_t blo = lo >> 16, bhi = hi_excl ? (hi_excl >> 16) : (uint32_t)FLT_BINS;      // keys are multiples of 0x10000
    unsigned long long c = 0ull;
    double a = 0.0;
#pragma unroll 8
    for (int b = (int)threadIdx.x; b < FLT_BINS; b += FLT_TPB) {       // lane-interleaved bins: consecutive lanes, consecutive LDS banks (a thread
        const uint32_t n = ((uint32_t)b >= blo && (uint32_t)b < bhi) ? hist[b] : 0u;   // owning 64 ADJACENT bins puts all 64 lanes on one bank)
        c += n;
        a += (double)n * (double)__uint_as_float((uint32_t)b << 16);
    }
    cnt = c; sum = a;
    flt_reduce(sh, cnt, sum);
}
// ---- float32 rows: three LEVELS of counters.  The 32-bit pattern of a probability splits into a BIN (its top 15 bits: 8 129 of them)
// and 17 low bits; a 64-bit LDS word per bin holds (count << 35) | (sum of the low parts), from which the bin's exact float64 sum
// follows — count x (the bin's first value) + (sum of low parts) x (the bin's spacing): an integer multiple of the spacing below
// 2^43, exact — so any count / sum whose limits fall on bin edges is ~1 us of LDS work as for bf16.  A limit INSIDE a bin is served
// by two sub-tables built by one pass over the row each: FOCUS A holds bits 16:8 of the ids of one bin (512 entries), FOCUS B the
// last 8 bits of the ids under one 24-bit prefix.  The bisections run coarse to fine (bin edges, then 256-aligned keys inside the
// bin found, then keys), so each builds A once and B once: ~8 passes over the row per call instead of a pass per bisection step
// (~70).  64 KB of bins + 5 KB of sub-tables: two workgroups per CU, like bf16.  Rows of more than 2^18 ids keep the
// pass-per-step variant (the low-part sums would carry into the counts).
constexpr int FLT_F32_SHIFT = 17, FLT_F32_BINS = 8192, FLT_SUBA = 512, FLT_SUBB = 256;
constexpr unsigned FLT_LDS_F32 = FLT_F32_BINS * 8 + FLT_SUBA * 8 + FLT_SUBB * 4;
constexpr uint32_t FLT_F32_LOW = (1u << FLT_F32_SHIFT) - 1u;
constexpr unsigned long long FLT_LOW_MASK = (1ull << 35) - 1ull;
static_assert(FLT_TPB >= FLT_SUBA && (0x3F800000u >> FLT_F32_SHIFT) < (unsigned)FLT_F32_BINS - 1u, "a thread per sub-table entry; 1.0 has a bin");
__device__ __forceinline__ double flt_bin_ulp(uint32_t bin) {             // spacing of the float32 values of a bin (subnormals: 2^-149)
    uint32_t e = bin >> (23 - FLT_F32_SHIFT);
    e = e ? e : 1u;
    return __longlong_as_double((long long)(e + 873u) << 52);              // 2^(e - 150)
}
struct FltF32 {
    unsigned long long *hist, *subA;
    uint32_t *subB;
    uint32_t focA, focB;                                                   // the bin / the 24-bit prefix the sub-tables hold (0xFFFFFFFF: none)
    uint32_t l1_from;                                                      // cached: count and sum of the bins >= l1_from
    unsigned long long l1_cnt;
    double l1_sum;
    __device__ __forceinline__ void reset() { focA = focB = l1_from = 0xFFFFFFFFu; }
};
__device__ __forceinline__ void flt_hist64_zero(unsigned long long *hist) {
    for (int b = threadIdx.x; b < FLT_F32_BINS; b += FLT_TPB) hist[b] = 0ull;
    __syncthreads();
}
__device__ __forceinline__ void flt_hist64_add(unsigned long long *hist, uint32_t key) {       // (a probability is <= 1.0: the clamp only keeps a
    const uint32_t b = key >> FLT_F32_SHIFT;                                                     // value that cannot occur from indexing outside LDS)
    atomicAdd(hist + (b < (uint32_t)FLT_F32_BINS ? b : (uint32_t)FLT_F32_BINS - 1u), (1ull << 35) | (unsigned long long)(key & FLT_F32_LOW));
}
__device__ __forceinline__ void flt_hist_add(uint32_t *hist, uint32_t key) {                    // bf16: one counter per 16-bit pattern
    const uint32_t b = key >> 16;
    atomicAdd(hist + (b < (uint32_t)FLT_BINS ? b : (uint32_t)FLT_BINS - 1u), 1u);
}
// count and exact sum of the row's positive elements with key >= `key` (every thread gets them)
__device__ __forceinline__ void flt_f32_ge(FltShared &sh, FltF32 &f, const void *row, int64_t V, uint32_t key, unsigned long long &cnt, double &sum) {
    const int tid = threadIdx.x;
    const uint32_t B = key >> FLT_F32_SHIFT, a = (key & FLT_F32_LOW) >> 8, c = key & 0xFFu;
    const bool pa = (key & FLT_F32_LOW) != 0u, pb = c != 0u;
    const uint32_t from = pa ? B + 1u : B;
    if (pa && f.focA != B) {
        __syncthreads();
        if (tid < FLT_SUBA) f.subA[tid] = 0ull;
        __syncthreads();
        flt_for_each<JF_F32>(row, V, [&](int64_t, uint32_t k) { if ((k >> FLT_F32_SHIFT) == B && k != 0u) atomicAdd(&f.subA[(k & FLT_F32_LOW) >> 8], (1ull << 35) | (unsigned long long)(k & 0xFFu)); });
        __syncthreads();
        f.focA = B;
    }
    if (pb && f.focB != (key >> 8)) {
        const uint32_t P = key >> 8;
        __syncthreads();
        if (tid < FLT_SUBB) f.subB[tid] = 0u;
        __syncthreads();
        flt_for_each<JF_F32>(row, V, [&](int64_t, uint32_t k) { if ((k >> 8) == P && k != 0u) atomicAdd(&f.subB[k & 0xFFu], 1u); });
        __syncthreads();
        f.focB = P;
    }
    if (f.l1_from != from) {
        unsigned long long n = 0ull;
        double s = 0.0;
#pragma unroll 8
        for (int b = tid; b < FLT_F32_BINS; b += FLT_TPB) {
            const unsigned long long w = (uint32_t)b >= from ? f.hist[b] : 0ull;
            const unsigned long long cn = w >> 35;
            n += cn;
            s += (double)cn * (double)__uint_as_float((uint32_t)b << FLT_F32_SHIFT) + (double)(w & FLT_LOW_MASK) * flt_bin_ulp((uint32_t)b);
        }
        flt_reduce(sh, n, s);
        f.l1_from = from; f.l1_cnt = n; f.l1_sum = s;
    }
    cnt = f.l1_cnt; sum = f.l1_sum;
    if (pa) {
        unsigned long long n = 0ull;
        double s = 0.0;
        if (tid < FLT_SUBA && (uint32_t)tid >= (pb ? a + 1u : a)) {
            const unsigned long long w = f.subA[tid], cn = w >> 35;
            n = cn;
            s = (double)cn * (double)__uint_as_float((B << FLT_F32_SHIFT) | ((uint32_t)tid << 8)) + (double)(w & FLT_LOW_MASK) * flt_bin_ulp(B);
        }
        if (pb && tid < FLT_SUBB && (uint32_t)tid >= c) {
            const unsigned long long nb = f.subB[tid];
            n += nb; s += (double)nb * (double)__uint_as_float((key & ~0xFFu) | (uint32_t)tid);
        }
        flt_reduce(sh, n, s);
        cnt += n; sum += s;
    }
}
// index of the c-th (1-based) element, in index order, whose key equals `key` (V if there are fewer).  Two steps: the ties are
// counted per TILE of 256 vectors (LDS counters, one vectorised pass), the tile that holds the c-th one is ranked thread by thread
// (a thread's vector of a tile is 16 contiguous bytes).  (The first version walked a contiguous chunk per thread with 2-byte loads:
// 70-140 us per row, the largest single piece of the kernel.)  Rows of more than 256 tiles take that walk.
template <int DT>
__device__ __forceinline__ int64_t flt_nth_equal(FltShared &sh, const void *row, int64_t V, uint32_t key, long long c) {
    constexpr int EPV = Elem<DT>::EPV;
    const int tid = threadIdx.x;
    const int64_t tile_elems = (int64_t)FLT_TPB * EPV, ntiles = (V + tile_elems - 1) / tile_elems;
    __shared__ long long s_hit, s_rank;
    __shared__ int s_tile;
    if (ntiles > FLT_TPB) {
        const int64_t chunk = (V + FLT_TPB - 1) / FLT_TPB, a0 = (int64_t)tid * chunk, b0 = a0 + chunk < V ? a0 + chunk : V;
        int mine = 0;
        for (int64_t i = a0; i < b0; ++i) mine += flt_key<DT>(row, i) == key ? 1 : 0;
        __syncthreads();
        sh.scan[tid] = mine;
        __syncthreads();
        long long before = 0;
        for (int q = 0; q < tid; ++q) before += sh.scan[q];
        if (tid == 0) s_hit = V;
        __syncthreads();
        if (before < c && c <= before + mine) {
            long long seen = before;
            for (int64_t i = a0; i < b0; ++i) if (flt_key<DT>(row, i) == key && ++seen == c) { s_hit = i; break; }
        }
        __syncthreads();
        return (int64_t)s_hit;
    }
    __syncthreads();
    sh.scan[tid] = 0;
    if (tid == 0) { s_hit = V; s_tile = -1; s_rank = 0; }
    __syncthreads();
    flt_for_each<DT>(row, V, [&](int64_t i, uint32_t k) { if (k == key) atomicAdd(&sh.scan[(int)(i / tile_elems)], 1); });
    __syncthreads();
    if (tid == 0) {
        long long before = 0;
        for (int t = 0; t < (int)ntiles; ++t) {
            const long long n = sh.scan[t];
            if (before < c && c <= before + n) { s_tile = t; s_rank = c - before; break; }
            before += n;
        }
    }
    __syncthreads();
    const int tile = s_tile;
    if (tile < 0) return V;
    const long long rank = s_rank;
    const int64_t e0 = (int64_t)tile * tile_elems + (int64_t)tid * EPV;
    int mine = 0;
    for (int j = 0; j < EPV; ++j) mine += (e0 + j < V && flt_key<DT>(row, e0 + j) == key) ? 1 : 0;
    __syncthreads();
    sh.scan[tid] = mine;
    __syncthreads();
    long long before = 0;
    for (int q = 0; q < tid; ++q) before += sh.scan[q];
    if (before < rank && rank <= before + mine) {
        long long seen = before;
        for (int j = 0; j < EPV; ++j) if (e0 + j < V && flt_key<DT>(row, e0 + j) == key && ++seen == rank) { s_hit = e0 + j; break; }
    }
    __syncthreads();
    return (int64_t)s_hit;
}
// renormalise in place: ids with key > thr, and ids AT thr up to index tie_last, keep value / denom; the others become 0
template <int DT, int HM>
__device__ __forceinline__ void flt_renorm(void *row, int64_t V, uint32_t thr, int64_t tie_last, float denom, void *hist /* nullable: rebuilt from the results (HM 1: counts, 2: float32 words) */) {
    constexpr int EPV = Elem<DT>::EPV, NB = 8;
    __syncthreads();
    if (hist) { if constexpr (HM == 2) flt_hist64_zero((unsigned long long *)hist); else flt_hist_zero((uint32_t *)hist); }
    RsRow rr;
    rr.p = row; rr.V = V; rr.vec = (((uintptr_t)row) % 16) == 0;
    for (int64_t b0 = (int64_t)threadIdx.x * EPV; b0 < V; b0 += (int64_t)NB * FLT_TPB * EPV) {      // (a thread rewrites exactly the ids it read)
        u32x4 v[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) { const int64_t e0 = b0 + (int64_t)k * FLT_TPB * EPV; if (e0 < V) v[k] = rs_load_vec<DT>(rr, e0); }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int64_t e0 = b0 + (int64_t)k * FLT_TPB * EPV;
            if (e0 >= V) continue;
            const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
            float q[EPV];
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                uint32_t key;
                if constexpr (DT == JF_F32) key = w[j]; else key = (j & 1) ? (w[j >> 1] & 0xFFFF0000u) : (w[j >> 1] << 16);
                const int64_t i = e0 + j;
                const bool keep = i < V && (key > thr || (key == thr && i <= tie_last));
                q[j] = 0.f;
                if (keep) {                                               // a BRANCH: the IEEE division is ~12 instructions, and after a top-k
                    q[j] = flt_div<DT>(__uint_as_float(key), denom);      // hardly any id is kept (as a select the pass was division-bound: 60 us)
                    // (zeros are not counted: nothing asks for them, and after a top-k nearly every id would hit that one counter)
                    if (hist && q[j] > 0.f) {
                        const uint32_t qb = __float_as_uint(q[j]);
                        if constexpr (HM == 2) flt_hist64_add((unsigned long long *)hist, qb);
                        else flt_hist_add((uint32_t *)hist, qb);
                    }
                }
            }
            if (rr.vec && e0 + EPV <= V) {                                  // one 16-byte store (2-byte stores made this pass 59 us per row)
                u32x4 o;
                if constexpr (DT == JF_F32) o = u32x4{__float_as_uint(q[0]), __float_as_uint(q[1]), __float_as_uint(q[2]), __float_as_uint(q[3])};
                else o = u32x4{(__float_as_uint(q[0]) >> 16) | (__float_as_uint(q[1]) & 0xFFFF0000u), (__float_as_uint(q[2]) >> 16) | (__float_as_uint(q[3]) & 0xFFFF0000u),
                               (__float_as_uint(q[4]) >> 16) | (__float_as_uint(q[5]) & 0xFFFF0000u), (__float_as_uint(q[6]) >> 16) | (__float_as_uint(q[7]) & 0xFFFF0000u)};
                *(u32x4 *)((char *)row + e0 * (DT == JF_F32 ? 4 : 2)) = o;
            } else {
#pragma unroll
                for (int j = 0; j < EPV; ++j) if (e0 + j < V) flt_store<DT>(row, e0 + j, q[j]);
            }
        }
    }
    __syncthreads();
}
#ifdef JF_EXP_FLT_TRACE
__device__ unsigned long long g_flttrace[16];
extern "C" __attribute__((visibility("default"))) int jf_exp_flt_trace(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_flttrace), sizeof(g_flttrace)) == hipSuccess ? 0 : -1;
}
#define FLT_STAMP(k) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) g_flttrace[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FLT_STAMP(k) do { } while (0)
#endif
// HIST: 0 a pass over the row per bisection step, 1 bf16 value counts in LDS, 2 the float32 levels
template <int DT, int HIST>
__global__ __launch_bounds__(FLT_TPB, 4) void rs_filter_kernel(const void *logits,   // (4 waves per SIMD = two workgroups per CU: <= 128 VGPRs)
                                                          int64_t R, int64_t V, int64_t row_stride, const int64_t *draft_next,
                                                         float t, int top_k, double top_p_d, void *probs /* scratch [R, V] */, jf_rs_filter_row *filt,
                                                         float *p_draft, float *row_max, float *row_sumexp) {
    const float top_p = (top_p_d > 0.0 && top_p_d < 1.0) ? 0.5f : 0.f;          // (only "is the stage on" below; the value is cast where it is compared)
    static_assert(HIST == 0 || (HIST == 1 && DT == JF_BF16) || (HIST == 2 && DT == JF_F32), "");
    constexpr int EPV = Elem<DT>::EPV;
    constexpr uint32_t STEP = DT == JF_F32 ? 1u : 0x10000u;                 // distance of two neighbouring values' keys
    constexpr uint32_t FLT_KEY_TOP = HIST == 2 ? 0x3F800000u + (1u << FLT_F32_SHIFT) : 0x3F800000u + STEP;   // a key above 1.0 on the (coarsest) key grid: no probability reaches it
    __shared__ FltShared sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char flt_dyn[];   // (64-bit LDS atomics on it: ADVICE r05)
    uint32_t *hist = HIST == 1 ? (uint32_t *)flt_dyn : nullptr;              // bf16: the row's value counts (FLT_BINS words of dynamic LDS)
    FltF32 f32;
    f32.hist = (unsigned long long *)flt_dyn; f32.subA = f32.hist + FLT_F32_BINS; f32.subB = (uint32_t *)(f32.subA + FLT_SUBA);
    f32.reset();
    const int tid = threadIdx.x;
    const int64_t r = blockIdx.x;
    const float M = row_max[r];
    void *out = (char *)probs + r * V * (DT == JF_F32 ? 4 : 2);
    // count and exact sum of the ids with key >= lo_ (and, hi_ != 0, key < hi_); HIST != 0 leaves the zeros out (they add nothing)
    auto count_sum = [&](uint32_t lo_, uint32_t hi_, unsigned long long &c_, double &s_) {
        if constexpr (HIST == 1) flt_hist_count_sum(sh, hist, lo_, hi_, c_, s_);
        else if constexpr (HIST == 2) {
            flt_f32_ge(sh, f32, out, V, lo_, c_, s_);
            if (hi_) { unsigned long long c2; double s2_; flt_f32_ge(sh, f32, out, V, hi_, c2, s2_); c_ -= c2; s_ -= s2_; }
        } else flt_count_sum<DT>(sh, out, V, lo_, hi_, c_, s_);
    };
    // the largest key `lo` (on the key grid) for which pred(count, sum of the ids >= lo) holds — pred is monotone, holds at 0 and fails at
    // FLT_KEY_TOP.  Coarse to fine for the float32 levels (bin edges, 256-aligned keys, keys): a phase stays inside what the one before found.
    auto bisect = [&](auto pred) -> uint32_t {
        uint32_t lo = 0u, hi = FLT_KEY_TOP;
        unsigned long long c_; double s_;
#pragma unroll
        for (int ph = (HIST == 2 ? 0 : 2); ph < 3; ++ph) {
            const uint32_t G = HIST == 2 ? (ph == 0 ? 1u << FLT_F32_SHIFT : ph == 1 ? 0x100u : 1u) : STEP;
            while (hi - lo > G) {
                const uint32_t mid = (lo + (hi - lo) / 2u) & ~(G - 1u);
                count_sum(mid, 0u, c_, s_);
                if (pred(c_, s_)) lo = mid; else hi = mid;
            }
        }
        return lo;
    };
    rs_load_tab(sh.tab);
    if constexpr (HIST == 1) flt_hist_zero(hist);
    if constexpr (HIST == 2) flt_hist64_zero(f32.hist);
    __syncthreads();
    const RsRow row = rs_make_row<DT>(logits, r, V, row_stride, t, M, 1.f);
    // ---- 1. exactly rounded probabilities of the whole row
    const bool finite = (__float_as_uint(M) & 0x7F800000u) != 0x7F800000u;
    FLT_STAMP(0);
    const double S = finite ? flt_row_s64<DT>(sh, row) : 0.0;
    FLT_STAMP(1);
    jf_rs_filter_row rec;
    rec.sum = S; rec.row_max = M; rec.x_keep = -INFINITY; rec.cut1 = 0u; rec.tie1 = (int32_t)V - 1; rec.s1 = 1.f; rec.cut2 = 0u; rec.tie2 = (int32_t)V - 1; rec.s2 = 1.f;
    rec.flags = ((top_k > 0 && (int64_t)top_k < V) ? JF_RS_FILT_TOPK : 0u) | (top_p > 0.f ? JF_RS_FILT_TOPP : 0u); rec.rsv = 0u;
    {
        RsRow plain = row;
        plain.S = row_sumexp[r];                                               // (NaN / inf rows: jf_rs_probs' float32 statistics, plain formula)
        const double invS = S > 0.0 ? 1.0 / S : 0.0;
        constexpr int NB = 8;                                                 // eight loads in flight per thread (one at a time: ~110 us of round trips per row)
        for (int64_t b0 = (int64_t)tid * EPV; b0 < V; b0 += (int64_t)NB * FLT_TPB * EPV) {
            u32x4 vv[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) { const int64_t e0 = b0 + (int64_t)k * FLT_TPB * EPV; if (e0 < V) vv[k] = rs_load_vec<DT>(row, e0); }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int64_t e0 = b0 + (int64_t)k * FLT_TPB * EPV;
                if (e0 >= V) continue;
                float p[EPV];
                rs_any_probs_from_vec<DT>(invS > 0.0 ? row : plain, vv[k], invS, sh.tab, p);
#pragma unroll
                for (int j = 0; j < EPV; ++j) {
                    p[j] = (e0 + j < V && p[j] >= 0.f) ? p[j] : 0.f;          // (a NaN row filters to zeros)
                    if constexpr (HIST == 1) { if (p[j] > 0.f) flt_hist_add(hist, __float_as_uint(p[j])); }
                    if constexpr (HIST == 2) { if (p[j] > 0.f) flt_hist64_add(f32.hist, __float_as_uint(p[j])); }
                }
                if (((uintptr_t)out % 16) == 0 && e0 + EPV <= V) {
                    u32x4 o;
                    if constexpr (DT == JF_F32) o = u32x4{__float_as_uint(p[0]), __float_as_uint(p[1]), __float_as_uint(p[2]), __float_as_uint(p[3])};
                    else o = u32x4{(__float_as_uint(p[0]) >> 16) | (__float_as_uint(p[1]) & 0xFFFF0000u), (__float_as_uint(p[2]) >> 16) | (__float_as_uint(p[3]) & 0xFFFF0000u),
                                   (__float_as_uint(p[4]) >> 16) | (__float_as_uint(p[5]) & 0xFFFF0000u), (__float_as_uint(p[6]) >> 16) | (__float_as_uint(p[7]) & 0xFFFF0000u)};
                    *(u32x4 *)((char *)out + e0 * (DT == JF_F32 ? 4 : 2)) = o;
                } else {
#pragma unroll
                    for (int j = 0; j < EPV; ++j) if (e0 + j < V) flt_store<DT>(out, e0 + j, p[j]);
                }
            }
        }
    }
    __syncthreads();
    FLT_STAMP(2);
    const float floor_d = rs_round_prob<DT>(1e-12);                           // sum.clamp_min(1e-12) in the dtype
    unsigned long long cnt;
    double sum;
    const bool want_p = top_p > 0.f && top_p < 1.f;
    // ---- 2. top-k (JDN:73-84)
    if (top_k > 0 && (int64_t)top_k < V) {
        const uint32_t thr = bisect([&](unsigned long long c_, double) { return c_ >= (unsigned long long)top_k; });      // the k-th largest value
        FLT_STAMP(3);                                                         // (fewer than k positive ids: 0, and every zero is "kept")
        count_sum(thr + STEP, 0u, cnt, sum);              // the ids above it, and their sum
        const long long need = (long long)top_k - (long long)cnt;            // ids AT the threshold to keep (>= 1), lowest index first
        unsigned long long n_at; double s_at;
        count_sum(thr, thr + STEP, n_at, s_at);
        const int64_t tie_last = (unsigned long long)need >= n_at ? V : flt_nth_equal<DT>(sh, out, V, thr, need);
        const double kept = sum + (double)need * (double)__uint_as_float(thr);
        float s1 = rs_round_prob<DT>(kept);
        s1 = s1 > floor_d ? s1 : floor_d;
        rec.cut1 = thr; rec.tie1 = (int32_t)(tie_last >= V ? V - 1 : tie_last); rec.s1 = s1;
        FLT_STAMP(4);
        flt_renorm<DT, HIST>(out, V, thr, tie_last, s1, !want_p ? nullptr : HIST == 1 ? (void *)hist : HIST == 2 ? (void *)f32.hist : nullptr);
        f32.reset();
        FLT_STAMP(5);
    }
    // ---- 3. top-p (JDN:91-107)
    if (want_p) {
        const float tp = rs_round_prob<DT>((double)(float)top_p_d);           // `cdf <= tp`: the Python float is cast to the tensor's dtype
        count_sum(0u, 0u, cnt, sum);                      // everything
        const bool all = rs_round_prob<DT>(sum) <= tp;
        uint32_t thr = 0u;
        int64_t tie_last = V;
        float s2;
        if (all) {                                                            // the nucleus holds every id with mass (and the zeros behind them)
            s2 = rs_round_prob<DT>(sum);
        } else {
            // a group of equal values is kept WHOLE when the cumulative sum at its end (everything >= it), rounded, is <= tp.  The group
            // the cut falls into is the largest value that is NOT: the largest key whose (sum of everything >= key) still exceeds tp — a
            // present value, since the sum changes there.
            thr = bisect([&](unsigned long long, double s_) { return !(rs_round_prob<DT>(s_) <= tp); });
            unsigned long long n_whole; double c_whole;
            count_sum(thr + STEP, 0u, n_whole, c_whole);   // the groups kept whole
            unsigned long long n_at; double s_at;
            count_sum(thr, thr + STEP, n_at, s_at);
            const double v = (double)__uint_as_float(thr);
            long long a = 0, b = (long long)n_at;                             // ids of the group whose own cumulative sum passes: pass(a) holds, pass(b) fails
            while (b - a > 1) { const long long m2 = a + (b - a) / 2; if (rs_round_prob<DT>(c_whole + (double)m2 * v) <= tp) a = m2; else b = m2; }
            long long c = a;
            if (n_whole == 0ull && c == 0) c = 1;                             // keep[..., 0] = True
            tie_last = c == 0 ? -1 : flt_nth_equal<DT>(sh, out, V, thr, c);
            s2 = rs_round_prob<DT>(c_whole + (double)c * v);
        }
        s2 = s2 > floor_d ? s2 : floor_d;
        rec.cut2 = thr; rec.tie2 = (int32_t)(tie_last >= V ? V - 1 : tie_last); rec.s2 = s2;
        FLT_STAMP(6);                                                          // (no renormalising pass: the record says what an id becomes)
    }
    if (tid == 0) {
        const int64_t tok = draft_next[r];
        float pd = 0.f;
        if (tok >= 0 && tok < V && S > 0.0) {                                   // the scratch row holds y (p when top-k is off): the top-p stage of the map is left
            const float y = __uint_as_float(flt_key<DT>(out, tok));
            const uint32_t b = __float_as_uint(y);
            pd = !want_p ? y : ((b > rec.cut2 || (b == rec.cut2 && tok <= (int64_t)rec.tie2)) ? flt_div<DT>(y, rec.s2) : 0.f);
        }
        filt[r] = rec;
        p_draft[r] = pd;
        row_max[r] = INFINITY;
        row_sumexp[r] = RS_PROB_ROW;
    }
}

// the dense tensor of the records (the reference's `probs`): one workgroup per (row, chunk of 8 192 ids)
template <int DT>
__global__ __launch_bounds__(256) void rs_filter_expand_kernel(const void *logits, int64_t V, int64_t row_stride, float t, const jf_rs_filter_row *filt,
                                                                void *probs, int chunks) {
    constexpr int EPV = Elem<DT>::EPV;
    __shared__ double s_tab[64];
    rs_load_tab(s_tab);
    __syncthreads();
    const int64_t r = blockIdx.x / chunks, c = blockIdx.x % chunks;
    const jf_rs_filter_row &f = filt[r];
    RsRow row = rs_make_row<DT>(logits, r, V, row_stride, t, f.row_max, 1.f);
    row.flt = &f;
    const double invS = f.sum > 0.0 ? 1.0 / f.sum : 0.0;
    void *out = (char *)probs + r * V * (DT == JF_F32 ? 4 : 2);
    const int64_t per = (V + chunks - 1) / chunks, lo = (c * per + EPV - 1) / EPV * EPV;
    int64_t hi = ((c + 1) * per + EPV - 1) / EPV * EPV;
    if (hi > V) hi = V;
    for (int64_t e0 = lo + (int64_t)threadIdx.x * EPV; e0 < hi; e0 += 256 * EPV) {
        float p[EPV];
        rs_row_probs_from_vec<DT>(row, rs_load_vec<DT>(row, e0), e0, invS, s_tab, p);
#pragma unroll
        for (int j = 0; j < EPV; ++j) if (e0 + j < V) flt_store<DT>(out, e0 + j, p[j]);
    }
}

extern "C" size_t jf_rs_filter_workspace_bytes(int dtype, int64_t R, int64_t V) {
    if (R <= 0 || V <= 0) return 0;
    if (dtype == JF_BF16 && V <= (int64_t)FH_MAX_TILES * FH_TILE) return 0;      // the pattern counts live in LDS
    return (size_t)R * (size_t)V * (dtype == JF_F32 ? 4 : 2);                     // a scratch row per row
}

extern "C" int jf_rs_filter(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride, const int64_t *draft_next,
                            float temperature, int32_t top_k, double top_p, jf_rs_filter_row *filt, float *p_draft, float *row_max,
                            float *row_sumexp, void *workspace, size_t workspace_bytes, void *stream) {
    if (R <= 0) return JF_OK;
    if (!logits || !draft_next || !filt || !p_draft || !row_max || !row_sumexp) return fail(JF_E_INVALID, "jf_rs_filter: null pointer");
    if (dtype != JF_F32 && dtype != JF_BF16) return fail(JF_E_INVALID, "jf_rs_filter: dtype %d", dtype);
    if (V <= 0 || V > 0x7FFFFFFFll || row_stride < V) return fail(JF_E_INVALID, "jf_rs_filter: bad shape V=%lld stride=%lld", (long long)V, (long long)row_stride);
    if (!(top_p == top_p)) return fail(JF_E_INVALID, "jf_rs_filter: top_p is NaN");
    const size_t need = jf_rs_filter_workspace_bytes(dtype, R, V);
    if (need && (!workspace || workspace_bytes < need || ((uintptr_t)workspace % 16) != 0))
        return fail(JF_E_INVALID, "jf_rs_filter: workspace of %zu bytes (16-byte aligned) needed, %zu given", need, workspace_bytes);
    const float t = (temperature <= 0.f) ? 1.f : temperature;               // JDN:66-67
    hipStream_t s = (hipStream_t)stream;
    // the product form of the bf16 scaling where the host proves it exact for this T (as jf_rs_probs / jf_rs_step: -T says so)
    const float tt = (dtype == JF_BF16 && t != 1.f && rs_scale_is_exact(t)) ? -t : t;
    // dynamic LDS beyond the default 64 KB per workgroup: opt in once PER DEVICE (a process may drive several);
    // JF_RS_FILTER_HIST=0 keeps the float32 rows on the pass-per-bisection-step variant (what rows of more than 2^18 ids take)
    static const bool hist = [] { const char *e = getenv("JF_RS_FILTER_HIST"); return !(e && e[0] == '0'); }();
    static std::atomic<int> opted[64];                                      // per device: 0 not asked, 1 granted, -1 refused
    auto lds_ok = [&](const void *fn, unsigned bytes, int slot) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return false;
        std::atomic<int> &st = opted[dev * 2 + slot];
        int v = st.load(std::memory_order_acquire);
        if (v == 0) {
            v = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 1 : -1;
            if (v < 0) (void)hipGetLastError();
            st.store(v, std::memory_order_release);
        }
        return v > 0;
    };
    if (dtype == JF_BF16 && need == 0) {
        if (!lds_ok((const void *)rs_filter_hist_kernel, FH_LDS, 1)) return fail(JF_E_LAUNCH, "jf_rs_filter: the device refuses %u bytes of LDS per workgroup", FH_LDS);
        static const bool zone = [] { const char *e = getenv("JF_RS_FILTER_ZONE"); return !(e && e[0] == '0'); }();   // A/B knob, read once
        if (zone && V <= (int64_t)FH_MAX_TILES * ZN_TPB * 8) {
            // two launches: the rows whose patterns fit the 16 KB zone table (two rows per CU), then the rows that did not (none, with a model's logits)
            rs_filter_zone_kernel<<<dim3((unsigned)R), dim3(ZN_TPB), ZN_LDS, s>>>(logits, R, V, row_stride, draft_next, tt, top_k, top_p, filt, p_draft, row_max, row_sumexp);
            const unsigned g = (unsigned)(R < 256 ? R : 256);
            rs_filter_hist_kernel<<<dim3(g), dim3(FH_TPB), FH_LDS, s>>>(logits, R, V, row_stride, draft_next, tt, top_k, top_p, filt, p_draft, row_max, row_sumexp, 1);
            return check_launch("rs_filter_zone_kernel + rs_filter_hist_kernel");
        }
        rs_filter_hist_kernel<<<dim3((unsigned)R), dim3(FH_TPB), FH_LDS, s>>>(logits, R, V, row_stride, draft_next, tt, top_k, top_p, filt, p_draft, row_max, row_sumexp, 0);
        return check_launch("rs_filter_hist_kernel");
    }
    if (dtype == JF_F32) {
        if (hist && V <= (1ll << 18) && lds_ok((const void *)rs_filter_kernel<JF_F32, 2>, FLT_LDS_F32, 0))
            rs_filter_kernel<JF_F32, 2><<<dim3((unsigned)R), dim3(FLT_TPB), FLT_LDS_F32, s>>>(logits, R, V, row_stride, draft_next, tt, top_k, top_p, workspace, filt, p_draft, row_max, row_sumexp);
        else rs_filter_kernel<JF_F32, 0><<<dim3((unsigned)R), dim3(FLT_TPB), 0, s>>>(logits, R, V, row_stride, draft_next, tt, top_k, top_p, workspace, filt, p_draft, row_max, row_sumexp);
    } else {                                                                 // bf16 rows of more than 524 288 ids
        rs_filter_kernel<JF_BF16, 0><<<dim3((unsigned)R), dim3(FLT_TPB), 0, s>>>(logits, R, V, row_stride, draft_next, tt, top_k, top_p, workspace, filt, p_draft, row_max, row_sumexp);
    }
    return check_launch("rs_filter_kernel");
}

extern "C" int jf_rs_filter_expand(const void *logits, int dtype, int64_t R, int64_t V, int64_t row_stride, float temperature,
                                   const jf_rs_filter_row *filt, void *probs, void *stream) {
    if (R <= 0) return JF_OK;
    if (!logits || !filt || !probs) return fail(JF_E_INVALID, "jf_rs_filter_expand: null pointer");
    if (dtype != JF_F32 && dtype != JF_BF16) return fail(JF_E_INVALID, "jf_rs_filter_expand: dtype %d", dtype);
    if (V <= 0 || V > 0x7FFFFFFFll || row_stride < V) return fail(JF_E_INVALID, "jf_rs_filter_expand: bad shape V=%lld stride=%lld", (long long)V, (long long)row_stride);
    const float t = (temperature <= 0.f) ? 1.f : temperature;
    const float tt = (dtype == JF_BF16 && t != 1.f && rs_scale_is_exact(t)) ? -t : t;
    const int chunks = (int)((V + 8191) / 8192);
    if ((int64_t)R * chunks > 0x7FFFFFFFll) return fail(JF_E_CAPACITY, "jf_rs_filter_expand: too many rows");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == JF_F32) rs_filter_expand_kernel<JF_F32><<<dim3((unsigned)(R * chunks)), dim3(256), 0, s>>>(logits, V, row_stride, tt, filt, probs, chunks);
    else rs_filter_expand_kernel<JF_BF16><<<dim3((unsigned)(R * chunks)), dim3(256), 0, s>>>(logits, V, row_stride, tt, filt, probs, chunks);
    return check_launch("rs_filter_expand_kernel");
}

// ------------------------------------------------------------------------------------------------
// Inverse-CDF draws (the injected stand-in for torch.multinomial, JDN:126-153 / JDO:150-168): the smallest index whose
// float64 running sum of (exactly rounded) probabilities, in vocabulary order, exceeds u * total.
//
// The row is summed ONCE, hierarchically, by RS_SEG workgroups (one vocabulary segment each):
//   phase A   float64 sum of exp(xs - M) over the segment -> s64part; the row's S is the sum of the RS_SEG partials in
//             segment order (every workgroup forms the same S).  One launch: the partials are exchanged through agent-scope
//             words inside it (the 16 workgroups of a row have consecutive block ids); several launches: rs_rowsum_a_kernel.
//   phase B   every element's exact probability, summed in float64 per lane vector, wavefront scans per tile: the mass of
//             every (tile, wavefront) of the segment -> wtsum, their sum in order -> segsum.  The workgroup whose segment
//             holds the row's PROPOSED token also records the running sum in front of it (lo_part, relative to the segment)
//             and its probability: with the segment sums that is the token's interval [c_lo, c_hi) of the CDF, so "this draw
//             returns the proposed token again" (JDN:140-146) is a comparison of u * total with two numbers.
// A draw then needs ONE wavefront (rs_pick_wave): segment prefix -> wave-tile prefix -> the 64 vectors of that wavefront's
// tile slice (one 16-byte load per lane, L2-resident) -> scan -> the crossing lane resolves inside its vector.  All running
// sums are formed the same way on both sides (sequential prefixes, one scan), so the interval test and the walk agree except
// at float64 rounding of lane boundaries; should the walk return the proposed token after all, the masked argmax decides.
// Vocabularies whose segments have more than RS_WT wave-tiles (V > 524 288 bf16 / 262 144 fp32) walk the segment with the
// whole workgroup instead (rs_pick_wg).
// ------------------------------------------------------------------------------------------------
struct RsSumShared {
    double tab[64];
    double red[4];
    double wt[RS_WT];               // wave-tile sums of this segment (phase B), tile-major
    double part[RS_SEG];
    double pv[8];                   // the avoided token's vector, its lane's exclusive prefix in front
    double excl;
    float mx[4];                    // the wavefronts' largest scaled logit (phase A)
    int ok;
};

// phase A of one (row, segment): float64 sum of exp(xs - M); KEEP: the float32 images of the exps stay in e32 for phase B
template <int DT, bool KEEP>
__device__ __forceinline__ double rs_seg_exp_sum(const RsRow &row, int64_t lo, int64_t hi, const double *tab,
                                                 float (&e32)[RsKeep<DT>::NV][Elem<DT>::EPV], u32x4 (&v)[RsKeep<DT>::NV], float &lmax) {
    constexpr int EPV = Elem<DT>::EPV, NV = RsKeep<DT>::NV;
    double acc = 0.0;
    lmax = -INFINITY;                                                  // this lane's largest scaled logit (slots beyond V hold -inf)
    const float mcut = row.M + (float)RS_EXP_CUT + 1.f;                // one above the cut (float rounding of the sum): vectors wholly above it skip the clamp
    for (int64_t b0 = lo + (int64_t)threadIdx.x * EPV; b0 < hi; b0 += (int64_t)NV * 256 * EPV) {
#pragma unroll
        for (int k = 0; k < NV; ++k) { const int64_t e0 = b0 + (int64_t)k * 256 * EPV; if (e0 < hi) v[k] = rs_load_vec<DT>(row, e0); }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int64_t e0 = b0 + (int64_t)k * 256 * EPV;
            if (e0 >= hi) { if constexpr (KEEP) { for (int j = 0; j < EPV; ++j) e32[k][j] = 0.f; } continue; }
            float xs[EPV];
            rs_scaled_from_vec<DT>(row, v[k], xs);
            float xmin = xs[0], xmax = xs[0];
#pragma unroll
            for (int j = 1; j < EPV; ++j) { xmin = fminf(xmin, xs[j]); xmax = fmaxf(xmax, xs[j]); }
            lmax = fmaxf(lmax, xmax);
            if (__ballot(!(xmin >= mcut)) == 0ull) {           // (wave-uniform, the usual case) nothing of this vector is near the cut: no clamp, no select
#pragma unroll
                for (int j = 0; j < EPV; ++j) {
                    const double e = rs_exp64((double)xs[j] - (double)row.M, tab);
                    acc += e;
                    if constexpr (KEEP) e32[k][j] = (float)e;
                }
            } else {
#pragma unroll
                for (int j = 0; j < EPV; ++j) {
                    const double e = rs_e64(xs[j], (double)row.M, tab);
                    acc += e;
                    if constexpr (KEEP) e32[k][j] = (float)e;
                }
            }
        }
    }
    return acc;
}

// exact probabilities of one vector from the kept float32 exps (bf16 logits): the float32 product errs by < 2^-22, so the
// rounding is certain unless a bf16 boundary lies inside the 2^-21 band around it (1 element in ~8 000: float64 quotient)
template <int DT>
__device__ __forceinline__ void rs_probs_from_kept(const RsRow &row, int64_t e0, const float (&e)[Elem<DT>::EPV], double invS, float invS32,
                                                   uint32_t tiny_m1, const double *tab, float (&p)[Elem<DT>::EPV]) {
    constexpr int EPV = Elem<DT>::EPV;
    // two elements per instruction: v_pk_mul_f32 for the product and its two band ends, v_cvt_pk_bf16_f32 for the roundings
    // (gfx950's conversion is RNE for every non-NaN pattern: tools/experiments/cvt_bf16_exhaustive.hip); the pair is certain when
    // both ends round to the same bf16.  tiny_m1: bits of (1e-36 / invS32) - 1 — a nonzero exp below it has a product near
    // float32's subnormal range, inexact in itself (unsigned compare of bits - 1: zero wraps to the top)
    const rs_f32x2 s2 = {invS32, invS32}, lo2 = {0.99999952316284179688f, 0.99999952316284179688f}, hi2 = {1.00000047683715820312f, 1.00000047683715820312f};   // 1 -+ 2^-21
    uint32_t diff = 0u, emin = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < EPV; j += 2) {
        const rs_f32x2 q = rs_f32x2{e[j], e[j + 1]} * s2;
        const uint32_t ha = __builtin_bit_cast(uint32_t, __builtin_convertvector(q * lo2, rs_bf16x2));
        const uint32_t hb = __builtin_bit_cast(uint32_t, __builtin_convertvector(q * hi2, rs_bf16x2));
        p[j] = __uint_as_float(ha << 16);
        p[j + 1] = __uint_as_float(ha & 0xFFFF0000u);
        diff |= ha ^ hb;
        const uint32_t b0 = __float_as_uint(e[j]) - 1u, b1 = __float_as_uint(e[j + 1]) - 1u;
        emin = b0 < emin ? b0 : emin;
        emin = b1 < emin ? b1 : emin;
    }
    const bool slow = diff != 0u || emin < tiny_m1;
    // out of the straight-line path; the vector is read again (L2) rather than kept in registers across the exchange of the partials
    if (__builtin_expect(slow, 0)) rs_exact_probs_from_vec<DT>(row, rs_load_vec<DT>(row, e0), invS, tab, p);
}

// phase B of one (row, segment): exact probabilities, float64 sums per lane vector, one scan per tile.  SIG: results go out
// as agent-scope atomic stores followed by the segment's generation word (the consumers are other workgroups of the SAME
// launch).  S <= 0: the row keeps the plain float32 formula (NaN / inf rows).  HIER: the segment's wave-tile sums fit the
// table (every real vocabulary); else the sums are formed tile by tile as rs_pick_wg re-forms them.
template <int DT, bool SIG, bool KEEP>
__device__ __forceinline__ void rs_seg_prob_sums(const RsRow &row, int64_t lo, int64_t hi, double S, const RsWs &w, int item, int seg,
                                                 int64_t av, uint32_t gen, RsSumShared &sh, const float (&e32)[RsKeep<DT>::NV][Elem<DT>::EPV],
                                                 u32x4 (&v)[RsKeep<DT>::NV], int empty_from = RS_SEG /* SIG: this workgroup also stands in for the empty segments [empty_from, RS_SEG) */) {
    constexpr int EPV = Elem<DT>::EPV, NV = RsKeep<DT>::NV;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool mine = av >= lo && av < hi;                   // workgroup-uniform: this segment holds the avoided token
    const int64_t avo = av - lo;
    const int kstar = mine ? (int)(avo / (256 * EPV)) : -1;  // its tile, and its thread inside the tile
    const int tstar = mine ? (int)((avo % (256 * EPV)) / EPV) : -1;
    const double invS = S > 0.0 ? 1.0 / S : 0.0;
    const float invS32 = (float)invS;
    const uint32_t tiny_m1 = __float_as_uint(1.0e-36f / (invS32 > 0.f ? invS32 : 1.f)) - 1u;   // (S >= 1: the quotient is a normal float)
    const int ntiles = (int)((hi - lo + 256 * EPV - 1) / (256 * EPV));
    const bool hier = ntiles * 4 <= RS_WT;
    double base = 0.0, front = 0.0;                          // !hier: running sum of the tiles before / in front of the token's wavefront
    for (int k0 = 0; k0 < ntiles; k0 += NV) {
        if constexpr (!KEEP) {
#pragma unroll
            for (int k = 0; k < NV; ++k) { const int64_t e0 = lo + ((int64_t)(k0 + k) * 256 + tid) * EPV; if (k0 + k < ntiles && e0 < hi) v[k] = rs_load_vec<DT>(row, e0); }
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (k0 + k >= ntiles) continue;                  // workgroup-uniform
            const int64_t e0 = lo + ((int64_t)(k0 + k) * 256 + tid) * EPV;
            float p[EPV];
#pragma unroll
            for (int j = 0; j < EPV; ++j) p[j] = 0.f;
            if (e0 < hi && row.flt) {                        // (workgroup-uniform) the record's map of the exact probabilities; phase A did not run
                rs_row_probs_from_vec<DT>(row, rs_load_vec<DT>(row, e0), e0, invS, sh.tab, p);
            } else if (e0 < hi) {
                if constexpr (KEEP && DT == JF_BF16) {
                    if (__builtin_expect(invS > 0.0, 1)) rs_probs_from_kept<DT>(row, e0, e32[k], invS, invS32, tiny_m1, sh.tab, p);
                    else rs_probs_from_vec<DT>(row, rs_load_vec<DT>(row, e0), p);
                } else {
                    // (KEEP: v[] is what phase A loaded — and phase A is skipped for rows without an exact sum: NaN / inf rows and
                    //  probability rows read their vector here.  Round 5: such float32 rows summed whatever the registers held.)
                    if (KEEP && !(invS > 0.0)) rs_probs_from_vec<DT>(row, rs_load_vec<DT>(row, e0), p);
                    else rs_any_probs_from_vec<DT>(row, v[k], invS, sh.tab, p);
                }
            }
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < EPV; ++j) a += (double)p[j];
            const double incl = wave_incl_scan_f64(a, lane);  // all lanes: the wavefront's total is the scan's last value
            const double up = wave_shift_up_f64(incl, lane);  // all lanes active: never shift under the lane test
            const double wt = __shfl(incl, 63, 64);
            if (mine && k0 + k == kstar && tid == tstar) {    // the avoided token's lane: exclusive prefix inside its wavefront, its vector
                sh.excl = up;
#pragma unroll
                for (int j = 0; j < EPV; ++j) sh.pv[j] = (double)p[j];
            }
            if (hier) {
                if (lane == 0) sh.wt[(k0 + k) * 4 + wave] = wt;
            } else {                                          // tile by tile, as rs_pick_wg walks: (w0 + w1) + (w2 + w3) per tile
                __syncthreads();
                if (lane == 0) sh.red[wave] = wt;
                __syncthreads();
                if (mine && k0 + k == kstar) { double wb = base; for (int q = 0; q < (tstar >> 6); ++q) wb += sh.red[q]; front = wb; }
                base += (sh.red[0] + sh.red[1]) + (sh.red[2] + sh.red[3]);
            }
        }
    }
    RS_PHASE(item, seg, 4);                                   // 4: phase B probabilities + scans done
    __syncthreads();
    double sum = base;
    if (hier && tid < 64) {                                  // the table's running sum in order: one LDS read per lane, then broadcasts
        const double mywt = tid < ntiles * 4 ? sh.wt[tid] : 0.0;   // (a read per step of the sum was ~1 us of this workgroup's tail)
        sum = 0.0;
        const int istar = mine ? kstar * 4 + (tstar >> 6) : -1;
        for (int i = 0; i < ntiles * 4; ++i) {
            if (i == istar) front = sum;
            sum += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mywt), i), __builtin_amdgcn_readlane(__double2loint(mywt), i));
        }
    }
    if (tid == 0) {
        double lo_p = 0.0, p_av = 0.0;
        if (mine) {                                          // the running sum in front of the token, formed as the walk forms it
            double rr = front + sh.excl;
            const int jstar = (int)(avo % EPV);
            for (int j = 0; j < jstar; ++j) rr += sh.pv[j];
            lo_p = rr; p_av = sh.pv[jstar];
        }
        if constexpr (SIG) {
            st_agent_f64(w.segsum + (int64_t)item * RS_SEG + seg, sum);
            for (int e = empty_from; e < RS_SEG; ++e) st_agent_f64(w.segsum + (int64_t)item * RS_SEG + e, 0.0);
            if (mine) { st_agent_f64(w.lo_part + item, lo_p); st_agent_f64(w.p_avoid + item, p_av); }
            if (seg == 0) st_agent_f64(w.s64 + item, S);
        } else {
            w.segsum[(int64_t)item * RS_SEG + seg] = sum;
            if (mine) { w.lo_part[item] = lo_p; w.p_avoid[item] = p_av; }
            if (seg == 0) w.s64[item] = S;
        }
    }
    if (hier && tid < ntiles * 4) {                          // the wave-tile table of this segment, for the one-wavefront walk
        double *dst = w.wtsum + ((int64_t)item * RS_SEG + seg) * RS_WT;
        if constexpr (SIG) st_agent_f64(dst + tid, sh.wt[tid]); else dst[tid] = sh.wt[tid];
    }
    RS_PHASE(item, seg, 5);                                   // 5: sums formed, stores issued
    if constexpr (SIG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the sums are performed before the word that announces them
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(w.segdone + (int64_t)item * RS_SEG + seg, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int e = empty_from; e < RS_SEG; ++e) __hip_atomic_store(w.segdone + (int64_t)item * RS_SEG + e, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    RS_PHASE(item, seg, 6);                                   // 6: announced
}

// Both phases of one (row, segment) inside ONE launch (rs_step_fused_kernel): the 16 workgroups of a row exchange their
// float64 partials through agent-scope words.  Returns false when a peer's partial did not arrive within the wait bound.
template <int DT, bool FLT>
__device__ __forceinline__ bool rs_rowsum_fused(const void *logits, int64_t V, int64_t row_stride, const float *row_max, const float *row_sumexp,
                                                float t, const RsWs &w, int item, int seg, int nact, int r, int64_t av, uint32_t gen, RsSumShared &sh) {
    constexpr int EPV = Elem<DT>::EPV, NV = RsKeep<DT>::NV;
    const int tid = threadIdx.x;
    const int empty_from = seg == nact - 1 ? nact : RS_SEG;    // the row's last active segment stands in for the empty ones (zeros)
    const RsRow row = rs_step_row<DT>(logits, r, V, row_stride, t, row_max, row_sumexp, FLT ? w.filt : nullptr);   // (!FLT: a literal null — the unfiltered kernel compiles without the record paths)
    const int64_t segE = rs_seg_elems(V, EPV);
    const int64_t lo = (int64_t)seg * segE;
    int64_t hi = lo + segE;
    if (hi > V) hi = V;
    if (lo > V) hi = lo;                                       // (tiny vocabularies: empty trailing segments)
    const bool exact = !row.flt && rs_row_is_exact(row.M, row.S);      // (a filtered row's sum is its record's: no phase A)
    const int ntiles = (int)((hi - lo + 256 * EPV - 1) / (256 * EPV));
    rs_load_tab(sh.tab);
    if (tid == 0) sh.ok = 1;
    __syncthreads();
    RS_PHASE(item, seg, 0);                                   // 0: table in LDS
    float e32[NV][EPV];
    u32x4 v[NV];
    double S = row.flt ? rs_flt_sum(row) : 0.0;
    const bool keep = ntiles <= NV;                           // workgroup-uniform (true for every vocabulary up to 163 840)
    if (exact) {
        float lmax;
        double acc = keep ? rs_seg_exp_sum<DT, true>(row, lo, hi, sh.tab, e32, v, lmax) : rs_seg_exp_sum<DT, false>(row, lo, hi, sh.tab, e32, v, lmax);
        RS_PHASE(item, seg, 1);                               // 1: phase A loads + exps done
        acc = wave_sum_f64(acc);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off, 64));
        if ((tid & 63) == 0) { sh.red[tid >> 6] = acc; sh.mx[tid >> 6] = lmax; }
        __syncthreads();
        RS_PHASE(item, seg, 2);                               // 2: reduced
        if (tid == 0) {
            st_agent_f64(w.s64part + (int64_t)item * RS_SEG + seg, (sh.red[0] + sh.red[1]) + (sh.red[2] + sh.red[3]));
            st_agent_f32(w.segmax + (int64_t)item * RS_SEG + seg, fmaxf(fmaxf(sh.mx[0], sh.mx[1]), fmaxf(sh.mx[2], sh.mx[3])));
            for (int e = empty_from; e < RS_SEG; ++e) { st_agent_f64(w.s64part + (int64_t)item * RS_SEG + e, 0.0); st_agent_f32(w.segmax + (int64_t)item * RS_SEG + e, -INFINITY); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(w.s64done + (int64_t)item * RS_SEG + seg, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int e = empty_from; e < RS_SEG; ++e) __hip_atomic_store(w.s64done + (int64_t)item * RS_SEG + e, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < RS_SEG) {                                   // the 15 peers: consecutive block ids, dispatched together
            if (!rs_wait_word(w.s64done + (int64_t)item * RS_SEG + tid, gen)) sh.ok = 0;
            sh.part[tid] = ld_agent_f64(w.s64part + (int64_t)item * RS_SEG + tid);
        }
        __syncthreads();
        RS_PHASE(item, seg, 3);                               // 3: all 16 partials in
        if (!sh.ok) return false;
#pragma unroll
        for (int s = 0; s < RS_SEG; ++s) S += sh.part[s];    // segment order: every workgroup of the row forms the same S
    }
    if (keep) rs_seg_prob_sums<DT, true, true>(row, lo, hi, S, w, item, seg, av, gen, sh, e32, v, empty_from);
    else rs_seg_prob_sums<DT, true, false>(row, lo, hi, S, w, item, seg, av, gen, sh, e32, v, empty_from);
    return true;
}

// The same as two launches (batches the one-launch step does not take, and the on-policy step)
template <int DT>
__global__ __launch_bounds__(256) void rs_rowsum_a_kernel(const void *logits, int64_t V, int64_t row_stride, const float *row_max,
                                                           const float *row_sumexp, float t, RsWs w) {
    constexpr int EPV = Elem<DT>::EPV, NV = RsKeep<DT>::NV;
    __shared__ RsSumShared sh;
    const int item = blockIdx.x / RS_SEG, seg = blockIdx.x % RS_SEG, tid = threadIdx.x;
    const int r = w.sel_row[item];
    if (r < 0) return;
    const RsRow row = rs_step_row<DT>(logits, r, V, row_stride, t, row_max, row_sumexp, w.filt);
    if (row.flt || !rs_row_is_exact(row.M, row.S)) { if (tid == 0) w.s64part[(int64_t)item * RS_SEG + seg] = 0.0; return; }
    const int64_t segE = rs_seg_elems(V, EPV);
    const int64_t lo = (int64_t)seg * segE;
    int64_t hi = lo + segE;
    if (hi > V) hi = V;
    if (lo > V) hi = lo;
    rs_load_tab(sh.tab);
    __syncthreads();
    float e32[NV][EPV];
    u32x4 v[NV];
    float lmax;
    double acc = rs_seg_exp_sum<DT, false>(row, lo, hi, sh.tab, e32, v, lmax);
    acc = wave_sum_f64(acc);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off, 64));
    if ((tid & 63) == 0) { sh.red[tid >> 6] = acc; sh.mx[tid >> 6] = lmax; }
    __syncthreads();
    if (tid == 0) {
        w.s64part[(int64_t)item * RS_SEG + seg] = (sh.red[0] + sh.red[1]) + (sh.red[2] + sh.red[3]);
        w.segmax[(int64_t)item * RS_SEG + seg] = fmaxf(fmaxf(sh.mx[0], sh.mx[1]), fmaxf(sh.mx[2], sh.mx[3]));
    }
}
template <int DT>
__global__ __launch_bounds__(256) void rs_rowsum_b_kernel(const void *logits, int64_t V, int64_t row_stride, const float *row_max,
                                                           const float *row_sumexp, float t, RsWs w) {
    constexpr int EPV = Elem<DT>::EPV, NV = RsKeep<DT>::NV;
    __shared__ RsSumShared sh;
    const int item = blockIdx.x / RS_SEG, seg = blockIdx.x % RS_SEG;
    const int r = w.sel_row[item];
    if (r < 0) return;
    const RsRow row = rs_step_row<DT>(logits, r, V, row_stride, t, row_max, row_sumexp, w.filt);
    const int64_t segE = rs_seg_elems(V, EPV);
    const int64_t lo = (int64_t)seg * segE;
    int64_t hi = lo + segE;
    if (hi > V) hi = V;
    if (lo > V) hi = lo;
    rs_load_tab(sh.tab);
    double S = 0.0;
    if (row.flt) S = rs_flt_sum(row);
    else if (rs_row_is_exact(row.M, row.S)) {
#pragma unroll
        for (int s = 0; s < RS_SEG; ++s) S += w.s64part[(int64_t)item * RS_SEG + s];
    }
    __syncthreads();
    float e32[NV][EPV];
    u32x4 v[NV];
    rs_seg_prob_sums<DT, false, false>(row, lo, hi, S, w, item, seg, (int64_t)w.avoid[item], 0u, sh, e32, v);
}

// total mass of a row and the CDF interval [c_lo, c_hi) of its avoided token, from the segment sums.  `total` and the prefix
// in front of the token's segment are formed exactly as the walks form them (segments in order).
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
    c_hi = sstar >= 0 ? before + (lo_p + p_av) : 0.0;         // the walk's running sum behind the token: prefix + (relative sum + p)
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
    *u_final = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(u), f));   // f is wave-uniform: v_readlane, not a shuffle
    return f + 1;
}

// One inverse-CDF draw by ONE wavefront (all 64 lanes, uniform control flow) over the hierarchical sums of a row:
// segment prefix -> wave-tile prefix -> the 64 vectors of that wavefront's slice -> scan -> element.  S: the row's float64
// sum (<= 0: plain float32 formula).  AGENT: the sums were stored by other workgroups of this launch.
template <int DT, bool AGENT>
__device__ __forceinline__ int rs_pick_wave(const RsRow &row, const RsWs &w, int item, double S, float u, const double *tab, int lane) {
    constexpr int EPV = Elem<DT>::EPV;
    auto ld = [](const double *p) { return AGENT ? ld_agent_f64(p) : *p; };
    const double *segsum = w.segsum + (int64_t)item * RS_SEG;
    double sg[RS_SEG];
#pragma unroll
    for (int s = 0; s < RS_SEG; ++s) sg[s] = ld(segsum + s);
    double total = 0.0;
#pragma unroll
    for (int s = 0; s < RS_SEG; ++s) total += sg[s];
    const double thr = (double)u * total;
    int sstar = -1;
    double before = 0.0, run = 0.0;
#pragma unroll
    for (int s = 0; s < RS_SEG; ++s) {
        const double nx = run + sg[s];
        if (sstar < 0 && nx > thr) { sstar = s; before = run; }
        run = nx;
    }
    if (sstar < 0) return (int)(row.V - 1);                 // thr >= total: clamp like min(idx, V - 1)
    const int64_t segE = rs_seg_elems(row.V, EPV);
    const int64_t lo = (int64_t)sstar * segE;
    int64_t hi = lo + segE;
    if (hi > row.V) hi = row.V;
    const int nwt = (int)((hi - lo + 256 * EPV - 1) / (256 * EPV)) * 4;
    // wave-tile prefix, sequential in table order (as the segment sum was formed): every lane holds one entry, the running
    // sum is walked with broadcasts
    const double *wtp = w.wtsum + ((int64_t)item * RS_SEG + sstar) * RS_WT;
    const double mywt = lane < nwt ? ld(wtp + lane) : 0.0;
    auto wt_at = [&](int i) {                               // i is wave-uniform: two v_readlane (a shuffle is a trip through the LDS crossbar per step)
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mywt), i), __builtin_amdgcn_readlane(__double2loint(mywt), i));
    };
    int istar = -1;
    double rel = 0.0;                                       // relative running sum in front of wave-tile istar
    {
        double r2 = 0.0;
        for (int i = 0; i < nwt; ++i) {
            const double nx = r2 + wt_at(i);
            if (before + nx > thr) { istar = i; rel = r2; break; }
            r2 = nx;
        }
        if (istar < 0) {                                    // rounding only: the segment sum said it crosses here -> last wave-tile with mass
            double r3 = 0.0;
            for (int i = 0; i < nwt; ++i) { const double wv = wt_at(i); if (wv > 0.0) { istar = i; rel = r3; } r3 += wv; }
            if (istar < 0) return (int)(hi - 1);
        }
    }
    const int64_t e0 = lo + ((int64_t)(istar >> 2) * 256 + (istar & 3) * 64 + lane) * EPV;
    float p[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) p[j] = 0.f;
    if (e0 < hi) rs_row_probs_from_vec<DT>(row, rs_load_vec<DT>(row, e0), e0, S > 0.0 ? 1.0 / S : 0.0, tab, p);
    double a = 0.0;
#pragma unroll
    for (int j = 0; j < EPV; ++j) a += (double)p[j];
    const double incl = wave_incl_scan_f64(a, lane);
    const double ex = rel + wave_shift_up_f64(incl, lane);   // relative running sum in front of this lane's vector
    const unsigned long long crossing = __ballot(before + (rel + incl) > thr);
    int src = crossing ? __builtin_ctzll(crossing) : -1;
    if (src < 0) {                                           // rounding only: last lane with mass
        const unsigned long long pos = __ballot(a > 0.0);
        src = pos ? 63 - __builtin_clzll(pos) : 63;
    }
    int hit = -1, lastpos = -1;
    {
        double rr = ex;
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
            rr += (double)p[j];
            if (p[j] > 0.f) lastpos = j;
            if (hit < 0 && before + rr > thr) hit = j;
        }
        if (hit < 0) hit = lastpos >= 0 ? lastpos : EPV - 1;
    }
    int64_t e = e0 + hit;
    if (e >= row.V) e = row.V - 1;
    return __shfl((int)e, src, 64);
}

struct RsPickShared {
    double tab[64];
    double red[4];
    double seg[RS_SEG];
    double wtx[4];
    unsigned long long best[4];
    int pick;
};

// The same draw by the whole workgroup walking the crossing segment tile by tile (vocabularies whose segments do not fit
// the wave-tile table).  Forms its sums as rs_seg_prob_sums' flat branch does.
template <int DT>
__device__ int rs_pick_wg(const RsRow &row, const double *segsum, double S, float u, RsPickShared &sh) {
    constexpr int EPV = Elem<DT>::EPV;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __syncthreads();
    if (tid < RS_SEG) sh.seg[tid] = ld_agent_f64(segsum + tid);
    if (tid == 0) sh.pick = 0x7FFFFFFF;
    __syncthreads();
    double total = 0.0;
#pragma unroll
    for (int s = 0; s < RS_SEG; ++s) total += sh.seg[s];
    const double thr = (double)u * total;
    int sstar = -1;
    double before = 0.0, run = 0.0;
#pragma unroll
    for (int s = 0; s < RS_SEG; ++s) {
        const double nx = run + sh.seg[s];
        if (sstar < 0 && nx > thr) { sstar = s; before = run; }
        run = nx;
    }
    if (sstar < 0) return (int)(row.V - 1);
    const int64_t segE = rs_seg_elems(row.V, EPV);
    const int64_t lo = (int64_t)sstar * segE;
    int64_t hi = lo + segE;
    if (hi > row.V) hi = row.V;
    const int ntiles = (int)((hi - lo + 256 * EPV - 1) / (256 * EPV));
    const double invS = S > 0.0 ? 1.0 / S : 0.0;
    double base = 0.0;
    bool found = false;
    int cand = 0x7FFFFFFF;
    for (int k = 0; k < ntiles; ++k) {
        const int64_t e0 = lo + ((int64_t)k * 256 + tid) * EPV;
        float p[EPV];
#pragma unroll
        for (int j = 0; j < EPV; ++j) p[j] = 0.f;
        if (e0 < hi) rs_row_probs_from_vec<DT>(row, rs_load_vec<DT>(row, e0), e0, invS, sh.tab, p);
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < EPV; ++j) a += (double)p[j];
        const double in = wave_incl_scan_f64(a, lane);
        const double up = wave_shift_up_f64(in, lane);
        __syncthreads();
        if (lane == 63) sh.wtx[wave] = in;
        __syncthreads();
        double wb = base;
        for (int q = 0; q < wave; ++q) wb += sh.wtx[q];
        if (!found && before + (wb + in) > thr) {
            double rr = wb + up;
            int hit = -1, lastpos = -1;
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                rr += (double)p[j];
                if (p[j] > 0.f) lastpos = j;
                if (hit < 0 && before + rr > thr) hit = j;
            }
            if (hit < 0) hit = lastpos >= 0 ? lastpos : EPV - 1;
            int64_t e = e0 + hit;
            if (e >= row.V) e = row.V - 1;
            cand = (int)e;
            found = true;
        }
        base += (sh.wtx[0] + sh.wtx[1]) + (sh.wtx[2] + sh.wtx[3]);
    }
    if (found) atomicMin(&sh.pick, cand);
    __syncthreads();
    int pick = sh.pick;
    if (pick == 0x7FFFFFFF) pick = (int)(hi - 1);           // rounding only: the segment sum said it crosses here
    return pick;
}

// argmax of the distribution with `proposed` masked (JDN:147-153 / JDO:164-168): first index of the largest probability —
// (this version forms EVERY probability: rows on the plain float32 formula; exact rows take rs_masked_argmax below)
// for bf16 logits that is a tie among every id whose ROUNDED probability equals the maximum; all mass on it -> keep it.
template <int DT>
__device__ int rs_masked_argmax_full(const RsRow &row, int64_t proposed, double S, RsPickShared &sh) {
    constexpr int EPV = Elem<DT>::EPV;
    const int tid = threadIdx.x;
    const double invS = S > 0.0 ? 1.0 / S : 0.0;
    unsigned long long best = 0ull;
    for (int64_t e0 = (int64_t)tid * EPV; e0 < row.V; e0 += 256 * EPV) {
        float p[EPV];
        rs_row_probs_from_vec<DT>(row, rs_load_vec<DT>(row, e0), e0, invS, sh.tab, p);
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

// The same for a row with an exact sum, without forming 152 064 float64 exps (~100 us for one workgroup — and the case is not
// rare with a trained model: a proposal that holds 0.97 of the mass and is rejected collides in all 16 draws).  The exactly
// rounded probability is a non-decreasing function of the scaled logit, so the largest probability belongs to the largest
// other logit; ids with smaller logits can only TIE with it after rounding, and only within ln 3 of it (the widest ratio two
// values can have and still round to the same number: 0.5 and 1.5 units of the smallest subnormal).  Phase A left every
// segment's largest scaled logit in the workspace (segmax): the largest other logit is their maximum — the proposed id's own
// segment is scanned again without it when it holds that segment's maximum — and only segments whose maximum lies within that
// distance of it (a unit in the last place for a normal pmax: almost always one segment) can hold the answer: they are scanned in vocabulary order (one round of loads per segment: 20 KB for a workgroup),
// exact probabilities for the few ids in range, first index that equals the maximum.  3-8 us instead of ~100.
template <int DT, bool FIND /* false: largest scaled logit without `proposed`; true: first id in [lo_x, inf) whose probability is pmax */>
__device__ __forceinline__ void rs_scan_segment(const RsRow &row, int seg, int64_t proposed, float lo_x, float pmax, double invS, const double *tab,
                                                float &best, unsigned long long &cand) {
    constexpr int EPV = Elem<DT>::EPV, NB = 5;
    const int64_t segE = rs_seg_elems(row.V, EPV);
    const int64_t lo = (int64_t)seg * segE;
    int64_t hi = lo + segE;
    if (hi > row.V) hi = row.V;
    for (int64_t b0 = lo + (int64_t)threadIdx.x * EPV; b0 < hi; b0 += (int64_t)NB * 256 * EPV) {
        u32x4 v[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) { const int64_t e0 = b0 + (int64_t)k * 256 * EPV; if (e0 < hi) v[k] = rs_load_vec<DT>(row, e0); }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int64_t e0 = b0 + (int64_t)k * 256 * EPV;
            if (e0 >= hi) continue;
            float xs[EPV];
            rs_scaled_from_vec<DT>(row, v[k], xs);
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                const int64_t i = e0 + j;
                if (i >= row.V || i == proposed) continue;
                if constexpr (!FIND) best = fmaxf(best, xs[j]);
                else if (xs[j] >= lo_x && (unsigned long long)i < cand) {
                    const float pj = rs_round_prob<DT>(rs_e64(xs[j], (double)row.M, tab) * invS);
                    if (pj == pmax) cand = (unsigned long long)i;
                }
            }
        }
    }
}
template <int DT, bool AGENT>
__device__ int rs_masked_argmax(const RsRow &row, int64_t proposed, double S, RsPickShared &sh, const float *segmax) {
    if (!(S > 0.0) || !segmax || row.flt) return rs_masked_argmax_full<DT>(row, proposed, S, sh);   // (a filtered row: few ids pass x_keep, few exps)
    constexpr int EPV = Elem<DT>::EPV;
    const int tid = threadIdx.x;
    const double invS = 1.0 / S;
    const int64_t segE = rs_seg_elems(row.V, EPV);
    __syncthreads();
    if (tid < RS_SEG) sh.seg[tid] = (double)(AGENT ? ld_agent_f32(segmax + tid) : segmax[tid]);
    __syncthreads();
    const int sp = (proposed >= 0 && proposed < row.V) ? (int)(proposed / segE) : -1;
    if (sp >= 0 && sp < RS_SEG) {
        // the proposed id's segment: when the id holds the segment's maximum, the segment's largest OTHER logit takes its place
        const float xa = rs_scaled<DT>(load_f<DT>(row.p, proposed), row.t, row.inv_t, row.unit_t, row.fast);
        if (xa >= (float)sh.seg[sp]) {                           // (workgroup-uniform)
            float m = -INFINITY;
            unsigned long long none = ~0ull;
            rs_scan_segment<DT, false>(row, sp, proposed, 0.f, 0.f, invS, sh.tab, m, none);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
            if ((tid & 63) == 0) sh.red[tid >> 6] = (double)m;
            __syncthreads();
            if (tid == 0) sh.seg[sp] = fmax(fmax(sh.red[0], sh.red[1]), fmax(sh.red[2], sh.red[3]));
            __syncthreads();
        }
    }
    float best = -INFINITY;
#pragma unroll
    for (int q = 0; q < RS_SEG; ++q) best = fmaxf(best, (float)sh.seg[q]);
    if (best == -INFINITY) return (int)proposed;                 // no other id holds any mass
    const float pmax = rs_round_prob<DT>(rs_e64(best, (double)row.M, sh.tab) * invS);
    if (!(pmax > 0.f)) return (int)proposed;
    // how far below `best` a logit can lie and still round to pmax: ln 3 where pmax is (near) subnormal, else the ratio of one
    // unit in the last place (bf16: < 1 + 2^-7, float32: < 1 + 2^-22), with a margin
    const float lo_x = best - (pmax >= 7.5e-37f ? (DT == JF_BF16 ? 0.0157f : 1.0e-5f) : 1.125f);
    for (int q = 0; q < RS_SEG; ++q) {                           // vocabulary order: the first segment with a hit holds the first index
        if (!((float)sh.seg[q] >= lo_x)) continue;               // (workgroup-uniform)
        unsigned long long cand = ~0ull;
        float unused = 0.f;
        rs_scan_segment<DT, true>(row, q, proposed, lo_x, pmax, invS, sh.tab, unused, cand);
        cand = ~wave_max_u64(~cand);                             // smallest index
        __syncthreads();
        if ((tid & 63) == 0) sh.best[tid >> 6] = cand;
        __syncthreads();
        unsigned long long mm = sh.best[0];
        for (int w = 1; w < 4; ++w) mm = sh.best[w] < mm ? sh.best[w] : mm;
        if (mm != ~0ull) { __syncthreads(); return (int)mm; }
    }
    return (int)proposed;                                        // (cannot happen: the segment that holds `best` holds a hit)
}

// the draw that counts for one row, by a whole workgroup (sh.tab loaded): wavefront 0 walks at u (u >= 0); the masked argmax
// after RS_MAX_TRIES collisions (u < 0), or should the walk return the proposed token after all (float64 rounding of the
// interval test against the walk: never the proposal).
template <int DT, bool AGENT>
__device__ __forceinline__ int rs_final_pick(const RsRow &row, const RsWs &w, int item, double S, int64_t proposed, float u, RsPickShared &sh) {
    int y = -1;
    if (u >= 0.f) {
        if (rs_hier_ok(row.V, Elem<DT>::EPV)) {
            if (threadIdx.x < 64) { const int yy = rs_pick_wave<DT, AGENT>(row, w, item, S, u, sh.tab, threadIdx.x); if (threadIdx.x == 0) sh.pick = yy; }
            __syncthreads();
            y = sh.pick;
        } else {
            y = rs_pick_wg<DT>(row, w.segsum + (int64_t)item * RS_SEG, S, u, sh);
        }
    }
    if (y < 0 || (int64_t)y == proposed) y = rs_masked_argmax<DT, AGENT>(row, proposed, S, sh, w.segmax + (int64_t)item * RS_SEG);
    return y;
}

// ------------------------------------------------------------------------------------------------
// Accept/reject of every row of a batch (JDN:581-639).  The reference visits the rows in order and draws torch.rand /
// torch.multinomial / torch.randint as it goes, so the position of every draw in the injected streams depends on the rows
// before it.  Everything wide runs in parallel around one short serial walk:
//   rs_accept_kernel  (1 workgroup)   every accept test's probability (float64 exp of the gathered logit over the float32
//                                     row sum: two candidate roundings, see "Exact probabilities") and the uniforms staged
//                                     in LDS by 256 threads, then ONE wavefront walks the rows: a row's L-1 accept tests are
//                                     one ballot, so the serial chain is B steps of LDS latency, not B*(L-1).  A test whose
//                                     uniform falls between the two candidates stops the walk: the workgroup forms that
//                                     row's float64 sum, patches the entry and the walk resumes at that row.
//   rs_rowsum_a/b     (B * RS_SEG)    the rejected rows' float64 sums and exact CDF sums (above)
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

constexpr int RS_STAGE = 6144;      // accept tests / uniforms staged in LDS by the accept scan (B * (L-1) <= this, else global)
constexpr int RS_ROWS_LDS = 2048;   // rows whose scan results are kept in LDS (half of it in the chain kernel)
constexpr uint32_t RS_PE_EOS = 0x80000000u, RS_PE_AMB = 0x40000000u, RS_PE_VAL = 0x3FFFFFFFu;
constexpr uint32_t RS_PATCH_NONE = 0xFFFFFFFFu;    // (no resolved word looks like this: a resolved test never carries RS_PE_AMB)
constexpr int RS_RES_NONE = -1, RS_RES_ABORT = -2;   // s_res words of rows the walk has not decided / stopped at

struct RsAcceptIn {                 // what an accept test's probability is made of
    const void *logits; int64_t V, row_stride; float t; const float *row_max, *row_sumexp, *p_draft;
};
// One accept test as an LDS word: the LOWER candidate probability (a float <= 1: bits 31 / 30 are free), RS_PE_AMB when the
// float32 row sum cannot decide the rounding (bf16: the upper candidate is the next bf16; float32: lo * (1 + 2^-12) bounds
// it), RS_PE_EOS when the proposed token is the EOS id.  Rows without finite statistics keep jf_rs_probs' plain float32 value.
// the word of one test from jf_rs_probs' p_draft (see rs_probs_finish_kernel) and, float32 logits, the row maximum
template <int DT>
__device__ __forceinline__ uint32_t rs_accept_word(float pd, float M, bool is_eos) {
    const uint32_t eos = is_eos ? RS_PE_EOS : 0u;
    if constexpr (DT == JF_BF16) {
        const float a = fabsf(pd);
        if (a != a) return eos;                              // NaN never accepts
        return (__float_as_uint(a > 1.f ? 1.f : a) & RS_PE_VAL) | ((__float_as_uint(pd) >> 31) ? RS_PE_AMB : 0u) | eos;
    } else {
        if (!(pd > 0.f)) return eos;
        if ((__float_as_uint(M) & 0x7F800000u) == 0x7F800000u) return (__float_as_uint(pd > 1.f ? 1.f : pd) & RS_PE_VAL) | eos;   // plain formula row
        // The band [lo, hi] around the centre value that contains the exact probability: hi = lo (1 + 2^-12) while eps <= 1e-4
        // (|max / T| < ~179), lo (1 + 2^-8) up to eps = 1.9e-3 (|max / T| < ~5 400: e.g. a maximum logit of 20 at T = 0.1 — round 5;
        // before, such rows marked EVERY test undecided and resolved them one by one with a float64 row sum each).  The width is
        // the word's lowest mantissa bit (lo is a lower bound: forcing that bit only lowers it by an ulp).
        const float eps = (float)rs_eps_row(M);
        if (eps > 1.9e-3f) return RS_PE_AMB | eos;            // no band known: every rejection is resolved exactly
        const bool wide = eps > 1.0e-4f;
        const float lo = pd * (1.f - eps - 1.2e-7f);
        uint32_t lb = __float_as_uint(lo > 1.f ? 1.f : lo) & RS_PE_VAL;
        if (lb < 2u) return RS_PE_AMB | eos;
        if ((lb & 1u) != (wide ? 1u : 0u)) lb -= 1u;
        return lb | RS_PE_AMB | eos;
    }
}
template <int DT>
__device__ __forceinline__ uint32_t rs_accept_entry(const RsAcceptIn &in, int64_t i, int64_t tok, int eos_id, const double *tab) {
    (void)tab;
    return rs_accept_word<DT>(in.p_draft[i], in.row_max[i], eos_id >= 0 && tok == (int64_t)eos_id);
}
template <int DT>
__device__ __forceinline__ float rs_accept_hi(uint32_t pe) {   // upper candidate of an ambiguous entry
    const uint32_t b = pe & RS_PE_VAL;
    if constexpr (DT == JF_BF16) return __uint_as_float(b + 0x00010000u);
    else return b ? __uint_as_float(b) * ((b & 1u) ? 1.00390625f : 1.000244140625f) : INFINITY;   // (1 + 2^-12) lo >= p (1 + eps) while eps <= 1e-4, (1 + 2^-8) lo up to 1.9e-3 (flagged in bit 0); 0: no band known
}

// STAGED: the batch fits the LDS tables (B * (L-1) <= RS_STAGE, B <= RS_ROWS_LDS) — the serial part touches LDS only.
// SIG (one-launch step): the walker announces every row the moment it is decided — flag[b] = (gen << 32) | eos << 30 |
// n_accepted << 16 | (reject_pos + 2), one self-contained 8-byte agent-scope store — and the workgroup ends with a release +
// the accept-done word; the row records' n_committed / eos / n_pads / active_next then belong to the row's bonus workgroup.
template <int DT, bool STAGED, bool SIG, int ROWS_N = RS_ROWS_LDS>
__device__ __forceinline__ void rs_accept_body(const RsAcceptIn &in, const int64_t *draft, int B, int L, int eos_id,
                                               const float *u_stream, int64_t u_len, const int64_t *u_cursor,
                                               int64_t *committed, jf_rs_row *rows, const RsWs &w, uint32_t gen,
                                               uint32_t *s_p, float *s_u /* LDS, B * (L-1) entries each when STAGED */) {
    __shared__ int s_res[STAGED ? ROWS_N : 1];                                 // nacc | eos << 15 | (rej + 1) << 16 per row
    __shared__ int s_off[STAGED ? ROWS_N + 1 : 1];                             // stream offset of every row's first test (-1: not known yet)
    __shared__ int s_pipe;                                                     // first row the offset chain (not a leading run) handles, -1: not yet
    __shared__ int s_eunc[2], s_erow[2], s_eoff[2];                            // the evaluating wavefronts' first undecided test (index, row, its offset)
    __shared__ double s_tab[64], s_red[4];
    __shared__ int s_unc, s_resume, s_used;
    // !STAGED: the resolved tests of the row the walk stands at, one word per position of the row (RS_PATCH_NONE = not resolved),
    // in the part of the workspace the later launches overwrite (w.wtsum).  A row can need a patch for every accepted test in
    // front of its stop, and the rows in front of it never need theirs again: a per-ROW table (round 4 kept eight entries for the
    // whole batch and overwrote the last one when full — a row with two live patches then never finished: ADVICE r04).
    uint32_t *g_patch = (uint32_t *)w.wtsum;
    __shared__ int s_patch_row;
    const int tid = threadIdx.x;
    const int W = L - 1;
    const int n = B * W;
    const int64_t uc0 = *u_cursor;
    auto tok_at = [&](int i) { const int b = i / W; return draft[(int64_t)b * L + (i - b * W) + 1]; };
    rs_load_tab(s_tab);
    if (tid == 0) s_patch_row = -1;
    __syncthreads();
    if constexpr (STAGED) {
        const int ul = (int)u_len, ub = (int)(uc0 % u_len);
        // every test's word and uniform in ONE round of loads (at most n uniforms can be used); the loads that do not wait for
        // the stream cursor are issued first, and the stream index wraps by subtraction (an integer division is ~40 instructions)
        for (int i0 = tid; i0 < n; i0 += 8 * 256) {
            int64_t tk[8];
            float M[8], pd[8], uu[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k * 256;
                tk[k] = -1; M[k] = 0.f;
                if (i < n) {
                    pd[k] = in.p_draft[i];
                    if (eos_id >= 0) tk[k] = tok_at(i);
                    if constexpr (DT == JF_F32) M[k] = in.row_max[i];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k * 256;
                if (i < n) {
                    unsigned x = (unsigned)ub + (unsigned)i;
                    if (x >= (unsigned)ul) x -= (unsigned)ul;
                    if (x >= (unsigned)ul) x %= (unsigned)ul;       // streams shorter than the batch's tests (tests only)
                    uu[k] = u_stream[x];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k * 256;
                if (i < n) { s_u[i] = uu[k]; s_p[i] = rs_accept_word<DT>(pd[k], M[k], eos_id >= 0 && tk[k] == (int64_t)eos_id); }
            }
        }
    }
    __syncthreads();
    int b_start = 0, used_start = 0;
    for (;;) {                                                      // the walk; re-entered after an undecided test was resolved
        if constexpr (STAGED) {                                     // (the evaluating / announcing wavefronts poll these words)
            for (int b = b_start + tid; b < B; b += 256) { s_res[b] = RS_RES_NONE; s_off[b + 1] = -1; }
            if (tid == 0) { s_off[b_start] = -1; s_pipe = -1; s_eunc[0] = s_eunc[1] = -1; s_erow[0] = s_erow[1] = 0x7FFFFFFF; }
            __syncthreads();
        }
        if (tid < 64 && STAGED && W <= 64) {
            // one ballot per row; the row's words were loaded while the row before it was decided, so the only dependent
            // access of a step is the uniform at this row's stream offset.  SIG: the row's result goes to LDS only — wavefront 1
            // announces it (below): composing and storing the flag word is off this wavefront's instruction stream (~0.1 us per row)
            const int lane = tid;
            const unsigned long long rowmask = W >= 64 ? ~0ull : ((1ull << W) - 1ull);
            const int lw = lane < W ? lane : W - 1;                 // lanes past the row read a valid entry and are masked out of every ballot
            int off = used_start, b = b_start;
            // Leading runs of rows at once, a lane per row, on the assumption that every row of the run STOPS AT ITS FIRST TEST (the
            // proposal at position 0 is rejected — what a draft that is not yet right gets — or is an accepted EOS): then each of
            // them consumes exactly one uniform and the offsets of the whole run are known.  A run ends in front of the first
            // row that accepts its first proposal (or whose first test is undecided).
            for (;;) {
                const int rb = b + lane;
                bool ok0 = false, rej0 = false;
                if (rb < B) {
                    const uint32_t pe0 = s_p[rb * W];
                    const float u0 = s_u[off + lane];
                    rej0 = !(u0 < __uint_as_float(pe0 & RS_PE_VAL));
                    const bool unc0 = rej0 && (pe0 & RS_PE_AMB) && u0 < rs_accept_hi<DT>(pe0);
                    ok0 = (rej0 || (pe0 & RS_PE_EOS)) && !unc0;
                }
                const unsigned long long nok = ~__ballot(ok0);
                const int run = nok ? __builtin_ctzll(nok) : 64;
                if (lane < run)                                     // rejected at position 0: nothing accepted; else the EOS at position 0 was accepted
                    __hip_atomic_store(&s_res[rb], rej0 ? (1 << 16) : (1 | (1 << 15)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                b += run;
                off += run;
                if (run < 64 || b >= B) break;
            }
            // From here on this wavefront does NOTHING but the dependent chain — where does the next row's stream of uniforms
            // start: one LDS read at the row's offset, one ballot against the row's thresholds (the next row's words are read a
            // row ahead; an EOS word is a negative float, never accepted: "stops here" either way), find-first, add — and drops
            // the offset into LDS (~0.1 us per row).  Wavefronts 2 and 3 pick the offsets up, alternate rows, and work out what the
            // row's stop MEANS (rejected / EOS accepted / undecided rounding: s_res), wavefront 1 announces the leading run of
            // decided rows: the three stages overlap, a walked batch of 64 rows is through in ~8 us instead of 15.  An undecided
            // first stop is found by the evaluating wavefronts; the offsets behind it were speculation and are formed again
            // after the test has been resolved.
            if (lane == 0) {
                __hip_atomic_store(&s_off[b < B ? b : B], off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&s_pipe, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (b < B) {
                uint32_t pe = s_p[b * W + lw];
                while (b < B) {
                    const float uu = s_u[off + lw];
                    const int bn = b + 1 < B ? b + 1 : b;
                    const uint32_t pen = s_p[bn * W + lw];
                    const unsigned long long stop = ~__ballot(uu < __uint_as_float(pe & ~RS_PE_AMB)) & rowmask;
                    const int f = stop ? __builtin_ctzll(stop) : W - 1;    // no stop: all W tests consumed
                    off += f + 1;
                    ++b;
                    if (lane == 0) __hip_atomic_store(&s_off[b], off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    pe = pen;
                }
            }
        } else if (STAGED && W <= 64 && tid >= 128) {
            // wavefronts 2 and 3: what each row's stop means, alternate rows, as soon as the row's offset is known
            const int lane = tid & 63, e = (tid >> 6) - 2;
            const unsigned long long rowmask = W >= 64 ? ~0ull : ((1ull << W) - 1ull);
            const int lw = lane < W ? lane : W - 1;
            int start;
            while ((start = __hip_atomic_load(&s_pipe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < 0) __builtin_amdgcn_s_sleep(1);
            int my_unc = -1, my_row = 0x7FFFFFFF, my_off = 0;
            for (int r = start + e; r < B; r += 2) {
                int o;
                while ((o = __hip_atomic_load(&s_off[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < 0) __builtin_amdgcn_s_sleep(0);
                const uint32_t pe = s_p[r * W + lw];
                const float uu = s_u[o + lw];
                const unsigned long long accm = __ballot(uu < __uint_as_float(pe & RS_PE_VAL));
                const unsigned long long eosm = __ballot((pe & RS_PE_EOS) != 0u);
                const unsigned long long rejm = ~accm & rowmask;
                const unsigned long long stopm = (rejm | eosm) & rowmask;
                int res = W;                                        // no stop: all W accepted (eos 0, rej -1)
                if (stopm) {
                    const int f = __builtin_ctzll(stopm);
                    if ((rejm >> f) & 1ull) {
                        const unsigned long long uncm = __ballot((pe & RS_PE_AMB) != 0u && uu < rs_accept_hi<DT>(pe));
                        if ((uncm >> f) & 1ull) { my_unc = r * W + f; my_row = r; my_off = o; break; }   // the first stop is undecided: resolve it
                        res = f | ((f + 1) << 16);                  // rejected at f: f accepted
                    } else {
                        res = (f + 1) | (1 << 15);                  // EOS accepted at f
                    }
                }
                if (lane == 0) __hip_atomic_store(&s_res[r], res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (lane == 0) {
                s_eunc[e] = my_unc; s_erow[e] = my_row; s_eoff[e] = my_off;
                if (SIG && my_unc >= 0) __hip_atomic_store(&s_res[my_row], RS_RES_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else if (SIG && STAGED && W <= 64 && tid >= 64 && tid < 128) {
            // wavefront 1: announces the rows as wavefront 0 decides them — a lane per row of the leading run of decided rows
            const int lane = tid - 64;
            int base = b_start;
            while (base < B) {
                const int r = base + lane;
                const int v = r < B ? __hip_atomic_load(&s_res[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : RS_RES_NONE;
                const unsigned long long nr = ~__ballot(v >= 0);
                const int run = nr ? __builtin_ctzll(nr) : 64;
                if (lane < run) {
                    const int nacc = v & 0x7FFF, eos = (v >> 15) & 1, rej = (v >> 16) - 1;
                    __hip_atomic_store(w.flag + (int64_t)r * RS_FLAG_STRIDE, ((unsigned long long)gen << 32) | ((unsigned long long)eos << 30) |
                                       ((unsigned long long)nacc << 16) | (unsigned long long)(rej + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    RS_ROWSTAMP(0, r);
                }
                base += run;
                if (run < 64 && __shfl(v, run, 64) == RS_RES_ABORT) break;      // an evaluating wavefront stopped at an undecided test
                if (run == 0) __builtin_amdgcn_s_sleep(1);
            }
#ifdef JF_EXP_RS_TRACE
            if (base >= B && lane == 0) atomicMax(&g_rstrace[15], (unsigned long long)__builtin_amdgcn_s_memrealtime());   // 15: the last row is decided and announced
#endif
        } else if (tid < 64) {
            const int lane = tid;
            int used_total = used_start, unc_at = -1, b = b_start;
            for (; b < B; ++b) {                                    // JDN:326-348, rows in order
                int nacc = 0, eos = 0, rej = -1, used = 0;
                for (int t0 = 0; t0 < W; t0 += 64) {
                    const int tt = t0 + lane;
                    bool stop = false, rejb = false, unc = false;
                    if (tt < W) {
                        const int i = b * W + tt;
                        uint32_t pe;
                        float uu;
                        if constexpr (STAGED) { pe = s_p[i]; uu = s_u[used_total + tt]; }
                        else {
                            pe = rs_accept_entry<DT>(in, i, tok_at(i), eos_id, s_tab);
                            if (b == s_patch_row) {
                                const uint32_t q = __hip_atomic_load(g_patch + tt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (q != RS_PATCH_NONE) pe = q;
                            }
                            uu = u_stream[(uc0 + used_total + tt) % u_len];
                        }
                        const bool sure = uu < __uint_as_float(pe & RS_PE_VAL);
                        rejb = !sure;
                        unc = rejb && (pe & RS_PE_AMB) && uu < rs_accept_hi<DT>(pe);
                        stop = rejb || (pe & RS_PE_EOS);            // rejected, or accepted EOS
                    }
                    const unsigned long long bal = __ballot(stop);
                    if (bal) {
                        const int f = __builtin_ctzll(bal);
                        if ((__ballot(unc) >> f) & 1ull) { unc_at = b * W + t0 + f; break; }   // the first stop is undecided: resolve it
                        const bool is_rej = (__ballot(rejb) >> f) & 1ull;
                        if (is_rej) { rej = t0 + f; nacc = t0 + f; } else { eos = 1; nacc = t0 + f + 1; }
                        used = t0 + f + 1;
                        break;
                    }
                    const int wd = (W - t0) < 64 ? (W - t0) : 64;
                    nacc = t0 + wd; used = t0 + wd;
                }
                if (unc_at >= 0) break;
                if (lane == 0) {
                    if constexpr (STAGED) s_res[b] = nacc | (eos << 15) | ((rej + 1) << 16);
                    else { rows[b].n_committed = nacc; rows[b].eos = eos; rows[b].reject_pos = rej; }
                    if constexpr (SIG) {
                        __hip_atomic_store(w.flag + (int64_t)b * RS_FLAG_STRIDE,
                                           ((unsigned long long)gen << 32) | ((unsigned long long)eos << 30) | ((unsigned long long)nacc << 16) |
                                               (unsigned long long)(rej + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        RS_ROWSTAMP(0, b);
                    }
                }
                used_total += used;
            }
            if (lane == 0) { s_unc = unc_at; s_resume = b; s_used = used_total; }
            if constexpr (SIG) { if (unc_at < 0) RS_STAMP_MAX(15); }   // 15: the accept walk has decided the last row
        }
        __syncthreads();
        if (STAGED && W <= 64) {                                    // the evaluating wavefronts' first undecided tests: the earlier row is the one to resolve
            if (tid == 0) {
                const int e = s_erow[0] <= s_erow[1] ? 0 : 1;
                s_unc = s_eunc[e]; s_resume = s_erow[e]; s_used = s_eoff[e];
            }
            __syncthreads();
        }
        const int ui = s_unc;
        if (ui < 0) break;
        // ---- the float32 row sum cannot decide this test: the row's float64 sum, by the whole workgroup (~1e-4 p of the tests)
        const int64_t tk = tok_at(ui);
        const float pex = rs_exact_prob_wg<DT>(in.logits, ui, in.V, in.row_stride, in.t, in.row_max[ui], tk, s_tab, s_red);
        if (tid == 0) {
            const uint32_t word = (__float_as_uint(pex > 1.f ? 1.f : pex) & RS_PE_VAL) | ((eos_id >= 0 && tk == (int64_t)eos_id) ? RS_PE_EOS : 0u);
            if constexpr (STAGED) s_p[ui] = word;
        }
        if constexpr (!STAGED) {
            const int prow = ui / W;
            if (prow != s_patch_row) {                              // the walk has moved on to another row: its table starts empty
                for (int t = tid; t < W; t += 256) __hip_atomic_store(g_patch + t, RS_PATCH_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                if (tid == 0) s_patch_row = prow;
            }
            if (tid == 0) {
                const uint32_t word = (__float_as_uint(pex > 1.f ? 1.f : pex) & RS_PE_VAL) | ((eos_id >= 0 && tk == (int64_t)eos_id) ? RS_PE_EOS : 0u);
                __hip_atomic_store(g_patch + (ui - prow * W), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        b_start = s_resume; used_start = s_used;
        __syncthreads();
    }
    __threadfence_block();
    // everything else about a row in parallel: row record, the rejected row's work item, the accepted tokens
    auto res_of = [&](int b, int &nacc, int &eos, int &rej) {
        if constexpr (STAGED) { const int r = s_res[b]; nacc = r & 0x7FFF; eos = (r >> 15) & 1; rej = (r >> 16) - 1; }
        else { nacc = rows[b].n_committed; eos = rows[b].eos; rej = rows[b].reject_pos; }
    };
    if (tid < 64 && !SIG) {                                         // rejected rows in front of each row: ballot prefix, 64 rows a pass
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
        rw.reject_pos = rej;
        rw.n_uniforms = rej >= 0 ? rej + 1 : nacc;                  // one uniform per tested position (JDN:329)
        if constexpr (!SIG) { rw.n_committed = nacc; rw.eos = eos; rw.n_bonus_draws = 0; rw.n_pads = 0; rw.active_next = 0; }
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

template <int DT, bool STAGED>
__global__ __launch_bounds__(256) void rs_accept_kernel(RsAcceptIn in, const int64_t *draft, int B, int L, int eos_id,
                                                         const float *u_stream, int64_t u_len, const int64_t *u_cursor,
                                                         int64_t *committed, jf_rs_row *rows, RsWs w) {
    __shared__ uint32_t s_p[STAGED ? RS_STAGE : 1];
    __shared__ float s_u[STAGED ? RS_STAGE : 1];
    rs_accept_body<DT, STAGED, false>(in, draft, B, L, eos_id, u_stream, u_len, u_cursor, committed, rows, w, 0u, s_p, s_u);
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
// single-workgroup launch between the row sums and this one (batches up to RS_BONUS_CHAIN_ROWS rows with a staged window;
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
    rs_load_tab(sh.tab);
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
        __syncthreads();
        u_final = w.pick_u[b];
    }
    const int64_t r = (int64_t)b * (L - 1) + rej;
    const RsRow row = rs_step_row<DT>(logits, r, V, row_stride, temp, row_max, row_sumexp, w.filt);
    const int bonus = rs_final_pick<DT, false>(row, w, b, w.s64[b], draft[(int64_t)b * L + rej + 1], u_final, sh);
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

// next-draft shape of a row that keeps decoding with n committed tokens (JDN:444-466 / 619-638)
__device__ __forceinline__ void rs_next_shape(int n, int L, int &off, int &copy_len) {
    const int acc_len = 1 + n;
    off = 0; copy_len = 1;
    if (acc_len < L) {
        off = acc_len > 1 ? acc_len - 1 : 1;
        const int rem = (L - 1) - off;
        copy_len = rem < L - 1 ? rem : L - 1;
    } else {
        off = L - 2;
    }
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
            int off, copy_len;
            rs_next_shape(n, L, off, copy_len);
            n_pads = L - 1 - copy_len;
        }
        rw.n_pads = n_pads;
    }
    __threadfence_block();
    __syncthreads();
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
    const int64_t pc0 = *pad_cursor;
    for (int64_t idx = tid; idx < (int64_t)B * L; idx += 256) {        // every next-draft element independently
        const int b = (int)(idx / L), i = (int)(idx - (int64_t)b * L);
        const jf_rs_row rw = rows[b];
        if (!rw.active_next) continue;
        const int64_t r0 = (int64_t)b * (L - 1);
        const int n = rw.n_committed;
        int off, copy_len;
        rs_next_shape(n, L, off, copy_len);
        int64_t v;
        if (i == 0) v = committed[(int64_t)b * L + n - 1];
        else if (i - 1 < copy_len) v = jfmb::decode_packed(packed[r0 + off + (i - 1)]);
        else v = pad_stream[(pc0 + rw.rsv + (i - 1 - copy_len)) % pad_len];
        next_draft[idx] = v;
    }
    __syncthreads();
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
//   block 2                    the end: waits for every row's finish word, then the pad offsets (one scan), the pads themselves
//                              and the stream cursors — all that depends on more than one row
//   blocks 3 ..                nact per row (nact = the segments that hold any element: 15 of RS_SEG = 16 at V = 152 064), block
//                              3 + b * nact + s = segment s of row b: wait for the row's flag; if the row was rejected, phase A,
//                              the exchange of the float64 partials with the row's other workgroups, phase B (the last active
//                              segment's workgroup also stores the zeros of the empty segments).  SEGMENT 0's workgroup then
//                              stays as the row's own: the bonus draw (ONE wavefront walks the hierarchical sums for the
//                              uniform the chain hands it, or the masked argmax), then everything about the row that depends
//                              on this row alone — EOS, row record, the next draft's seed and greedy tail, its argmax slots
//                              re-zeroed, the row's finish word.  (A row that was not rejected: its segment 0 finishes it at once.)
// 64 rows are 3 + 960 workgroups: all resident at once at four per CU.  (Until round 4 every row had a sixteenth, empty
// segment workgroup and a separate bonus workgroup, 1 091 in all: the last 67 — rows 60-63 — started when the first
// ones left, ~17 us late, and the launch ended with them: profiles/rs_step_r04.txt.)  Waits only point backwards in dispatch
// order — a row's workgroups wait for block 0 and for each other (consecutive ids), segment 0 for the chain, the chain for the
// rows in order — so the launch cannot starve itself whatever the device keeps resident; jf_rs_step still asks the runtime
// that 3 + 2 * RS_SEG workgroups of the kernel fit at once.  Every wait is bounded (2 s): a row that
// times out reports JF_E_LAUNCH through rows[0].rsv.  All hand-off words carry the call's generation number (nothing to
// re-zero, no stale reads); payloads cross workgroups as agent-scope atomics (a release fence per producer would write back
// an L2 full of freshly written logits — profiles/verify_release_ab_r03.txt).
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
__device__ __forceinline__ void rs_report_timeout(jf_rs_row *rows) {
    __hip_atomic_store(&rows[0].rsv, (int32_t)JF_E_LAUNCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int DT, bool FLT /* the rows carry jf_rs_filter's records (a.w.filt): the unfiltered kernel compiles without those paths */>
__global__ __launch_bounds__(256, 4) void rs_step_fused_kernel(RsFusedArgs a) {   // 4 workgroups per CU (<= 128 VGPRs, < 40 KB LDS): 64 rows' 963 workgroups are resident at once
    const int blk = blockIdx.x, tid = threadIdx.x;
    const int B = a.B, L = a.L, W = a.L - 1;
    const RsWs &w = a.w;
    // the roles' large LDS tables share one buffer (a workgroup has one role): accept words + uniforms (32 KB), the chain's
    // staged uniforms (8 KB), the end's pad window (8 KB) — a sum of them would cost the short roles their residency
    __shared__ __attribute__((aligned(16))) unsigned char s_big[2 * RS_FUSED_STAGE * 4];
    // (raised wave priority — s_setprio 3 — for the role workgroups and a row's own workgroup, which share their CUs with three
    // segment workgroups each, changed nothing: the serial paths are their own scalar bookkeeping and round trips)
    if (blk == 0) {
        RS_STAMP_MIN(0);                                             // 0: launch start (accept workgroup)
        const RsAcceptIn in{a.logits, a.V, a.row_stride, a.t, a.row_max, a.row_sumexp, a.p_draft};
        rs_accept_body<DT, true, true, RS_FUSED_ROWS>(in, a.draft, B, L, a.eos_id, a.u_stream, a.u_len, a.u_cursor,
                                                      a.committed, a.rows, w, a.gen, (uint32_t *)s_big, (float *)(s_big + RS_FUSED_STAGE * 4));
        RS_STAMP_MAX(1);                                             // 1: accept workgroup done (records + accept-done word)
        return;
    }
    if (blk == 1) {                                                 // ---- the chain: draws of all rows in stream order
        // Wavefronts 1-3 turn rows into CDF intervals as their flags and segment sums come in (a thread per row); wavefront 0
        // counts the draws in row order on LDS and hands every rejected row its uniform the moment the rows before it are
        // counted — the walk of row i starts a few us after row i was decided, not after the last row's sums.
        float *s_u = (float *)s_big;                                 // [RS_MAX_TRIES * RS_FUSED_ROWS]
        double *s_tot = (double *)(s_big + RS_MAX_TRIES * RS_FUSED_ROWS * 4), *s_lo = s_tot + RS_FUSED_ROWS, *s_hi = s_lo + RS_FUSED_ROWS;   // s_tot < 0: not rejected
        unsigned long long *s_hand = (unsigned long long *)(s_hi + RS_FUSED_ROWS);   // per row: 0 = not counted yet, else HAND_* | draws << 32 | uniform bits
        int *s_ready = (int *)(s_hand + RS_FUSED_ROWS);
        static_assert(RS_MAX_TRIES * RS_FUSED_ROWS * 4 + 4 * RS_FUSED_ROWS * 8 + RS_FUSED_ROWS * 4 <= 2 * RS_FUSED_STAGE * 4, "chain tables");
        constexpr unsigned long long HAND_REJ = 1ull << 63, HAND_SKIP = 1ull << 62;
        const int64_t bc0 = *a.b_cursor;
        const bool staged = a.b_len < 0x7FFFFFFFll;
        for (int i = tid; i < B; i += 256) { s_ready[i] = 0; s_hand[i] = 0ull; }
        if (staged) {                                                // every entry the walk can touch: RS_MAX_TRIES per row
            const int bl = (int)a.b_len, bb = (int)(bc0 % a.b_len);
            batched_for<8, float>(RS_MAX_TRIES * B, tid, 256, [&](int64_t i) { return a.b_stream[(bb + (int)i) % bl]; }, [&](int64_t i, float v) { s_u[i] = v; });
        }
        __syncthreads();
        if (tid >= 192) {
            // wavefront 3: hands the counted rows over — a lane per row of the leading run of rows wavefront 0 has counted: the
            // two global stores per row (and their address arithmetic) are off the counting wavefront's instruction stream
            const int lane = tid - 192;
            int base = 0;
            while (base < B) {
                const int r = base + lane;
                const unsigned long long hw = r < B ? __hip_atomic_load(&s_hand[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0ull;
                const unsigned long long nr = ~__ballot(hw != 0ull);
                const int run = nr ? __builtin_ctzll(nr) : 64;
                if (lane < run && (hw & HAND_REJ)) {
                    __hip_atomic_store(&a.rows[r].n_bonus_draws, (int)((hw >> 32) & 0xFFFFu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the end workgroup
                    __hip_atomic_store(w.pick + r, ((unsigned long long)a.gen << 32) | (hw & 0xFFFFFFFFull), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);   // the uniform for the row's bonus workgroup: a self-contained word
                    RS_ROWSTAMP(3, r);
                }
                base += run;
                if (run == 0) __builtin_amdgcn_s_sleep(1);
            }
#ifdef JF_EXP_RS_TRACE
            if (lane == 0) atomicMax(&g_rstrace[7], (unsigned long long)__builtin_amdgcn_s_memrealtime());   // 7: last row handed its uniform
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every count is performed before the word that says so
            if (lane == 0) __hip_atomic_store(w.acceptdone + 1, a.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (tid >= 64) {
            // a thread per row (B <= RS_FUSED_ROWS = 128: wavefronts 1 and 2), every lane polling for ITS row without blocking the
            // others of its wavefront: a lane that waited in a loop of its own would hold all 64 rows back until the last of them is in
            const int i = tid - 64;
            bool fin = i >= B;
            int rp = -3;                                             // -3: the row's flag has not been seen yet
            unsigned spins = 0;
            unsigned long long t0 = 0;
            for (;;) {                                               // wave-uniform exit (the ballot below): with a per-lane `while (!fin)` the
                if (!fin) {                                          // compiler sinks a lane's publishing behind the loop, i.e. behind ALL 64 rows
                if (rp == -3) {
                    const unsigned long long v = __hip_atomic_load(w.flag + (int64_t)i * RS_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)(v >> 32) == a.gen) rp = (int)(v & 0xFFFFull) - 2;
                }
                if (rp != -3) {
                    const bool sums_in = rp < 0 || __hip_atomic_load(w.ivdone + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.gen;
                    if (sums_in) {
                        double t_ = -1.0, lo_ = 0.0, hi_ = 0.0;
                        if (rp >= 0) { t_ = ld_agent_f64(w.iv + 3 * (int64_t)i); lo_ = ld_agent_f64(w.iv + 3 * (int64_t)i + 1); hi_ = ld_agent_f64(w.iv + 3 * (int64_t)i + 2); }
                        s_tot[i] = t_; s_lo[i] = lo_; s_hi[i] = hi_;
                        __hip_atomic_store(&s_ready[i], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        fin = true;
                    }
                }
                }
                if (__ballot(!fin) == 0ull) break;
                __builtin_amdgcn_s_sleep(4);
                if ((++spins & 4095u) == 0u) {                       // bounded: a row whose sums never arrive is reported, not waited for
                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                    if (t0 == 0) t0 = now;
                    else if (now - t0 > RS_WAIT_TICKS) {
                        if (!fin) { s_tot[i] = -1.0; rs_report_timeout(a.rows); __hip_atomic_store(&s_ready[i], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); fin = true; }
                    }
                }
            }
            return;
        }
        // wavefront 0: one look at the ready words of the next 64 rows (a lane each), then every row of the leading run of ready
        // ones without polling again — a poll per row (an acquire on LDS each) was 0.7 us per row.  The run's intervals are
        // loaded a lane per row and broadcast from registers: the one dependent access of a row's step is its uniforms.
        int off = 0, i = 0;
        while (i < B) {
            const int r = i + tid;
            const bool rdy = r < B && __hip_atomic_load(&s_ready[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
            const unsigned long long nr = ~__ballot(rdy);
            const int run = nr ? __builtin_ctzll(nr) : 64;           // rows i .. i + run - 1 are ready
            if (run == 0) { __builtin_amdgcn_s_sleep(1); continue; }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // their intervals were stored before their ready words
            const double my_tot = tid < run ? s_tot[r] : -1.0, my_lo = tid < run ? s_lo[r] : 0.0, my_hi = tid < run ? s_hi[r] : 0.0;
            auto lane_f64 = [](double v, int k) {               // k is wave-uniform: two v_readlane (a shuffle goes through the LDS crossbar)
                return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), k), __builtin_amdgcn_readlane(__double2loint(v), k));
            };
            auto u_at = [&](int x) { return staged ? s_u[x] : a.b_stream[(bc0 + x) % a.b_len]; };
            auto hand = [&](int row, int draws, float uf) {      // counted: wavefront 3 stores the count and the uniform where they are read
                __hip_atomic_store(&s_hand[row], HAND_REJ | ((unsigned long long)draws << 32) | (unsigned long long)__float_as_uint(uf), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
            };
            // The run in parallel first, a lane per row, on the assumption that every rejected row in front takes ONE draw (no
            // collision with its proposed token: the usual case — a rejected proposal rarely holds much mass): the rows up to the
            // first lane whose first draw collides are final at once.  From that row on the run is counted row by row (<= 16
            // draws each, all lanes): where collisions are frequent a second parallel attempt would only cost its own pass.
            {
                const bool act = tid < run && !(my_tot < 0.0);      // (a NaN total is still a rejected row)
                const unsigned long long actm = __ballot(act);
                const int before = __builtin_popcountll(actm & ((1ull << tid) - 1ull));
                float u1 = 0.f;
                bool coll = false;
                if (act) { u1 = u_at(off + before); const double thr = (double)u1 * my_tot; coll = thr >= my_lo && thr < my_hi; }
                const unsigned long long cm = __ballot(coll);
                const int f = cm ? __builtin_ctzll(cm) : 64;
                if (act && tid < f) hand(i + tid, 1, u1);
                if (!act && tid < run && tid < f) __hip_atomic_store(&s_hand[i + tid], HAND_SKIP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // not rejected
                off += __builtin_popcountll(f < 64 ? (actm & ((1ull << f) - 1ull)) : actm);
                for (int k = f; k < run; ++k) {
                    const double t_ = lane_f64(my_tot, k);
                    if (t_ < 0.0) { if (tid == 0) __hip_atomic_store(&s_hand[i + k], HAND_SKIP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); continue; }
                    float uf;
                    const int o = off;
                    const int draws = rs_count_draws([&](int tr) { return u_at(o + tr); }, t_, lane_f64(my_lo, k), lane_f64(my_hi, k), tid, &uf);
                    if (tid == 0) hand(i + k, draws, uf);
                    off += draws;
                }
            }
            i += run;
        }
        return;                                                      // (wavefront 3 announces the end of the chain)
    }
    if (blk >= 3) {                                                 // ---- row (blk-3)/nact: one segment's sums; segment 0's workgroup then draws the row's bonus token and finishes the row
        const int nact = rs_active_segs(a.V, Elem<DT>::EPV);
        const int item = (blk - 3) / nact, seg = (blk - 3) % nact, b = item;
        __shared__ RsSumShared shs;
        __shared__ RsPickShared sh;
        __shared__ int s_rej, s_nacc, s_eos;
        __shared__ float s_uf;
        __shared__ double s_S;
        if (tid == 0) {
            unsigned long long f;
            if (!rs_wait_flag(w.flag + (int64_t)item * RS_FLAG_STRIDE, a.gen, &f)) s_rej = -3;
            else { s_rej = (int)(f & 0xFFFFull) - 2; s_nacc = (int)((f >> 16) & 0x3FFFull); s_eos = (int)((f >> 30) & 1ull); }
        }
        __syncthreads();
        int rej = s_rej;
        bool sums_ok = rej != -3;
        if (rej >= 0) {
            if (tid == 0 && seg == 0) RS_ROWSTAMP(1, item);
            sums_ok = rs_rowsum_fused<DT, FLT>(a.logits, a.V, a.row_stride, a.row_max, a.row_sumexp, a.t, w, item, seg, nact, item * W + rej,
                                          a.draft[(int64_t)item * L + rej + 1], a.gen, shs);
            if (tid == 0 && seg == 0) RS_ROWSTAMP(4, item);
        }
        if (!sums_ok && tid == 0) rs_report_timeout(a.rows);
        if (seg != 0) return;
        // ---- segment 0's workgroup stays as the row's own.  First the row's CDF interval for the chain, as soon as the row's
        // other segments are in (a lane per segment polls its word): the chain's thread of this row then waits for ONE word and
        // reads three numbers (until round 4 it polled the RS_SEG words and formed the interval itself, in a loop shared with
        // 63 other rows: a row's sums were noticed up to 6-10 us after they were stored)
        if (rej >= 0) {
            bool ok = sums_ok;
            if (ok && tid < RS_SEG) ok = rs_wait_word(w.segdone + (int64_t)item * RS_SEG + tid, a.gen);
            sums_ok = __syncthreads_and(ok ? 1 : 0) != 0;
            if (tid == 0) {
                double t_ = -1.0, lo_ = 0.0, hi_ = 0.0;
                if (sums_ok) rs_interval<true>(w, b, a.V, Elem<DT>::EPV, t_, lo_, hi_, a.draft[(int64_t)b * L + rej + 1]);
                else rs_report_timeout(a.rows);
                st_agent_f64(w.iv + 3 * (int64_t)b, t_); st_agent_f64(w.iv + 3 * (int64_t)b + 1, lo_); st_agent_f64(w.iv + 3 * (int64_t)b + 2, hi_);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(w.ivdone + b, a.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                RS_ROWSTAMP(2, b);
            }
        }
        // ---- then the bonus draw (when the chain hands the uniform over) and the finish.  (Copying the row's sums and wave-tile
        // tables into LDS while the uniform is on its way did not shorten the walk: its time is the vector loads and the exact
        // probabilities of the 512 elements it ends in.)
        rs_load_tab(sh.tab);
        if (tid == 0 && sums_ok && rej >= 0) {
            unsigned long long pk = 0ull;
            if (!rs_wait_flag(w.pick + b, a.gen, &pk)) s_rej = -3;
            s_uf = __uint_as_float((uint32_t)pk);
            s_S = ld_agent_f64(w.s64 + b);                           // stored before the row's segment-done words the chain has seen
        }
        if (tid == 0 && !sums_ok) s_rej = -3;
        __syncthreads();
        rej = s_rej;
        if (rej == -3) { if (tid == 0) { rs_report_timeout(a.rows); __hip_atomic_store(w.fin + b, (unsigned long long)a.gen << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } return; }
        if (tid == 0) RS_ROWSTAMP(5, b);
        int n = s_nacc, eos = s_eos, bonus = -1;
        // the next draft's greedy tail depends on the NUMBER of committed tokens only, which is known before the bonus token is
        // (a rejected row commits its accepted tokens + 1): its argmax words are loaded ahead of the walk, not behind it
        int pre_off = 0, pre_len = 0;
        rs_next_shape(n + (rej >= 0 ? 1 : 0), L, pre_off, pre_len);
        unsigned long long pre_tail = 0ull;
        if (tid >= 1 && tid < 1 + pre_len && tid < 256) pre_tail = a.packed[(int64_t)b * W + pre_off + (tid - 1)];
        if (rej >= 0) {
            const int64_t r = (int64_t)b * W + rej;
            if (b == B - 1) RS_STAMP_MAX(17);                        // 17: last row's bonus workgroup has its uniform
            const RsRow row = rs_step_row<DT>(a.logits, r, a.V, a.row_stride, a.t, a.row_max, a.row_sumexp, FLT ? w.filt : nullptr);
            bonus = rs_final_pick<DT, true>(row, w, b, s_S, a.draft[(int64_t)b * L + rej + 1], s_uf, sh);
            if (b == B - 1) RS_STAMP_MAX(19);                        // 19: ... has walked
            if (a.eos_id >= 0 && bonus == a.eos_id) eos = 1;
            n += 1;                                                  // n_accepted of a rejected row == reject_pos (rs_accept_body)
        }
        // everything that depends on this row alone (JDN:444-466 / 619-638): the next draft's seed and greedy tail
        const int active = (!eos && n < a.remaining[b]) ? 1 : 0;
        int off = 0, copy_len = 0, n_pads = 0;
        if (active) { rs_next_shape(n, L, off, copy_len); n_pads = L - 1 - copy_len; }
        const int64_t r0 = (int64_t)b * W;
        if (active) {
            for (int i = tid; i < 1 + copy_len; i += 256) {
                int64_t v;
                if (i == 0) v = rej >= 0 ? (int64_t)bonus : a.draft[(int64_t)b * L + n];     // the last committed token
                else v = jfmb::decode_packed(i < 256 ? pre_tail : a.packed[r0 + off + (i - 1)]);   // (off == pre_off, copy_len == pre_len)
                a.next_draft[(int64_t)b * L + i] = v;
            }
        }
        __syncthreads();                                             // the greedy tokens are read: the row's argmax slots go back to zero
        for (int i = tid; i < W; i += 256) a.packed[r0 + i] = 0ull;
        if (tid == 0) {
            if (rej >= 0) a.committed[(int64_t)b * L + rej] = bonus;
            jf_rs_row &rw = a.rows[b];
            rw.n_committed = n; rw.eos = eos; rw.active_next = active; rw.n_pads = n_pads;
            if (rej < 0) rw.n_bonus_draws = 0;
            if (b != 0) rw.rsv = 0;                                  // rows[0].rsv carries a timeout report (cleared by the host before the call)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(w.fin + b, ((unsigned long long)a.gen << 32) | (unsigned long long)n_pads, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            RS_ROWSTAMP(6, b);
        }
        if (b == B - 1) RS_STAMP_MAX(21);                            // 21: ... has stored its token and its finish word
        return;
    }
    // ---- block 2: what depends on more than one row — pad offsets, the pads, the stream cursors
    __shared__ int s_np[RS_FUSED_ROWS], s_off[RS_FUSED_ROWS];
    int32_t *s_pad = (int32_t *)s_big;                                // [2048]
    __shared__ int s_bad, s_pads;
    const int64_t pc0 = *a.pad_cursor;                               // (nobody else writes the cursors: loaded while the rows are still at work,
    const int64_t uc0 = *a.u_cursor, bc0 = *a.b_cursor;              //  not as a read-modify-write at the very end)
    // the pad stream's window this call can touch (at most L - 2 per row), staged while the rows are still at work
    const int padwin = B * (L - 2) < 2048 ? B * (L - 2) : 2048;
    batched_for<8, int64_t>(padwin, tid, 256, [&](int64_t i) { return a.pad_stream[(pc0 + i) % a.pad_len]; }, [&](int64_t i, int64_t v) { s_pad[i] = (int32_t)v; });
    if (tid == 0) s_bad = 0;
    __syncthreads();
    // wavefronts 2 and 3: the two stream cursors' increments as soon as their producers are done — the accept workgroup's row
    // records (uniforms per row), the chain's draw counts — long before the last row finishes; wavefronts 0 and 1: a thread per
    // row waits for the row's finish word (its pads).  Every word that crosses workgroups is read as an agent-scope load.
    __shared__ int s_uc, s_bc;
    if (tid >= 128) {
        const int lane = tid & 63, which = (tid >> 6) - 2;           // 0: accept uniforms, 1: bonus draws
        bool ok = true;
        if (lane == 0) ok = rs_wait_word(w.acceptdone, a.gen) && (which == 0 || rs_wait_word(w.acceptdone + 1, a.gen));
        if (__ballot(!ok) != 0ull) { if (lane == 0) s_bad = 1; }
        else {
            int tot = 0;
            for (int b0 = 0; b0 < B; b0 += 64) {
                const int r = b0 + lane;
                int v = 0;
                if (r < B) {
                    if (which == 0) v = __hip_atomic_load(&a.rows[r].n_uniforms, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else if (__hip_atomic_load(&a.rows[r].reject_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0)
                        v = __hip_atomic_load(&a.rows[r].n_bonus_draws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                int t_;
                (void)wave_excl_scan_i32(v, lane, &t_);
                tot += t_;
            }
            if (lane == 0) { if (which == 0) s_uc = tot; else s_bc = tot; }
        }
    } else {
        for (int i = tid; i < B; i += 128) {
            unsigned long long f;
            if (!rs_wait_flag(w.fin + i, a.gen, &f)) s_bad = 1;
            s_np[i] = (int)(f & 0xFFFFull);
        }
    }
    __syncthreads();
    RS_STAMP_MIN(12);                                                // 12: end workgroup has everything
    if (s_bad) { if (tid == 0) rs_report_timeout(a.rows); return; }
    if (tid < 64) {
        int pc = 0;
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int r = b0 + tid;
            const int np = r < B ? s_np[r] : 0;
            int tp;
            const int ep = wave_excl_scan_i32(np, tid, &tp);
            if (r < B) s_off[r] = pc + ep;                            // this row's offset into the pad stream
            pc += tp;
        }
        if (tid == 0) { *a.u_cursor = uc0 + s_uc; *a.b_cursor = bc0 + s_bc; s_pads = pc; }
    }
    __syncthreads();
    RS_STAMP_MAX(25);                                                // 25: end: scans done
    for (int idx = tid; idx < B * 32; idx += 256) {                  // the pads of every row that keeps decoding: 32 lanes per row
        const int b = idx >> 5;
        const int np = s_np[b];
        for (int j = idx & 31; j < np; j += 32) {
            const int o = s_off[b] + j;
            a.next_draft[(int64_t)b * L + (L - np) + j] = o < padwin ? (int64_t)s_pad[o] : a.pad_stream[(pc0 + o) % a.pad_len];
        }
    }
    __syncthreads();
    if (tid == 0) *a.pad_cursor = pc0 + s_pads;
    RS_STAMP_MAX(13);                                                // 13: finished
}

// ------------------------------------------------------------------------------------------------
// On-policy rollout step (JDO = inference_engine/engine/jacobi_decoding_nongreedy_on_policy.py): sequential accept /
// reject of ONE sequence's proposed tokens with a stop-token SET (JDO:270-327), then a fresh sample of every not yet
// accepted position from this forward's distribution (JDO:465-477).
//   rs_op_accept_kernel   one workgroup: the R accept tests are one ballot of its first wavefront (probabilities as in
//                         rs_accept_entry; an undecided test is resolved with the row's float64 sum by the workgroup)
//   rs_rowsum_a/b         float64 sums and exact CDF sums of the rejected row (+ its proposed token's interval) and of every
//                         row after it (the re-draft candidates)
//   rs_op_bonus_kernel    one workgroup: draws counted against the interval, ONE walk, stop flag, stream cursors
//   rs_op_redraft_kernel  one workgroup per re-drafted row: one draw each; re-zeroes packed
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool op_is_stop(const int32_t *stop_ids, int n_stop, int64_t tok) {
    for (int k = 0; k < n_stop; ++k) if (tok == (int64_t)stop_ids[k]) return true;
    return false;
}

constexpr int RS_OP_STAGE = 2048;
template <int DT>
__global__ __launch_bounds__(256) void rs_op_accept_kernel(RsAcceptIn in, const int64_t *proposed, int R,
                                                            const int32_t *stop_ids, int n_stop, const float *u_stream,
                                                            int64_t u_len, const int64_t *u_cursor, int64_t *committed,
                                                            jf_op_row *out, RsWs w) {
    __shared__ uint32_t s_p[RS_OP_STAGE];
    __shared__ double s_tab[64], s_red[4];
    __shared__ int s_unc, s_n, s_stop, s_rej, s_used;
    uint32_t *g_patch = (uint32_t *)w.wtsum;      // !staged: one word per test, RS_PATCH_NONE = not resolved (the later launches overwrite it)
    const int tid = threadIdx.x;
    const int64_t uc = *u_cursor;
    const bool staged = R <= RS_OP_STAGE;
    rs_load_tab(s_tab);
    if (staged) for (int i = tid; i < R; i += 256) s_p[i] = rs_accept_entry<DT>(in, i, proposed[i], -1, s_tab);
    else for (int i = tid; i < R; i += 256) __hip_atomic_store(g_patch + i, RS_PATCH_NONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (;;) {
        if (tid < 64) {
            const int lane = tid;
            int n = 0, stop = 0, rej = -1, used = 0, unc_at = -1;
            for (int t0 = 0; t0 < R; t0 += 64) {                              // JDO:293-320
                const int t = t0 + lane;
                bool st = false, rejb = false, unc = false;
                if (t < R) {
                    uint32_t pe;
                    if (staged) pe = s_p[t];
                    else {
                        pe = rs_accept_entry<DT>(in, t, proposed[t], -1, s_tab);
                        const uint32_t q = __hip_atomic_load(g_patch + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (q != RS_PATCH_NONE) pe = q;
                    }
                    const float uu = u_stream[(uc + t) % u_len];
                    rejb = !(uu < __uint_as_float(pe & RS_PE_VAL));
                    unc = rejb && (pe & RS_PE_AMB) && uu < rs_accept_hi<DT>(pe);
                    st = rejb || op_is_stop(stop_ids, n_stop, proposed[t]);
                }
                const unsigned long long bal = __ballot(st);
                if (bal) {
                    const int f = __builtin_ctzll(bal);
                    if ((__ballot(unc) >> f) & 1ull) { unc_at = t0 + f; break; }
                    const bool is_rej = (__ballot(rejb) >> f) & 1ull;
                    if (is_rej) { rej = t0 + f; n = t0 + f; } else { stop = 1; n = t0 + f + 1; }
                    used = t0 + f + 1;
                    break;
                }
                const int wd = (R - t0) < 64 ? (R - t0) : 64;
                n = t0 + wd; used = t0 + wd;
            }
            if (lane == 0) { s_unc = unc_at; s_n = n; s_stop = stop; s_rej = rej; s_used = used; }
        }
        __syncthreads();
        const int ui = s_unc;
        if (ui < 0) break;
        const float pex = rs_exact_prob_wg<DT>(in.logits, ui, in.V, in.row_stride, in.t, in.row_max[ui], proposed[ui], s_tab, s_red);
        if (tid == 0) {
            const uint32_t word = __float_as_uint(pex > 1.f ? 1.f : pex) & RS_PE_VAL;
            if (staged) s_p[ui] = word;
            else __hip_atomic_store(g_patch + ui, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    const int n = s_n, rej = s_rej;
    for (int i = tid; i < n; i += 256) committed[i] = proposed[i];
    for (int i = tid; i < R; i += 256) {
        w.sel_row[i] = (rej >= 0 && i >= rej) ? i : -1;
        w.avoid[i] = (i == rej) ? (int32_t)proposed[i] : -1;
    }
    if (tid == 0) {
        out->n_committed = n; out->stop_hit = s_stop; out->reject_pos = rej; out->n_bonus_draws = 0;
        out->n_uniforms = s_used; out->n_redraft = 0; out->redraft_base_lo = 0; out->redraft_base_hi = 0;
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
    rs_load_tab(sh.tab);
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
        const RsRow row = rs_step_row<DT>(logits, rej, V, row_stride, temp, row_max, row_sumexp, w.filt);
        const int bonus = rs_final_pick<DT, false>(row, w, rej, w.s64[rej], proposed[rej], s_uf, sh);
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
                                                             RsWs w, int64_t *redraft, unsigned long long *packed) {
    __shared__ RsPickShared sh;
    const int li = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) packed[li] = 0ull;
    const int n = res->n_committed;
    if (res->n_redraft <= 0 || li < n) return;
    rs_load_tab(sh.tab);
    __syncthreads();
    const int64_t base = ((int64_t)res->redraft_base_hi << 32) | (int64_t)(uint32_t)res->redraft_base_lo;
    const RsRow row = rs_step_row<DT>(logits, li, V, row_stride, temp, row_max, row_sumexp, w.filt);
    const float u = m_stream[(base + (li - n)) % m_len];
    int y;
    if (rs_hier_ok(V, Elem<DT>::EPV)) {
        if (tid < 64) { const int yy = rs_pick_wave<DT, false>(row, w, li, w.s64[li], u, sh.tab, tid); if (tid == 0) sh.pick = yy; }
        __syncthreads();
        y = sh.pick;
    } else {
        y = rs_pick_wg<DT>(row, w.segsum + (int64_t)li * RS_SEG, w.s64[li], u, sh);
    }
    if (tid == 0) redraft[li] = y;
}

extern "C" int jf_rs_onpolicy_step(const void *logits, int dtype, int64_t V, int64_t row_stride, const int64_t *proposed, int R,
                                   const float *p_draft, const float *row_max, const float *row_sumexp, uint64_t *packed,
                                   float temperature, const int32_t *stop_ids, int n_stop, const float *u_stream, int64_t u_len,
                                   int64_t *u_cursor, const float *m_stream, int64_t m_len, int64_t *m_cursor,
                                   int64_t *committed, int64_t *redraft, jf_op_row *row, void *workspace,
                                   size_t workspace_bytes, const jf_rs_filter_row *filt, void *stream) {
    if (R <= 0) return JF_OK;
    if (!logits || !proposed || !p_draft || !row_max || !row_sumexp || !packed || !u_stream || !u_cursor || !m_stream ||
        !m_cursor || !committed || !redraft || !row || !workspace || (n_stop > 0 && !stop_ids))
        return fail(JF_E_INVALID, "jf_rs_onpolicy_step: null pointer");
    if (u_len <= 0 || m_len <= 0 || n_stop < 0) return fail(JF_E_INVALID, "jf_rs_onpolicy_step: empty random stream");
    if (workspace_bytes < rs_ws_bytes(R) || ((uintptr_t)workspace % 16) != 0) return fail(JF_E_INVALID, "jf_rs_onpolicy_step: workspace too small or not 16-byte aligned");
    if (dtype != JF_F32 && dtype != JF_BF16) return fail(JF_E_INVALID, "jf_rs_onpolicy_step: dtype %d", dtype);
    if (V <= 0 || V > 0x7FFFFFFFll) return fail(JF_E_INVALID, "jf_rs_onpolicy_step: V=%lld", (long long)V);
    float t = (temperature <= 0.f) ? 1.f : temperature;
    if (dtype == JF_BF16 && t != 1.f && rs_scale_is_exact(t)) t = -t;    // the kernels' rows take the product form of the scaling (rs_make_row)
    hipStream_t s = (hipStream_t)stream;
    RsWs w = rs_ws(workspace, R);
    w.filt = filt;
    unsigned long long *pk = (unsigned long long *)packed;
    const RsAcceptIn in{logits, V, row_stride, t, row_max, row_sumexp, p_draft};
#define JF_OP(DT)                                                                                                                              \
    rs_op_accept_kernel<DT><<<1, 256, 0, s>>>(in, proposed, R, stop_ids, n_stop, u_stream, u_len, u_cursor, committed, row, w);                 \
    rs_rowsum_a_kernel<DT><<<R * RS_SEG, 256, 0, s>>>(logits, V, row_stride, row_max, row_sumexp, t, w);                                        \
    rs_rowsum_b_kernel<DT><<<R * RS_SEG, 256, 0, s>>>(logits, V, row_stride, row_max, row_sumexp, t, w);                                        \
    rs_op_bonus_kernel<DT><<<1, 256, 0, s>>>(logits, V, row_stride, proposed, R, row_max, row_sumexp, t, stop_ids, n_stop, u_cursor, m_stream,  \
                                             m_len, m_cursor, committed, row, w);                                                               \
    rs_op_redraft_kernel<DT><<<R, 256, 0, s>>>(logits, V, row_stride, R, row_max, row_sumexp, t, m_stream, m_len, row, w, redraft, pk)
    if (dtype == JF_F32) { JF_OP(JF_F32); } else { JF_OP(JF_BF16); }
#undef JF_OP
    return check_launch("rs_onpolicy kernels");
}

// May the one-launch step run on this device?  Its three role workgroups and the workgroups of two rows (a row's wait for each
// other) must fit at once in HALF of what the runtime says it keeps resident; everything else only waits backwards
// in dispatch order.
static bool rs_fused_fits(const void *kern, int variant, int B) {
    static std::mutex mu;
    static int cap[4] = {-1, -1, -1, -1}, capdev[4] = {-1, -1, -1, -1};
    std::lock_guard<std::mutex> g(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (cap[variant] < 0 || capdev[variant] != dev) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
            (void)hipGetLastError();
            per_cu = 0;
        }
        const long long c = (long long)per_cu * cus / 2;
        cap[variant] = (int)(c > 0x7FFFFFFF ? 0x7FFFFFFF : c);
        capdev[variant] = dev;
    }
    (void)B;
    return 3 + 2 * RS_SEG <= cap[variant];
}

extern "C" int jf_rs_step(const void *logits, int dtype, int64_t V, int64_t row_stride, const int64_t *draft, int B, int L,
                          const float *p_draft, const float *row_max, const float *row_sumexp, uint64_t *packed,
                          float temperature, int32_t eos_id, const int32_t *remaining, const float *u_stream, int64_t u_len,
                          int64_t *u_cursor, const float *bonus_stream, int64_t bonus_len, int64_t *bonus_cursor,
                          const int64_t *pad_stream, int64_t pad_len, int64_t *pad_cursor, int64_t *committed,
                          int64_t *next_draft, jf_rs_row *rows, void *workspace, size_t workspace_bytes, const jf_rs_filter_row *filt,
                          void *stream) {
    const JfTiming tm = jf_take_timing();                            // events armed for this call (jf_timing_arm): taken whatever happens below
    if (B <= 0) return JF_OK;
    if (L < 2) return fail(JF_E_INVALID, "Draft must have at least 2 tokens (seed + 1 speculative)");
    if (L > 0x3FFF) return fail(JF_E_INVALID, "jf_rs_step: L=%d too large", L);
    if (!logits || !draft || !p_draft || !row_max || !row_sumexp || !packed || !remaining || !u_stream || !u_cursor ||
        !bonus_stream || !bonus_cursor || !pad_stream || !pad_cursor || !committed || !next_draft || !rows || !workspace)
        return fail(JF_E_INVALID, "jf_rs_step: null pointer");
    if (u_len <= 0 || bonus_len <= 0 || pad_len <= 0) return fail(JF_E_INVALID, "jf_rs_step: empty random stream");
    if (workspace_bytes < rs_ws_bytes(B) || ((uintptr_t)workspace % 16) != 0) return fail(JF_E_INVALID, "jf_rs_step: workspace too small or not 16-byte aligned");
    if (V <= 0 || V > 0x7FFFFFFFll) return fail(JF_E_INVALID, "jf_rs_step: V=%lld", (long long)V);
    float t = (temperature <= 0.f) ? 1.f : temperature;
    if (dtype != JF_F32 && dtype != JF_BF16) return fail(JF_E_INVALID, "jf_rs_step: dtype %d", dtype);
    if (dtype == JF_BF16 && t != 1.f && rs_scale_is_exact(t)) t = -t;    // the kernels' rows take the product form of the scaling (rs_make_row)
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *pk = (unsigned long long *)packed;
    RsWs w = rs_ws(workspace, B);
    w.filt = filt;
    static const bool fused_ok = !(getenv("JF_RS_FUSED") && getenv("JF_RS_FUSED")[0] == '0');   // A/B knob, read once
    const bool flt = filt != nullptr;
    const void *fk = dtype == JF_F32 ? (flt ? (const void *)rs_step_fused_kernel<JF_F32, true> : (const void *)rs_step_fused_kernel<JF_F32, false>)
                                     : (flt ? (const void *)rs_step_fused_kernel<JF_BF16, true> : (const void *)rs_step_fused_kernel<JF_BF16, false>);
    if (fused_ok && (int64_t)B * (L - 1) <= RS_FUSED_STAGE && B <= RS_FUSED_ROWS && u_len < 0x7FFFFFFFll &&
        rs_hier_ok(V, dtype == JF_F32 ? 4 : 8) && rs_fused_fits(fk, (dtype == JF_F32 ? 0 : 1) + (flt ? 2 : 0), B)) {
        static std::atomic<uint32_t> g_gen{0};                      // generation of the hand-off words (0 never used: a zeroed workspace)
        uint32_t gen = ++g_gen;
        if (gen == 0) gen = ++g_gen;
        RsFusedArgs a{logits, V, row_stride, draft, B, L, p_draft, row_max, row_sumexp, pk, t, eos_id, remaining, u_stream, u_len, u_cursor,
                      bonus_stream, bonus_len, bonus_cursor, pad_stream, pad_len, pad_cursor, committed, next_draft, rows, w, gen};
        const unsigned grid = (unsigned)(3 + B * rs_active_segs(V, dtype == JF_F32 ? 4 : 8));
#define JF_FUSED(K)                                                                                                     \
        if (tm.any() && !jf_timing_bracket()) hipExtLaunchKernelGGL(K, dim3(grid), dim3(256), 0, s, tm.begin, tm.end, 0, a);   \
        else {                                                                                                          \
            if (tm.begin) (void)hipEventRecord(tm.begin, s);                                                            \
            K<<<grid, 256, 0, s>>>(a);                                                                                   \
            if (tm.end) (void)hipEventRecord(tm.end, s);                                                                \
        }
        // (events attached to the dispatch: the launch's own start / stop timestamps)
        if (dtype == JF_F32) { if (flt) { JF_FUSED((rs_step_fused_kernel<JF_F32, true>)) } else { JF_FUSED((rs_step_fused_kernel<JF_F32, false>)) } }
        else { if (flt) { JF_FUSED((rs_step_fused_kernel<JF_BF16, true>)) } else { JF_FUSED((rs_step_fused_kernel<JF_BF16, false>)) } }
#undef JF_FUSED
        return check_launch("rs_step_fused_kernel");
    }
    if (tm.begin) (void)hipEventRecord(tm.begin, s);                 // several launches: the events bracket them
    const RsAcceptIn in{logits, V, row_stride, t, row_max, row_sumexp, p_draft};
    const bool staged = (int64_t)B * (L - 1) <= RS_STAGE && B <= RS_ROWS_LDS && u_len < 0x7FFFFFFFll;
    // the walk outside LDS keeps a row's resolved tests in the workspace's wave-tile block (one word per position of a row)
    if (!staged && (int64_t)(L - 1) > (int64_t)((B + 3) / 4 * 4) * RS_SEG * RS_WT * 2)
        return fail(JF_E_CAPACITY, "jf_rs_step: a block of %d tokens with %d rows does not fit the step workspace's patch table", L, B);
    const bool chain_in_bonus = B <= RS_BONUS_CHAIN_ROWS;
#define JF_RS_MULTI(DT)                                                                                                                        \
    if (staged) rs_accept_kernel<DT, true><<<1, 256, 0, s>>>(in, draft, B, L, eos_id, u_stream, u_len, u_cursor, committed, rows, w);           \
    else rs_accept_kernel<DT, false><<<1, 256, 0, s>>>(in, draft, B, L, eos_id, u_stream, u_len, u_cursor, committed, rows, w);                 \
    rs_rowsum_a_kernel<DT><<<B * RS_SEG, 256, 0, s>>>(logits, V, row_stride, row_max, row_sumexp, t, w);                                        \
    rs_rowsum_b_kernel<DT><<<B * RS_SEG, 256, 0, s>>>(logits, V, row_stride, row_max, row_sumexp, t, w);                                        \
    if (chain_in_bonus) rs_bonus_kernel<DT, true><<<B, 256, 0, s>>>(logits, V, row_stride, draft, L, row_max, row_sumexp, t, bonus_stream,      \
                                                                    bonus_len, bonus_cursor, committed, rows, w);                               \
    else {                                                                                                                                      \
        rs_chain_kernel<<<1, 256, 0, s>>>(B, V, Elem<DT>::EPV, bonus_stream, bonus_len, bonus_cursor, rows, w);                                 \
        rs_bonus_kernel<DT, false><<<B, 256, 0, s>>>(logits, V, row_stride, draft, L, row_max, row_sumexp, t, bonus_stream, bonus_len,          \
                                                     bonus_cursor, committed, rows, w);                                                         \
    }
    if (dtype == JF_F32) { JF_RS_MULTI(JF_F32) } else { JF_RS_MULTI(JF_BF16) }
#undef JF_RS_MULTI
    rs_finish_kernel<<<1, 256, 0, s>>>(B, L, pk, eos_id, remaining, u_cursor, bonus_cursor, pad_stream, pad_len, pad_cursor, committed, next_draft, rows);
    if (tm.end) (void)hipEventRecord(tm.end, s);
    return check_launch("rs_step kernels");
}

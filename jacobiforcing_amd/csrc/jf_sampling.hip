// jf_sampling.hip — (a19) non-greedy verify: fused softmax-gather + argmax (jf_rs_probs), the batch accept / bonus / finish
// step (jf_rs_step) and the on-policy rollout step (jf_rs_onpolicy_step).
#include "jf_common.h"

// ------------------------------------------------------------------------------------------------
// (a19) non-greedy verify: fused online-softmax gather + argmax, logits read once
// ------------------------------------------------------------------------------------------------
// Stage 1 — one workgroup per (row, chunk): 16 B per lane per load, four vectors (16/32 elements) per lane per round.
// (hot loop: hardware v_exp_f32 via __expf, ~1e-6 relative; the verify tolerance is 2e-5)
// Per round the lane first raises its running max over the whole round (register-resident values), rescales its sum once,
// then adds exp(x - m) for every element: one exp per element plus one per round, branch-free.  The argmax tracker runs on
// the same registers.  Partials (m, s) go to the workspace, the argmax to `packed` by atomicMax.
template <int DT, int NV>
__device__ __forceinline__ void rs_round(const u32x4 (&vv)[NV], float cs, float &m, float &s) {
    // (m, s) live in the scaled log2 domain: x' = w * cs with cs = log2(e) / T, s = sum of 2^(x' - m); one multiply and one
    // v_exp_f32 per element (exp(x/T - M) == 2^(x' - m) up to the rounding of the product)
    constexpr int EPV = Elem<DT>::EPV;
    constexpr int NE = NV * EPV;
    float x[NE];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const uint32_t w[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (DT == JF_F32) {
                x[u * 4 + j] = __uint_as_float(w[j]) * cs;
            } else {
                x[u * 8 + 2 * j] = __uint_as_float(w[j] << 16) * cs;
                x[u * 8 + 2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u) * cs;
            }
        }
    }
    float mx = x[0];
#pragma unroll
    for (int j = 1; j < NE; ++j) mx = fmaxf(mx, x[j]);
    const float mn = fmaxf(m, mx);
    if (mn == -INFINITY) return;                        // nothing finite yet: keep (m, s) = (-inf, 0), never form inf - inf
    float acc = (m == -INFINITY) ? 0.f : s * __builtin_amdgcn_exp2f(m - mn);
#pragma unroll
    for (int j = 0; j < NE; ++j) acc += __builtin_amdgcn_exp2f(x[j] - mn);
    s = acc;
    m = mn;
}

// raw fp32 value behind an order key (inverse of order_key for non-NaN values)
__device__ __forceinline__ float key_to_float(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
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
    const float cs = inv_t * 1.44269504088896340736f;     // log2(e) / T
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
            rs_round<DT, 4>(vv, cs, m, s);
        }
        for (; k < nvec; k += 256, q += 256) {          // this lane's remaining vectors, one at a time
            const u32x4 vv[1] = {JF_LOAD(q)};
            ft.consume(vv[0], ebase + (uint32_t)k * EPV);
            rs_round<DT, 1>(vv, cs, m, s);
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
        const float xv = load_f<DT>(p, i) * cs;
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
        // the chunk's RAW maximum (from the argmax key) and its sum relative to fl(raw max * cs) == M: multiplying by a
        // positive constant is monotone, so the largest scaled value belongs to the largest raw value
        partial[item] = make_float2(M == -INFINITY ? -INFINITY : key_to_float((uint32_t)(mm >> 32)), Ssum);
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
    const float cs = inv_t * 1.44269504088896340736f;
    float Mraw = -INFINITY;
    for (int c = 0; c < cpr; ++c) Mraw = fmaxf(Mraw, partial[row * cpr + c].x);
    float S = 0.f;
    for (int c = 0; c < cpr; ++c) {
        const float2 ps = partial[row * cpr + c];
        S += (ps.x == -INFINITY) ? 0.f : ps.y * exp2f(ps.x * cs - Mraw * cs);
    }
    const float M = Mraw * inv_t;                         // consumers form exp(x * inv_t - M): exactly 1 at the maximum
    row_max[row] = M;
    row_sumexp[row] = S;
    const int64_t tok = draft_next[row];
    const void *p = (const char *)logits + row * row_stride * (DT == JF_F32 ? 4 : 2);
    p_draft[row] = (tok >= 0 && tok < V) ? expf(load_f<DT>(p, tok) * inv_t - M) / S : 0.f;
}

static int64_t rs_chunk(int dtype, int64_t R, int64_t V, int64_t *cpr_out) {
    const int64_t gran = 4 * (int64_t)256 * (dtype == JF_F32 ? 4 : 8);     // one full round per workgroup
    const char *e = getenv("JF_RS_ITEMS");                                  // workgroups to aim for (sweeps in tools/)
    const int64_t target = (e && *e) ? atoll(e) : 1024;                 // measured: fewer, longer items win (profiles/rs_probs_microbench_r01.txt)
    int64_t per_row = (target + R - 1) / R;
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


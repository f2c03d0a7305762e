// jf_kv.hip — (a18) KV append, fused RoPE + query re-layout + KV append, SwiGLU gate; (a9/a10) KV commit of the winning
// candidate row.
#include "jf_common.h"

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

// The same arithmetic with 16-byte accesses: one lane rotates EPV consecutive pairs (x1[i], x2[i]) of one (token, head) —
// two 16-byte loads from the fused QKV row, the cos/sin entries as float4s, two 16-byte stores per destination.  D = 128 bf16:
// 8 lanes per head, a wavefront covers 8 heads of a token (2 KB contiguous in, 256-byte rows out).
template <typename T, int EPV>
__global__ __launch_bounds__(256) void rope_kv_append_vec_kernel(const T *__restrict__ qkv, int64_t N, int Tlen, int nq, int nkv, int D,
                                                                  const int32_t *__restrict__ positions,
                                                                  const float *__restrict__ cos_t, const float *__restrict__ sin_t,
                                                                  T *__restrict__ q_out, T *__restrict__ k_cache,
                                                                  T *__restrict__ v_cache, const int64_t *__restrict__ slot_main,
                                                                  int64_t S_max, T *__restrict__ k_cand, T *__restrict__ v_cand,
                                                                  const int64_t *__restrict__ slot_cand, int64_t T_max) {
    const int half = D >> 1;
    const int lph = half / EPV;                               // lanes per head
    const int heads = nq + 2 * nkv;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= N * heads * lph) return;
    const int i0 = (int)(gid % lph) * EPV;
    const int64_t th = gid / lph;
    const int h = (int)(th % heads);
    const int64_t tok = th / heads;
    const T *src = qkv + (tok * heads + h) * D;
    alignas(16) T a[EPV], b[EPV], oa[EPV], ob[EPV];
    *reinterpret_cast<uint4 *>(a) = *reinterpret_cast<const uint4 *>(src + i0);
    *reinterpret_cast<uint4 *>(b) = *reinterpret_cast<const uint4 *>(src + half + i0);
    if (h < nq + nkv) {                                       // rotate q and k heads, v passes through
        const int64_t pos = positions[tok];
        alignas(16) float c[EPV], sn[EPV];
#pragma unroll
        for (int j = 0; j < EPV; j += 4) {
            *reinterpret_cast<float4 *>(c + j) = *reinterpret_cast<const float4 *>(cos_t + pos * half + i0 + j);
            *reinterpret_cast<float4 *>(sn + j) = *reinterpret_cast<const float4 *>(sin_t + pos * half + i0 + j);
        }
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
            const float x1 = ld_f(a + j), x2 = ld_f(b + j);
            st_f(oa + j, x1 * c[j] - x2 * sn[j]);
            st_f(ob + j, x2 * c[j] + x1 * sn[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < EPV; ++j) { oa[j] = a[j]; ob[j] = b[j]; }
    }
    auto put = [&](T *dst) {
        *reinterpret_cast<uint4 *>(dst + i0) = *reinterpret_cast<const uint4 *>(oa);
        *reinterpret_cast<uint4 *>(dst + half + i0) = *reinterpret_cast<const uint4 *>(ob);
    };
    if (h < nq) {
        const int G = nq / nkv;
        const int kvh = h / G, g = h - kvh * G;
        const int64_t r = tok / Tlen, t = tok - r * Tlen;
        put(q_out + (((r * nkv + kvh) * (int64_t)G * Tlen) + (int64_t)g * Tlen + t) * D);
        return;
    }
    const bool is_v = h >= nq + nkv;
    const int kvh = is_v ? h - nq - nkv : h - nq;
    const int64_t sm = slot_main[tok];
    if (sm >= 0) {
        const int64_t brow = sm / S_max, pos = sm - brow * S_max;
        put((is_v ? v_cache : k_cache) + ((brow * nkv + kvh) * S_max + pos) * D);
    }
    if (slot_cand) {
        const int64_t sc = slot_cand[tok];
        if (sc >= 0) {
            const int64_t brow = sc / T_max, pos = sc - brow * T_max;
            put((is_v ? v_cand : k_cand) + ((brow * nkv + kvh) * T_max + pos) * D);
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
    hipStream_t s = (hipStream_t)stream;
    // 16-byte path: every row start (QKV rows, q_out, caches) and the table rows are 16-byte aligned
    const int esz = dtype == JF_F32 ? 4 : 2, epv = 16 / esz;
    auto al16 = [](const void *p) { return ((uintptr_t)p) % 16 == 0; };
    static const bool force_scalar = [] { const char *e = getenv("JF_ROPE_SCALAR"); return e && *e == '1'; }();   // A/B in tools/
    const bool vec = !force_scalar && (dtype == JF_F32 || dtype == JF_BF16) && (D / 2) % epv == 0 && (D / 2) % 4 == 0 && al16(qkv) && al16(q_out) &&
                     al16(k_cache) && al16(v_cache) && al16(cos_table) && al16(sin_table) && (!slot_cand || (al16(k_cand) && al16(v_cand)));
    if (vec) {
        const int64_t total = N * (nq + 2 * nkv) * ((D / 2) / epv);
        const dim3 grid((unsigned)((total + 255) / 256)), block(256);
        if (dtype == JF_F32)
            rope_kv_append_vec_kernel<float, 4><<<grid, block, 0, s>>>((const float *)qkv, N, T, nq, nkv, D, positions, cos_table, sin_table,
                                                                      (float *)q_out, (float *)k_cache, (float *)v_cache, slot_main, S_max,
                                                                      (float *)k_cand, (float *)v_cand, slot_cand, T_max);
        else
            rope_kv_append_vec_kernel<uint16_t, 8><<<grid, block, 0, s>>>((const uint16_t *)qkv, N, T, nq, nkv, D, positions, cos_table,
                                                                         sin_table, (uint16_t *)q_out, (uint16_t *)k_cache,
                                                                         (uint16_t *)v_cache, slot_main, S_max, (uint16_t *)k_cand,
                                                                         (uint16_t *)v_cand, slot_cand, T_max);
        return check_launch("rope_kv_append_vec_kernel");
    }
    const int64_t total = N * (nq + 2 * nkv) * (D / 2);
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
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

// ---- SwiGLU gate: two 16-byte vectors of gate and of up per lane (four independent loads in flight), silu through the
// hardware exp2 / rcp (x * rcp(1 + exp2(-x * log2 e)): ~6 VALU operations per element; the exact expf + IEEE division it
// replaces cost ~35 and made the kernel VALU-bound at 2.6 TB/s, profiles/kernel_classes_r02.txt)
template <typename T, int EPV>
__global__ __launch_bounds__(256) void swiglu_kernel(const T *__restrict__ gu, int64_t M, int64_t I, T *__restrict__ out) {
    constexpr int VPT = 2;                                     // vectors per thread, 256 * EPV elements apart
    const int64_t vec_per_row = I / EPV;
    const int64_t chunks_per_row = (vec_per_row + 256 * VPT - 1) / (256 * VPT);
    const int64_t m = blockIdx.x / chunks_per_row;
    const int64_t v0 = (blockIdx.x - m * chunks_per_row) * (256 * VPT) + threadIdx.x;
    const T *g = gu + m * 2 * I;
    const T *u = g + I;
    T *o = out + m * I;
    alignas(16) T gv[VPT][EPV], uv[VPT][EPV], ov[EPV];
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int64_t v = v0 + k * 256;
        if (v < vec_per_row) {
            *reinterpret_cast<u32x4 *>(gv[k]) = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(g + v * EPV));
            *reinterpret_cast<u32x4 *>(uv[k]) = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(u + v * EPV));
        }
    }
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int64_t v = v0 + k * 256;
        if (v >= vec_per_row) continue;
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
            const float x = ld_f(gv[k] + j), y = ld_f(uv[k] + j);
            const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
            st_f(ov + j, x * sg * y);
        }
        *reinterpret_cast<uint4 *>(o + v * EPV) = *reinterpret_cast<const uint4 *>(ov);
    }
}

extern "C" int jf_swiglu(const void *gu, int dtype, int64_t M, int64_t I, void *out, void *stream) {
    if (M <= 0 || I <= 0) return JF_OK;
    if (!gu || !out) return fail(JF_E_INVALID, "jf_swiglu: null pointer");
    const int epv = dtype == JF_F32 ? 4 : 8;
    if ((dtype != JF_F32 && dtype != JF_BF16) || I % epv != 0 || ((uintptr_t)gu) % 16 || ((uintptr_t)out) % 16)
        return fail(JF_E_INVALID, "jf_swiglu: dtype/alignment (I must be a multiple of %d)", epv);
    const int64_t chunks = ((I / epv) + 511) / 512;            // 256 threads x 2 vectors per workgroup, one row per workgroup
    if (M * chunks > 0x7FFFFFFFll) return fail(JF_E_CAPACITY, "jf_swiglu: grid too large");
    const dim3 grid((unsigned)(M * chunks)), block(256);
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


// jf_multiblock.hip — (a1, a4-a12) kernels of the multiblock Jacobi state machine (jf_mb_core.h): one 64-lane wavefront
// per prompt.
#ifdef JF_EXP_MB_TRACE
// experiment build only (tools/mb_step_trace.py): shader-clock stamps of prompt 0's step, read back by jf_exp_read_trace
#include <hip/hip_runtime.h>
__device__ unsigned long long g_mb_trace[32];
#define JF_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_mb_trace[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#endif
#include "jf_common.h"

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


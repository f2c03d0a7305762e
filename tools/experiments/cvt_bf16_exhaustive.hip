// One-off check (gfx950): does v_cvt_pk_bf16_f32 round every fp32 bit pattern like the integer RNE formula the softmax-gather
// kernels used (q + 0x7FFF + lsb)?  Walks all 2^32 patterns; NaN inputs only have to stay NaN.
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/cvt_bf16_exhaustive tools/experiments/cvt_bf16_exhaustive.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void walk(unsigned long long *bad, uint32_t *first) {
    const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    for (uint64_t q0 = base; q0 < (1ull << 32); q0 += (uint64_t)gridDim.x * blockDim.x * 2) {
        const uint32_t qa = (uint32_t)q0, qb = (uint32_t)q0 + 1u;
        f2 v = {__uint_as_float(qa), __uint_as_float(qb)};
        const uint32_t hw = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2));
        const uint32_t in[2] = {qa, qb}, out[2] = {hw & 0xFFFFu, hw >> 16};
        for (int k = 0; k < 2; ++k) {
            const uint32_t q = in[k];
            const bool nan = (q & 0x7FFFFFFFu) > 0x7F800000u;
            const uint32_t ref = ((q + 0x7FFFu + ((q >> 16) & 1u)) >> 16) & 0xFFFFu;
            const bool ok = nan ? ((out[k] & 0x7FFFu) > 0x7F80u) : (out[k] == ref);
            if (!ok) { const unsigned long long n = atomicAdd(bad, 1ull); if (n < 16) first[n] = q; }
        }
    }
}
int main() {
    unsigned long long *bad; uint32_t *first;
    hipMalloc(&bad, 8); hipMalloc(&first, 64); hipMemset(bad, 0, 8); hipMemset(first, 0, 64);
    walk<<<4096, 256>>>(bad, first);
    unsigned long long h = 0; uint32_t f[16];
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(f, first, 64, hipMemcpyDeviceToHost);
    printf("v_cvt_pk_bf16_f32 vs integer RNE over 2^32 inputs: %llu mismatches\n", h);
    for (unsigned long long i = 0; i < h && i < 16; ++i) printf("  input 0x%08x\n", f[i]);
    return h ? 1 : 0;
}

// Issue rate of the VALU instructions the softmax stream is made of (gfx950): N independent chains per lane, 8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o tools/experiments/valu_rate tools/experiments/valu_rate.hip && tools/experiments/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
#define CHAINS 8
#define ITERS 4096
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
    float a[CHAINS]; f2 p[CHAINS];
    for (int i = 0; i < CHAINS; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = f2{a[i], a[i] + 0.5f}; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (OP == 0) a[i] = __builtin_amdgcn_exp2f(a[i]);
            else if (OP == 1) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
            else if (OP == 2) p[i] = __builtin_elementwise_fma(p[i], f2{1.0001f, 1.0001f}, f2{0.5f, 0.5f});
            else if (OP == 3) a[i] = __builtin_elementwise_maximum(__builtin_elementwise_maximum(a[i], a[(i + 1) % CHAINS]), seed);
            else if (OP == 4) { bf2 h = __builtin_convertvector(p[i], bf2); uint32_t u = __builtin_bit_cast(uint32_t, h); p[i] = f2{__uint_as_float(u << 16), __uint_as_float(u & 0xFFFF0000u)}; }
            else if (OP == 5) p[i] = p[i] * f2{1.0001f, 1.0001f};
            else if (OP == 6) a[i] = __uint_as_float((__float_as_uint(a[i]) << 16) ^ 0x3f000000u);      // shift + xor
            else if (OP == 8) a[i] = __uint_as_float(__float_as_uint(a[i]) & (0xFFFF0000u | it));   // and
            else if (OP == 9) a[i] = a[i] + 1.25f;
            else if (OP == 10) a[i] = a[i] * 1.0001f;
            else if (OP == 11) a[i] = __builtin_elementwise_maximum(a[i], a[(i + 1) % CHAINS]);
            else if (OP == 12) p[i] = p[i] + f2{1.25f, 0.5f};
            else if (OP == 13) { bf2 h = __builtin_convertvector(p[i], bf2); p[i].x += __uint_as_float(__builtin_bit_cast(uint32_t, h)); }   // cvt + add
            else if (OP == 14) a[i] = a[i] > a[(i + 1) % CHAINS] ? a[i] : seed;     // cmp + cndmask
            else if (OP == 15) a[i] = __uint_as_float(__float_as_uint(a[i]) + (uint32_t)it);
            else if (OP == 7) a[i] = __builtin_amdgcn_rcpf(a[i]);
        }
    }
    float r = 0; for (int i = 0; i < CHAINS; ++i) r += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int OP> void run(const char *name, int per_iter) {
    float *o; hipMalloc(&o, 256 * 8 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<256 * 8, 256>>>(o, 0.25f); hipDeviceSynchronize();
    hipEventRecord(a); k<OP><<<256 * 8, 256>>>(o, 0.25f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // 8 workgroups per CU = 8 waves per SIMD; instructions per SIMD = 8 waves * ITERS * CHAINS * per_iter
    const double inst = 8.0 * ITERS * CHAINS * per_iter;
    printf("%-28s %8.3f ms  %6.2f cycles per wave-instruction at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / inst);
    hipFree(o);
}
int main() {
    run<0>("v_exp_f32", 1); run<7>("v_rcp_f32", 1); run<1>("v_fma_f32", 1); run<2>("v_pk_fma_f32", 1); run<5>("v_pk_mul_f32", 1);
    run<3>("v_maximum3_f32", 1); run<4>("cvt_pk_bf16 + 2 unpack", 3); run<6>("v_lshlrev + v_xor", 2); run<8>("v_and_b32 (+or)", 1);
    run<9>("v_add_f32", 1); run<10>("v_mul_f32", 1); run<11>("v_maximum_f32 (2 in)", 1); run<12>("v_pk_add_f32", 1); run<13>("cvt_pk_bf16 + v_add_f32", 2);
    run<14>("v_cmp + v_cndmask", 2); run<15>("v_add_u32", 1);
    return 0;
}

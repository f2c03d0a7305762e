// tools/probe_stream.hip — NOT part of the product.  Load-only kernel with the same grid/chunk/4-deep
// unrolled 16-B nontemporal access pattern as argmax_partial_kernel: measures the streaming ceiling that
// pattern can reach on this chip, to separate "memory pattern" from "VALU work" when reading the roofline.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int TPB = 256;
__global__ __launch_bounds__(TPB) void probe_kernel(const void *logits, int64_t V, int64_t row_stride_bytes, int esz,
                                                     unsigned long long *out, int cpr, int64_t chunk) {
    const int64_t item = blockIdx.x, row = item / cpr;
    const int c = (int)(item - row * cpr);
    const int EPV = 16 / esz;
    const int64_t begin = (int64_t)c * chunk;
    int64_t end = begin + chunk; if (end > V) end = V;
    const u32x4 *pv = (const u32x4 *)((const char *)logits + row * row_stride_bytes);
    const int tid = threadIdx.x;
    uint32_t acc = 0;
    int64_t i = begin + (int64_t)tid * EPV;
    const int64_t STEP = (int64_t)TPB * EPV;
    for (; i + 3 * STEP + EPV <= end; i += 4 * STEP) {
        const u32x4 v0 = __builtin_nontemporal_load(pv + i / EPV);
        const u32x4 v1 = __builtin_nontemporal_load(pv + (i + STEP) / EPV);
        const u32x4 v2 = __builtin_nontemporal_load(pv + (i + 2 * STEP) / EPV);
        const u32x4 v3 = __builtin_nontemporal_load(pv + (i + 3 * STEP) / EPV);
        acc ^= v0.x ^ v0.y ^ v0.z ^ v0.w ^ v1.x ^ v1.y ^ v1.z ^ v1.w ^ v2.x ^ v2.y ^ v2.z ^ v2.w ^ v3.x ^ v3.y ^ v3.z ^ v3.w;
    }
    for (; i + EPV <= end; i += STEP) { const u32x4 v0 = __builtin_nontemporal_load(pv + i / EPV); acc ^= v0.x ^ v0.y ^ v0.z ^ v0.w; }
    if (acc == 0x12345678u) atomicMax(out + row, 1ull);   // keeps the loads alive, (almost) never taken
}
extern "C" int probe_stream(const void *logits, int esz, int64_t R, int64_t V, int64_t stride_elems, void *out, int64_t chunk, void *stream) {
    const int64_t cpr = (V + chunk - 1) / chunk;
    probe_kernel<<<dim3((unsigned)(R * cpr)), TPB, 0, (hipStream_t)stream>>>(logits, V, stride_elems * esz, esz, (unsigned long long *)out, (int)cpr, chunk);
    return (int)hipGetLastError();
}

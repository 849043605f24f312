// Sustained MFMA rate probe: v_mfma_f32_32x32x16_bf16 from registers only (no memory traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) probe(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(threadIdx.x * 3 + e); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters, const char* name) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NACC><<<blocks, 256>>>(out, 10);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        probe<NACC><<<blocks, 256>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 * iters * 4 * NACC * 2.0 * 32 * 32 * 16;
        printf("%-34s blocks %5d  %8.3f ms  %8.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
    }
    hipFree(out);
}
int main() {
    run<4>(256, 20000, "1 wave/SIMD, 4 independent acc");
    run<4>(512, 20000, "2 waves/SIMD, 4 independent acc");
    run<4>(1024, 20000, "4 waves/SIMD, 4 independent acc");
    run<1>(512, 40000, "2 waves/SIMD, 1 dependent chain");
    run<4>(512, 200000, "2 waves/SIMD, 4 acc, long (~10x)");
    return 0;
}

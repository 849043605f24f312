// Does the sustained rate of v_mfma_f32_32x32x16_bf16 depend on the DATA?  (tools/probes/mfma_peak.hip multiplies small constant
// integers and sustains ~2.06 PFLOP/s with 2 waves/SIMD.)  Real GEMM operands toggle every multiplier input between consecutive
// instructions; if the chip is power-limited the matrix clock drops.  Variants: operands all zero / small constants / random bf16,
// one operand pair or four pairs rotating (fresh inputs every instruction, as in a GEMM K loop).  Registers only, no memory traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE, int NPAIR>   // MODE 0 zeros, 1 small constants, 2 random bf16 in [-2, 2)
__global__ void __launch_bounds__(256) probe(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[NPAIR], b[NPAIR];
    for (int p = 0; p < NPAIR; ++p)
        for (int e = 0; e < 8; ++e) {
            float va = 0.f, vb = 0.f;
            if (MODE == 1) { va = (float)((threadIdx.x + e + p) & 7); vb = (float)((threadIdx.x * 3 + e + p) & 7); }
            if (MODE == 2) {
                const unsigned h = hash32(threadIdx.x * 977u + e * 131u + p * 7919u + blockIdx.x);
                va = (float)(int)(h & 0xFFFF) / 16384.f - 2.f;
                vb = (float)(int)(h >> 16) / 16384.f - 2.f;
            }
            a[p][e] = (__bf16)va; b[p][e] = (__bf16)vb;
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < NPAIR; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[p], b[(p + i) % NPAIR], acc[i], 0, 0, 0);
        if (MODE == 2) {          // keep the accumulators bounded (and their bits busy): scale down now and then
            if ((it & 63) == 63) for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] *= 1e-3f;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NPAIR>
void run(int blocks, int iters, const char* name) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, NPAIR><<<blocks, 256>>>(out, 10);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        probe<MODE, NPAIR><<<blocks, 256>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 * (double)iters * NPAIR * 4 * 2.0 * 32 * 32 * 16;
        printf("%-52s %8.3f ms  %8.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
    }
    hipFree(out);
}
int main() {
    // 512 blocks x 4 waves = 2 waves / SIMD on 256 CUs; ~50 ms per launch
    run<0, 1>(512, 400000, "zeros, 1 operand pair");
    run<1, 1>(512, 400000, "small constants, 1 operand pair");
    run<2, 1>(512, 400000, "random bf16, 1 operand pair");
    run<2, 4>(512, 100000, "random bf16, 4 rotating operand pairs");
    run<1, 4>(512, 100000, "small constants, 4 rotating operand pairs");
    run<2, 4>(256, 100000, "random bf16, 4 pairs, 1 wave/SIMD");
    return 0;
}

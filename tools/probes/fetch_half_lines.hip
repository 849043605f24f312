// Does rocprofv3's FETCH_SIZE (x2 on gfx950, MI355X_MICROARCH.md) over-count a kernel that touches only 64 bytes of every 128-byte line?
// The 3 x 3 GEMMs with 64 input channels fetch a 32-channel chunk (64 B) of each 128-byte pixel per K pass.
//   full_lines : every byte of a buffer once, 16 B per lane, consecutive lanes consecutive addresses
//   half_lines : bytes [0, 64) of every 128-byte line (4 lanes per line), i.e. HALF the buffer's bytes, once
//   half_twice : first the low halves of all lines, then (second launch) the high halves: every byte once, in two passes
// build: hipcc --offload-arch=gfx950 -O3 -o tools/_ablate/fetch/fetch_half_lines tools/probes/fetch_half_lines.hip
// run  : rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <dir> -- tools/_ablate/fetch/fetch_half_lines   (then tools/pmc_summary.py or read the csv)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void full_lines(const f4* p, long long n16, float* out) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) acc += p[i];
    if (acc.x == 1234.5f) out[0] = acc.y;
}
__global__ void half_lines(const f4* p, long long nlines, int which, float* out) {      // which: 0 low half, 1 high half of each 128-byte line
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nlines * 4; i += (long long)gridDim.x * blockDim.x)
        acc += p[(i >> 2) * 8 + which * 4 + (i & 3)];
    if (acc.x == 1234.5f) out[0] = acc.y;
}
int main() {
    const long long bytes = 1LL << 30;                       // 1 GiB: far beyond L2 + MALL
    f4* buf; float* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 64);
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    for (int r = 0; r < 2; ++r) {
        full_lines<<<4096, 256>>>(buf, bytes / 16, out);
        hipDeviceSynchronize();
        half_lines<<<4096, 256>>>(buf, bytes / 128, 0, out);
        hipDeviceSynchronize();
        half_lines<<<4096, 256>>>(buf, bytes / 128, 1, out);
        hipDeviceSynchronize();
    }
    printf("buffer %lld MiB: full_lines reads it all, each half_lines launch reads half of it\n", bytes >> 20);
    return 0;
}

// k_data.hip — device-side input pipeline.
//
// Replaces Dataset_Aug1 / Dataset + DataLoader of deblurring_diffusion_pytorch.py:983-1026, 1094-1096 (8-16 PIL worker
// processes: Resize 1.12x -> RandomCrop / CenterCrop -> RandomHorizontalFlip -> ToTensor -> t * 2 - 1).  The Resize of that
// chain is deterministic, so it is applied ONCE when the dataset is cached: the cache is [N][S][S][C] uint8 in HBM (S =
// int(1.12 image_size); CelebA's 202 599 images at S = 143 are 12.4 GB of the 288 GB).  Per batch ONE kernel does the rest:
// gather B cached images by index, crop at (oy, ox), mirror, and convert with exactly ToTensor's arithmetic
// (float(v) / 255, IEEE division) followed by t * 2 - 1 in fp32, writing the NCHW fp32 batch the diffusion classes take.
// HBM-bound and tiny: 49 KB read + 196 KB written per 128 x 128 image.
#include "cdf_common.h"
#include "colddiff.h"

__global__ void augment_batch_kernel(const unsigned char* cache, const long long* idx, const int* oy, const int* ox, const int* flip,
                                     float* out, int SH, int SW, int C, int pad, int H, int W) {
    // (oy, ox) address the image after RandomCrop's `padding=pad` border (zero fill); pad = 0 for the plain chains
    const int b = blockIdx.y, y = blockIdx.x;
    const int sy = oy[b] + y - pad;
    const bool row_in = sy >= 0 && sy < SH;
    const unsigned char* src = cache + ((size_t)idx[b] * SH + (size_t)(row_in ? sy : 0)) * SW * C;
    const int x0 = ox[b] - pad, fl = flip[b];
    float* dst = out + ((size_t)b * C * H + y) * W;          // + c * H * W + x
    for (int i = threadIdx.x; i < W * C; i += blockDim.x) {
        const int c = i / W, x = i - c * W;                  // consecutive lanes: consecutive x of one channel plane (coalesced stores)
        const int sx = x0 + (fl ? W - 1 - x : x);
        const bool in = row_in && sx >= 0 && sx < SW;
        const unsigned char u = src[(in ? sx : 0) * C + c];  // unconditional load from a clamped address, then select
        const float v = (float)(in ? u : (unsigned char)0) / 255.0f;     // ToTensor: uint8 -> float32, div(255)
        dst[(size_t)c * H * W + x] = v * 2.0f - 1.0f;        // Lambda(t * 2 - 1)   (no FMA contraction: -ffp-contract=off)
    }
}

extern "C" int cdf_augment_batch_pad(const void* cache, long long N, int SH, int SW, int C, int pad, const long long* idx, const int* oy,
                                     const int* ox, const int* flip, float* out, int B, int H, int W, void* stream) {
    CDF_REQUIRE(cache && idx && oy && ox && flip && out, "cdf_augment_batch: null pointer");
    CDF_REQUIRE(N > 0 && B > 0 && C >= 1 && C <= 4 && H > 0 && W > 0 && pad >= 0 && H <= SH + 2 * pad && W <= SW + 2 * pad,
                "cdf_augment_batch: bad geometry (crop %dx%d of %dx%d + %d border, %d channels)", H, W, SH, SW, pad, C);
    CDF_LAUNCH(augment_batch_kernel, dim3(H, B), dim3(256), 0, CDF_S, (const unsigned char*)cache, idx, oy, ox, flip, out, SH, SW, C, pad, H, W);
    return cdf_check_launch("augment_batch");
}

extern "C" int cdf_augment_batch(const void* cache, long long N, int S, int C, const long long* idx, const int* oy, const int* ox,
                                 const int* flip, float* out, int B, int H, int W, void* stream) {
    return cdf_augment_batch_pad(cache, N, S, S, C, 0, idx, oy, ox, flip, out, B, H, W, stream);
}

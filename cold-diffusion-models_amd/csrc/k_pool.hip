// k_pool.hip — the non-GEMM layers of the FID feature extractor (deblurring-diffusion-pytorch/Fid/inception.py).
//
// InceptionV3 as pytorch-fid patches it is BasicConv2d (conv + folded BatchNorm + ReLU: the GEMM kernels with act = 3) plus
//   * nn.MaxPool2d(3, stride 2)                                        (inception.py:91, 100; InceptionB / InceptionD pool branches)
//   * F.avg_pool2d(3, stride 1, padding 1, count_include_pad=False)    (inception.py:214, 243, 282: "Tensorflow's average pool")
//   * F.max_pool2d(3, stride 1, padding 1)                             (inception.py:323: FIDInceptionE_2)
//   * nn.AdaptiveAvgPool2d((1, 1))                                     (inception.py:122)
//   * F.interpolate(size=(299, 299), bilinear, align_corners=False) and 2 x - 1   (inception.py:146-153)
// All HBM-bound streams over NHWC feature maps (float4 lanes over channels; a pixel pitch `ld` so that a branch can write its
// channel slice of the concatenated block output directly).
#include "cdf_common.h"
#include "colddiff.h"

// mode 0: max (padding never wins: -inf), 1: average over the taps INSIDE the image (count_include_pad = False)
__global__ void pool2d_kernel(const float* x, int ldx, float* y, int ldy, int H, int W, int C4, int OH, int OW, int k, int stride, int pad,
                              int mode, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        long long p = i / C4;
        const int ox = (int)(p % OW);
        p /= OW;
        const int oy = (int)(p % OH), b = (int)(p / OH);
        const int y0 = oy * stride - pad, x0 = ox * stride - pad;
        float4 acc = mode == 0 ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY) : make_float4(0.f, 0.f, 0.f, 0.f);
        int cnt = 0;
        for (int dy = 0; dy < k; ++dy) {
            const int sy = y0 + dy;
            if (sy < 0 || sy >= H) continue;
            for (int dx = 0; dx < k; ++dx) {
                const int sx = x0 + dx;
                if (sx < 0 || sx >= W) continue;
                const float4 v = *(const float4*)(x + (((long long)b * H + sy) * W + sx) * ldx + c);
                if (mode == 0) {
                    acc.x = fmaxf(acc.x, v.x); acc.y = fmaxf(acc.y, v.y); acc.z = fmaxf(acc.z, v.z); acc.w = fmaxf(acc.w, v.w);
                } else {
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;     // row-major tap order, like ATen's avg_pool2d loop
                }
                ++cnt;
            }
        }
        if (mode == 1) {
            const float d = (float)cnt;
            acc.x /= d; acc.y /= d; acc.z /= d; acc.w /= d;
        }
        *(float4*)(y + (((long long)b * OH + oy) * OW + ox) * ldy + c) = acc;
    }
}

extern "C" int cdf_pool2d(const float* x, int ldx, float* y, int ldy, int B, int H, int W, int C, int k, int stride, int pad, int mode,
                          void* stream) {
    CDF_REQUIRE(x && y, "cdf_pool2d: null pointer");
    CDF_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C,
                "cdf_pool2d: channels / pitches must be multiples of 4 (C %d, ldx %d, ldy %d)", C, ldx, ldy);
    CDF_REQUIRE(k >= 1 && stride >= 1 && pad >= 0 && 2 * pad <= k && (mode == 0 || mode == 1), "cdf_pool2d: bad window (k %d, stride %d, pad %d, mode %d)", k,
                stride, pad, mode);
    const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
    CDF_REQUIRE(OH > 0 && OW > 0, "cdf_pool2d: window larger than the image");
    const long long total = (long long)B * OH * OW * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 256 * 64 ? (total + 255) / 256 : 256 * 64);
    CDF_LAUNCH(pool2d_kernel, dim3(blocks), dim3(256), 0, CDF_S, x, ldx, y, ldy, H, W, C / 4, OH, OW, k, stride, pad, mode, total);
    return cdf_check_launch("pool2d");
}

// y[b][c] = mean over the H x W pixels (AdaptiveAvgPool2d((1, 1))): one block per (image, 64-channel group), 4 pixel strides per block
__global__ void global_avgpool_kernel(const float* x, int ldx, float* y, int ldy, int HW, int C) {
    __shared__ float part[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < C)
        for (int p = g; p < HW; p += 4) acc += x[((long long)b * HW + p) * ldx + c];
    part[g][threadIdx.x & 63] = acc;
    __syncthreads();
    if (g == 0 && c < C) y[(long long)b * ldy + c] = (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]) / (float)HW;
}

extern "C" int cdf_global_avgpool(const float* x, int ldx, float* y, int ldy, int B, int HW, int C, void* stream) {
    CDF_REQUIRE(x && y && B > 0 && HW > 0 && C > 0 && ldx >= C && ldy >= C, "cdf_global_avgpool: bad arguments");
    CDF_LAUNCH(global_avgpool_kernel, dim3((C + 63) / 64, B), dim3(256), 0, CDF_S, x, ldx, y, ldy, HW, C);
    return cdf_check_launch("global_avgpool");
}

// NCHW image batch -> NHWC (pitch ldy, pad channels zeroed by the caller) bilinear resize with ATen's align_corners=False source
// index (scale = in / out in fp32, src = scale (dst + 0.5) - 0.5 clamped at 0, neighbour clamped at in - 1), then a * v + s.
__global__ void resize_bilinear_kernel(const float* x, float* y, int ldy, int C, int H, int W, int OH, int OW, float sh, float sw, float a, float s,
                                       long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % OW);
        long long p = i / OW;
        const int oy = (int)(p % OH);
        p /= OH;
        const int c = (int)(p % C), b = (int)(p / C);
        float fy = sh * ((float)oy + 0.5f) - 0.5f, fx = sw * ((float)ox + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float* src = x + ((long long)b * C + c) * H * W;
        const float top = (1.f - lx) * src[(long long)y0 * W + x0] + lx * src[(long long)y0 * W + x1];
        const float bot = (1.f - lx) * src[(long long)y1 * W + x0] + lx * src[(long long)y1 * W + x1];
        const float v = (1.f - ly) * top + ly * bot;
        y[(((long long)b * OH + oy) * OW + ox) * ldy + c] = a * v + s;
    }
}

extern "C" int cdf_resize_bilinear_nhwc(const float* x, float* y, int ldy, int B, int C, int H, int W, int OH, int OW, float mul, float add,
                                        void* stream) {
    CDF_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && ldy >= C, "cdf_resize_bilinear_nhwc: bad arguments");
    const long long total = (long long)B * C * OH * OW;
    const int blocks = (int)((total + 255) / 256 < 256 * 64 ? (total + 255) / 256 : 256 * 64);
    CDF_LAUNCH(resize_bilinear_kernel, dim3(blocks), dim3(256), 0, CDF_S, x, y, ldy, C, H, W, OH, OW, (float)H / (float)OH, (float)W / (float)OW, mul,
               add, total);
    return cdf_check_launch("resize_bilinear_nhwc");
}

// k_metrics.hip — the metric step after sampling: SSIM of two image batches.
//
// Replaces pytorch_msssim.ssim(X, Y, data_range, size_average=True) as called by the reference's evaluation code
// (deblurring_diffusion_pytorch.py:1677-1702): 11-tap Gaussian window (sigma 1.5), "valid" separable filtering of X, Y, X^2, Y^2,
// XY per channel plane, ssim_map = (2 mu1 mu2 + C1)/(mu1^2 + mu2^2 + C1) * (2 s12 + C2)/(s1 + s2 + C2), mean over the
// (H-10) x (W-10) valid positions of every plane.  pytorch_msssim does this with ten grouped conv2d launches and a dozen
// elementwise kernels, every intermediate (5 filtered maps x 2 passes) round-tripping through HBM; here a workgroup keeps a
// 42 x 42 tile of both images in LDS, filters the five products along H then along W (the library's order) on chip and
// reduces its 32 x 32 SSIM values to one partial sum: the images are read once (8 B / pixel), nothing else is written.
// RMSE is cdf_loss_fwd(l2 = 1) followed by a square root (k_degrade.hip).
#include "cdf_common.h"
#include "colddiff.h"

#define SSIM_WIN 11
#define SSIM_T 32                       // output tile edge
#define SSIM_IN (SSIM_T + SSIM_WIN - 1) // 42

struct SsimWin { float w[SSIM_WIN]; };

__global__ void __launch_bounds__(256) ssim_partial_kernel(const float* X, const float* Y, float* partial, int H, int W, int tiles_x,
                                                          int tiles_y, float C1, float C2, SsimWin win) {
    __shared__ float sx[SSIM_IN][SSIM_IN + 1], sy[SSIM_IN][SSIM_IN + 1];
    __shared__ float v[5][SSIM_T][SSIM_IN + 1];            // after the pass along H: [quantity][out row][in col]
    __shared__ float red[4];
    const int plane = blockIdx.z, ty = blockIdx.y, tx = blockIdx.x, tid = threadIdx.x;
    const int OH = H - SSIM_WIN + 1, OW = W - SSIM_WIN + 1;
    const int y0 = ty * SSIM_T, x0 = tx * SSIM_T;
    const float* px = X + (size_t)plane * H * W;
    const float* py = Y + (size_t)plane * H * W;
    for (int i = tid; i < SSIM_IN * SSIM_IN; i += 256) {
        const int r = i / SSIM_IN, c = i - r * SSIM_IN;
        const int gy = y0 + r, gx = x0 + c;
        const bool ok = gy < H && gx < W;
        sx[r][c] = ok ? px[(size_t)gy * W + gx] : 0.f;
        sy[r][c] = ok ? py[(size_t)gy * W + gx] : 0.f;
    }
    __syncthreads();
    // pass along H (dim 2 first, as pytorch_msssim's gaussian_filter does): out row r, input column c
    for (int i = tid; i < SSIM_T * SSIM_IN; i += 256) {
        const int r = i / SSIM_IN, c = i - r * SSIM_IN;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
        for (int k = 0; k < SSIM_WIN; ++k) {
            const float xv = sx[r + k][c], yv = sy[r + k][c], w = win.w[k];
            a0 += w * xv;
            a1 += w * yv;
            a2 += w * (xv * xv);
            a3 += w * (yv * yv);
            a4 += w * (xv * yv);
        }
        v[0][r][c] = a0; v[1][r][c] = a1; v[2][r][c] = a2; v[3][r][c] = a3; v[4][r][c] = a4;
    }
    __syncthreads();
    // pass along W, SSIM value, tile sum
    float acc = 0.f;
    for (int i = tid; i < SSIM_T * SSIM_T; i += 256) {
        const int r = i / SSIM_T, c = i - r * SSIM_T;
        if (y0 + r < OH && x0 + c < OW) {
            float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
            for (int k = 0; k < SSIM_WIN; ++k) {
                const float w = win.w[k];
                m1 += w * v[0][r][c + k];
                m2 += w * v[1][r][c + k];
                e11 += w * v[2][r][c + k];
                e22 += w * v[3][r][c + k];
                e12 += w * v[4][r][c + k];
            }
            const float mu1_sq = m1 * m1, mu2_sq = m2 * m2, mu12 = m1 * m2;
            const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;       // (compensation = 1.0)
            const float cs = (2.f * s12 + C2) / (s1 + s2 + C2);
            acc += ((2.f * mu12 + C1) / (mu1_sq + mu2_sq + C1)) * cs;
        }
    }
    acc = cdf_wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) partial[((size_t)plane * tiles_y + ty) * tiles_x + tx] = red[0] + red[1] + red[2] + red[3];
}

extern "C" int cdf_ssim_tiles(int H, int W) {
    if (H < SSIM_WIN || W < SSIM_WIN) return 0;
    return cdf_cdiv(H - SSIM_WIN + 1, SSIM_T) * cdf_cdiv(W - SSIM_WIN + 1, SSIM_T);
}

extern "C" int cdf_ssim_partial(const float* x, const float* y, float* partial, int planes, int H, int W, const float* window11,
                                float C1, float C2, void* stream) {
    CDF_REQUIRE(x && y && partial && window11 && planes > 0, "cdf_ssim_partial: null pointer");
    CDF_REQUIRE(H >= SSIM_WIN && W >= SSIM_WIN, "cdf_ssim_partial: images must be at least 11 x 11 (got %d x %d)", H, W);
    SsimWin win;
    for (int k = 0; k < SSIM_WIN; ++k) win.w[k] = window11[k];      // (host pointer: 11 taps by value)
    const int tx = cdf_cdiv(W - SSIM_WIN + 1, SSIM_T), ty = cdf_cdiv(H - SSIM_WIN + 1, SSIM_T);
    CDF_LAUNCH(ssim_partial_kernel, dim3(tx, ty, planes), dim3(256), 0, CDF_S, x, y, partial, H, W, tx, ty, C1, C2, win);
    return cdf_check_launch("ssim_partial");
}

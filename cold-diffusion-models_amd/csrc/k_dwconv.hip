// k_dwconv.hip — depthwise 7x7 convolution of the ConvNeXt block on NHWC feature maps.
//   reference: nn.Conv2d(dim, dim, 7, padding=3, groups=dim) + time-embedding bias
//   deblurring_diffusion_pytorch.py:145,157-162
// HBM-bound (49 MAC per element): each thread produces a 4-pixel x 4-channel strip so that one
// 10x7 window of float4 loads feeds 196 float4 FMAs; lanes run along channels (coalesced 16 B).
// Weights are packed [49][Cp] (cdf_pack_weight with R=1), Cp = C rounded up to 4, zero padded.
#include "cdf_common.h"
#include "colddiff.h"

#define DW_K 7
#define DW_TAPS 49

__device__ __forceinline__ void f4_fma(float4& acc, const float4& a, const float4& b) {
    acc.x = fmaf(a.x, b.x, acc.x);
    acc.y = fmaf(a.y, b.y, acc.y);
    acc.z = fmaf(a.z, b.z, acc.z);
    acc.w = fmaf(a.w, b.w, acc.w);
}

// y[b,y,x,c] = sum_k x[b,y+ky-3,x+kx-3,c] * w[k][c] (+ bias[c] + sbias[b][c]);  flip => mirrored taps (dgrad)
__global__ void __launch_bounds__(256) dwconv7_kernel(const float* x, int ldx, const float* w, int ldw, const float* bias,
                                                      const float* sbias, int ld_sbias, float* y, int ldy, int B, int H,
                                                      int W, int C4, int flip, int accumulate) {
    const int strips_w = (W + 3) / 4;
    const long long n = (long long)B * H * strips_w * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        long long r = i / C4;
        const int xs = (int)(r % strips_w) * 4;
        r /= strips_w;
        const int yy = (int)(r % H), b = (int)(r / H);
        float4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ky = 0; ky < DW_K; ++ky) {
            const int iy = yy + ky - 3;
            if (iy < 0 || iy >= H) continue;
            const float* row = x + (((long long)b * H + iy) * W) * ldx + c;
            float4 win[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                const int ix = xs + q - 3;
                win[q] = (ix >= 0 && ix < W) ? *(const float4*)(row + (long long)ix * ldx) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int kx = 0; kx < DW_K; ++kx) {
                const int tap = flip ? (6 - ky) * DW_K + (6 - kx) : ky * DW_K + kx;
                const float4 wv = *(const float4*)(w + (long long)tap * ldw + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) f4_fma(acc[j], win[kx + j], wv);
            }
        }
        float4 add = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) {
            const float4 bv = *(const float4*)(bias + c);
            add.x += bv.x; add.y += bv.y; add.z += bv.z; add.w += bv.w;
        }
        if (sbias) {
            const float4 sv = *(const float4*)(sbias + (long long)b * ld_sbias + c);
            add.x += sv.x; add.y += sv.y; add.z += sv.z; add.w += sv.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (xs + j >= W) break;
            float* dst = y + (((long long)b * H + yy) * W + xs + j) * ldy + c;
            float4 o = make_float4(acc[j].x + add.x, acc[j].y + add.y, acc[j].z + add.z, acc[j].w + add.w);
            if (accumulate) {
                const float4 old = *(const float4*)dst;
                o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            }
            *(float4*)dst = o;
        }
    }
}

// weight gradient partials: part[((b*nchunk + chunk)*50 + tap)*C + c], tap 49 = sum of dy
// grid = (ceil(C/64), nchunk, B), block 256 = 4 waves; lane = channel.
__global__ void __launch_bounds__(256) dwconv7_wgrad_partial_kernel(const float* x, int ldx, const float* dy, int lddy,
                                                                   float* part, int H, int W, int C, int rows_per_chunk) {
    __shared__ float red[4][DW_TAPS + 1][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, b = blockIdx.z;
    const bool cv = c < C;
    const int y0 = blockIdx.y * rows_per_chunk;
    int y1 = y0 + rows_per_chunk;
    if (y1 > H) y1 = H;
    const int strips_w = (W + 3) / 4, nstrips = (y1 - y0) * strips_w;
    float acc[DW_TAPS + 1];
#pragma unroll
    for (int t = 0; t <= DW_TAPS; ++t) acc[t] = 0.f;
    for (int s = wave; s < nstrips; s += 4) {
        const int yy = y0 + s / strips_w, xs = (s % strips_w) * 4;
        float d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            d[j] = (cv && xs + j < W) ? dy[(((long long)b * H + yy) * W + xs + j) * lddy + c] : 0.f;
            acc[DW_TAPS] += d[j];
        }
#pragma unroll
        for (int ky = 0; ky < DW_K; ++ky) {
            const int iy = yy + ky - 3;
            const bool rowok = iy >= 0 && iy < H;
            const float* row = x + (((long long)b * H + (rowok ? iy : 0)) * W) * ldx + c;
            float win[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                const int ix = xs + q - 3;
                win[q] = (cv && rowok && ix >= 0 && ix < W) ? row[(long long)ix * ldx] : 0.f;
            }
#pragma unroll
            for (int kx = 0; kx < DW_K; ++kx)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[ky * DW_K + kx] = fmaf(win[kx + j], d[j], acc[ky * DW_K + kx]);
        }
    }
    // cross-wave reduction through LDS (static indices only: acc[] must stay in registers)
#pragma unroll
    for (int t = 0; t <= DW_TAPS; ++t) red[wave][t][lane] = acc[t];
    __syncthreads();
    float* dst = part + (((long long)b * gridDim.y + blockIdx.y) * (DW_TAPS + 1)) * C;
    for (int i = threadIdx.x; i < (DW_TAPS + 1) * 64; i += 256) {
        const int t = i >> 6, l = i & 63, cc = blockIdx.x * 64 + l;
        if (cc < C) dst[(long long)t * C + cc] = (red[0][t][l] + red[1][t][l]) + (red[2][t][l] + red[3][t][l]);
    }
}

// dw[c*49 + tap] (+)= sum_{b,chunk} part ; dbias[c] (+)= sum part[..][49] ; dsb[b][c] = sum_chunk part[b][..][49]
// grid = (ceil(C/64), 50 taps); block 256 = 4 partial lanes x 64 channels (lanes split the b*nchunk partials)
__global__ void dwconv7_wgrad_final_kernel(const float* part, int B, int nchunk, int C, float* dw, float* dbias,
                                           float* dsb, int ld_dsb, int accumulate) {
    __shared__ float red[4][64];
    const int l = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + l, t = blockIdx.y;
    const int n = B * nchunk;
    float s = 0.f;
    if (c < C)
        for (int i = rl; i < n; i += 4) s += part[((long long)i * (DW_TAPS + 1) + t) * C + c];
    red[rl][l] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        const float tot = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
        if (t < DW_TAPS) {
            float* dst = dw + (long long)c * DW_TAPS + t;
            *dst = accumulate ? *dst + tot : tot;
        } else if (dbias) {
            dbias[c] = accumulate ? dbias[c] + tot : tot;
        }
    }
    if (t == DW_TAPS && dsb) {          // per-sample time-bias gradient: sum over the chunks of each sample
        __syncthreads();
        for (int b = rl; b < B; b += 4) {
            if (c < C) {
                float sb = 0.f;
                for (int k = 0; k < nchunk; ++k) sb += part[(((long long)b * nchunk + k) * (DW_TAPS + 1) + DW_TAPS) * C + c];
                dsb[(long long)b * ld_dsb + c] = sb;
            }
        }
    }
}

// ================================================================================================
extern "C" int cdf_dwconv7(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* sbias,
                           int ld_sbias, float* y, int ldy, int B, int H, int W, int C, int flip, int accumulate,
                           void* stream) {
    CDF_REQUIRE(x && w && y, "cdf_dwconv7: null pointer");
    const int Cp = (C + 3) & ~3;
    CDF_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ldw % 4 == 0 && ldx >= Cp && ldy >= Cp && ldw >= Cp, "cdf_dwconv7: pitches must be multiples of 4 and >= roundup4(C)");
    CDF_REQUIRE(!bias || (C % 4 == 0), "cdf_dwconv7: bias with C %% 4 != 0 needs a padded bias (pass a padded vector and C rounded up)");
    const long long n = (long long)B * H * ((W + 3) / 4) * (Cp / 4);
    long long grid = (n + 255) / 256;
    if (grid > 8192) grid = 8192;
    CDF_LAUNCH(dwconv7_kernel, dim3((int)grid), dim3(256), 0, CDF_S, x, ldx, w, ldw, bias, sbias, ld_sbias, y, ldy, B, H, W, Cp / 4, flip, accumulate);
    return cdf_check_launch("dwconv7");
}

extern "C" int cdf_dwconv7_wgrad_nchunk(int H) {
    int n = H / 4;
    if (n < 1) n = 1;
    if (n > 32) n = 32;
    return n;
}

// ws >= B * nchunk * 50 * C floats; dw in the PyTorch layout [C][1][7][7]; dsb [B][ld_dsb] (overwritten)
extern "C" int cdf_dwconv7_wgrad(const float* x, int ldx, const float* dy, int lddy, float* dw, float* dbias,
                                 float* dsb, int ld_dsb, float* ws, int B, int H, int W, int C, int accumulate,
                                 void* stream) {
    CDF_REQUIRE(x && dy && dw && ws, "cdf_dwconv7_wgrad: null pointer");
    const int nchunk = cdf_dwconv7_wgrad_nchunk(H), rpc = cdf_cdiv(H, nchunk);
    CDF_LAUNCH(dwconv7_wgrad_partial_kernel, dim3(cdf_cdiv(C, 64), nchunk, B), dim3(256), 0, CDF_S, x, ldx, dy, lddy, ws, H, W, C, rpc);
    CDF_LAUNCH(dwconv7_wgrad_final_kernel, dim3(cdf_cdiv(C, 64), DW_TAPS + 1), dim3(256), 0, CDF_S, (const float*)ws, B, nchunk, C, dw, dbias, dsb, ld_dsb, accumulate);
    return cdf_check_launch("dwconv7_wgrad");
}

// k_degrade.hip — the deterministic degradation operators D(x,t) of cold diffusion and the
// Algorithm-2 (x0_step_down) combine, as gfx950 kernels.  All image tensors here are the public
// NCHW fp32 tensors of the GaussianDiffusion API ([B,C,H,W]); a (b,c) plane is the unit of work.
//
// Reference behaviour restated (file:line in /root/reference):
//   blur   : deblurring_diffusion_pytorch.py:351-361 (depthwise Conv2d, circular|reflect),
//            :927-960 (q_sample: apply K_0..K_t[b], pick per sample), :436-451 (Alg. 2)
//   mask   : defading_diffusion_gaussian.py:496-535, :405-420
//   pixel  : resolution_diffusion_pytorch.py:354-385 (F.interpolate down, nearest-exact up)
//   noise  : denoising_diffusion_pytorch.py:517-522, :413-432
//
// Design (MI355X): a 128x128 fp32 plane is 64 KB, so ONE workgroup keeps the whole plane in the
// CU's 160 KB LDS and runs the complete step chain 0..t[b] on-chip: HBM traffic is one read and
// one write of the plane regardless of t (the reference stacks every intermediate in HBM).
#include "cdf_common.h"
#include "colddiff.h"

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int cdf_pad_index(int i, int n, int mode) {
    // mode 0: circular, 1: reflect (no edge repeat), 2: replicate
    if (mode == 0) {
        i %= n;
        if (i < 0) i += n;
        return i;
    } else if (mode == 1) {
        if (n == 1) return 0;
        const int period = 2 * (n - 1);
        i %= period;
        if (i < 0) i += period;
        return i < n ? i : period - i;
    }
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// 8-bit quantisation used by discrete=True (DEBLUR:954-958): truncation toward zero.
__device__ __forceinline__ float cdf_quantise8(float v) {
    float a = (v + 1.0f) * 0.5f;
    a = a * 255.0f;
    a = (float)((int)a);
    a = a / 255.0f;
    return a * 2.0f - 1.0f;
}

// block-wide sum (blockDim.x multiple of 64, <= 1024); result valid in all threads
__device__ __forceinline__ float cdf_block_sum(float v, float* red /*>=17 floats LDS*/) {
    v = cdf_wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < nw; ++i) s += red[i];
        red[16] = s;
    }
    __syncthreads();
    return red[16];
}

struct BlurArgs {
    const float* x;       // [B,C,H,W] input
    float* y;             // [B,C,H,W] output (state after step hi(b)) or combine result
    float* snap;          // nullable: state after step hi(b)-1 (input itself if hi(b)==lo)
    const float* img;     // nullable: if set, y = img - D_hi + D_{hi-1}   (Alg. 2)
    const float* taps;    // [nsteps][C][k][k]
    const int64_t* t;     // nullable: per-sample last step index (inclusive)
    int B, C, H, W, k;
    int step_lo;          // first step applied
    int step_hi;          // last step applied (inclusive) when t == nullptr; < step_lo => identity
    int pad_mode;         // 0 circular, 1 reflect
    int collapse_step;    // -1 or step index after which the plane is replaced by its mean
    int quantise;         // apply cdf_quantise8 to the final output
};

// ------------------------------------------------------------------------------------------------
// Fused multi-step blur, plane resident in LDS.  grid = B*C, block = NT threads.
// LDS: P = padded plane [(H+2h)][PW] (PW = roundup4(W+2h)), U = plane [H][W], red[32]
// ------------------------------------------------------------------------------------------------
template <int K, int SW>
__global__ void __launch_bounds__(1024) blur_plane_lds_kernel(BlurArgs a) {
    CDF_DYN_SMEM(smem);
    const int H = a.H, W = a.W, k = (K > 0 ? K : a.k), h = k / 2;
    const int PW = (W + 2 * h + 3) & ~3, PH = H + 2 * h;
    float* P = (float*)smem;
    float* U = P + (size_t)PW * PH;
    float* red = U + (size_t)H * W;
    const int plane = blockIdx.x, b = plane / a.C, c = plane % a.C;
    const int nt = blockDim.x, tid = threadIdx.x;
    const size_t poff = (size_t)plane * H * W;
    const int hi = a.t ? (int)a.t[b] : a.step_hi;

    for (int i = tid; i < H * W; i += nt) U[i] = a.x[poff + i];
    __syncthreads();
    const int strips_per_row = W / SW, nstrips = strips_per_row * H;
    for (int s = a.step_lo; s <= hi; ++s) {
        // halo fill: P[py][px] = U[map(py-h)][map(px-h)]
        for (int i = tid; i < PH * (W + 2 * h); i += nt) {
            const int py = i / (W + 2 * h), px = i - py * (W + 2 * h);
            const int sy = cdf_pad_index(py - h, H, a.pad_mode), sx = cdf_pad_index(px - h, W, a.pad_mode);
            P[py * PW + px] = U[sy * W + sx];
        }
        __syncthreads();
        const float* wt = a.taps + ((size_t)s * a.C + c) * k * k;
        for (int st = tid; st < nstrips; st += nt) {
            const int y = st / strips_per_row, x0 = (st - y * strips_per_row) * SW;
            float acc[SW];
#pragma unroll
            for (int j = 0; j < SW; ++j) acc[j] = 0.f;
            if (K > 0) {
                constexpr int RW = (K > 0 ? SW + K - 1 : 1);
                constexpr int RW4 = (RW + 3) / 4 * 4;
#pragma unroll 1
                for (int ky = 0; ky < K; ++ky) {
                    float r[RW4];
                    const float* row = P + (y + ky) * PW + x0;
#pragma unroll
                    for (int q = 0; q < RW4 / 4; ++q) {
                        const float4 v = *(const float4*)(row + 4 * q);
                        r[4 * q + 0] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
                    }
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        const float wv = wt[ky * K + kx];
#pragma unroll
                        for (int j = 0; j < SW; ++j) acc[j] = fmaf(wv, r[kx + j], acc[j]);
                    }
                }
            } else {
                for (int ky = 0; ky < k; ++ky) {
                    const float* row = P + (y + ky) * PW + x0;
                    for (int kx = 0; kx < k; ++kx) {
                        const float wv = wt[ky * k + kx];
#pragma unroll
                        for (int j = 0; j < SW; ++j) acc[j] = fmaf(wv, row[kx + j], acc[j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < SW; ++j) U[y * W + x0 + j] = acc[j];
        }
        __syncthreads();
        if (s == a.collapse_step) {
            float part = 0.f;
            for (int i = tid; i < H * W; i += nt) part += U[i];
            const float mean = cdf_block_sum(part, red) / (float)(H * W);
            for (int i = tid; i < H * W; i += nt) U[i] = mean;
            __syncthreads();
        }
    }
    // D(x, hi-1): the interior of P still holds the state the last step started from; with no
    // step applied (hi < lo) it is the input itself.
    const bool stepped = hi >= a.step_lo;
    if (a.snap) {
        for (int i = tid; i < H * W; i += nt) {
            const int y = i / W, x = i - y * W;
            a.snap[poff + i] = stepped ? P[(y + h) * PW + x + h] : U[i];
        }
    }
    if (a.img) {
        // Alg. 2: x = img - D(x0,t) + D(x0,t-1)   (DEBLUR:451)
        for (int i = tid; i < H * W; i += nt) {
            const int y = i / W, x = i - y * W;
            const float prev = stepped ? P[(y + h) * PW + x + h] : U[i];
            const float v = a.img[poff + i] - U[i];
            a.y[poff + i] = v + prev;
        }
    } else {
        for (int i = tid; i < H * W; i += nt) {
            float v = U[i];
            if (a.quantise) v = cdf_quantise8(v);
            a.y[poff + i] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Separable variant.  The reference's kernels are Gaussians g (x) g (DEBLUR:363-389), so a step is a 1-D pass along
// x followed by a 1-D pass along y: 2k instead of k^2 FMAs per pixel (30 vs 225 at k = 15 -- the dense version is
// VALU-bound at ~80 us per step and 128x128 plane, and a q_sample at T = 200 runs up to 200 of them back to back).
// taps = [nsteps][C][2][k] (row 0: factor along y, row 1: along x), supplied by the host only when every kernel is
// rank one to fp32 rounding (|w - gy gx^T| <= 1e-7 max w); results differ from the dense conv by rounding only.
// LDS: U = state [H][W], T = row-pass output [H][W]; border handling by index mapping (no padded copy).
// The state BEFORE the last step (needed by snap / Alg. 2) is written to global memory just before that step.
// ------------------------------------------------------------------------------------------------
// border index for -n < i < 2n (k/2 < n is required by the caller): no integer division
__device__ __forceinline__ int cdf_pad_near(int i, int n, int mode) {
    if (mode == 0) return i < 0 ? i + n : (i >= n ? i - n : i);
    return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i);
}

template <int K>
__global__ void __launch_bounds__(1024) blur_plane_sep_kernel(BlurArgs a) {
    CDF_DYN_SMEM(smem);
    const int H = a.H, W = a.W, k = (K > 0 ? K : a.k), h = k / 2;
    float* U = (float*)smem;
    float* T = U + (size_t)H * W;
    float* red = T + (size_t)H * W;                 // 32 floats
    float* wt = red + 32;                           // [2 stages][2][64]: the step's 1-D factors (k <= 64), double buffered
    const int plane = blockIdx.x, b = plane / a.C, c = plane % a.C;
    const int nt = blockDim.x, tid = threadIdx.x;
    const size_t poff = (size_t)plane * H * W;
    const int hi = a.t ? (int)a.t[b] : a.step_hi;
    float* prev_out = a.snap ? a.snap : (a.img ? a.y : nullptr);      // Alg. 2 without snap: y doubles as scratch

    for (int i = tid; i < H * W; i += nt) U[i] = a.x[poff + i];
    // the factors of step s live in wt[(s & 1)]: fetched by the first 2k threads one step ahead (a uniform scalar read of
    // them inside the strip loops turns into per-strip global loads with a full round trip each)
    auto fetch_w = [&](int s) {
        if (tid < 2 * k && s <= hi) {
            const int r = tid >= k ? 1 : 0, e = tid - r * k;
            wt[((s & 1) * 2 + r) * 64 + e] = a.taps[(((size_t)s * a.C + c) * 2 + r) * k + e];
        }
    };
    fetch_w(a.step_lo);
    __syncthreads();
    const int strips_per_row = W / 4, nstrips = strips_per_row * H;
    constexpr int H4 = K > 0 ? ((K / 2 + 3) & ~3) : 0;               // aligned left reach of the fast row pass
    constexpr int NV = K > 0 ? (2 * H4 + 4) / 4 : 1;                 // float4s covering [x0 - H4, x0 + 4 + H4)
    // Round 4: 4 x 4-pixel register tiles for the compile-time kernel sizes (planes whose height is a multiple of 4).  With one
    // 4-pixel strip per thread the column pass read a float4 of T per tap (15 ds_read_b128 + 15 weight reads per 60 FMAs at k = 15)
    // and mapped its row index per tap; a 4 x 4 tile slides over k + 3 rows once (18 reads per 240 FMAs, the factors in registers),
    // and the row pass keeps the same 20-float window per row.  Every output still sums its taps in ascending order: bit-identical
    // to the strip form.  k = 15 at 128 x 128: 17.9 -> 12.2 us per step (profiles/round4_blur_tiles_ab.txt).
    const bool tiled = K > 0 && K <= 15 && (H & 3) == 0;     // (k = 27: the tile's window and factors do not fit 128 registers)
    const int tiles_per_row = W / 4, ntiles = tiles_per_row * (H / 4);
    for (int s = a.step_lo; s <= hi; ++s) {
        if (s == hi && prev_out)
            for (int i = tid; i < H * W; i += nt) prev_out[poff + i] = U[i];
        const float* gy = wt + (s & 1) * 128;
        const float* gx = gy + 64;
        if (tiled) {
            constexpr int KK = (K > 0 && K <= 15) ? K : 1;
            // (the factors are read from LDS where they are used -- broadcast reads; k of them in registers, duplicated into pairs for
            // the packed FMAs, cost 30 registers and spilled)
            // pass along x: T[y][x] = sum_kx gx[kx] U[y][map(x + kx - h)], four rows of a tile one after the other
            for (int tl = tid; tl < ntiles; tl += nt) {
                const int ty = tl / tiles_per_row, x0 = (tl - ty * tiles_per_row) * 4, y0 = ty * 4;
                const bool inner = x0 >= H4 && x0 + 4 + H4 <= W;
#pragma unroll 1
                for (int rr = 0; rr < 4; ++rr) {
                    const float* row = U + (y0 + rr) * W;
                    float acc[4] = {0.f, 0.f, 0.f, 0.f};
                    if (inner) {
                        float r[NV * 4];
#pragma unroll
                        for (int q = 0; q < NV; ++q) {
                            const float4 v = *(const float4*)(row + x0 - H4 + 4 * q);
                            r[4 * q + 0] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
                        }
#pragma unroll
                        for (int kx = 0; kx < KK; ++kx) {
                            const float wv = gx[kx];
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[j] = fmaf(wv, r[H4 - KK / 2 + kx + j], acc[j]);
                        }
                    } else {
                        for (int e = 0; e < k + 3; ++e) {        // the strip's k + 3 source pixels, each mapped once
                            const float v = row[cdf_pad_near(x0 + e - h, W, a.pad_mode)];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int kx = e - j;
                                if (kx >= 0 && kx < k) acc[j] = fmaf(gx[kx], v, acc[j]);
                            }
                        }
                    }
                    *(float4*)(T + (y0 + rr) * W + x0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                }
            }
            fetch_w(s + 1);                                  // lands before the barrier that ends this step
            __syncthreads();
            // pass along y: U[y][x] = sum_ky gy[ky] T[map(y + ky - h)][x]: the tile's k + 3 source rows, each read once; source row e
            // feeds output row j with factor gy[e - j] -- a window of four factors that slides by one per row
            for (int tl = tid; tl < ntiles; tl += nt) {
                const int ty = tl / tiles_per_row, x0 = (tl - ty * tiles_per_row) * 4, y0 = ty * 4;
                float4 acc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                float wj[4] = {0.f, 0.f, 0.f, 0.f};          // wj[j] = gy[e - j]
#pragma unroll
                for (int e = 0; e < KK + 3; ++e) {
                    wj[3] = wj[2]; wj[2] = wj[1]; wj[1] = wj[0];
                    wj[0] = e < KK ? gy[e] : 0.f;
                    const int sy = cdf_pad_near(y0 + e - h, H, a.pad_mode);
                    const float4 v = *(const float4*)(T + sy * W + x0);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (e - j >= 0 && e - j < KK) {
                            acc[j].x = fmaf(wj[j], v.x, acc[j].x); acc[j].y = fmaf(wj[j], v.y, acc[j].y);
                            acc[j].z = fmaf(wj[j], v.z, acc[j].z); acc[j].w = fmaf(wj[j], v.w, acc[j].w);
                        }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) *(float4*)(U + (y0 + j) * W + x0) = acc[j];
            }
            __syncthreads();
            if (s == a.collapse_step) {
                float part = 0.f;
                for (int i = tid; i < H * W; i += nt) part += U[i];
                const float mean = cdf_block_sum(part, red) / (float)(H * W);
                for (int i = tid; i < H * W; i += nt) U[i] = mean;
                __syncthreads();
            }
            continue;
        }
        // pass along x: T[y][x] = sum_kx gx[kx] U[y][map(x + kx - h)]
        for (int st = tid; st < nstrips; st += nt) {
            const int y = st / strips_per_row, x0 = (st - y * strips_per_row) * 4;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            const float* row = U + y * W;
            if (K > 0 && x0 >= H4 && x0 + 4 + H4 <= W) {
                float r[NV * 4];
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    const float4 v = *(const float4*)(row + x0 - H4 + 4 * q);
                    r[4 * q + 0] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
                }
#pragma unroll
                for (int kx = 0; kx < (K > 0 ? K : 1); ++kx) {
                    const float wv = gx[kx];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaf(wv, r[H4 - K / 2 + kx + j], acc[j]);
                }
            } else {
                for (int e = 0; e < k + 3; ++e) {            // the strip's k + 3 source pixels, each mapped once
                    const float v = row[cdf_pad_near(x0 + e - h, W, a.pad_mode)];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int kx = e - j;
                        if (kx >= 0 && kx < k) acc[j] = fmaf(gx[kx], v, acc[j]);
                    }
                }
            }
            *(float4*)(T + y * W + x0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        fetch_w(s + 1);                                      // lands before the barrier that ends this step
        __syncthreads();
        // pass along y: U[y][x] = sum_ky gy[ky] T[map(y + ky - h)][x]
        for (int st = tid; st < nstrips; st += nt) {
            const int y = st / strips_per_row, x0 = (st - y * strips_per_row) * 4;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 5
            for (int ky = 0; ky < k; ++ky) {
                const int sy = cdf_pad_near(y + ky - h, H, a.pad_mode);
                const float4 v = *(const float4*)(T + sy * W + x0);
                const float wv = gy[ky];
                acc.x = fmaf(wv, v.x, acc.x); acc.y = fmaf(wv, v.y, acc.y); acc.z = fmaf(wv, v.z, acc.z); acc.w = fmaf(wv, v.w, acc.w);
            }
            *(float4*)(U + y * W + x0) = acc;
        }
        __syncthreads();
        if (s == a.collapse_step) {
            float part = 0.f;
            for (int i = tid; i < H * W; i += nt) part += U[i];
            const float mean = cdf_block_sum(part, red) / (float)(H * W);
            for (int i = tid; i < H * W; i += nt) U[i] = mean;
            __syncthreads();
        }
    }
    const bool stepped = hi >= a.step_lo;
    if (!stepped && a.snap)
        for (int i = tid; i < H * W; i += nt) a.snap[poff + i] = U[i];
    if (a.img) {
        // Alg. 2: x = img - D(x0,t) + D(x0,t-1)   (DEBLUR:451); D(x0,t-1) was parked in prev_out by this same thread
        for (int i = tid; i < H * W; i += nt) {
            const float prev = stepped ? prev_out[poff + i] : U[i];
            const float v = a.img[poff + i] - U[i];
            a.y[poff + i] = v + prev;
        }
    } else {
        for (int i = tid; i < H * W; i += nt) {
            float v = U[i];
            if (a.quantise) v = cdf_quantise8(v);
            a.y[poff + i] = v;
        }
    }
}

// Generic single-step blur straight from global memory (any plane size / per-step kernel size).
__global__ void blur_step_global_kernel(const float* x, float* y, const float* taps /*[C][k][k]*/, int B, int C, int H,
                                         int W, int k, int pad_mode) {
    const long long n = (long long)B * C * H * W;
    const int h = k / 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int xw = (int)(i % W), yh = (int)((i / W) % H);
        const long long plane = i / ((long long)H * W);
        const int c = (int)(plane % C);
        const float* src = x + plane * H * W;
        const float* wt = taps + (size_t)c * k * k;
        float acc = 0.f;
        for (int ky = 0; ky < k; ++ky) {
            const int sy = cdf_pad_index(yh + ky - h, H, pad_mode);
            for (int kx = 0; kx < k; ++kx) {
                const int sx = cdf_pad_index(xw + kx - h, W, pad_mode);
                acc = fmaf(wt[ky * k + kx], src[sy * W + sx], acc);
            }
        }
        y[i] = acc;
    }
}

// plane mean collapse (discrete=True) for the global fallback: one block per plane
__global__ void plane_mean_kernel(float* x, int HW) {
    __shared__ float red[32];
    float* p = x + (size_t)blockIdx.x * HW;
    float part = 0.f;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) part += p[i];
    const float mean = cdf_block_sum(part, red) / (float)HW;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) p[i] = mean;
}

// ------------------------------------------------------------------------------------------------
// Gaussian-mask fade: x <- m_i * x for i = lo..hi(b)  (sequential products, same order as DEFADE:518)
// masks [T][MH][MW]; per-sample crop offset (off_y[b], off_x[b]) for the Random_* routines.
// ------------------------------------------------------------------------------------------------
struct MaskArgs {
    const float* x;
    float* y;
    float* snap;       // nullable: state after step hi-1
    const float* img;  // nullable: Alg. 2 combine
    const float* masks;
    const int64_t* t;
    const int64_t* off_y;  // nullable
    const int64_t* off_x;  // nullable
    int B, C, H, W, MH, MW;
    int step_lo, step_hi;
    int quantise;
};

__global__ void mask_apply_kernel(MaskArgs a) {
    const long long n = (long long)a.B * a.C * a.H * a.W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int xw = (int)(i % a.W), yh = (int)((i / a.W) % a.H);
        const int b = (int)(i / ((long long)a.C * a.H * a.W));
        const int hi = a.t ? (int)a.t[b] : a.step_hi;
        const int oy = a.off_y ? (int)a.off_y[b] : 0, ox = a.off_x ? (int)a.off_x[b] : 0;
        const float* m = a.masks + (size_t)(yh + oy) * a.MW + (xw + ox);
        const size_t ms = (size_t)a.MH * a.MW;
        float v = a.x[i], prev = v;
        for (int s = a.step_lo; s <= hi; ++s) {
            prev = v;
            v = m[s * ms] * v;
        }
        if (a.snap) a.snap[i] = prev;
        if (a.img) {
            const float d = a.img[i] - v;
            a.y[i] = d + prev;
        } else {
            a.y[i] = a.quantise ? cdf_quantise8(v) : v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Pixelation: x <- up_nearest_exact(down_mode(x, S_i)) for i = lo..hi(b), plane resident in LDS.
// mode 0: area (adaptive average), 1: bilinear, 2: bicubic (A=-0.75); align_corners=False,
// antialias=False — the index/weight formulas follow ATen's UpSample.h (area_pixel_compute_*,
// nearest_exact_idx, get_cubic_upsample_coefficients) evaluated in the same float/double mix.
// ------------------------------------------------------------------------------------------------
struct PixArgs {
    const float* x;
    float* y;
    float* snap;
    const float* img;
    const int* sizes;  // [nsteps] down-sampled edge length per step (device memory)
    const int64_t* t;
    int B, C, H;  // square planes (H == W), as the reference asserts
    int step_lo, step_hi;
    int mode;
};

__device__ __forceinline__ float cdf_cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cdf_cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

__global__ void __launch_bounds__(1024) pixelate_plane_kernel(PixArgs a) {
    CDF_DYN_SMEM(smem);
    const int H = a.H;
    float* U = (float*)smem;          // [H][H]
    float* D = U + (size_t)H * H;     // [S][S]
    int* tidx = (int*)(D + (size_t)H * H);  // [H][4]
    float* twt = (float*)(tidx + 4 * H);    // [H][4]
    int* upi = (int*)(twt + 4 * H);         // [H]
    const int plane = blockIdx.x, b = plane / a.C;
    const int nt = blockDim.x, tid = threadIdx.x;
    const size_t poff = (size_t)plane * H * H;
    const int hi = a.t ? (int)a.t[b] : a.step_hi;

    for (int i = tid; i < H * H; i += nt) U[i] = a.x[poff + i];
    __syncthreads();
    if (a.snap && hi - 1 < a.step_lo)
        for (int i = tid; i < H * H; i += nt) a.snap[poff + i] = U[i];

    for (int s = a.step_lo; s <= hi; ++s) {
        const int S = a.sizes[s];
        // ---- per-axis tables -------------------------------------------------------------------
        for (int o = tid; o < S; o += nt) {
            if (a.mode == 0) {
                // adaptive pooling window [start,end): start=floor(o*H/S), end=ceil((o+1)*H/S)
                const int st = (int)(((long long)o * H) / S);
                const int en = (int)((((long long)(o + 1)) * H + S - 1) / S);
                tidx[4 * o] = st;
                tidx[4 * o + 1] = en;
            } else {
                const float scale = (float)H / (float)S;
                float real = (float)((double)scale * ((double)o + 0.5) - 0.5);
                if (a.mode == 1 && real < 0.f) real = 0.f;
                int i0 = (int)floorf(real);
                if (i0 > H - 1) i0 = H - 1;
                float lam = real - (float)i0;
                lam = fminf(fmaxf(lam, 0.f), 1.f);
                if (a.mode == 1) {
                    tidx[4 * o] = i0;
                    tidx[4 * o + 1] = (i0 + 1 < H - 1) ? i0 + 1 : H - 1;
                    twt[4 * o] = 1.f - lam;
                    twt[4 * o + 1] = lam;
                } else {
                    const float A = -0.75f;
                    const float x2 = 1.f - lam;
                    twt[4 * o + 0] = cdf_cubic2(lam + 1.f, A);
                    twt[4 * o + 1] = cdf_cubic1(lam, A);
                    twt[4 * o + 2] = cdf_cubic1(x2, A);
                    twt[4 * o + 3] = cdf_cubic2(x2 + 1.f, A);
                    for (int j = 0; j < 4; ++j) {
                        int id = i0 + j - 1;
                        id = id < 0 ? 0 : (id > H - 1 ? H - 1 : id);
                        tidx[4 * o + j] = id;
                    }
                }
            }
        }
        for (int o = tid; o < H; o += nt) {
            const float scale = (float)S / (float)H;
            int id = (int)floorf((float)(((double)o + 0.5) * (double)scale));
            upi[o] = id < S - 1 ? id : S - 1;
        }
        __syncthreads();
        // ---- down ----------------------------------------------------------------------------
        for (int i = tid; i < S * S; i += nt) {
            const int oy = i / S, ox = i - oy * S;
            float v;
            if (a.mode == 0) {
                const int y0 = tidx[4 * oy], y1 = tidx[4 * oy + 1], x0 = tidx[4 * ox], x1 = tidx[4 * ox + 1];
                float sum = 0.f;
                for (int yy = y0; yy < y1; ++yy)
                    for (int xx = x0; xx < x1; ++xx) sum += U[yy * H + xx];
                v = sum / (float)(y1 - y0) / (float)(x1 - x0);
            } else if (a.mode == 1) {
                const int y0 = tidx[4 * oy], y1 = tidx[4 * oy + 1], x0 = tidx[4 * ox], x1 = tidx[4 * ox + 1];
                const float wy0 = twt[4 * oy], wy1 = twt[4 * oy + 1], wx0 = twt[4 * ox], wx1 = twt[4 * ox + 1];
                const float r0 = wx0 * U[y0 * H + x0] + wx1 * U[y0 * H + x1];
                const float r1 = wx0 * U[y1 * H + x0] + wx1 * U[y1 * H + x1];
                v = wy0 * r0 + wy1 * r1;
            } else {
                v = 0.f;
                for (int j = 0; j < 4; ++j) {
                    const int yy = tidx[4 * oy + j];
                    float r = 0.f;
                    for (int q = 0; q < 4; ++q) r += twt[4 * ox + q] * U[yy * H + tidx[4 * ox + q]];
                    v += twt[4 * oy + j] * r;
                }
            }
            D[i] = v;
        }
        __syncthreads();
        // ---- nearest-exact up -----------------------------------------------------------------
        for (int i = tid; i < H * H; i += nt) {
            const int yh = i / H, xw = i - yh * H;
            U[i] = D[upi[yh] * S + upi[xw]];
        }
        __syncthreads();
        if (a.snap && s == hi - 1)
            for (int i = tid; i < H * H; i += nt) a.snap[poff + i] = U[i];
    }
    if (a.img) {
        for (int i = tid; i < H * H; i += nt) {
            const float v = a.img[poff + i] - U[i];
            a.y[poff + i] = v + a.snap[poff + i];
        }
    } else {
        for (int i = tid; i < H * H; i += nt) a.y[poff + i] = U[i];
    }
}

// ------------------------------------------------------------------------------------------------
// elementwise: Alg.2 combine, Gaussian-noise q_sample, x2_bar, and the fused denoising Alg.2 step
// ------------------------------------------------------------------------------------------------
__global__ void combine_kernel(const float* img, const float* d_t, const float* d_tm1, float* out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = img[i] - d_t[i];
        out[i] = v + d_tm1[i];
    }
}

// x_t = ca[t[b]] * x0 + cb[t[b]] * eps      (DENOISE:517-522)
__global__ void noise_qsample_kernel(const float* x0, const float* eps, const float* ca, const float* cb,
                                     const int64_t* t, float* out, long long per_sample, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per_sample);
        const int64_t tt = t[b];
        const float p = ca[tt] * x0[i];
        const float q = cb[tt] * eps[i];
        out[i] = p + q;
    }
}

// One reverse step of the Gaussian-noise sampler (DENOISE:342-375 / :383-434):
//   x2 = est_noise ? (img - ca[t-1]*x1) / cb[t-1] : noise
//   xt_bar = ca[t-1]*x1 + cb[t-1]*x2 ; xt_sub1 = (t-1 != 0) ? ca[t-2]*x1 + cb[t-2]*x2 : x1
//   out = img - xt_bar + xt_sub1
__global__ void noise_step_kernel(const float* img, const float* x1, const float* noise, const float* ca,
                                  const float* cb, int t, int est_noise, float* out, long long n) {
    const float a1 = ca[t - 1], b1 = cb[t - 1];
    const float a2 = t - 1 != 0 ? ca[t - 2] : 0.f, b2 = t - 1 != 0 ? cb[t - 2] : 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float im = img[i], xa = x1[i];
        float x2;
        if (est_noise) {
            const float p = a1 * xa;
            x2 = (im - p) / b1;
        } else {
            x2 = noise[i];
        }
        const float p1 = a1 * xa, q1 = b1 * x2;
        const float xt_bar = p1 + q1;
        float xt_sub1 = xa;
        if (t - 1 != 0) {
            const float p2 = a2 * xa, q2 = b2 * x2;
            xt_sub1 = p2 + q2;
        }
        const float d = im - xt_bar;
        out[i] = d + xt_sub1;
    }
}

// Per-pixel blend of two images by mask tables [T][H*W] (the "defading generation" forward process,
// defading-generation-diffusion-pytorch/defading_diffusion_pytorch/defading_diffusion_pytorch.py:543-548):
//   x_t[b,c,p] = alphas[t[b]][p] * x1[b,c,p] + one_minus_alphas[t[b]][p] * x2[b,c,p]
__global__ void blend_qsample_kernel(const float* x1, const float* x2, const float* al, const float* om, const int64_t* t, float* out,
                                     int C, long long HW, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long p = i % HW;
        const int b = (int)(i / (HW * C));
        const long long k = (long long)t[b] * HW + p;
        const float u = al[k] * x1[i];
        const float v = om[k] * x2[i];
        out[i] = u + v;
    }
}

// One reverse step with a FIXED second image x2 (same file :386-419, :428-457):
//   xt_bar = al[t-1] x1 + om[t-1] x2 ; xt_sub1 = (t-1 != 0) ? al[t-2] x1 + om[t-2] x2 : x1 ; out = img - xt_bar + xt_sub1
__global__ void blend_step_kernel(const float* img, const float* x1, const float* x2, const float* al, const float* om, int t, float* out,
                                  long long HW, long long n) {
    const float* a1 = al + (long long)(t - 1) * HW;
    const float* o1 = om + (long long)(t - 1) * HW;
    const float* a2 = t - 1 != 0 ? al + (long long)(t - 2) * HW : a1;
    const float* o2 = t - 1 != 0 ? om + (long long)(t - 2) * HW : o1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long p = i % HW;
        const float xa = x1[i], xb = x2[i];
        const float u1 = a1[p] * xa, v1 = o1[p] * xb;
        const float xt_bar = u1 + v1;
        float xt_sub1 = xa;
        if (t - 1 != 0) {
            const float u2 = a2[p] * xa, v2 = o2[p] * xb;
            xt_sub1 = u2 + v2;
        }
        const float d = img[i] - xt_bar;
        out[i] = d + xt_sub1;
    }
}

// ------------------------------------------------------------------------------------------------
// losses: mean|x-y| (l1) / mean (x-y)^2 (l2).  Two-stage deterministic reduction.
// ------------------------------------------------------------------------------------------------
__global__ void loss_partial_kernel(const float* x, const float* y, float* partial, long long n, int l2) {
    __shared__ float red[32];
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = x[i] - y[i];
        acc += l2 ? d * d : fabsf(d);
    }
    const float s = cdf_block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ void loss_final_kernel(const float* partial, int np, float* out, float inv_n) {
    __shared__ float red[32];
    float acc = 0.f;
    for (int i = threadIdx.x; i < np; i += blockDim.x) acc += partial[i];
    const float s = cdf_block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = s * inv_n;
}
// d loss / d y  (y = prediction): l1: -sign(x-y)/n * g ; l2: -2(x-y)/n * g ; g read from device scalar
__global__ void loss_bwd_kernel(const float* x, const float* y, const float* gout, float* gy, long long n, int l2) {
    const float g = gout[0] / (float)n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = x[i] - y[i];
        float s;
        if (l2) s = -2.f * d;
        else s = d > 0.f ? -1.f : (d < 0.f ? 1.f : 0.f);
        gy[i] = s * g;
    }
}

// ------------------------------------------------------------------------------------------------
// layout shuffles between the public NCHW tensors and the engine's NHWC activations
// ------------------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc_kernel(const float* x, float* y, int B, int C, int HW, int ldy) {
    const long long n = (long long)B * C * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long p = i / C;  // b*HW + pix
        const int b = (int)(p / HW), pix = (int)(p % HW);
        y[p * ldy + c] = x[((long long)b * C + c) * HW + pix];
    }
}
__global__ void nhwc_to_nchw_kernel(const float* x, float* y, const float* add /*nullable NCHW*/, int B, int C, int HW,
                                    int ldx) {
    const long long n = (long long)B * C * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(i % HW);
        const long long bc = i / HW;
        const int c = (int)(bc % C), b = (int)(bc / C);
        float v = x[((long long)b * HW + pix) * ldx + c];
        if (add) v += add[i];
        y[i] = v;
    }
}

// ================================================================================================
// C ABI
// ================================================================================================
static inline int ew_grid(long long n, int block) {
    long long g = (n + block - 1) / block;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

template <int K, int SW>
static int launch_blur_plane(const BlurArgs& a, int nt, size_t lds, hipStream_t s) {
    CDF_LAUNCH((blur_plane_lds_kernel<K, SW>), dim3(a.B * a.C), dim3(nt), lds, s, a);
    return cdf_check_launch("blur_plane_lds");
}

extern "C" size_t cdf_blur_lds_bytes(int H, int W, int k) {
    const int h = k / 2;
    const size_t PW = (size_t)((W + 2 * h + 3) & ~3), PH = (size_t)H + 2 * h;
    return (PW * PH + (size_t)H * W + 32) * sizeof(float);
}

extern "C" int cdf_blur_chain(const float* x, float* y, float* snap, const float* img, const float* taps,
                              const int64_t* t, int B, int C, int H, int W, int k, int step_lo, int step_hi,
                              int pad_mode, int collapse_step, int quantise, void* stream) {
    CDF_REQUIRE(x && y && taps, "cdf_blur_chain: null pointer");
    CDF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && k > 0 && (k & 1), "cdf_blur_chain: bad shape B=%d C=%d H=%d W=%d k=%d", B, C, H, W, k);
    CDF_REQUIRE(pad_mode == 0 || pad_mode == 1, "cdf_blur_chain: pad_mode must be 0 (circular) or 1 (reflect)");
    CDF_REQUIRE(pad_mode == 0 || (k / 2 < H && k / 2 < W), "cdf_blur_chain: reflect padding needs k/2 < H,W");
    const size_t lds = cdf_blur_lds_bytes(H, W, k);
    CDF_REQUIRE(lds <= 160 * 1024 && (W % 4) == 0, "cdf_blur_chain: plane %dx%d k=%d does not fit the LDS-resident kernel (use cdf_blur_step)", H, W, k);
    BlurArgs a{x, y, snap, img, taps, t, B, C, H, W, k, step_lo, step_hi, pad_mode, collapse_step, quantise};
    const bool sw8 = (W % 8) == 0;
    const int nstrips = H * W / (sw8 ? 8 : 4);
    int nt = ((nstrips + 63) / 64) * 64;
    if (nt > 1024) nt = 1024;
    if (nt < 64) nt = 64;
#ifndef CDF_EMU
    static CdfDeviceLatch attr_done;
    if (attr_done.first()) {
        // allow > 64 KB dynamic LDS for every instantiation
#define CDF_SET_LDS(K, SW) (void)hipFuncSetAttribute((const void*)blur_plane_lds_kernel<K, SW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
        CDF_SET_LDS(3, 8); CDF_SET_LDS(11, 8); CDF_SET_LDS(15, 8); CDF_SET_LDS(27, 8); CDF_SET_LDS(0, 8);
        CDF_SET_LDS(3, 4); CDF_SET_LDS(11, 4); CDF_SET_LDS(15, 4); CDF_SET_LDS(27, 4); CDF_SET_LDS(0, 4);
#undef CDF_SET_LDS
    }
#endif
#define CDF_BLUR_CASE(K)                                                      \
    case K:                                                                   \
        return sw8 ? launch_blur_plane<K, 8>(a, nt, lds, CDF_S) : launch_blur_plane<K, 4>(a, nt, lds, CDF_S);
    switch (k) {
        CDF_BLUR_CASE(3)
        CDF_BLUR_CASE(11)
        CDF_BLUR_CASE(15)
        CDF_BLUR_CASE(27)
        default:
            return sw8 ? launch_blur_plane<0, 8>(a, nt, lds, CDF_S) : launch_blur_plane<0, 4>(a, nt, lds, CDF_S);
    }
#undef CDF_BLUR_CASE
}

template <int K>
static int launch_blur_sep(const BlurArgs& a, int nt, size_t lds, hipStream_t s) {
#ifndef CDF_EMU
    static CdfDeviceLatch attr_done;
    if (attr_done.first()) {
        (void)hipFuncSetAttribute((const void*)blur_plane_sep_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
#endif
    CDF_LAUNCH((blur_plane_sep_kernel<K>), dim3(a.B * a.C), dim3(nt), lds, s, a);
    return cdf_check_launch("blur_plane_sep");
}

extern "C" size_t cdf_blur_sep_lds_bytes(int H, int W) { return ((size_t)2 * H * W + 32 + 256) * sizeof(float); }

// taps1d = [nsteps][C][2][k]: per step and channel the factor along y, then the factor along x
extern "C" int cdf_blur_chain_sep(const float* x, float* y, float* snap, const float* img, const float* taps1d,
                                  const int64_t* t, int B, int C, int H, int W, int k, int step_lo, int step_hi,
                                  int pad_mode, int collapse_step, int quantise, void* stream) {
    CDF_REQUIRE(x && y && taps1d, "cdf_blur_chain_sep: null pointer");
    CDF_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && k > 0 && (k & 1), "cdf_blur_chain_sep: bad shape B=%d C=%d H=%d W=%d k=%d", B, C, H, W, k);
    CDF_REQUIRE(pad_mode == 0 || pad_mode == 1, "cdf_blur_chain_sep: pad_mode must be 0 (circular) or 1 (reflect)");
    CDF_REQUIRE(k / 2 < H && k / 2 < W && k <= 64, "cdf_blur_chain_sep: needs k/2 < H,W and k <= 64");
    CDF_REQUIRE(!img || (x != y && img != y), "cdf_blur_chain_sep: y must not alias x / img in the Alg. 2 form");
    const size_t lds = cdf_blur_sep_lds_bytes(H, W);
    CDF_REQUIRE(lds <= 160 * 1024 && (W % 4) == 0, "cdf_blur_chain_sep: plane %dx%d does not fit the LDS-resident kernel", H, W);
    BlurArgs a{x, y, snap, img, taps1d, t, B, C, H, W, k, step_lo, step_hi, pad_mode, collapse_step, quantise};
    int nt = ((H * W / 4 + 63) / 64) * 64;
    if (nt > 1024) nt = 1024;
    if (nt < 64) nt = 64;
    switch (k) {
        case 3: return launch_blur_sep<3>(a, nt, lds, CDF_S);
        case 11: return launch_blur_sep<11>(a, nt, lds, CDF_S);
        case 15: return launch_blur_sep<15>(a, nt, lds, CDF_S);
        case 27: return launch_blur_sep<27>(a, nt, lds, CDF_S);
        default: return launch_blur_sep<0>(a, nt, lds, CDF_S);
    }
}

extern "C" int cdf_blur_step(const float* x, float* y, const float* taps, int B, int C, int H, int W, int k,
                             int pad_mode, void* stream) {
    CDF_REQUIRE(x && y && taps && x != y, "cdf_blur_step: null / aliased pointer");
    CDF_REQUIRE(k > 0 && (k & 1), "cdf_blur_step: k must be odd");
    const long long n = (long long)B * C * H * W;
    CDF_LAUNCH(blur_step_global_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, CDF_S, x, y, taps, B, C, H, W, k, pad_mode);
    return cdf_check_launch("blur_step_global");
}

extern "C" int cdf_plane_mean(float* x, int planes, int HW, void* stream) {
    CDF_REQUIRE(x && planes > 0 && HW > 0, "cdf_plane_mean: bad args");
    CDF_LAUNCH(plane_mean_kernel, dim3(planes), dim3(256), 0, CDF_S, x, HW);
    return cdf_check_launch("plane_mean");
}

extern "C" int cdf_mask_chain(const float* x, float* y, float* snap, const float* img, const float* masks,
                              const int64_t* t, const int64_t* off_y, const int64_t* off_x, int B, int C, int H, int W,
                              int MH, int MW, int step_lo, int step_hi, int quantise, void* stream) {
    CDF_REQUIRE(x && y && masks, "cdf_mask_chain: null pointer");
    CDF_REQUIRE(MH >= H && MW >= W, "cdf_mask_chain: mask table smaller than the image");
    CDF_REQUIRE(!img || snap, "cdf_mask_chain: Alg.2 combine needs a snap buffer");
    MaskArgs a{x, y, snap, img, masks, t, off_y, off_x, B, C, H, W, MH, MW, step_lo, step_hi, quantise};
    const long long n = (long long)B * C * H * W;
    CDF_LAUNCH(mask_apply_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, CDF_S, a);
    return cdf_check_launch("mask_apply");
}

extern "C" int cdf_pixelate_chain(const float* x, float* y, float* snap, const float* img, const int* sizes,
                                  const int64_t* t, int B, int C, int H, int step_lo, int step_hi, int mode,
                                  void* stream) {
    CDF_REQUIRE(x && y && sizes, "cdf_pixelate_chain: null pointer");
    CDF_REQUIRE(mode >= 0 && mode <= 2, "cdf_pixelate_chain: mode must be 0 area, 1 bilinear, 2 bicubic");
    CDF_REQUIRE(!img || snap, "cdf_pixelate_chain: Alg.2 combine needs a snap buffer");
    const size_t lds = ((size_t)2 * H * H + 9 * (size_t)H) * 4;
    CDF_REQUIRE(lds <= 160 * 1024, "cdf_pixelate_chain: %dx%d plane does not fit LDS", H, H);
    PixArgs a{x, y, snap, img, sizes, t, B, C, H, step_lo, step_hi, mode};
    int nt = H * H >= 4096 ? 1024 : 256;
#ifndef CDF_EMU
    static CdfDeviceLatch attr_done;
    if (attr_done.first()) {
        (void)hipFuncSetAttribute((const void*)pixelate_plane_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
#endif
    CDF_LAUNCH(pixelate_plane_kernel, dim3(B * C), dim3(nt), lds, CDF_S, a);
    return cdf_check_launch("pixelate_plane");
}

extern "C" int cdf_x0_step_down(const float* img, const float* d_t, const float* d_tm1, float* out, long long n,
                                void* stream) {
    CDF_REQUIRE(img && d_t && d_tm1 && out && n > 0, "cdf_x0_step_down: bad args");
    CDF_LAUNCH(combine_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, CDF_S, img, d_t, d_tm1, out, n);
    return cdf_check_launch("combine");
}

extern "C" int cdf_noise_qsample(const float* x0, const float* eps, const float* ca, const float* cb, const int64_t* t,
                                 float* out, int B, long long per_sample, void* stream) {
    CDF_REQUIRE(x0 && eps && ca && cb && t && out, "cdf_noise_qsample: null pointer");
    const long long n = (long long)B * per_sample;
    CDF_LAUNCH(noise_qsample_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, CDF_S, x0, eps, ca, cb, t, out, per_sample, n);
    return cdf_check_launch("noise_qsample");
}

extern "C" int cdf_noise_step(const float* img, const float* x1, const float* noise, const float* ca, const float* cb,
                              int t, int est_noise, float* out, long long n, void* stream) {
    CDF_REQUIRE(img && x1 && ca && cb && out && t >= 1, "cdf_noise_step: bad args");
    CDF_REQUIRE(est_noise || noise, "cdf_noise_step: fixed-noise mode needs the noise tensor");
    CDF_LAUNCH(noise_step_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, CDF_S, img, x1, noise, ca, cb, t, est_noise, out, n);
    return cdf_check_launch("noise_step");
}

extern "C" int cdf_blend_qsample(const float* x1, const float* x2, const float* alphas, const float* one_minus, const int64_t* t,
                                 float* out, int B, int C, long long HW, void* stream) {
    CDF_REQUIRE(x1 && x2 && alphas && one_minus && t && out && B > 0 && C > 0 && HW > 0, "cdf_blend_qsample: bad args");
    const long long n = (long long)B * C * HW;
    CDF_LAUNCH(blend_qsample_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, CDF_S, x1, x2, alphas, one_minus, t, out, C, HW, n);
    return cdf_check_launch("blend_qsample");
}

extern "C" int cdf_blend_step(const float* img, const float* x1, const float* x2, const float* alphas, const float* one_minus, int t,
                              float* out, long long HW, long long n, void* stream) {
    CDF_REQUIRE(img && x1 && x2 && alphas && one_minus && out && t >= 1 && HW > 0 && n > 0, "cdf_blend_step: bad args");
    CDF_LAUNCH(blend_step_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, CDF_S, img, x1, x2, alphas, one_minus, t, out, HW, n);
    return cdf_check_launch("blend_step");
}

extern "C" int cdf_loss_fwd(const float* x, const float* y, float* out, float* partial /*>=1024 floats*/, long long n,
                            int l2, void* stream) {
    CDF_REQUIRE(x && y && out && partial && n > 0, "cdf_loss_fwd: bad args");
    int np = ew_grid(n, 256);
    if (np > 1024) np = 1024;
    CDF_LAUNCH(loss_partial_kernel, dim3(np), dim3(256), 0, CDF_S, x, y, partial, n, l2);
    CDF_LAUNCH(loss_final_kernel, dim3(1), dim3(256), 0, CDF_S, (const float*)partial, np, out, 1.0f / (float)n);
    return cdf_check_launch("loss_fwd");
}

extern "C" int cdf_loss_bwd(const float* x, const float* y, const float* gout, float* gy, long long n, int l2,
                            void* stream) {
    CDF_REQUIRE(x && y && gout && gy && n > 0, "cdf_loss_bwd: bad args");
    CDF_LAUNCH(loss_bwd_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, CDF_S, x, y, gout, gy, n, l2);
    return cdf_check_launch("loss_bwd");
}

extern "C" int cdf_nchw_to_nhwc(const float* x, float* y, int B, int C, int HW, int ldy, void* stream) {
    CDF_REQUIRE(x && y && ldy >= C, "cdf_nchw_to_nhwc: bad args");
    const long long n = (long long)B * C * HW;
    CDF_LAUNCH(nchw_to_nhwc_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, CDF_S, x, y, B, C, HW, ldy);
    return cdf_check_launch("nchw_to_nhwc");
}

extern "C" int cdf_nhwc_to_nchw(const float* x, float* y, const float* add, int B, int C, int HW, int ldx,
                                void* stream) {
    CDF_REQUIRE(x && y && ldx >= C, "cdf_nhwc_to_nchw: bad args");
    const long long n = (long long)B * C * HW;
    CDF_LAUNCH(nhwc_to_nchw_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, CDF_S, x, y, add, B, C, HW, ldx);
    return cdf_check_launch("nhwc_to_nchw");
}

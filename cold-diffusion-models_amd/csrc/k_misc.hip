// k_misc.hip — small fused kernels around the UNets and the optimizer tail.
//   sinusoidal time embedding : deblurring_diffusion_pytorch.py:91-103, Model2.py:6-24
//   GELU / SiLU on [B,K] vectors (time MLP)          : DEBLUR:140-143,211-216 ; MODEL2:27-29,120,293-296
//   nearest x2 upsample / its adjoint, dropout       : MODEL2:36-50,94,124
//   Adam (torch.optim.Adam defaults) and EMA         : DEBLUR:59-81,1117,1200-1204
// The optimizer kernels run over ONE flat fp32 arena holding every UNet parameter (and a second
// one for the gradients), so an optimizer step is a single launch at HBM speed.
#include "cdf_common.h"
#include "colddiff.h"

static inline int ew_grid(long long n) {
    long long g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    return g < 1 ? 1 : (int)g;
}
#define CDF_EW_LOOP(i, n) \
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

// emb[b][j] = sin(t[b] * f_j), emb[b][half + j] = cos(t[b] * f_j); the frequency table
// f_j = exp(-j * ln(10000)/(half-1)) is an init-time constant supplied by the caller (a 1e-7 relative
// difference in f_j is amplified by t ~ 1000 into a 1e-4 phase error, so it is taken as data).
__global__ void sinusoidal_kernel(const int64_t* t, const float* freq, float* out, int ldo, int B, int dim) {
    const int half = dim / 2;
    CDF_EW_LOOP(i, (long long)B * half) {
        const int b = (int)(i / half), j = (int)(i % half);
        const float a = (float)t[b] * freq[j];
        out[(long long)b * ldo + j] = sinf(a);
        out[(long long)b * ldo + half + j] = cosf(a);
    }
}

// y = act(x) / dx = dy * act'(x) on row-major [rows, C] with pitches; act 1 GELU, 2 SiLU, 3 ReLU (forward only: the FID extractor)
__global__ void act_fwd_kernel(const float* x, int ldx, float* y, int ldy, long long rows, int C, int act) {
    CDF_EW_LOOP(i, rows * C) {
        const long long r = i / C;
        const int c = (int)(i % C);
        const float v = x[r * ldx + c];
        y[r * ldy + c] = act == 1 ? cdf_gelu(v) : (act == 2 ? cdf_silu(v) : fmaxf(v, 0.0f));
    }
}
__global__ void act_bwd_kernel(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, long long rows,
                               int C, int act, int accumulate) {
    CDF_EW_LOOP(i, rows * C) {
        const long long r = i / C;
        const int c = (int)(i % C);
        const float v = x[r * ldx + c];
        float g = dy[r * lddy + c] * (act == 1 ? cdf_gelu_grad(v) : cdf_silu_grad(v));
        float* dst = dx + r * lddx + c;
        *dst = accumulate ? *dst + g : g;
    }
}

// dst[r][c] = alpha*dst + beta*src over [rows, C] with pitches (skip-connection gradient sums, copies)
__global__ void axpby_kernel(float* dst, int ldd, const float* src, int lds, long long rows, int C, float alpha, float beta) {
    CDF_EW_LOOP(i, rows * C) {
        const long long r = i / C;
        const int c = (int)(i % C);
        float* d = dst + r * ldd + c;
        const float s = beta * src[r * lds + c];
        *d = alpha == 0.f ? s : alpha * *d + s;
    }
}

// nearest x2 upsample NHWC: y[b, 2h+i, 2w+j, c] = x[b,h,w,c]
__global__ void upsample2_kernel(const float* x, int ldx, float* y, int ldy, int B, int H, int W, int C4) {
    CDF_EW_LOOP(i, (long long)B * 2 * H * 2 * W * C4) {
        const int c = (int)(i % C4) * 4;
        long long r = i / C4;
        const int ox = (int)(r % (2 * W));
        r /= 2 * W;
        const int oy = (int)(r % (2 * H)), b = (int)(r / (2 * H));
        *(float4*)(y + (((long long)b * 2 * H + oy) * 2 * W + ox) * ldy + c) =
            *(const float4*)(x + (((long long)b * H + (oy >> 1)) * W + (ox >> 1)) * ldx + c);
    }
}
// adjoint: dx[b,h,w,c] (+)= sum of the 2x2 block of dy
__global__ void upsample2_bwd_kernel(const float* dy, int lddy, float* dx, int lddx, int B, int H, int W, int C4, int accumulate) {
    CDF_EW_LOOP(i, (long long)B * H * W * C4) {
        const int c = (int)(i % C4) * 4;
        long long r = i / C4;
        const int w = (int)(r % W);
        r /= W;
        const int h = (int)(r % H), b = (int)(r / H);
        const float* p = dy + (((long long)b * 2 * H + 2 * h) * 2 * W + 2 * w) * lddy + c;
        const float4 a = *(const float4*)p, b2 = *(const float4*)(p + lddy);
        const float4 c2 = *(const float4*)(p + (long long)2 * W * lddy), d2 = *(const float4*)(p + (long long)2 * W * lddy + lddy);
        float4 o = make_float4((a.x + b2.x) + (c2.x + d2.x), (a.y + b2.y) + (c2.y + d2.y), (a.z + b2.z) + (c2.z + d2.z), (a.w + b2.w) + (c2.w + d2.w));
        float* dst = dx + (((long long)b * H + h) * W + w) * lddx + c;
        if (accumulate) {
            const float4 old = *(const float4*)dst;
            o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
        }
        *(float4*)dst = o;
    }
}

// counter-based dropout mask (cdf_hash32, cdf_common.h): keep iff hash(seed, index) >= p*2^32 ; y = x * keep / (1-p)
__global__ void dropout_kernel(const float* x, int ldx, float* y, int ldy, long long rows, int C, float p,
                               unsigned long long seed) {
    const unsigned thr = (unsigned)((double)p * 4294967296.0);
    const float inv = 1.0f / (1.0f - p);
    CDF_EW_LOOP(i, rows * C) {
        const long long r = i / C;
        const int c = (int)(i % C);
        const bool keep = cdf_hash32(seed, (unsigned long long)i) >= thr;
        y[r * ldy + c] = keep ? x[r * ldx + c] * inv : 0.f;
    }
}

// torch.optim.Adam (amsgrad=False, weight_decay=0, maximize=False), single-tensor CPU op order:
//   m = m + (g - m)*(1-b1) ; v = v*b2 + ((1-b2)*g)*g ; denom = sqrt(v)/bc2_sqrt + eps ;
//   p = p + (-step_size * m) / denom             with step_size = lr / bc1
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long long n, float one_minus_b1, float b2,
                            float one_minus_b2, float neg_step_size, float bc2_sqrt, float eps) {
    CDF_EW_LOOP(i, n) {
        const float gi = g[i];
        float mi = m[i], vi = v[i];
        mi = mi + (gi - mi) * one_minus_b1;
        vi = vi * b2;
        vi = vi + (one_minus_b2 * gi) * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] + (neg_step_size * mi) / denom;
        m[i] = mi;
        v[i] = vi;
    }
}
// EMA: ma = ma*beta + (1-beta)*p   (DEBLUR:78-81)
__global__ void ema_kernel(float* ma, const float* p, long long n, float beta, float one_minus_beta) {
    CDF_EW_LOOP(i, n) {
        const float a = ma[i] * beta;
        const float b = one_minus_beta * p[i];
        ma[i] = a + b;
    }
}
__global__ void scale_kernel(float* x, long long n, float s) {
    CDF_EW_LOOP(i, n) x[i] = x[i] * s;
}

// ================================================================================================
// ------------------------------------------------------------------------------------------------
// Skinny linear layers (time-embedding MLPs: DEBLUR:96-103,142-144,160; MODEL2:44-48,238-245): M = batch rows only.
// On the 128-row GEMM tiles these are one block walking K in 16-element steps behind a barrier each (~70 us for a
// 32 x 512 x 64 product); here the rows are register accumulators and the contraction is split over 4 waves.
//   out[m][j] = bias[j] + sum_i in[m][i] * Wm[i * ldw + j]
// forward: i = k, j = n, Wm = the [K][N] packed weight;  data gradient: i = n, j = k, Wm = the PyTorch [N][K] weight.
// grid = (ceil(J/64), ceil(M/32)); block 256 = 4 contraction slices x 64 output columns.
// ------------------------------------------------------------------------------------------------
template <int MB>                                            // rows per block: 32, or 8 when that leaves most CUs without a block
__global__ void __launch_bounds__(256) linear_small_kernel(const float* in, int ldi, const float* Wm, int ldw, const float* bias,
                                                           float* out, int ldo, int M, int I, int J) {
    constexpr int IC = 128;                                  // contraction chunk staged in LDS
    __shared__ __attribute__((aligned(16))) float xs[MB][IC];
    __shared__ float red[4][MB][64];
    const int tid = threadIdx.x, lane = tid & 63, ks = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = blockIdx.x * 64 + lane, m0 = blockIdx.y * MB;
    const int jc = j < J ? j : J - 1;
    float acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = 0.f;
    for (int c0 = 0; c0 < I; c0 += IC) {
        // stage in[m0 .. m0+31][c0 .. c0+127] (rows / columns past the end as zeros): element-wise, any pitch
        // (all 16 loads of a thread in flight: unconditional, clamped address + select)
        float stage[MB * IC / 256];
#pragma unroll
        for (int r = 0; r < MB * IC / 256; ++r) {
            const int e = tid + 256 * r, m = e / IC, i = e - m * IC;
            const bool ok = m0 + m < M && c0 + i < I;
            const float v = in[(long long)(ok ? m0 + m : 0) * ldi + (ok ? c0 + i : 0)];
            stage[r] = ok ? v : 0.f;
        }
        // wave ks multiplies its quarter of the chunk: its 32 weight rows are fetched up front, in flight together with the
        // staging loads (one round trip per chunk); the activations are then LDS broadcasts
        float w[IC / 4];
#pragma unroll
        for (int u = 0; u < IC / 4; ++u) {
            const int i = c0 + ks * (IC / 4) + u;
            w[u] = Wm[(long long)(i < I ? i : I - 1) * ldw + jc];       // (xs is zero past I)
        }
#pragma unroll
        for (int r = 0; r < MB * IC / 256; ++r) {
            const int e = tid + 256 * r, m = e / IC, i = e - m * IC;
            xs[m][i] = stage[r];
        }
        __syncthreads();
#pragma unroll
        for (int i4 = 0; i4 < IC / 4; i4 += 4) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const float4 xv = *(const float4*)&xs[m][ks * (IC / 4) + i4];
                acc[m] = fmaf(xv.x, w[i4 + 0], acc[m]);
                acc[m] = fmaf(xv.y, w[i4 + 1], acc[m]);
                acc[m] = fmaf(xv.z, w[i4 + 2], acc[m]);
                acc[m] = fmaf(xv.w, w[i4 + 3], acc[m]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < MB; ++m) red[ks][m][lane] = acc[m];
    __syncthreads();
    const float b = (bias && j < J) ? bias[j] : 0.f;
    for (int m = ks; m < MB; m += 4) {
        if (m0 + m >= M || j >= ldo) continue;
        const float v = (red[0][m][lane] + red[1][m][lane]) + (red[2][m][lane] + red[3][m][lane]) + b;
        out[(long long)(m0 + m) * ldo + j] = j < J ? v : 0.f;     // columns J..ldo-1 are pitch padding
    }
}

// dW[n][k] += sum_m dy[m][n] x[m][k];  db[n] += sum_m dy[m][n].  grid = (ceil(K/64), ceil(N/4)); wave = one n, lanes = k.
__global__ void __launch_bounds__(256) linear_small_wgrad_kernel(const float* dy, int ldd, const float* x, int ldx, float* dW, float* db,
                                                                 int M, int N, int K) {
    const int lane = threadIdx.x & 63, n = blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (n >= N) return;
    const int k = blockIdx.x * 64 + lane, kc = k < K ? k : K - 1;
    float acc = 0.f, bs = 0.f;
    int m = 0;
    for (; m + 8 <= M; m += 8) {                             // 8 row pairs in flight
        float d[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            d[u] = dy[(long long)(m + u) * ldd + n];         // wave-uniform
            xv[u] = x[(long long)(m + u) * ldx + kc];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = fmaf(d[u], xv[u], acc);
            bs += d[u];
        }
    }
    for (; m < M; ++m) {
        const float d = dy[(long long)m * ldd + n];
        acc = fmaf(d, x[(long long)m * ldx + kc], acc);
        bs += d;
    }
    if (k < K) dW[(long long)n * K + k] += acc;
    if (db && blockIdx.x == 0 && lane == 0) db[n] += bs;
}

extern "C" int cdf_sinusoidal(const int64_t* t, const float* freq, float* out, int ldo, int B, int dim, void* stream) {
    CDF_REQUIRE(t && freq && out && B > 0 && dim >= 4 && (dim & 1) == 0 && ldo >= dim, "cdf_sinusoidal: bad args");
    CDF_LAUNCH(sinusoidal_kernel, dim3(ew_grid((long long)B * dim / 2)), dim3(256), 0, CDF_S, t, freq, out, ldo, B, dim);
    return cdf_check_launch("sinusoidal");
}
extern "C" int cdf_linear_small(const float* in, int ldi, const float* Wm, int ldw, const float* bias, float* out, int ldo, int M,
                                int I, int J, void* stream) {
    CDF_REQUIRE(in && Wm && out && M > 0 && I > 0 && J > 0 && ldi >= I && ldw >= J && ldo >= J, "cdf_linear_small: bad args");
    // the layers are latency bound (a few thousand FMAs per thread): with few 32-row blocks, 8-row blocks cut the serial part 4x
    if ((long long)cdf_cdiv(ldo, 64) * cdf_cdiv(M, 32) < 128)
        CDF_LAUNCH(linear_small_kernel<8>, dim3(cdf_cdiv(ldo, 64), cdf_cdiv(M, 8)), dim3(256), 0, CDF_S, in, ldi, Wm, ldw, bias, out, ldo, M, I, J);
    else
        CDF_LAUNCH(linear_small_kernel<32>, dim3(cdf_cdiv(ldo, 64), cdf_cdiv(M, 32)), dim3(256), 0, CDF_S, in, ldi, Wm, ldw, bias, out, ldo, M, I, J);
    return cdf_check_launch("linear_small");
}

extern "C" int cdf_linear_small_wgrad(const float* dy, int ldd, const float* x, int ldx, float* dW, float* db, int M, int N, int K,
                                      void* stream) {
    CDF_REQUIRE(dy && x && dW && M > 0 && N > 0 && K > 0 && ldd >= N && ldx >= K, "cdf_linear_small_wgrad: bad args");
    CDF_LAUNCH(linear_small_wgrad_kernel, dim3(cdf_cdiv(K, 64), cdf_cdiv(N, 4)), dim3(256), 0, CDF_S, dy, ldd, x, ldx, dW, db, M, N, K);
    return cdf_check_launch("linear_small_wgrad");
}

extern "C" int cdf_act_fwd(const float* x, int ldx, float* y, int ldy, long long rows, int C, int act, void* stream) {
    CDF_REQUIRE(x && y && rows > 0 && C > 0 && act >= 1 && act <= 3, "cdf_act_fwd: bad args");
    CDF_LAUNCH(act_fwd_kernel, dim3(ew_grid(rows * C)), dim3(256), 0, CDF_S, x, ldx, y, ldy, rows, C, act);
    return cdf_check_launch("act_fwd");
}
extern "C" int cdf_act_bwd(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, long long rows, int C,
                           int act, int accumulate, void* stream) {
    CDF_REQUIRE(x && dy && dx && rows > 0 && C > 0 && (act == 1 || act == 2), "cdf_act_bwd: bad args");
    CDF_LAUNCH(act_bwd_kernel, dim3(ew_grid(rows * C)), dim3(256), 0, CDF_S, x, ldx, dy, lddy, dx, lddx, rows, C, act, accumulate);
    return cdf_check_launch("act_bwd");
}
extern "C" int cdf_axpby(float* dst, int ldd, const float* src, int lds, long long rows, int C, float alpha, float beta,
                         void* stream) {
    CDF_REQUIRE(dst && src && rows > 0 && C > 0, "cdf_axpby: bad args");
    CDF_LAUNCH(axpby_kernel, dim3(ew_grid(rows * C)), dim3(256), 0, CDF_S, dst, ldd, src, lds, rows, C, alpha, beta);
    return cdf_check_launch("axpby");
}
extern "C" int cdf_upsample2(const float* x, int ldx, float* y, int ldy, int B, int H, int W, int C, void* stream) {
    CDF_REQUIRE(x && y && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "cdf_upsample2: bad args");
    CDF_LAUNCH(upsample2_kernel, dim3(ew_grid((long long)B * 4 * H * W * (C / 4))), dim3(256), 0, CDF_S, x, ldx, y, ldy, B, H, W, C / 4);
    return cdf_check_launch("upsample2");
}
extern "C" int cdf_upsample2_bwd(const float* dy, int lddy, float* dx, int lddx, int B, int H, int W, int C,
                                 int accumulate, void* stream) {
    CDF_REQUIRE(dy && dx && C % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0, "cdf_upsample2_bwd: bad args");
    CDF_LAUNCH(upsample2_bwd_kernel, dim3(ew_grid((long long)B * H * W * (C / 4))), dim3(256), 0, CDF_S, dy, lddy, dx, lddx, B, H, W, C / 4, accumulate);
    return cdf_check_launch("upsample2_bwd");
}
extern "C" int cdf_dropout(const float* x, int ldx, float* y, int ldy, long long rows, int C, float p,
                           long long seed, void* stream) {
    CDF_REQUIRE(x && y && rows > 0 && C > 0 && p >= 0.f && p < 1.f, "cdf_dropout: bad args");
    CDF_LAUNCH(dropout_kernel, dim3(ew_grid(rows * C)), dim3(256), 0, CDF_S, x, ldx, y, ldy, rows, C, p, (unsigned long long)seed);
    return cdf_check_launch("dropout");
}
extern "C" int cdf_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1,
                             double beta2, double eps, int step, void* stream) {
    CDF_REQUIRE(p && g && m && v && n > 0 && step >= 1, "cdf_adam_step: bad args");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const double step_size = lr / bc1;
    CDF_LAUNCH(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, CDF_S, p, g, m, v, n, (float)(1.0 - beta1), (float)beta2,
               (float)(1.0 - beta2), (float)(-step_size), (float)sqrt(bc2), (float)eps);
    return cdf_check_launch("adam");
}
extern "C" int cdf_ema_update(float* ma, const float* p, long long n, double beta, void* stream) {
    CDF_REQUIRE(ma && p && n > 0, "cdf_ema_update: bad args");
    CDF_LAUNCH(ema_kernel, dim3(ew_grid(n)), dim3(256), 0, CDF_S, ma, p, n, (float)beta, (float)(1.0 - beta));
    return cdf_check_launch("ema");
}
extern "C" int cdf_scale(float* x, long long n, float s, void* stream) {
    CDF_REQUIRE(x && n > 0, "cdf_scale: bad args");
    CDF_LAUNCH(scale_kernel, dim3(ew_grid(n)), dim3(256), 0, CDF_S, x, n, s);
    return cdf_check_launch("scale");
}
extern "C" int cdf_zero(void* p, long long bytes, void* stream) {
    CDF_REQUIRE(p && bytes >= 0, "cdf_zero: bad args");
    if (bytes == 0) return CDF_OK;
    hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, CDF_S);
    if (e != hipSuccess) {
        cdf_set_error("cdf_zero: hipMemsetAsync failed: %s", hipGetErrorString(e));
        return CDF_E_LAUNCH;
    }
    return CDF_OK;
}

// k_norm.hip — normalisation layers on NHWC feature maps (pixel pitch ld).
//
//   channel LayerNorm  : deblurring_diffusion_pytorch.py:111-121  (biased var, eps 1e-5,
//                        (x-mean)/sqrt(var+eps)*g+b) — per-pixel reduction over C, contiguous in
//                        NHWC, done with wave64 sub-group shuffles (LP lanes per pixel).
//   GroupNorm(32)+SiLU : Model2.py:27-33,116-123 (eps 1e-6, affine, swish) — per (sample, group)
//                        statistics over (C/32 channels x all pixels); two-level reduction.
// Both are HBM-bound: forward = 1 read + 1 write, backward = 2 reads + 1 write (+ small stats).
#include "cdf_common.h"
#include "colddiff.h"

// ------------------------------------------------------------------------------------------------
// channel LayerNorm
// ------------------------------------------------------------------------------------------------
// HBM-bound (8 B / element forward, 12-16 B backward).  Each wave handles 64/LP pixels at a time, and every lane keeps
// the loads of U consecutive pixel groups in flight (unconditional: clamped address + select) -- with one load per
// lane in flight and the 512-block grid the kernels sat at a quarter of the HBM rate.
// XB (bf16 activation storage): x is a bf16 tensor (pitch in bf16 elements); statistics and the output arithmetic stay fp32.
template <int NV, bool XB>
__global__ void __launch_bounds__(256) layernorm_c_fwd_kernel(const void* x, int ldx, float* y, int ldy, const float* g,
                                                              const float* bta, float* mean_out, float* rstd_out,
                                                              long long M, int C, int LP, float eps, unsigned short* ys_hi,
                                                              unsigned short* ys_lo, int ld_ys) {
    constexpr int U = NV <= 2 ? 4 : 2;
    const int lane = threadIdx.x & 63, sub = lane / LP, li = lane - sub * LP;
    const int groups_per_wave = 64 / LP;
    const long long wave_global = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long npg = (M + groups_per_wave - 1) / groups_per_wave;  // pixel groups
    float4 gg[NV], bb[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (li + j * LP) * 4;
        gg[j] = *(const float4*)(g + (c < C ? c : 0));
        bb[j] = *(const float4*)(bta + (c < C ? c : 0));
    }
    for (long long pg0 = wave_global * U; pg0 < npg; pg0 += nwaves * U) {
        float4 v[U][NV];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long m = (pg0 + u) * groups_per_wave + sub;
            const long long mc = m < M ? m : M - 1;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int c = (li + j * LP) * 4;
                const float4 t = cdf_quad_cvt(cdf_quad_ld<XB>(x, mc * ldx + (c < C ? c : 0)));
                v[u][j] = (m < M && c < C) ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long m = (pg0 + u) * groups_per_wave + sub;
            const bool valid = m < M;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) s += (v[u][j].x + v[u][j].y) + (v[u][j].z + v[u][j].w);
            s = cdf_group_sum(s, LP);
            const float mean = s / (float)C;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int c = (li + j * LP) * 4;
                if (c < C) {
                    const float a0 = v[u][j].x - mean, a1 = v[u][j].y - mean, a2 = v[u][j].z - mean, a3 = v[u][j].w - mean;
                    q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                }
            }
            q = cdf_group_sum(q, LP);
            const float sd = sqrtf(q / (float)C + eps);
            if (valid) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int c = (li + j * LP) * 4;
                    if (c < C) {
                        float4 o;
                        o.x = (v[u][j].x - mean) / sd * gg[j].x + bb[j].x;
                        o.y = (v[u][j].y - mean) / sd * gg[j].y + bb[j].y;
                        o.z = (v[u][j].z - mean) / sd * gg[j].z + bb[j].z;
                        o.w = (v[u][j].w - mean) / sd * gg[j].w + bb[j].w;
                        if (y) *(float4*)(y + m * ldy + c) = o;
                        if (ys_hi) {                         // operand split of the following conv fused in
                            const float ov[4] = {o.x, o.y, o.z, o.w};
                            cdf_split_store4(ys_hi + m * ld_ys + c, ys_lo ? ys_lo + m * ld_ys + c : nullptr, ov);
                        }
                    }
                }
                if (li == 0 && mean_out) {
                    mean_out[m] = mean;
                    rstd_out[m] = 1.0f / sd;
                }
            }
        }
    }
}

// backward: dx, and per-block partial sums of dg / db in part[block][2][C]
// IO (bf16 activation storage), bits CDF_LN_*_BF16: which of dy / x / dx / add are bf16 tensors (pitches in their own elements).
// Used: 0 (all fp32), dy | x | dx (the ConvNeXt block: dhn, h -> dh), x | dx | add (the attention block: fp32 dxn, bf16 stream).
#define CDF_LN_DY_BF16 1
#define CDF_LN_X_BF16 2
#define CDF_LN_DX_BF16 4
#define CDF_LN_ADD_BF16 8
template <int NV, int IO>
__global__ void __launch_bounds__(256) layernorm_c_bwd_kernel(const void* dy, int lddy, const void* x, int ldx,
                                                              const float* g, const float* mean_in, const float* rstd_in,
                                                              void* dx, int lddx, float* part, long long M, int C,
                                                              int LP, const void* add, int ldadd, unsigned short* ph, unsigned short* pl, int ldp) {
    constexpr bool DYB = (IO & CDF_LN_DY_BF16) != 0, XB = (IO & CDF_LN_X_BF16) != 0, DXB = (IO & CDF_LN_DX_BF16) != 0, ADB = (IO & CDF_LN_ADD_BF16) != 0;
    constexpr int U = NV <= 2 ? 2 : 1;
    const bool accumulate_dx = add != nullptr;               // dx = grad + add (add == dx: accumulate in place)
    CDF_DYN_SMEM(smem);
    float* sred = (float*)smem;  // [G][2][C]
    const int lane = threadIdx.x & 63, sub = lane / LP, li = lane - sub * LP;
    const int groups_per_wave = 64 / LP;
    const int G = (blockDim.x >> 6) * groups_per_wave;
    const int gidx = (threadIdx.x >> 6) * groups_per_wave + sub;
    const long long wave_global = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long npg = (M + groups_per_wave - 1) / groups_per_wave;
    float4 adg[NV], adb[NV], gg[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (li + j * LP) * 4;
        adg[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        adb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        gg[j] = *(const float4*)(g + (c < C ? c : 0));
    }
    for (long long pg0 = wave_global * U; pg0 < npg; pg0 += nwaves * U) {
        float4 xv[U][NV], dv[U][NV];
        typename cdf_quad<ADB>::raw old[U][NV];
        float mean[U], rstd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long m = (pg0 + u) * groups_per_wave + sub;
            const long long mc = m < M ? m : M - 1;
            const bool valid = m < M;
            mean[u] = mean_in[mc];
            rstd[u] = valid ? rstd_in[mc] : 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int c = (li + j * LP) * 4, cc = c < C ? c : 0;
                const bool ok = valid && c < C;
                const typename cdf_quad<XB>::raw rx = cdf_quad_ld<XB>(x, mc * ldx + cc);
                const typename cdf_quad<DYB>::raw rd = cdf_quad_ld<DYB>(dy, mc * lddy + cc);
                if (accumulate_dx) old[u][j] = cdf_quad_ld<ADB>(add, mc * ldadd + cc);     // block-uniform branch
                const float4 tx = cdf_quad_cvt(rx), td = cdf_quad_cvt(rd);
                xv[u][j] = ok ? tx : make_float4(mean[u], mean[u], mean[u], mean[u]);      // => xhat = 0
                dv[u][j] = ok ? td : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long m = (pg0 + u) * groups_per_wave + sub;
            const bool valid = m < M;
            float4 xh[NV], dg_[NV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                xh[j].x = (xv[u][j].x - mean[u]) * rstd[u]; xh[j].y = (xv[u][j].y - mean[u]) * rstd[u];
                xh[j].z = (xv[u][j].z - mean[u]) * rstd[u]; xh[j].w = (xv[u][j].w - mean[u]) * rstd[u];
                const float4 d = dv[u][j];
                adg[j].x += d.x * xh[j].x; adg[j].y += d.y * xh[j].y; adg[j].z += d.z * xh[j].z; adg[j].w += d.w * xh[j].w;
                adb[j].x += d.x; adb[j].y += d.y; adb[j].z += d.z; adb[j].w += d.w;
                dg_[j].x = d.x * gg[j].x; dg_[j].y = d.y * gg[j].y; dg_[j].z = d.z * gg[j].z; dg_[j].w = d.w * gg[j].w;
                s1 += (dg_[j].x + dg_[j].y) + (dg_[j].z + dg_[j].w);
                s2 += (dg_[j].x * xh[j].x + dg_[j].y * xh[j].y) + (dg_[j].z * xh[j].z + dg_[j].w * xh[j].w);
            }
            s1 = cdf_group_sum(s1, LP) / (float)C;
            s2 = cdf_group_sum(s2, LP) / (float)C;
            if (valid) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int c = (li + j * LP) * 4;
                    if (c < C) {
                        float4 o;
                        o.x = rstd[u] * (dg_[j].x - s1 - xh[j].x * s2);
                        o.y = rstd[u] * (dg_[j].y - s1 - xh[j].y * s2);
                        o.z = rstd[u] * (dg_[j].z - s1 - xh[j].z * s2);
                        o.w = rstd[u] * (dg_[j].w - s1 - xh[j].w * s2);
                        if (accumulate_dx) {
                            const float4 ov = cdf_quad_cvt(old[u][j]);
                            o.x += ov.x; o.y += ov.y; o.z += ov.z; o.w += ov.w;
                        }
                        cdf_quad_st<DXB>(dx, m * lddx + c, o);
                        if (!DXB && ph) {                    // also as bf16 hi / lo planes (the consumer's GEMM operand: fused cdf_split_bf16)
                            const float ov4[4] = {o.x, o.y, o.z, o.w};
                            cdf_split_store4(ph + m * ldp + c, pl + m * ldp + c, ov4);
                        }
                    }
                }
            }
        }
    }
    // block reduction of the per-lane dg/db partials: sred[g][0/1][c]
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (li + j * LP) * 4;
        if (c < C) {
            *(float4*)(sred + ((size_t)gidx * 2 + 0) * C + c) = adg[j];
            *(float4*)(sred + ((size_t)gidx * 2 + 1) * C + c) = adb[j];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        float s = 0.f;
        for (int gg2 = 0; gg2 < G; ++gg2) s += sred[(size_t)gg2 * 2 * C + i];
        part[(size_t)blockIdx.x * 2 * C + i] = s;
    }
}

// out[c] (+)= sum_blocks part[block][which][c]; block 1024 = 16 partial lanes x 64 columns of the [2C] vector
// (the vector is short -- 2 blocks at C = 64 -- so the parallelism has to come from splitting the partials)
__global__ void __launch_bounds__(1024) norm_param_reduce_kernel(const float* part, int nblocks, int C, float* dg, float* db, int accumulate) {
    __shared__ float red[16][64];
    const int l = threadIdx.x & 63, rl = threadIdx.x >> 6, i = blockIdx.x * 64 + l;
    float s0 = 0.f, s1 = 0.f;
    if (i < 2 * C) {
        // eight independent loads in flight (1024 partials = 64 per lane: with two in flight the 32 dependent round trips were
        // 11 of the kernel's 12 us); the summation tree is fixed by nblocks: deterministic
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f, t4 = 0.f, t5 = 0.f, t6 = 0.f, t7 = 0.f;
        const float* p = part + i;
        const size_t st = (size_t)2 * C;
        int b = rl;
        for (; b + 112 < nblocks; b += 128) {
            const float v0 = p[(size_t)b * st], v1 = p[(size_t)(b + 16) * st], v2 = p[(size_t)(b + 32) * st], v3 = p[(size_t)(b + 48) * st];
            const float v4 = p[(size_t)(b + 64) * st], v5 = p[(size_t)(b + 80) * st], v6 = p[(size_t)(b + 96) * st], v7 = p[(size_t)(b + 112) * st];
            t0 += v0; t1 += v1; t2 += v2; t3 += v3; t4 += v4; t5 += v5; t6 += v6; t7 += v7;
        }
        for (; b + 16 < nblocks; b += 32) {
            s0 += p[(size_t)b * st];
            s1 += p[(size_t)(b + 16) * st];
        }
        if (b < nblocks) s0 += p[(size_t)b * st];
        s0 += (t0 + t1) + (t2 + t3);
        s1 += (t4 + t5) + (t6 + t7);
    }
    red[rl][l] = s0 + s1;
    __syncthreads();
    if (rl == 0 && i < 2 * C) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][l];
        float* dst = i < C ? dg + i : db + (i - C);
        *dst = accumulate ? *dst + t : t;
    }
}

// ------------------------------------------------------------------------------------------------
// GroupNorm(32 groups) [+ SiLU]
// ------------------------------------------------------------------------------------------------
// Per-(sample, chunk, channel) partial sums of two quantities:
//   mode 0 (fwd stats) : p0 = sum x            p1 = sum x^2
//   mode 1 (bwd)       : p0 = sum dz           p1 = sum dz * xhat,  dz = dy * act'(xhat*gamma+beta)
// part layout [B][nchunk][2][C]; grid = (ceil(C/64), nchunk, B); block 256 = 4 row lanes x 64 ch
__global__ void groupnorm_partial_kernel(const float* x, int ldx, const float* dy, int lddy, const float* gamma,
                                         const float* beta, const float* mean, const float* rstd, float* part, int HW,
                                         int rows_per_chunk, int C, int groups, int mode, int silu, float p_drop,
                                         unsigned long long seed) {
    // p_drop > 0 (mode 1): dy is the gradient of the DROPPED output, the mask (cdf_dropout's, element (b HW + r) C + c) is applied here
    const unsigned thr = (unsigned)((double)p_drop * 4294967296.0);
    const float inv = 1.0f / (1.0f - p_drop);
    __shared__ float red[2][4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, b = blockIdx.z;
    const int r0 = blockIdx.y * rows_per_chunk;
    int r1 = r0 + rows_per_chunk;
    if (r1 > HW) r1 = HW;
    float p0 = 0.f, p1 = 0.f;
    if (c < C) {
        const float* xp = x + (long long)b * HW * ldx + c;
        if (mode == 0) {
            int r = r0 + rl;
            for (; r + 12 < r1; r += 16) {                   // 4 rows in flight per lane
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = xp[(long long)(r + 4 * u) * ldx];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    p0 += v[u];
                    p1 += v[u] * v[u];
                }
            }
            for (; r < r1; r += 4) {
                const float v = xp[(long long)r * ldx];
                p0 += v;
                p1 += v * v;
            }
        } else {
            const int gi = c / (C / groups);
            const float mu = mean[b * groups + gi], rs = rstd[b * groups + gi], ga = gamma[c], be = beta[c];
            const float* dp = dy + (long long)b * HW * lddy + c;
            int r = r0 + rl;
            for (; r + 12 < r1; r += 16) {                   // 4 row pairs in flight per lane
                float xv[4], dv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xv[u] = xp[(long long)(r + 4 * u) * ldx];
                    dv[u] = dp[(long long)(r + 4 * u) * lddy];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float xh = (xv[u] - mu) * rs;
                    float dz = dv[u];
                    if (p_drop > 0.f) dz = cdf_hash32(seed, (unsigned long long)((long long)b * HW + r + 4 * u) * C + c) >= thr ? dz * inv : 0.f;
                    if (silu) dz *= cdf_silu_grad(xh * ga + be);
                    p0 += dz;
                    p1 += dz * xh;
                }
            }
            for (; r < r1; r += 4) {
                const float xh = (xp[(long long)r * ldx] - mu) * rs;
                float dz = dp[(long long)r * lddy];
                if (p_drop > 0.f) dz = cdf_hash32(seed, (unsigned long long)((long long)b * HW + r) * C + c) >= thr ? dz * inv : 0.f;
                if (silu) dz *= cdf_silu_grad(xh * ga + be);
                p0 += dz;
                p1 += dz * xh;
            }
        }
    }
    red[0][rl][threadIdx.x & 63] = p0;
    red[1][rl][threadIdx.x & 63] = p1;
    __syncthreads();
    if (rl == 0 && c < C) {
        const int l = threadIdx.x;
        float* dst = part + (((long long)b * gridDim.y + blockIdx.y) * 2) * C + c;
        dst[0] = (red[0][0][l] + red[0][1][l]) + (red[0][2][l] + red[0][3][l]);
        dst[C] = (red[1][0][l] + red[1][1][l]) + (red[1][2][l] + red[1][3][l]);
    }
}

// forward finalize: mean/rstd per (b, group). grid = B, block = 64 (>= groups... loops)
__global__ void groupnorm_stats_kernel(const float* part, int nchunk, int C, int groups, int HW, float eps, float* mean,
                                       float* rstd) {
    const int b = blockIdx.x, cg = C / groups;
    for (int gi = threadIdx.x; gi < groups; gi += blockDim.x) {
        double s = 0.0, q = 0.0;
        for (int k = 0; k < nchunk; ++k) {
            const float* p = part + (((long long)b * nchunk + k) * 2) * C;
            for (int c = gi * cg; c < (gi + 1) * cg; ++c) {
                s += (double)p[c];
                q += (double)p[C + c];
            }
        }
        const double n = (double)HW * cg;
        const double mu = s / n;
        double var = q / n - mu * mu;
        if (var < 0.0) var = 0.0;
        mean[b * groups + gi] = (float)mu;
        rstd[b * groups + gi] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// backward finalize, three small multi-block kernels (one 1024-thread block doing all of it serially took ~95 us):
//  (1) ab[b][2][C] = sum_chunk part           (A = sum dz, Bc = sum dz*xhat per sample and channel)
//  (2) per (b, group): s1 = sum_c gamma*A / n, s2 = sum_c gamma*Bc / n
//  (3) dgamma[c] (+)= sum_b Bc, dbeta[c] (+)= sum_b A      (16 batch lanes x 64 channels per block)
__global__ void groupnorm_bwd_reduce_kernel(const float* part, int nchunk, int B, int C, float* ab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 2 * C) return;
    const int b = i / (2 * C), j = i - b * 2 * C;
    float s = 0.f;
#pragma unroll 4
    for (int k = 0; k < nchunk; ++k) s += part[(((long long)b * nchunk + k) * 2) * C + j];
    ab[i] = s;
}
__global__ void groupnorm_bwd_group_kernel(const float* ab, int B, int C, int groups, int HW, const float* gamma, float* s12) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * groups) return;
    const int cg = C / groups, b = i / groups, gi = i - b * groups;
    float s1 = 0.f, s2 = 0.f;
    for (int c = gi * cg; c < (gi + 1) * cg; ++c) {
        s1 += gamma[c] * ab[(long long)b * 2 * C + c];
        s2 += gamma[c] * ab[(long long)b * 2 * C + C + c];
    }
    const float n = (float)HW * (float)cg;
    s12[2 * i] = s1 / n;
    s12[2 * i + 1] = s2 / n;
}
__global__ void __launch_bounds__(1024) groupnorm_bwd_param_kernel(const float* ab, int B, int C, float* dgamma, float* dbeta, int accumulate) {
    __shared__ float red[2][16][64];
    const int l = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + l;
    float a = 0.f, bc = 0.f;
    if (c < C)
        for (int b = rl; b < B; b += 16) {
            a += ab[(long long)b * 2 * C + c];
            bc += ab[(long long)b * 2 * C + C + c];
        }
    red[0][rl][l] = a;
    red[1][rl][l] = bc;
    __syncthreads();
    if (rl == 0 && c < C) {
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ta += red[0][r][l];
            tb += red[1][r][l];
        }
        dgamma[c] = accumulate ? dgamma[c] + tb : tb;
        dbeta[c] = accumulate ? dbeta[c] + ta : ta;
    }
}

// y = drop(act((x-mean)*rstd*gamma+beta)); float4 over channels.  Optional tail of the ResnetBlock chain (Model2.py:118-126): the
// dropout mask of cdf_dropout (same hash, same element index) and the result again as bf16 hi / lo planes (the next conv's operand
// split, bit-equal to cdf_split_bf16 of y); y may be null when only the planes are wanted.
__global__ void groupnorm_apply_kernel(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta,
                                       const float* mean, const float* rstd, int B, int HW, int C, int groups, int silu, float p_drop,
                                       unsigned long long seed, unsigned short* y_hi, unsigned short* y_lo, int ld_ys) {
    const int c4n = C / 4, cg = C / groups;
    const unsigned thr = (unsigned)((double)p_drop * 4294967296.0);
    const float inv = 1.0f / (1.0f - p_drop);
    const long long n = (long long)B * HW * c4n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const long long pix = i / c4n;
        const int b = (int)(pix / HW);
        const float4 v = *(const float4*)(x + pix * ldx + c);
        const float4 ga = *(const float4*)(gamma + c), be = *(const float4*)(beta + c);
        float in[4] = {v.x, v.y, v.z, v.w}, gaa[4] = {ga.x, ga.y, ga.z, ga.w}, bee[4] = {be.x, be.y, be.z, be.w}, o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int gi = (c + e) / cg;
            const float z = (in[e] - mean[b * groups + gi]) * rstd[b * groups + gi] * gaa[e] + bee[e];
            o[e] = silu ? cdf_silu(z) : z;
            if (p_drop > 0.f) o[e] = cdf_hash32(seed, (unsigned long long)(pix * C + c + e)) >= thr ? o[e] * inv : 0.f;
        }
        if (y) *(float4*)(y + pix * ldy + c) = make_float4(o[0], o[1], o[2], o[3]);
        if (y_hi) cdf_split_store4(y_hi + pix * ld_ys + c, y_lo ? y_lo + pix * ld_ys + c : nullptr, o);
    }
}

// dx = rstd * (dz*gamma - s1 - xhat*s2)
__global__ void groupnorm_bwd_apply_kernel(const float* dy, int lddy, const float* x, int ldx, const float* gamma,
                                           const float* beta, const float* mean, const float* rstd, const float* s12,
                                           float* dx, int lddx, int B, int HW, int C, int groups, int silu,
                                           int accumulate, float p_drop, unsigned long long seed) {
    const int c4n = C / 4, cg = C / groups;
    const unsigned thr = (unsigned)((double)p_drop * 4294967296.0);
    const float inv = 1.0f / (1.0f - p_drop);
    const long long n = (long long)B * HW * c4n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const long long pix = i / c4n;
        const int b = (int)(pix / HW);
        const float4 v = *(const float4*)(x + pix * ldx + c), d = *(const float4*)(dy + pix * lddy + c);
        const float4 ga = *(const float4*)(gamma + c), be = *(const float4*)(beta + c);
        float in[4] = {v.x, v.y, v.z, v.w}, dd[4] = {d.x, d.y, d.z, d.w}, gaa[4] = {ga.x, ga.y, ga.z, ga.w},
              bee[4] = {be.x, be.y, be.z, be.w}, o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int gi = (c + e) / cg, sg = b * groups + gi;
            const float rs = rstd[sg], xh = (in[e] - mean[sg]) * rs;
            float dz = dd[e];
            if (p_drop > 0.f) dz = cdf_hash32(seed, (unsigned long long)(pix * C + c + e)) >= thr ? dz * inv : 0.f;
            if (silu) dz *= cdf_silu_grad(xh * gaa[e] + bee[e]);
            o[e] = rs * (dz * gaa[e] - s12[2 * sg] - xh * s12[2 * sg + 1]);
        }
        float* dst = dx + pix * lddx + c;
        if (accumulate) {
            const float4 old = *(const float4*)dst;
            o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w;
        }
        *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ================================================================================================
// C ABI
// ================================================================================================
static int ln_geometry(int C, int* LP, int* NV) {
    int lp = 1;
    while (lp < 64 && lp * 4 < C) lp <<= 1;
    *LP = lp;
    *NV = cdf_cdiv(C, 4 * lp);
    return (*NV >= 1 && *NV <= 4) ? CDF_OK : CDF_E_UNSUPPORTED;
}

extern "C" int cdf_layernorm_blocks(long long M, int C) {
    int LP, NV;
    if (ln_geometry(C, &LP, &NV)) return 0;
    const long long groups_per_block = 4 * (64 / LP);
    long long nb = (M + groups_per_block - 1) / groups_per_block;
    if (nb > 1024) nb = 1024;                     // 4 blocks per CU (the backward's partial sums are [nb][2][C])
    return nb < 1 ? 1 : (int)nb;
}

// x_bf16 != 0: x is a bf16 tensor (pitch in bf16 elements, 8-byte aligned)
extern "C" int cdf_layernorm_c_fwd_io(const void* x, int ldx, float* y, int ldy, const float* g, const float* b,
                                      float* mean, float* rstd, long long M, int C, float eps, void* y_hi, void* y_lo, int ld_ys,
                                      int x_bf16, void* stream) {
    CDF_REQUIRE((!y_hi && !y_lo) || (y_hi && ld_ys % 4 == 0 && ld_ys >= C && ((((uintptr_t)y_hi) | ((uintptr_t)y_lo)) & 7) == 0),
                "cdf_layernorm_c_fwd: split output planes need ld_ys %% 4 == 0, ld_ys >= C, 8-byte alignment");
    CDF_REQUIRE(x && (y || y_hi) && g && b && M > 0, "cdf_layernorm_c_fwd: null / empty");
    CDF_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && (!y || ldy % 4 == 0) && C <= 1024, "cdf_layernorm_c_fwd: C=%d must be a multiple of 4 and <= 1024", C);
    CDF_REQUIRE((((uintptr_t)x) & (x_bf16 ? 7 : 15)) == 0, "cdf_layernorm_c_fwd: x must be 16-byte aligned (bf16: 8)");
    int LP, NV;
    CDF_REQUIRE(ln_geometry(C, &LP, &NV) == CDF_OK, "cdf_layernorm_c_fwd: unsupported C=%d", C);
    const int nb = cdf_layernorm_blocks(M, C);
#define CDF_LN_FWD(N, XB) CDF_LAUNCH((layernorm_c_fwd_kernel<N, XB>), dim3(nb), dim3(256), 0, CDF_S, x, ldx, y, ldy, g, b, mean, rstd, M, C, LP, eps, (unsigned short*)y_hi, (unsigned short*)y_lo, ld_ys)
#define CDF_LN_FWD_NV(XB)                  \
    switch (NV) {                          \
        case 1: CDF_LN_FWD(1, XB); break;  \
        case 2: CDF_LN_FWD(2, XB); break;  \
        case 3: CDF_LN_FWD(3, XB); break;  \
        default: CDF_LN_FWD(4, XB); break; \
    }
    if (x_bf16) { CDF_LN_FWD_NV(true) } else { CDF_LN_FWD_NV(false) }
#undef CDF_LN_FWD_NV
#undef CDF_LN_FWD
    return cdf_check_launch("layernorm_c_fwd");
}

extern "C" int cdf_layernorm_c_fwd(const float* x, int ldx, float* y, int ldy, const float* g, const float* b,
                                   float* mean, float* rstd, long long M, int C, float eps, void* y_hi, void* y_lo, int ld_ys,
                                   void* stream) {
    return cdf_layernorm_c_fwd_io(x, ldx, y, ldy, g, b, mean, rstd, M, C, eps, y_hi, y_lo, ld_ys, 0, stream);
}

// part: >= cdf_layernorm_blocks(M, C) * 2 * C floats
// io_bf16: CDF_LN_DY_BF16 (1) | CDF_LN_X_BF16 (2) | CDF_LN_DX_BF16 (4) | CDF_LN_ADD_BF16 (8) -- which tensors are bf16 (pitches in their own
// elements, 8-byte aligned); supported combinations: 0, 7 (dy, x, dx: the ConvNeXt block) and 14 (x, dx, add: the attention block)
static int ln_bwd_launch(const void* dy, int lddy, const void* x, int ldx, const float* g,
                                      const float* mean, const float* rstd, void* dx, int lddx, const void* add, int ldadd,
                                      float* dg, float* db, float* part, long long M, int C, int accumulate_dx, int accumulate_param,
                                      int io_bf16, void* dx_hi, void* dx_lo, int ld_planes, void* stream) {
    CDF_REQUIRE(dy && x && g && mean && rstd && dx && dg && db && part && M > 0, "cdf_layernorm_c_bwd: null / empty");
    CDF_REQUIRE((dx_hi != nullptr) == (dx_lo != nullptr) && (!dx_hi || (io_bf16 == 0 && ld_planes % 4 == 0 && ld_planes >= C && ((((uintptr_t)dx_hi) | ((uintptr_t)dx_lo)) & 7) == 0)),
                "cdf_layernorm_c_bwd_planes: both planes or neither, fp32 tensors only, 8-byte aligned, pitch %% 4 == 0");
    CDF_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && C <= 1024, "cdf_layernorm_c_bwd: bad C / pitch");
    CDF_REQUIRE(!(add && accumulate_dx) && (!add || (ldadd % 4 == 0 && (((uintptr_t)add) & ((io_bf16 & CDF_LN_ADD_BF16) ? 7 : 15)) == 0)), "cdf_layernorm_c_bwd: add and accumulate_dx exclude each other; add must be 16-byte aligned (bf16: 8) with a pitch % 4 == 0");
    CDF_REQUIRE(io_bf16 == 0 || io_bf16 == 7 || io_bf16 == 14, "cdf_layernorm_c_bwd_io: io_bf16 = %d is not one of 0 / 7 / 14", io_bf16);
    CDF_REQUIRE((((uintptr_t)dy) & ((io_bf16 & CDF_LN_DY_BF16) ? 7 : 15)) == 0 && (((uintptr_t)x) & ((io_bf16 & CDF_LN_X_BF16) ? 7 : 15)) == 0 &&
                (((uintptr_t)dx) & ((io_bf16 & CDF_LN_DX_BF16) ? 7 : 15)) == 0, "cdf_layernorm_c_bwd: dy / x / dx must be 16-byte aligned (bf16: 8)");
    if (accumulate_dx) {
        CDF_REQUIRE(io_bf16 != 7, "cdf_layernorm_c_bwd_io: accumulate_dx (dx read back in its own type) is not instantiated for io_bf16 = 7");
        add = dx; ldadd = lddx;
    }
    int LP, NV;
    CDF_REQUIRE(ln_geometry(C, &LP, &NV) == CDF_OK, "cdf_layernorm_c_bwd: unsupported C=%d", C);
    const int nb = cdf_layernorm_blocks(M, C);
    const int G = 4 * (64 / LP);
    const size_t lds = (size_t)G * 2 * C * sizeof(float);
#ifndef CDF_EMU
    static CdfDeviceLatch attr_done;
    if (attr_done.first()) {
#define CDF_LN_ATTR(N, IO) (void)hipFuncSetAttribute((const void*)layernorm_c_bwd_kernel<N, IO>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
        CDF_LN_ATTR(1, 0); CDF_LN_ATTR(2, 0); CDF_LN_ATTR(3, 0); CDF_LN_ATTR(4, 0);
        CDF_LN_ATTR(1, 7); CDF_LN_ATTR(2, 7); CDF_LN_ATTR(3, 7); CDF_LN_ATTR(4, 7);
        CDF_LN_ATTR(1, 14); CDF_LN_ATTR(2, 14); CDF_LN_ATTR(3, 14); CDF_LN_ATTR(4, 14);
#undef CDF_LN_ATTR
    }
#endif
#define CDF_LN_BWD(N, IO) CDF_LAUNCH((layernorm_c_bwd_kernel<N, IO>), dim3(nb), dim3(256), lds, CDF_S, dy, lddy, x, ldx, g, mean, rstd, dx, lddx, part, M, C, LP, add, ldadd, (unsigned short*)dx_hi, (unsigned short*)dx_lo, ld_planes)
#define CDF_LN_BWD_NV(IO)                  \
    switch (NV) {                          \
        case 1: CDF_LN_BWD(1, IO); break;  \
        case 2: CDF_LN_BWD(2, IO); break;  \
        case 3: CDF_LN_BWD(3, IO); break;  \
        default: CDF_LN_BWD(4, IO); break; \
    }
    if (io_bf16 == 7) { CDF_LN_BWD_NV(7) } else if (io_bf16 == 14) { CDF_LN_BWD_NV(14) } else { CDF_LN_BWD_NV(0) }
#undef CDF_LN_BWD_NV
#undef CDF_LN_BWD
    CDF_LAUNCH(norm_param_reduce_kernel, dim3(cdf_cdiv(2 * C, 64)), dim3(1024), 0, CDF_S, (const float*)part, nb, C, dg, db, accumulate_param);
    return cdf_check_launch("layernorm_c_bwd");
}

// dg[c] (+)= sum_b part[b][0][c], db[c] (+)= sum_b part[b][1][c]: the second stage of every LayerNorm backward (also behind the fused
// data-gradient epilogue, cdf_conv_gemm_bf16x_lnbwd)
extern "C" int cdf_norm_param_reduce(const float* part, int nblocks, int C, float* dg, float* db, int accumulate, void* stream) {
    CDF_REQUIRE(part && dg && db && nblocks > 0 && C > 0, "cdf_norm_param_reduce: bad args");
    CDF_LAUNCH(norm_param_reduce_kernel, dim3(cdf_cdiv(2 * C, 64)), dim3(1024), 0, CDF_S, part, nblocks, C, dg, db, accumulate);
    return cdf_check_launch("norm_param_reduce");
}

extern "C" int cdf_layernorm_c_bwd_io(const void* dy, int lddy, const void* x, int ldx, const float* g,
                                      const float* mean, const float* rstd, void* dx, int lddx, const void* add, int ldadd,
                                      float* dg, float* db, float* part, long long M, int C, int accumulate_dx, int accumulate_param,
                                      int io_bf16, void* stream) {
    return ln_bwd_launch(dy, lddy, x, ldx, g, mean, rstd, dx, lddx, add, ldadd, dg, db, part, M, C, accumulate_dx, accumulate_param, io_bf16, nullptr, nullptr, 0, stream);
}
// ... and dx ALSO as bf16 hi / lo planes (fp32 tensors): what the consumer of this gradient would otherwise produce with cdf_split_bf16
extern "C" int cdf_layernorm_c_bwd_planes(const float* dy, int lddy, const float* x, int ldx, const float* g, const float* mean, const float* rstd,
                                          float* dx, int lddx, const float* add, int ldadd, float* dg, float* db, float* part, long long M, int C,
                                          int accumulate_dx, int accumulate_param, void* dx_hi, void* dx_lo, int ld_planes, void* stream) {
    return ln_bwd_launch(dy, lddy, x, ldx, g, mean, rstd, dx, lddx, add, ldadd, dg, db, part, M, C, accumulate_dx, accumulate_param, 0, dx_hi, dx_lo, ld_planes, stream);
}

extern "C" int cdf_layernorm_c_bwd(const float* dy, int lddy, const float* x, int ldx, const float* g,
                                   const float* mean, const float* rstd, float* dx, int lddx, const float* add, int ldadd,
                                   float* dg, float* db, float* part, long long M, int C, int accumulate_dx, int accumulate_param,
                                   void* stream) {
    return cdf_layernorm_c_bwd_io(dy, lddy, x, ldx, g, mean, rstd, dx, lddx, add, ldadd, dg, db, part, M, C, accumulate_dx, accumulate_param, 0, stream);
}

extern "C" int cdf_groupnorm_nchunk(int HW) {
    int n = HW / 256;
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    return n;
}

// ws: >= B * nchunk * 2 * C floats ; mean/rstd: [B][groups]
extern "C" int cdf_groupnorm_fwd_ex(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta,
                                    float* mean, float* rstd, float* ws, int B, int HW, int C, int groups, float eps,
                                    int silu, float p_drop, long long seed, void* y_hi, void* y_lo, int ld_ys, void* stream) {
    CDF_REQUIRE(x && (y || y_hi) && gamma && beta && mean && rstd && ws, "cdf_groupnorm_fwd: null pointer");
    CDF_REQUIRE(C % groups == 0 && C % 4 == 0 && ldx % 4 == 0 && (!y || ldy % 4 == 0), "cdf_groupnorm_fwd: C=%d groups=%d", C, groups);
    CDF_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "cdf_groupnorm_fwd: dropout probability %g", (double)p_drop);
    CDF_REQUIRE(!y_hi || (ld_ys % 4 == 0 && ld_ys >= C && ((((uintptr_t)y_hi) | ((uintptr_t)y_lo)) & 7) == 0), "cdf_groupnorm_fwd: bad output planes");
    CDF_REQUIRE(y_hi || !y_lo, "cdf_groupnorm_fwd: lo plane without hi plane");
    const int nchunk = cdf_groupnorm_nchunk(HW), rpc = cdf_cdiv(HW, nchunk);
    CDF_LAUNCH(groupnorm_partial_kernel, dim3(cdf_cdiv(C, 64), nchunk, B), dim3(256), 0, CDF_S, x, ldx, (const float*)nullptr, 0,
               gamma, beta, (const float*)nullptr, (const float*)nullptr, ws, HW, rpc, C, groups, 0, 0, 0.f, 0ull);
    CDF_LAUNCH(groupnorm_stats_kernel, dim3(B), dim3(64), 0, CDF_S, (const float*)ws, nchunk, C, groups, HW, eps, mean, rstd);
    const long long n = (long long)B * HW * (C / 4);
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    CDF_LAUNCH(groupnorm_apply_kernel, dim3(grid), dim3(256), 0, CDF_S, x, ldx, y, ldy, gamma, beta, (const float*)mean, (const float*)rstd, B, HW, C, groups, silu,
               p_drop, (unsigned long long)seed, (unsigned short*)y_hi, (unsigned short*)y_lo, ld_ys);
    return cdf_check_launch("groupnorm_fwd");
}
extern "C" int cdf_groupnorm_fwd(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta,
                                 float* mean, float* rstd, float* ws, int B, int HW, int C, int groups, float eps,
                                 int silu, void* stream) {
    CDF_REQUIRE(y, "cdf_groupnorm_fwd: null pointer");
    return cdf_groupnorm_fwd_ex(x, ldx, y, ldy, gamma, beta, mean, rstd, ws, B, HW, C, groups, eps, silu, 0.f, 0, nullptr, nullptr, 0, stream);
}

// ws: >= B*nchunk*2*C + B*2*C + B*groups*2 floats
extern "C" int cdf_groupnorm_bwd_ex(const float* dy, int lddy, const float* x, int ldx, const float* gamma,
                                    const float* beta, const float* mean, const float* rstd, float* dx, int lddx,
                                    float* dgamma, float* dbeta, float* ws, int B, int HW, int C, int groups, int silu,
                                    int accumulate_dx, int accumulate_param, float p_drop, long long seed, void* stream) {
    CDF_REQUIRE(dy && x && gamma && beta && mean && rstd && dx && dgamma && dbeta && ws, "cdf_groupnorm_bwd: null pointer");
    CDF_REQUIRE(C % groups == 0 && C % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0, "cdf_groupnorm_bwd: bad C / pitch");
    CDF_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "cdf_groupnorm_bwd: dropout probability %g", (double)p_drop);
    const int nchunk = cdf_groupnorm_nchunk(HW), rpc = cdf_cdiv(HW, nchunk);
    float* part = ws;
    float* ab = part + (size_t)B * nchunk * 2 * C;
    float* s12 = ab + (size_t)B * 2 * C;
    CDF_LAUNCH(groupnorm_partial_kernel, dim3(cdf_cdiv(C, 64), nchunk, B), dim3(256), 0, CDF_S, x, ldx, dy, lddy, gamma, beta, mean, rstd,
               part, HW, rpc, C, groups, 1, silu, p_drop, (unsigned long long)seed);
    CDF_LAUNCH(groupnorm_bwd_reduce_kernel, dim3(cdf_cdiv(B * 2 * C, 256)), dim3(256), 0, CDF_S, (const float*)part, nchunk, B, C, ab);
    CDF_LAUNCH(groupnorm_bwd_group_kernel, dim3(cdf_cdiv(B * groups, 256)), dim3(256), 0, CDF_S, (const float*)ab, B, C, groups, HW, gamma, s12);
    CDF_LAUNCH(groupnorm_bwd_param_kernel, dim3(cdf_cdiv(C, 64)), dim3(1024), 0, CDF_S, (const float*)ab, B, C, dgamma, dbeta, accumulate_param);
    const long long n = (long long)B * HW * (C / 4);
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    CDF_LAUNCH(groupnorm_bwd_apply_kernel, dim3(grid), dim3(256), 0, CDF_S, dy, lddy, x, ldx, gamma, beta, mean, rstd, (const float*)s12,
               dx, lddx, B, HW, C, groups, silu, accumulate_dx, p_drop, (unsigned long long)seed);
    return cdf_check_launch("groupnorm_bwd");
}
extern "C" int cdf_groupnorm_bwd(const float* dy, int lddy, const float* x, int ldx, const float* gamma,
                                 const float* beta, const float* mean, const float* rstd, float* dx, int lddx,
                                 float* dgamma, float* dbeta, float* ws, int B, int HW, int C, int groups, int silu,
                                 int accumulate_dx, int accumulate_param, void* stream) {
    return cdf_groupnorm_bwd_ex(dy, lddy, x, ldx, gamma, beta, mean, rstd, dx, lddx, dgamma, dbeta, ws, B, HW, C, groups, silu, accumulate_dx,
                                accumulate_param, 0.f, 0, stream);
}

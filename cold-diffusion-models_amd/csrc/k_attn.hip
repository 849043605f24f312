// k_attn.hip — attention inner kernels.
//
// LinearAttention (deblurring_diffusion_pytorch.py:167-187), heads x 32 channels, n = H*W tokens:
//   q *= 32^-0.5 ; k = softmax_n(k) ; ctx[d,e] = sum_n k[d,n] v[e,n] ; out[e,n] = sum_d ctx[d,e] q[d,n]
// qkv is the NHWC output of to_qkv: [B, n, 3*HD] (q | k | v, channel = head*32 + d), HD = heads*32.
// The softmax over n (up to 16 384 tokens) is a column reduction in NHWC: pass 1 finds the per
// (b, channel) max, pass 2 accumulates exp(k-max)^T v on the matrix cores (32x32x2 f32 MFMA reads
// its fragments straight from global memory: lane (d, n&1)) together with the column sums, so k
// and v are each read once per pass; O(n d^2) work, HBM-bound.
//
// AttnBlock softmax (Model2.py:164-188): row softmax of the [n, n] score matrix with scale C^-0.5.
#include "cdf_common.h"
#include "colddiff.h"

#define LA_D 32

// ---- pass 1: per (b, chunk, channel) max of k ---------------------------------------------------
// grid = (HD/64, nchunk, B); block 256 = 4 row lanes x 64 channels ; part [B][nchunk][HD]
__global__ void linattn_kmax_kernel(const float* qkv, int ld, float* part, int n, int rows_per_chunk, int HD) {
    __shared__ float red[4][64];
    const int l = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + l, b = blockIdx.z;
    const int r0 = blockIdx.y * rows_per_chunk;
    int r1 = r0 + rows_per_chunk;
    if (r1 > n) r1 = n;
    const float* kp = qkv + (long long)b * n * ld + HD + c;
    float m = -3.0e38f;
    for (int r = r0 + rl; r < r1; r += 4) m = fmaxf(m, kp[(long long)r * ld]);
    red[rl][l] = m;
    __syncthreads();
    if (rl == 0) part[((long long)b * gridDim.y + blockIdx.y) * HD + c] = fmaxf(fmaxf(red[0][l], red[1][l]), fmaxf(red[2][l], red[3][l]));
}

// ---- pass 2: context partials -------------------------------------------------------------------
// EXPA: A = exp(k - kmax) (forward) ; else A = a_ptr rows as they are (backward: A = q, B = dout).
// grid = (heads, nsplit, B), block 256.  ctx_part [B][nsplit][heads][32][32], sum_part [B][nsplit][HD]
template <bool EXPA>
__global__ void __launch_bounds__(256) linattn_ctx_kernel(const float* a_ptr, int lda, const float* b_ptr, int ldb,
                                                          const float* kmax_part, int nchunk_max, float* ctx_part,
                                                          float* sum_part, int n, int rows_per_split, int HD) {
    __shared__ float red[4][LA_D * LA_D];
    __shared__ float sred[4][LA_D];
    __shared__ float smax[LA_D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.x, split = blockIdx.y, b = blockIdx.z, heads = gridDim.x;
    const int i = lane & 31, hh = lane >> 5;
    if (EXPA) {
        if (threadIdx.x < LA_D) {
            float m = -3.0e38f;
            for (int k = 0; k < nchunk_max; ++k) m = fmaxf(m, kmax_part[((long long)b * nchunk_max + k) * HD + h * LA_D + threadIdx.x]);
            smax[threadIdx.x] = m;
        }
        __syncthreads();
    }
    const float mx = EXPA ? smax[i] : 0.f;
    const int r0 = split * rows_per_split;
    int r1 = r0 + rows_per_split;
    if (r1 > n) r1 = n;
    const float* A = a_ptr + (long long)b * n * lda + h * LA_D + i;
    const float* Bp = b_ptr + (long long)b * n * ldb + h * LA_D + i;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float psum = 0.f;
    // each wave takes row pairs r0 + 2*(wave + 4*j)
    for (int r = r0 + 2 * wave; r < r1; r += 8) {
        const int row = r + hh;
        float av = 0.f, bv = 0.f;
        if (row < r1) {
            av = A[(long long)row * lda];
            bv = Bp[(long long)row * ldb];
            if (EXPA) av = expf(av - mx);
        }
        psum += av;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    psum += __shfl_xor(psum, 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][((r & 3) + 8 * (r >> 2) + 4 * hh) * LA_D + i] = acc[r];
    if (hh == 0) sred[wave][i] = psum;
    __syncthreads();
    float* dst = ctx_part + ((((long long)b * gridDim.y + split) * heads + h) * LA_D) * LA_D;
    for (int k = threadIdx.x; k < LA_D * LA_D; k += 256) dst[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    if (sum_part && threadIdx.x < LA_D)
        sum_part[((long long)b * gridDim.y + split) * HD + h * LA_D + threadIdx.x] =
            (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
}

// ---- finalize (forward): kmax, ksum, ctx = sum_split part / ksum ---------------------------------
// grid = (heads, B), block 256
__global__ void linattn_ctx_final_kernel(const float* ctx_part, const float* sum_part, const float* kmax_part,
                                         int nsplit, int nchunk_max, int HD, float* ctx, float* kmax, float* ksum) {
    __shared__ float ssum[LA_D];
    const int h = blockIdx.x, b = blockIdx.y, heads = gridDim.x;
    if (threadIdx.x < LA_D) {
        const int c = h * LA_D + threadIdx.x;
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += sum_part[((long long)b * nsplit + k) * HD + c];
        float m = -3.0e38f;
        for (int k = 0; k < nchunk_max; ++k) m = fmaxf(m, kmax_part[((long long)b * nchunk_max + k) * HD + c]);
        ssum[threadIdx.x] = s;
        ksum[(long long)b * HD + c] = s;
        kmax[(long long)b * HD + c] = m;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < LA_D * LA_D; k += blockDim.x) {
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += ctx_part[((((long long)b * nsplit + sp) * heads + h) * LA_D) * LA_D + k];
        ctx[(((long long)b * heads + h) * LA_D) * LA_D + k] = s / ssum[k / LA_D];
    }
}

// ---- finalize (backward): dctx = scale * sum_split part ; r[d] = sum_e dctx[d,e]*ctx[d,e] -------------
__global__ void linattn_dctx_final_kernel(const float* part, int nsplit, const float* ctx, float scale, float* dctx,
                                          float* rvec, int HD) {
    __shared__ float sd[LA_D * LA_D];
    const int h = blockIdx.x, b = blockIdx.y, heads = gridDim.x;
    const long long base = (((long long)b * heads + h) * LA_D) * LA_D;
    for (int k = threadIdx.x; k < LA_D * LA_D; k += blockDim.x) {
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += part[((((long long)b * nsplit + sp) * heads + h) * LA_D) * LA_D + k];
        s *= scale;
        dctx[base + k] = s;
        sd[k] = s * ctx[base + k];
    }
    __syncthreads();
    if (threadIdx.x < LA_D) {
        float s = 0.f;
        for (int e = 0; e < LA_D; ++e) s += sd[threadIdx.x * LA_D + e];
        rvec[(long long)b * HD + h * LA_D + threadIdx.x] = s;
    }
}

// ---- out[n][h*32+e] = scale * sum_d q[n][h*32+d] * ctx[h][d][e] -----------------------------------
// grid = (row blocks, B); block 256: thread = (row-in-block, head, e-quad); ctx[b] staged in LDS.
__global__ void __launch_bounds__(256) linattn_out_kernel(const float* qkv, int ld, const float* ctx, float* out, int ldo,
                                                          int n, int heads, float scale) {
    CDF_DYN_SMEM(smem);
    float* sctx = (float*)smem;  // [heads][32][32]
    const int b = blockIdx.y;
    for (int k = threadIdx.x; k < heads * LA_D * LA_D; k += blockDim.x) sctx[k] = ctx[(long long)b * heads * LA_D * LA_D + k];
    __syncthreads();
    const int quads = heads * 8;              // e-quads per row
    const int rows_per_block = 256 / quads;   // heads=4 -> 8 rows
    const int tq = threadIdx.x % quads, tr = threadIdx.x / quads;
    const int h = tq / 8, e0 = (tq & 7) * 4;
    for (int row = blockIdx.x * rows_per_block + tr; row < n; row += gridDim.x * rows_per_block) {
        if (tr >= rows_per_block) break;
        const float* qp = qkv + ((long long)b * n + row) * ld + h * LA_D;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int d4 = 0; d4 < 8; ++d4) {
            const float4 qv = *(const float4*)(qp + d4 * 4);
            const float qs[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 cv = *(const float4*)(sctx + (h * LA_D + d4 * 4 + u) * LA_D + e0);
                acc.x = fmaf(qs[u], cv.x, acc.x);
                acc.y = fmaf(qs[u], cv.y, acc.y);
                acc.z = fmaf(qs[u], cv.z, acc.z);
                acc.w = fmaf(qs[u], cv.w, acc.w);
            }
        }
        *(float4*)(out + ((long long)b * n + row) * ldo + h * LA_D + e0) = make_float4(acc.x * scale, acc.y * scale, acc.z * scale, acc.w * scale);
    }
}

// ---- backward per-row kernel: writes dqkv [B, n, 3*HD] ---------------------------------------------
//   dq[d] = scale * sum_e ctx[d,e] dout[e] ; P[d] = exp(k[d]-kmax[d])/ksum[d]
//   dv[e] = sum_d P[d] dctx[d,e] ; dP[d] = sum_e dctx[d,e] v[e] ; dk[d] = P[d] (dP[d] - r[d])
__global__ void __launch_bounds__(256) linattn_bwd_rows_kernel(const float* qkv, int ld, const float* dout, int lddo,
                                                               const float* ctx, const float* dctx, const float* kmax,
                                                               const float* ksum, const float* rvec, float* dqkv,
                                                               int lddq, int n, int heads, float scale) {
    CDF_DYN_SMEM(smem);
    const int HD = heads * LA_D;
    float* sctx = (float*)smem;              // [heads][32][32]
    float* sdctx = sctx + heads * LA_D * LA_D;
    float* smx = sdctx + heads * LA_D * LA_D;  // [HD]
    float* ssm = smx + HD;
    float* srv = ssm + HD;
    const int b = blockIdx.y;
    for (int k = threadIdx.x; k < heads * LA_D * LA_D; k += blockDim.x) {
        sctx[k] = ctx[(long long)b * heads * LA_D * LA_D + k];
        sdctx[k] = dctx[(long long)b * heads * LA_D * LA_D + k];
    }
    for (int k = threadIdx.x; k < HD; k += blockDim.x) {
        smx[k] = kmax[(long long)b * HD + k];
        ssm[k] = ksum[(long long)b * HD + k];
        srv[k] = rvec[(long long)b * HD + k];
    }
    __syncthreads();
    const int quads = heads * 8, rows_per_block = 256 / quads;
    const int tq = threadIdx.x % quads, tr = threadIdx.x / quads;
    const int h = tq / 8, j0 = (tq & 7) * 4;
    for (int row = blockIdx.x * rows_per_block + tr; row < n; row += gridDim.x * rows_per_block) {
        if (tr >= rows_per_block) break;
        const long long rbase = (long long)b * n + row;
        const float* kp = qkv + rbase * ld + HD + h * LA_D;
        const float* vp = qkv + rbase * ld + 2 * HD + h * LA_D;
        const float* dop = dout + rbase * lddo + h * LA_D;
        float dov[LA_D], vv[LA_D], pn[LA_D];
#pragma unroll
        for (int q4 = 0; q4 < 8; ++q4) {
            const float4 a = *(const float4*)(dop + q4 * 4), c = *(const float4*)(vp + q4 * 4), kk = *(const float4*)(kp + q4 * 4);
            dov[q4 * 4] = a.x; dov[q4 * 4 + 1] = a.y; dov[q4 * 4 + 2] = a.z; dov[q4 * 4 + 3] = a.w;
            vv[q4 * 4] = c.x; vv[q4 * 4 + 1] = c.y; vv[q4 * 4 + 2] = c.z; vv[q4 * 4 + 3] = c.w;
            const float kr[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int d = q4 * 4 + u;
                pn[d] = expf(kr[u] - smx[h * LA_D + d]) / ssm[h * LA_D + d];
            }
        }
        // dq[j0..j0+3], dP[j0..j0+3]: dot products over e with rows d = j0+u of ctx / dctx
        float dq[4], dp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* cr = sctx + (h * LA_D + j0 + u) * LA_D;
            const float* dr = sdctx + (h * LA_D + j0 + u) * LA_D;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e4 = 0; e4 < 8; ++e4) {
                const float4 c4 = *(const float4*)(cr + e4 * 4), d4 = *(const float4*)(dr + e4 * 4);
                s1 = fmaf(c4.x, dov[e4 * 4], s1); s1 = fmaf(c4.y, dov[e4 * 4 + 1], s1);
                s1 = fmaf(c4.z, dov[e4 * 4 + 2], s1); s1 = fmaf(c4.w, dov[e4 * 4 + 3], s1);
                s2 = fmaf(d4.x, vv[e4 * 4], s2); s2 = fmaf(d4.y, vv[e4 * 4 + 1], s2);
                s2 = fmaf(d4.z, vv[e4 * 4 + 2], s2); s2 = fmaf(d4.w, vv[e4 * 4 + 3], s2);
            }
            dq[u] = s1 * scale;
            dp[u] = s2;
        }
        // dv[e = j0..j0+3] = sum_d P[d] dctx[d][e]
        float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int d = 0; d < LA_D; ++d) {
            const float4 d4 = *(const float4*)(sdctx + (h * LA_D + d) * LA_D + j0);
            dv.x = fmaf(pn[d], d4.x, dv.x);
            dv.y = fmaf(pn[d], d4.y, dv.y);
            dv.z = fmaf(pn[d], d4.z, dv.z);
            dv.w = fmaf(pn[d], d4.w, dv.w);
        }
        float* o = dqkv + rbase * lddq + h * LA_D + j0;
        *(float4*)o = make_float4(dq[0], dq[1], dq[2], dq[3]);
        float dk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) dk[u] = pn[j0 + u] * (dp[u] - srv[h * LA_D + j0 + u]);
        *(float4*)(o + HD) = make_float4(dk[0], dk[1], dk[2], dk[3]);
        *(float4*)(o + 2 * HD) = dv;
    }
}

// ---- row softmax (AttnBlock) ---------------------------------------------------------------------
// p[r][j] = softmax_j(scale * s[r][j]); one wave per row, rows of length n with pitch ld
__global__ void softmax_rows_fwd_kernel(const float* s, float* p, long long rows, int n, int ld, float scale) {
    const int lane = threadIdx.x & 63;
    const long long w0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = w0; r < rows; r += nw) {
        const float* sp = s + r * ld;
        float m = -3.0e38f;
        for (int j = lane; j < n; j += 64) m = fmaxf(m, sp[j] * scale);
        m = cdf_wave_max(m);
        float sum = 0.f;
        for (int j = lane; j < n; j += 64) sum += expf(sp[j] * scale - m);
        sum = cdf_wave_sum(sum);
        float* pp = p + r * ld;
        for (int j = lane; j < n; j += 64) pp[j] = expf(sp[j] * scale - m) / sum;
    }
}
// ds[r][j] = scale * p * (dp - sum_j dp*p)
__global__ void softmax_rows_bwd_kernel(const float* p, const float* dp, float* ds, long long rows, int n, int ld, float scale) {
    const int lane = threadIdx.x & 63;
    const long long w0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long r = w0; r < rows; r += nw) {
        const float* pp = p + r * ld;
        const float* dpp = dp + r * ld;
        float dot = 0.f;
        for (int j = lane; j < n; j += 64) dot = fmaf(pp[j], dpp[j], dot);
        dot = cdf_wave_sum(dot);
        float* o = ds + r * ld;
        for (int j = lane; j < n; j += 64) o[j] = scale * pp[j] * (dpp[j] - dot);
    }
}

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" int cdf_linattn_nsplit(int n) {
    int s = n / 256;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return s;
}

// ws >= B*nsplit*HD (kmax partials) + B*nsplit*heads*1024 (ctx partials) + B*nsplit*HD (sum partials) floats
extern "C" size_t cdf_linattn_ws_floats(int B, int n, int heads) {
    const size_t ns = (size_t)cdf_linattn_nsplit(n), HD = (size_t)heads * LA_D;
    return (size_t)B * ns * (2 * HD + (size_t)heads * LA_D * LA_D);
}

// qkv [B,n,ld] ; out [B,n,ldo] (HD channels) ; ctx [B,heads,32,32] ; kmax, ksum [B,HD]
extern "C" int cdf_linattn_fwd(const float* qkv, int ld, float* out, int ldo, float* ctx, float* kmax, float* ksum,
                               float* ws, int B, int n, int heads, float scale, void* stream) {
    CDF_REQUIRE(qkv && out && ctx && kmax && ksum && ws, "cdf_linattn_fwd: null pointer");
    const int HD = heads * LA_D;
    CDF_REQUIRE(HD % 64 == 0 && ld % 4 == 0 && ldo % 4 == 0 && ld >= 3 * HD && ldo >= HD && heads * 8 <= 256, "cdf_linattn_fwd: heads=%d unsupported / bad pitch", heads);
    const int ns = cdf_linattn_nsplit(n), rps = cdf_cdiv(cdf_cdiv(n, ns), 2) * 2;
    float* kmax_part = ws;
    float* ctx_part = kmax_part + (size_t)B * ns * HD;
    float* sum_part = ctx_part + (size_t)B * ns * heads * LA_D * LA_D;
    CDF_LAUNCH(linattn_kmax_kernel, dim3(HD / 64, ns, B), dim3(256), 0, CDF_S, qkv, ld, kmax_part, n, rps, HD);
    CDF_LAUNCH((linattn_ctx_kernel<true>), dim3(heads, ns, B), dim3(256), 0, CDF_S, qkv + HD, ld, qkv + 2 * HD, ld, (const float*)kmax_part, ns, ctx_part, sum_part, n, rps, HD);
    CDF_LAUNCH(linattn_ctx_final_kernel, dim3(heads, B), dim3(256), 0, CDF_S, (const float*)ctx_part, (const float*)sum_part, (const float*)kmax_part, ns, ns, HD, ctx, kmax, ksum);
    const size_t lds = (size_t)heads * LA_D * LA_D * sizeof(float);
    const int rows_per_block = 256 / (heads * 8);
    int gx = cdf_cdiv(n, rows_per_block);
    if (gx > 1024) gx = 1024;
    CDF_LAUNCH(linattn_out_kernel, dim3(gx, B), dim3(256), lds, CDF_S, qkv, ld, (const float*)ctx, out, ldo, n, heads, scale);
    return cdf_check_launch("linattn_fwd");
}

// dctx, rvec are scratch outputs ([B,heads,32,32], [B,HD]); dqkv [B,n,lddq] receives (dq | dk | dv)
extern "C" int cdf_linattn_bwd(const float* qkv, int ld, const float* dout, int lddo, const float* ctx,
                               const float* kmax, const float* ksum, float* dqkv, int lddq, float* dctx, float* rvec,
                               float* ws, int B, int n, int heads, float scale, void* stream) {
    CDF_REQUIRE(qkv && dout && ctx && kmax && ksum && dqkv && dctx && rvec && ws, "cdf_linattn_bwd: null pointer");
    const int HD = heads * LA_D;
    CDF_REQUIRE(HD % 64 == 0 && ld % 4 == 0 && lddo % 4 == 0 && lddq % 4 == 0 && lddq >= 3 * HD, "cdf_linattn_bwd: bad pitch");
    const int ns = cdf_linattn_nsplit(n), rps = cdf_cdiv(cdf_cdiv(n, ns), 2) * 2;
    float* ctx_part = ws + (size_t)B * ns * HD;
    CDF_LAUNCH((linattn_ctx_kernel<false>), dim3(heads, ns, B), dim3(256), 0, CDF_S, qkv, ld, dout, lddo, (const float*)nullptr, 0, ctx_part, (float*)nullptr, n, rps, HD);
    CDF_LAUNCH(linattn_dctx_final_kernel, dim3(heads, B), dim3(256), 0, CDF_S, (const float*)ctx_part, ns, ctx, scale, dctx, rvec, HD);
    const size_t lds = ((size_t)2 * heads * LA_D * LA_D + 3 * HD) * sizeof(float);
    const int rows_per_block = 256 / (heads * 8);
    int gx = cdf_cdiv(n, rows_per_block);
    if (gx > 1024) gx = 1024;
    CDF_LAUNCH(linattn_bwd_rows_kernel, dim3(gx, B), dim3(256), lds, CDF_S, qkv, ld, dout, lddo, ctx, (const float*)dctx, kmax, ksum, (const float*)rvec, dqkv, lddq, n, heads, scale);
    return cdf_check_launch("linattn_bwd");
}

extern "C" int cdf_softmax_rows_fwd(const float* s, float* p, long long rows, int n, int ld, float scale, void* stream) {
    CDF_REQUIRE(s && p && rows > 0 && n > 0 && ld >= n, "cdf_softmax_rows_fwd: bad args");
    long long g = (rows + 3) / 4;
    if (g > 4096) g = 4096;
    CDF_LAUNCH(softmax_rows_fwd_kernel, dim3((int)g), dim3(256), 0, CDF_S, s, p, rows, n, ld, scale);
    return cdf_check_launch("softmax_rows_fwd");
}
extern "C" int cdf_softmax_rows_bwd(const float* p, const float* dp, float* ds, long long rows, int n, int ld, float scale, void* stream) {
    CDF_REQUIRE(p && dp && ds && rows > 0 && n > 0 && ld >= n, "cdf_softmax_rows_bwd: bad args");
    long long g = (rows + 3) / 4;
    if (g > 4096) g = 4096;
    CDF_LAUNCH(softmax_rows_bwd_kernel, dim3((int)g), dim3(256), 0, CDF_S, p, dp, ds, rows, n, ld, scale);
    return cdf_check_launch("softmax_rows_bwd");
}

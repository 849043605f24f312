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
#include <atomic>
#include "cdf_common.h"
#include "colddiff.h"

#define LA_D 32

// ---- pass 1: per (b, chunk, channel) max of k ---------------------------------------------------
// grid = (HD/64, nchunk, B); block 256 = 4 row lanes x 64 channels ; part [B][nchunk][HD]
__global__ void linattn_kmax_kernel(const float* qkv, int ld, float* part, int n, int rows_per_chunk, int HD, int koff) {
    __shared__ float red[4][64];
    const int l = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + l, b = blockIdx.z;
    const int r0 = blockIdx.y * rows_per_chunk;
    int r1 = r0 + rows_per_chunk;
    if (r1 > n) r1 = n;
    const float* kp = qkv + (long long)b * n * ld + koff + c;
    float m = -3.0e38f;
    {
        // 8 rows in flight per lane (one load per loop trip is a chain of memory round trips)
        float t[8];
        int r = r0 + rl;
        for (; r + 28 < r1; r += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = kp[(long long)(r + 4 * u) * ld];
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3])), fmaxf(fmaxf(t[4], t[5]), fmaxf(t[6], t[7]))));
        }
        for (; r < r1; r += 4) m = fmaxf(m, kp[(long long)r * ld]);
    }
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
#pragma unroll 8
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
    // each wave takes row pairs r0 + 2*(wave + 4*j); four pairs per trip with all 8 loads in flight, unconditional
    // (clamped row + select: a load behind a divergent branch waits for everything before it)
    for (int r = r0 + 2 * wave; r < r1; r += 32) {
        float av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = r + 8 * u + hh;
            const int rc = row < r1 ? row : r1 - 1;
            const float ta = A[(long long)rc * lda], tb = Bp[(long long)rc * ldb];
            av[u] = row < r1 ? (EXPA ? expf(ta - mx) : ta) : 0.f;
            bv[u] = row < r1 ? tb : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            psum += av[u];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        }
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

// ---- one-pass context partials (round 2) -----------------------------------------------------------
// The two passes above read k twice (column max, then exp(k - max)^T v with 4-byte loads per lane).  Here a block owns 256 rows of
// one (image, head): the k and v tiles [256][32] come in ONCE as float4 rows (eight loads in flight per thread and tensor), the
// column max is the TILE's own (m_loc), P~ = exp(k - m_loc) and v go to LDS ([row][32] unpadded: the MFMA fragment reads put rows
// 2s / 2s+1 on banks 0-31 / 32-63) and the 32 x 32 context partial runs on the fp32 matrix cores from there.  The finalize kernel
// rescales every partial by exp(m_loc - max_splits m_loc) -- the usual online-softmax identity; kmax / ksum come out as before.
// grid = (heads, ceil(n / 256), B), block 256.
__global__ void __launch_bounds__(256) linattn_ctx1p_kernel(const float* kv, int ld, int koff, float* max_part, float* ctx_part,
                                                           float* sum_part, int n, int HD) {
    CDF_DYN_SMEM(smem_raw);
    float* sp = (float*)smem_raw;                 // [256][32] P~
    float* sv = sp + 256 * LA_D;                  // [256][32] v
    float* red = sv + 256 * LA_D;                 // [32 row groups][32] column reductions
    float* scol = red + 32 * LA_D;                // [32] the tile's column maxima
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, split = blockIdx.y, b = blockIdx.z, heads = gridDim.x, nsplit = gridDim.y;
    const int rg = tid >> 3, c4 = (tid & 7) * 4;
    const int r0 = split * 256;
    const float* kb = kv + (size_t)b * n * ld + koff + h * LA_D + c4;
    const float* vb = kb + HD;
    float4 kq[8], vq[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = r0 + rg + 32 * q;
        const int rc = r < n ? r : n - 1;
        kq[q] = *(const float4*)(kb + (size_t)rc * ld);
        vq[q] = *(const float4*)(vb + (size_t)rc * ld);
    }
    float4 m4 = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (r0 + rg + 32 * q < n) {
            m4.x = fmaxf(m4.x, kq[q].x); m4.y = fmaxf(m4.y, kq[q].y); m4.z = fmaxf(m4.z, kq[q].z); m4.w = fmaxf(m4.w, kq[q].w);
        }
    *(float4*)(red + rg * LA_D + c4) = m4;
    __syncthreads();
    if (tid < LA_D) {
        float m = -3.0e38f;
#pragma unroll 8
        for (int g = 0; g < 32; ++g) m = fmaxf(m, red[g * LA_D + tid]);
        scol[tid] = m;
    }
    __syncthreads();
    const float4 mx = *(const float4*)(scol + c4);
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const bool ok = r0 + rg + 32 * q < n;
        float4 p = make_float4(expf(kq[q].x - mx.x), expf(kq[q].y - mx.y), expf(kq[q].z - mx.z), expf(kq[q].w - mx.w));
        if (!ok) { p = make_float4(0.f, 0.f, 0.f, 0.f); vq[q] = p; }
        s4.x += p.x; s4.y += p.y; s4.z += p.z; s4.w += p.w;
        *(float4*)(sp + (rg + 32 * q) * LA_D + c4) = p;
        *(float4*)(sv + (rg + 32 * q) * LA_D + c4) = vq[q];
    }
    *(float4*)(red + rg * LA_D + c4) = s4;
    __syncthreads();
    const size_t pb = ((size_t)b * nsplit + split);
    if (tid < LA_D) {
        float sum = 0.f;
#pragma unroll 8
        for (int g = 0; g < 32; ++g) sum += red[g * LA_D + tid];
        sum_part[pb * HD + h * LA_D + tid] = sum;
        max_part[pb * HD + h * LA_D + tid] = scol[tid];
    }
    // ctx_part[d][e] = sum_rows P~[row][d] v[row][e]: wave w takes rows 64 w .. 64 w + 63
    const int i = lane & 31, hh = lane >> 5;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* ap = sp + (wave * 64 + hh) * LA_D + i;
    const float* bp = sv + (wave * 64 + hh) * LA_D + i;
#pragma unroll 8
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s * LA_D], bp[2 * s * LA_D], acc, 0, 0, 0);
    __syncthreads();                              // every wave is done reading sp / sv (and the sums in red)
    float* rw = sp + wave * (LA_D * LA_D);            // (the four accumulator tiles reuse the P~ tile)
#pragma unroll
    for (int r = 0; r < 16; ++r) rw[((r & 3) + 8 * (r >> 2) + 4 * hh) * LA_D + i] = acc[r];
    __syncthreads();
    float* dst = ctx_part + (pb * heads + h) * (LA_D * LA_D);
    for (int k = tid; k < LA_D * LA_D; k += 256)
        dst[k] = (sp[k] + sp[LA_D * LA_D + k]) + (sp[2 * LA_D * LA_D + k] + sp[3 * LA_D * LA_D + k]);
}

// finalize of the one-pass form: m = max_s m_s; w_s = exp(m_s - m); ksum = sum_s w_s sum_s; ctx = sum_s w_s ctx_s / ksum
// grid = (heads, B), block 1024 = one thread per context element; the statistics phase runs as 32 channels x 32 split lanes.
__global__ void __launch_bounds__(1024) linattn_ctx1p_final_kernel(const float* ctx_part, const float* sum_part, const float* max_part,
                                                                   int nsplit, int HD, float* ctx, float* ctxs, float scale, float* kmax,
                                                                   float* ksum) {
    CDF_DYN_SMEM(wsm_raw);
    float* wsm = (float*)wsm_raw;                 // [nsplit][32] weights
    __shared__ float sred[32][33];
    __shared__ float smx[LA_D], sinv[LA_D];
    const int h = blockIdx.x, b = blockIdx.y, heads = gridDim.x;
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = h * LA_D + cl;
    const float* mp = max_part + (size_t)b * nsplit * HD + c;
    const float* sp = sum_part + (size_t)b * nsplit * HD + c;
    float m = -3.0e38f;
    for (int k = sl; k < nsplit; k += 32) m = fmaxf(m, mp[(size_t)k * HD]);
    sred[sl][cl] = m;
    __syncthreads();
    if (threadIdx.x < LA_D) {
        float mm = -3.0e38f;
#pragma unroll 8
        for (int g = 0; g < 32; ++g) mm = fmaxf(mm, sred[g][threadIdx.x]);
        smx[threadIdx.x] = mm;
    }
    __syncthreads();
    m = smx[cl];
    float s = 0.f;
    for (int k = sl; k < nsplit; k += 32) {
        const float w = expf(mp[(size_t)k * HD] - m);
        wsm[k * LA_D + cl] = w;
        s += w * sp[(size_t)k * HD];
    }
    sred[sl][cl] = s;
    __syncthreads();
    if (threadIdx.x < LA_D) {
        float t = 0.f;
#pragma unroll 8
        for (int g = 0; g < 32; ++g) t += sred[g][threadIdx.x];
        sinv[threadIdx.x] = 1.0f / t;
        ksum[(size_t)b * HD + h * LA_D + threadIdx.x] = t;
        kmax[(size_t)b * HD + h * LA_D + threadIdx.x] = smx[threadIdx.x];
    }
    __syncthreads();
    const int k = threadIdx.x, d = k >> 5;        // element [d][e] of the 32 x 32 context
    const float* cp = ctx_part + (((size_t)b * nsplit) * heads + h) * (LA_D * LA_D) + k;
    const size_t cstride = (size_t)heads * LA_D * LA_D;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int q = 0;
    for (; q + 3 < nsplit; q += 4) {
        const float v0 = cp[(size_t)q * cstride], v1 = cp[(size_t)(q + 1) * cstride], v2 = cp[(size_t)(q + 2) * cstride], v3 = cp[(size_t)(q + 3) * cstride];
        a0 += wsm[q * LA_D + d] * v0;
        a1 += wsm[(q + 1) * LA_D + d] * v1;
        a2 += wsm[(q + 2) * LA_D + d] * v2;
        a3 += wsm[(q + 3) * LA_D + d] * v3;
    }
    for (; q < nsplit; ++q) a0 += wsm[q * LA_D + d] * cp[(size_t)q * cstride];
    const float r = ((a0 + a1) + (a2 + a3)) * sinv[d];
    ctx[(((size_t)b * heads + h) * LA_D) * LA_D + k] = r;
    ctxs[(((size_t)b * heads + h) * LA_D) * LA_D + k] = r * scale;
}

// ---- finalize (forward): kmax, ksum, ctx = sum_split part / ksum ---------------------------------
// grid = (heads, B), block 256
__global__ void linattn_ctx_final_kernel(const float* ctx_part, const float* sum_part, const float* kmax_part,
                                         int nsplit, int nchunk_max, int HD, float* ctx, float* ctxs, float scale,
                                         float* kmax, float* ksum) {
    __shared__ float ssum[LA_D];
    const int h = blockIdx.x, b = blockIdx.y, heads = gridDim.x;
    if (threadIdx.x < LA_D) {
        const int c = h * LA_D + threadIdx.x;
        float s = 0.f;
#pragma unroll 8
        for (int k = 0; k < nsplit; ++k) s += sum_part[((long long)b * nsplit + k) * HD + c];
        float m = -3.0e38f;
#pragma unroll 8
        for (int k = 0; k < nchunk_max; ++k) m = fmaxf(m, kmax_part[((long long)b * nchunk_max + k) * HD + c]);
        ssum[threadIdx.x] = s;
        ksum[(long long)b * HD + c] = s;
        kmax[(long long)b * HD + c] = m;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < LA_D * LA_D; k += blockDim.x) {
        float s = 0.f;
#pragma unroll 8
        for (int sp = 0; sp < nsplit; ++sp) s += ctx_part[((((long long)b * nsplit + sp) * heads + h) * LA_D) * LA_D + k];
        const float c = s / ssum[k / LA_D];
        ctx[(((long long)b * heads + h) * LA_D) * LA_D + k] = c;
        ctxs[(((long long)b * heads + h) * LA_D) * LA_D + k] = c * scale;   // q *= scale folded into the context
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
#pragma unroll 8
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

// ---- P[n][c] = exp(k[n][c] - kmax[c]) / ksum[c]  (the softmax over n, materialised for backward) --------
__global__ void linattn_softk_kernel(const float* qkv, int ld, const float* kmax, const float* ksum, float* pn, int ldp,
                                     int B, int n, int HD) {
    const int c4n = HD / 4;
    const long long total = (long long)B * n * c4n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const long long row = i / c4n;
        const int b = (int)(row / n);
        const float4 kv = *(const float4*)(qkv + row * ld + HD + c);
        const float4 mx = *(const float4*)(kmax + (long long)b * HD + c), sm = *(const float4*)(ksum + (long long)b * HD + c);
        *(float4*)(pn + row * ldp + c) = make_float4(expf(kv.x - mx.x) / sm.x, expf(kv.y - mx.y) / sm.y, expf(kv.z - mx.z) / sm.z,
                                                     expf(kv.w - mx.w) / sm.w);
    }
}
// ---- dk[n][c] = P[n][c] * (dP[n][c] - r[b][c]) ---------------------------------------------------------
__global__ void linattn_dk_kernel(const float* pn, int ldp, const float* dp, int lddp, const float* rvec, float* dk, int lddk,
                                  int B, int n, int HD) {
    const int c4n = HD / 4;
    const long long total = (long long)B * n * c4n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const long long row = i / c4n;
        const int b = (int)(row / n);
        const float4 p = *(const float4*)(pn + row * ldp + c), d = *(const float4*)(dp + row * lddp + c);
        const float4 r = *(const float4*)(rvec + (long long)b * HD + c);
        *(float4*)(dk + row * lddk + c) = make_float4(p.x * (d.x - r.x), p.y * (d.y - r.y), p.z * (d.z - r.z), p.w * (d.w - r.w));
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
static void la_final_attr() {
#ifndef CDF_EMU
    static CdfDeviceLatch done;
    if (done.first()) {          // nsplit x 128 B of rescaling weights next to ~4 KB of static LDS: past the 64 KB default from ~480 partials up
        (void)hipFuncSetAttribute((const void*)linattn_ctx1p_final_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);   // (+ ~4 KB static: the sum must stay under the 160 KB of a CU)
    }
#endif
}

extern "C" int cdf_linattn_nsplit(int n) {       // 256 rows per split (the one-pass context kernel holds a split's k, v tile in LDS)
    int s = (n + 255) / 256;
    if (s < 1) s = 1;
    return s;
}

// ws >= B*nsplit*HD (kmax partials) + B*nsplit*heads*1024 (ctx partials) + B*nsplit*HD (sum partials) floats
extern "C" size_t cdf_linattn_ws_floats(int B, int n, int heads) {
    const size_t ns = (size_t)cdf_linattn_nsplit(n), HD = (size_t)heads * LA_D;
    return (size_t)B * ns * (2 * HD + (size_t)heads * LA_D * LA_D);
}

// Context pass: ctx[b,h,d,e] = sum_n softmax_n(k)[d,n] v[e,n]; ctxs = scale*ctx; kmax/ksum [B,HD] saved.
// The output  out[n, h*32+e] = sum_d q[n, h*32+d] ctxs[h,d,e]  is a K=32 GEMM per (b, head): cdf_conv_gemm.
extern "C" int cdf_linattn_context(const float* qkv, int ld, int koff, float* ctx, float* ctxs, float* kmax, float* ksum, float* ws,
                                   int B, int n, int heads, float scale, int onepass, void* stream) {
    CDF_REQUIRE(qkv && ctx && ctxs && kmax && ksum && ws, "cdf_linattn_context: null pointer");
    const int HD = heads * LA_D;
    CDF_REQUIRE(HD % 64 == 0 && ld % 4 == 0 && koff >= 0 && koff % 4 == 0 && ld >= koff + 2 * HD,
                "cdf_linattn_context: heads=%d unsupported / bad pitch or k offset", heads);
    const int ns = cdf_linattn_nsplit(n), rps = cdf_cdiv(cdf_cdiv(n, ns), 2) * 2;
    float* kmax_part = ws;
    float* ctx_part = kmax_part + (size_t)B * ns * HD;
    float* sum_part = ctx_part + (size_t)B * ns * heads * LA_D * LA_D;
    if (onepass && ns <= 1024) {
        const size_t lds = ((size_t)2 * 256 * LA_D + 32 * LA_D + LA_D) * sizeof(float);
#ifndef CDF_EMU
        static CdfDeviceLatch attr_done;
        if (attr_done.first()) {
            (void)hipFuncSetAttribute((const void*)linattn_ctx1p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
#endif
        CDF_LAUNCH(linattn_ctx1p_kernel, dim3(heads, ns, B), dim3(256), lds, CDF_S, qkv, ld, koff, kmax_part, ctx_part, sum_part, n, HD);
        la_final_attr();
        CDF_LAUNCH(linattn_ctx1p_final_kernel, dim3(heads, B), dim3(1024), (size_t)ns * LA_D * sizeof(float), CDF_S, (const float*)ctx_part,
                   (const float*)sum_part, (const float*)kmax_part, ns, HD, ctx, ctxs, scale, kmax, ksum);
        return cdf_check_launch("linattn_context");
    }
    CDF_LAUNCH(linattn_kmax_kernel, dim3(HD / 64, ns, B), dim3(256), 0, CDF_S, qkv, ld, kmax_part, n, rps, HD, koff);
    CDF_LAUNCH((linattn_ctx_kernel<true>), dim3(heads, ns, B), dim3(256), 0, CDF_S, qkv + koff, ld, qkv + koff + HD, ld, (const float*)kmax_part, ns, ctx_part, sum_part, n, rps, HD);
    CDF_LAUNCH(linattn_ctx_final_kernel, dim3(heads, B), dim3(256), 0, CDF_S, (const float*)ctx_part, (const float*)sum_part, (const float*)kmax_part, ns, ns, HD, ctx, ctxs, scale, kmax, ksum);
    return cdf_check_launch("linattn_context");
}

// Fold of nparts context partials per image and head (layout of cdf_linattn_context's workspace with nsplit = nparts:
// max [B][nparts][HD] | ctx [B][nparts][heads][32][32] | sum [B][nparts][HD]) as written by cdf_linattn_kvctx (k_conv_sp.hip).
extern "C" int cdf_linattn_finalize(const float* ws, int nparts, float* ctx, float* ctxs, float* kmax, float* ksum, int B, int heads,
                                    float scale, void* stream) {
    CDF_REQUIRE(ws && ctx && ctxs && kmax && ksum && nparts >= 1 && nparts <= 1024 && B > 0 && heads >= 1, "cdf_linattn_finalize: bad args");
    const int HD = heads * LA_D;
    const float* max_part = ws;
    const float* ctx_part = max_part + (size_t)B * nparts * HD;
    const float* sum_part = ctx_part + (size_t)B * nparts * heads * LA_D * LA_D;
    la_final_attr();
    CDF_LAUNCH(linattn_ctx1p_final_kernel, dim3(heads, B), dim3(1024), (size_t)nparts * LA_D * sizeof(float), CDF_S, ctx_part, sum_part, max_part,
               nparts, HD, ctx, ctxs, scale, kmax, ksum);
    return cdf_check_launch("linattn_finalize");
}

// Backward context pass: dctx[b,h,d,e] = scale * sum_n q[n,d] dout[n,e];  rvec[b, h*32+d] = sum_e dctx*ctx.
extern "C" int cdf_linattn_dcontext(const float* qkv, int ld, const float* dout, int lddo, const float* ctx, float* dctx,
                                    float* rvec, float* ws, int B, int n, int heads, float scale, void* stream) {
    CDF_REQUIRE(qkv && dout && ctx && dctx && rvec && ws, "cdf_linattn_dcontext: null pointer");
    const int HD = heads * LA_D;
    CDF_REQUIRE(HD % 64 == 0 && ld % 4 == 0 && lddo % 4 == 0, "cdf_linattn_dcontext: bad pitch");
    const int ns = cdf_linattn_nsplit(n), rps = cdf_cdiv(cdf_cdiv(n, ns), 2) * 2;
    float* ctx_part = ws + (size_t)B * ns * HD;
    CDF_LAUNCH((linattn_ctx_kernel<false>), dim3(heads, ns, B), dim3(256), 0, CDF_S, qkv, ld, dout, lddo, (const float*)nullptr, 0, ctx_part, (float*)nullptr, n, rps, HD);
    CDF_LAUNCH(linattn_dctx_final_kernel, dim3(heads, B), dim3(256), 0, CDF_S, (const float*)ctx_part, ns, ctx, scale, dctx, rvec, HD);
    return cdf_check_launch("linattn_dcontext");
}

static inline int la_grid(long long n) {
    long long g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    return g < 1 ? 1 : (int)g;
}

extern "C" int cdf_linattn_softk(const float* qkv, int ld, const float* kmax, const float* ksum, float* pn, int ldp, int B,
                                 int n, int heads, void* stream) {
    CDF_REQUIRE(qkv && kmax && ksum && pn && ld % 4 == 0 && ldp % 4 == 0, "cdf_linattn_softk: bad args");
    const int HD = heads * LA_D;
    CDF_LAUNCH(linattn_softk_kernel, dim3(la_grid((long long)B * n * HD / 4)), dim3(256), 0, CDF_S, qkv, ld, kmax, ksum, pn, ldp, B, n, HD);
    return cdf_check_launch("linattn_softk");
}

extern "C" int cdf_linattn_dk(const float* pn, int ldp, const float* dp, int lddp, const float* rvec, float* dk, int lddk,
                              int B, int n, int heads, void* stream) {
    CDF_REQUIRE(pn && dp && rvec && dk && ldp % 4 == 0 && lddp % 4 == 0 && lddk % 4 == 0, "cdf_linattn_dk: bad args");
    const int HD = heads * LA_D;
    CDF_LAUNCH(linattn_dk_kernel, dim3(la_grid((long long)B * n * HD / 4)), dim3(256), 0, CDF_S, pn, ldp, dp, lddp, rvec, dk, lddk, B, n, HD);
    return cdf_check_launch("linattn_dk");
}

// ---- fused k / v backward -------------------------------------------------------------------------------------------------
// Given dctx [B,heads,32,32] (gradient of the softmax-weighted context) and rvec, per pixel n and head h:
//   P[d]  = exp(k[n,d] - kmax[d]) / ksum[d]                        (softmax over n, recomputed)
//   dP[d] = sum_e v[n,e] dctx[d,e]          dk[n,d] = P[d] (dP[d] - rvec[d])
//   dv[e] = sum_d P[d] dctx[d,e]
// Upstream of this kernel these were four launches (cdf_linattn_softk, two K = 32 GEMMs, cdf_linattn_dk) that wrote and re-read
// three [B,n,HD] intermediates (P, dP, and P again): 4.6 KB of HBM traffic per pixel against the 2 KB this kernel needs (read k, v;
// write dk, dv).  One wave = one head: a 32-pixel tile of k and v goes global -> registers (float4, lanes along channels: full
// 128-byte rows) -> LDS, the two 32 x 32 x 32 products run on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, A fragments from
// the LDS tiles with lanes along pixels, the dctx fragments live in registers for the whole tile loop), and the results go
// back through an LDS tile so that they leave as float4 rows.  grid = (ceil(n / (32 TILES)), B), block = 64 heads threads.
#define LA_TP 36                                             // LDS row pitch (floats): 16-byte aligned rows
#define LA_TILES 8                                           // 32-pixel tiles per wave

// PL: dk | dv leave as bf16 hi / lo planes (pitch ldpl, channel offset dkoff) instead of fp32 -- the operand form of the two GEMMs that
// consume them (the k | v projection's data and weight gradients), same bytes, no split pass and no in-kernel split downstream.
template <bool PL>
__global__ void __launch_bounds__(256) linattn_bwd_kv_kernel(const float* qkv, int ld, const float* dctx, const float* rvec, const float* kmax,
                                                            const float* ksum, float* dqkv, int lddq, int n, int heads, int koff, int dkoff,
                                                            unsigned short* pl_hi, unsigned short* pl_lo, int ldpl) {
    CDF_DYN_SMEM(smem_raw);
    const int lane = threadIdx.x & 63, h = threadIdx.x >> 6, b = blockIdx.y;
    const int HD = heads * LA_D;
    float* sp = (float*)smem_raw + (size_t)h * 3 * LA_D * LA_TP;          // this wave's P tile
    float* sv = sp + LA_D * LA_TP;                                         // V tile
    float* so = sv + LA_D * LA_TP;                                         // output staging
    const int i = lane & 31, hh = lane >> 5;                               // MFMA lane geometry
    const int lr = lane >> 3, lc = (lane & 7) * 4;                          // load / store geometry: pixel row lr + 8 q, channels lc .. lc+3
    // dctx fragments: B1[s] = dctx[d = i][e = 2s + hh] (dP = V dctx^T), B2[s] = dctx[d = 2s + hh][e = i] (dv = P dctx)
    const float* dc = dctx + ((size_t)b * heads + h) * LA_D * LA_D;
    float B1[16], B2[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        B1[s] = dc[i * LA_D + 2 * s + hh];
        B2[s] = dc[(2 * s + hh) * LA_D + i];
    }
    const float rv = rvec[(size_t)b * HD + h * LA_D + i];
    const float4 km = *(const float4*)(kmax + (size_t)b * HD + h * LA_D + lc);
    const float4 ks = *(const float4*)(ksum + (size_t)b * HD + h * LA_D + lc);
    const float4 ri = make_float4(1.0f / ks.x, 1.0f / ks.y, 1.0f / ks.z, 1.0f / ks.w);
    const float* kbase = qkv + (size_t)b * n * ld + koff + h * LA_D + lc;
    const float* vbase = kbase + HD;
    float* dkbase = PL ? nullptr : dqkv + (size_t)b * n * lddq + dkoff + h * LA_D + lc;
    float* dvbase = PL ? nullptr : dkbase + HD;
    const size_t plk = (size_t)b * n * ldpl + dkoff + h * LA_D + lc, plv = plk + HD;        // (PL) element offsets of dk / dv in the planes
    auto put = [&](float* base, size_t ploff, int p, const float4& v) {
        if constexpr (PL) {
            const float vv[4] = {v.x, v.y, v.z, v.w};
            cdf_split_store4(pl_hi + ploff + (size_t)p * ldpl, pl_lo ? pl_lo + ploff + (size_t)p * ldpl : nullptr, vv);
        } else {
            *(float4*)(base + (size_t)p * lddq) = v;
        }
    };
    const int p_begin = blockIdx.x * (LA_D * LA_TILES);
    for (int t = 0; t < LA_TILES; ++t) {
        const int p0 = p_begin + t * LA_D;
        if (p0 >= n) break;                                                // (block-uniform)
        // ---- k, v tile -> P, V in LDS (rows past n: zeros)
        float4 kq[4], vq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = p0 + lr + 8 * q;
            const int pc = p < n ? p : n - 1;
            kq[q] = *(const float4*)(kbase + (size_t)pc * ld);
            vq[q] = *(const float4*)(vbase + (size_t)pc * ld);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool ok = p0 + lr + 8 * q < n;
            float4 pn = make_float4(expf(kq[q].x - km.x) * ri.x, expf(kq[q].y - km.y) * ri.y, expf(kq[q].z - km.z) * ri.z, expf(kq[q].w - km.w) * ri.w);
            if (!ok) { pn = make_float4(0.f, 0.f, 0.f, 0.f); vq[q] = pn; }
            *(float4*)(sp + (lr + 8 * q) * LA_TP + lc) = pn;
            *(float4*)(sv + (lr + 8 * q) * LA_TP + lc) = vq[q];
        }
        CDF_WAVE_SYNC();                                   // (one wave owns these tiles: LDS operations of a wave are in order)
        // ---- dP = V dctx^T ; dk = P (dP - rvec)
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[i * LA_TP + 2 * s + hh], B1[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int px = (r & 3) + 8 * (r >> 2) + 4 * hh;                // accumulator row of this lane's column d = i
            so[px * LA_TP + i] = sp[px * LA_TP + i] * (acc[r] - rv);
        }
        CDF_WAVE_SYNC();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = p0 + lr + 8 * q;
            if (p < n) put(dkbase, plk, p, *(const float4*)(so + (lr + 8 * q) * LA_TP + lc));
        }
        CDF_WAVE_SYNC();
        // ---- dv = P dctx
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sp[i * LA_TP + 2 * s + hh], B2[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) so[((r & 3) + 8 * (r >> 2) + 4 * hh) * LA_TP + i] = acc[r];
        CDF_WAVE_SYNC();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = p0 + lr + 8 * q;
            if (p < n) put(dvbase, plv, p, *(const float4*)(so + (lr + 8 * q) * LA_TP + lc));
        }
        CDF_WAVE_SYNC();                                   // the tiles are rewritten by the next trip
    }
}

extern "C" int cdf_linattn_bwd_kv(const float* qkv, int ld, int koff, const float* dctx, const float* rvec, const float* kmax, const float* ksum,
                                  float* dqkv, int lddq, int dkoff, int B, int n, int heads, void* stream) {
    CDF_REQUIRE(qkv && dctx && rvec && kmax && ksum && dqkv && B > 0 && n > 0, "cdf_linattn_bwd_kv: null pointer");
    CDF_REQUIRE(heads >= 1 && heads <= 4 && ld % 4 == 0 && lddq % 4 == 0 && koff >= 0 && dkoff >= 0 && koff % 4 == 0 && dkoff % 4 == 0 &&
                ld >= koff + 2 * heads * LA_D && lddq >= dkoff + 2 * heads * LA_D &&
                ((((uintptr_t)qkv) | ((uintptr_t)dqkv) | ((uintptr_t)kmax) | ((uintptr_t)ksum)) & 15) == 0,
                "cdf_linattn_bwd_kv: up to 4 heads; pitches / channel offsets must be multiples of 4 and hold k | v, pointers 16-byte aligned");
    const size_t lds = (size_t)heads * 3 * LA_D * LA_TP * sizeof(float);
    CDF_LAUNCH(linattn_bwd_kv_kernel<false>, dim3(cdf_cdiv(n, LA_D * LA_TILES), B), dim3(64 * heads), lds, CDF_S, qkv, ld, dctx, rvec, kmax, ksum, dqkv, lddq, n,
               heads, koff, dkoff, (unsigned short*)nullptr, (unsigned short*)nullptr, 0);
    return cdf_check_launch("linattn_bwd_kv");
}

// ... with dk | dv written as bf16 hi / lo planes [B n][ldpl] (lo nullable: single-plane bf16 mode) at channel offset dkoff
extern "C" int cdf_linattn_bwd_kv_planes(const float* qkv, int ld, int koff, const float* dctx, const float* rvec, const float* kmax, const float* ksum,
                                         void* dkv_hi, void* dkv_lo, int ldpl, int dkoff, int B, int n, int heads, void* stream) {
    CDF_REQUIRE(qkv && dctx && rvec && kmax && ksum && dkv_hi && B > 0 && n > 0, "cdf_linattn_bwd_kv_planes: null pointer");
    CDF_REQUIRE(heads >= 1 && heads <= 4 && ld % 4 == 0 && ldpl % 8 == 0 && koff >= 0 && dkoff >= 0 && koff % 4 == 0 && dkoff % 8 == 0 &&
                ld >= koff + 2 * heads * LA_D && ldpl >= dkoff + 2 * heads * LA_D &&
                ((((uintptr_t)qkv) | ((uintptr_t)dkv_hi) | ((uintptr_t)dkv_lo) | ((uintptr_t)kmax) | ((uintptr_t)ksum)) & 15) == 0,
                "cdf_linattn_bwd_kv_planes: up to 4 heads; the planes' pitch / channel offset must be multiples of 8 and hold dk | dv, pointers 16-byte aligned");
    const size_t lds = (size_t)heads * 3 * LA_D * LA_TP * sizeof(float);
    CDF_LAUNCH(linattn_bwd_kv_kernel<true>, dim3(cdf_cdiv(n, LA_D * LA_TILES), B), dim3(64 * heads), lds, CDF_S, qkv, ld, dctx, rvec, kmax, ksum, (float*)nullptr, 0, n,
               heads, koff, dkoff, (unsigned short*)dkv_hi, (unsigned short*)dkv_lo, ldpl);
    return cdf_check_launch("linattn_bwd_kv_planes");
}

// dctx = scale * raw ; rvec[row] = sum_e dctx[row][e] * ctx[row][e]  (rows of LA_D = 32 entries; one 32-lane group per row)
__global__ void linattn_dctx_finish_kernel(const float* raw, const float* ctx, float* dctx, float* rvec, long long rows, float scale) {
    const long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int e = threadIdx.x & 31;
    float p = 0.f;
    if (row < rows) {
        const float d = scale * raw[row * LA_D + e];
        dctx[row * LA_D + e] = d;
        p = d * ctx[row * LA_D + e];
    }
    p = cdf_group_sum(p, 32);
    if (row < rows && e == 0) rvec[row] = p;
}

extern "C" int cdf_linattn_dctx_finish(const float* raw, const float* ctx, float* dctx, float* rvec, long long rows, float scale, void* stream) {
    CDF_REQUIRE(raw && ctx && dctx && rvec && rows > 0, "cdf_linattn_dctx_finish: bad args");
    CDF_LAUNCH(linattn_dctx_finish_kernel, dim3(cdf_cdiv(rows * 32, 256)), dim3(256), 0, CDF_S, raw, ctx, dctx, rvec, rows, scale);
    return cdf_check_launch("linattn_dctx_finish");
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

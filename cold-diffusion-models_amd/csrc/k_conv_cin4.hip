// k_conv_cin4.hip — direct convolution for layers with at most 4 input channels (the image-side convs of the
// UNets: ConvNeXt block 0 conv1 3->128 3x3 and res_conv 3->64 1x1, DEBLUR:145-165 with dim = channels).
//
// As GEMMs these are K = taps*Cin <= 36 against M = B*H*W = 524 288 pixels: the 128-row MFMA tiles spend their time
// on padding, and the weight-gradient kernel re-reads the 128-channel gradient once per tap (2.4 GB).  The work is
// only 27 FMAs per output element, so it runs on the vector ALUs at the speed of the one big tensor of each pass:
//   forward   : x [M][4] (L1 broadcast) -> y / pre / bf16 planes [M][Cout], written once
//   data grad : dy [M][Cout] read per tap (L2), dx [M][4] written
//   weight grad: dy read ONCE, 27 x 4 register accumulators per lane, per-chunk partials + unpack_reduce
// Layout: NHWC, input pitch exactly 4 floats (channel 3 is padding and must be zero or ignored: its weights are zero),
// stride 1, odd k, "same" padding.  Weights arrive packed [taps][4][Cout] fp32 (cdf_pack_weight: T=taps, R=Cin<=4 ...).
// All arithmetic fp32 FMA.
#include "cdf_common.h"
#include "colddiff.h"

#define C4_MAX_TAPS 9

struct Cin4Args {
    const float* x;       // [B][H][W][4]
    const float* w;       // [taps][4][ldw]  (rows c >= Cin are zero)
    const float* bias;    // [Cout] nullable
    float* y;             // nullable (planes only)
    float* pre;           // nullable: pre-activation
    unsigned short* ys_hi;
    unsigned short* ys_lo;
    int ldw, ldy, ldp, ld_ys;
    int B, H, W, Cout, k, act;
};

// grid-stride over groups of PX consecutive pixels of an image row; block 256 = (256 / LP) groups x LP lanes, lane = 4 output channels.
// (hipcc keeps the 4 T weight vectors of a lane in registers across the loop.  PX = 4 -- one K x (PX + K - 1) input window per group --
// was measured at 128 x 128: 0.200 ms either way, the 144 FMAs + GELU + operand split per pixel and lane are the bound, and it needs every
// VGPR; the launcher uses PX = 1.)
template <int K, int PX>
__global__ void __launch_bounds__(256) conv_cin4_fwd_kernel(Cin4Args a) {
    constexpr int T = K * K, h = K / 2, WC = PX + K - 1;
    CDF_DYN_SMEM(smem);
    float4* wl = (float4*)smem;                    // [T][4][LP]
    const int LP = a.Cout / 4, PPB = 256 / LP;
    for (int i = threadIdx.x; i < T * 4 * LP; i += 256) {
        const int l = i % LP, tc = i / LP;
        wl[i] = *(const float4*)(a.w + (long long)tc * a.ldw + l * 4);
    }
    __syncthreads();
    const int l = threadIdx.x % LP, pl = threadIdx.x / LP;
    if (pl >= PPB) return;
    const int n = l * 4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) bv = *(const float4*)(a.bias + n);
    const int MG = a.B * a.H * a.W / PX;                     // (< 2^31 pixels: 32-bit index math; 64-bit div/mod is ~100 instructions)
    for (int g = blockIdx.x * PPB + pl; g < MG; g += gridDim.x * PPB) {
        const int m = g * PX;
        const int px = m % a.W, py = (m / a.W) % a.H;
        float4 acc[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) acc[p] = bv;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int iy = py + ky - h;
            const bool rok = iy >= 0 && iy < a.H;
            float4 xw[WC];
#pragma unroll
            for (int c = 0; c < WC; ++c) {
                const int ix = px + c - h;
                const bool ok = rok && ix >= 0 && ix < a.W;
                // pointer select against the zero page, no value select: "ok ? load : 0" becomes a branch around the load
                xw[c] = *(const float4*)(ok ? a.x + (long long)(m + (ky - h) * a.W + (c - h)) * 4 : cdf_zero_page);
            }
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int t = ky * K + kx;
                const float4 w0 = wl[(t * 4 + 0) * LP + l], w1 = wl[(t * 4 + 1) * LP + l], w2 = wl[(t * 4 + 2) * LP + l], w3 = wl[(t * 4 + 3) * LP + l];
#pragma unroll
                for (int p = 0; p < PX; ++p) {
                    const float4 xv = xw[p + kx];
                    acc[p].x = fmaf(xv.x, w0.x, acc[p].x); acc[p].y = fmaf(xv.x, w0.y, acc[p].y); acc[p].z = fmaf(xv.x, w0.z, acc[p].z); acc[p].w = fmaf(xv.x, w0.w, acc[p].w);
                    acc[p].x = fmaf(xv.y, w1.x, acc[p].x); acc[p].y = fmaf(xv.y, w1.y, acc[p].y); acc[p].z = fmaf(xv.y, w1.z, acc[p].z); acc[p].w = fmaf(xv.y, w1.w, acc[p].w);
                    acc[p].x = fmaf(xv.z, w2.x, acc[p].x); acc[p].y = fmaf(xv.z, w2.y, acc[p].y); acc[p].z = fmaf(xv.z, w2.z, acc[p].z); acc[p].w = fmaf(xv.z, w2.w, acc[p].w);
                    acc[p].x = fmaf(xv.w, w3.x, acc[p].x); acc[p].y = fmaf(xv.w, w3.y, acc[p].y); acc[p].z = fmaf(xv.w, w3.z, acc[p].z); acc[p].w = fmaf(xv.w, w3.w, acc[p].w);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const long long mp = m + p;
            if (a.pre) *(float4*)(a.pre + mp * a.ldp + n) = acc[p];
            float v[4] = {acc[p].x, acc[p].y, acc[p].z, acc[p].w};
            if (a.act == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = cdf_gelu(v[e]);
            } else if (a.act == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = cdf_silu(v[e]);
            }
            if (a.y) *(float4*)(a.y + mp * a.ldy + n) = make_float4(v[0], v[1], v[2], v[3]);
            if (a.ys_hi) cdf_split_store4(a.ys_hi + mp * a.ld_ys + n, a.ys_lo ? a.ys_lo + mp * a.ld_ys + n : nullptr, v);
        }
    }
}

// dx[m][c] = sum_t sum_n dy[m - off(t)][n] * w[t][c][n]   (transposed taps);  block = (256/LP) pixels x LP lanes
template <int K>
__global__ void __launch_bounds__(256) conv_cin4_dgrad_kernel(const float* dy, int ldd, const float* w, int ldw, float* dx, int B, int H,
                                                              int W, int Cout, int accumulate) {
    constexpr int T = K * K, h = K / 2;
    CDF_DYN_SMEM(smem);
    float4* wl = (float4*)smem;                    // [T][4][LP]
    const int LP = Cout / 4, PPB = 256 / LP;
    for (int i = threadIdx.x; i < T * 4 * LP; i += 256) {
        const int l = i % LP, tc = i / LP;
        wl[i] = *(const float4*)(w + (long long)tc * ldw + l * 4);
    }
    __syncthreads();
    const int l = threadIdx.x % LP, pl = threadIdx.x / LP;
    const int n = l * 4;
    const int M = B * H * W, mstep = gridDim.x * PPB;
    // every lane of a pixel group runs the same trip count (the cross-lane sums below need all LP lanes)
    for (int m0 = blockIdx.x * PPB; m0 < M; m0 += mstep) {
        const int m = m0 + pl;
        const bool live = pl < PPB && m < M;
        const int mc = live ? m : 0;
        const int px = mc % W, py = (mc / W) % H;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        float4 dv[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            // forward tap t reads x[m + off(t)]; its gradient flows from dy[m - off(t)]
            const int oy = t / K - h, ox = t % K - h;
            const int iy = py - oy, ix = px - ox;
            const bool ok = live && iy >= 0 && iy < H && ix >= 0 && ix < W;
            dv[t] = *(const float4*)(ok ? dy + (long long)(mc - oy * W - ox) * ldd + n : cdf_zero_page);
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float4 w0 = wl[(t * 4 + 0) * LP + l], w1 = wl[(t * 4 + 1) * LP + l], w2 = wl[(t * 4 + 2) * LP + l], w3 = wl[(t * 4 + 3) * LP + l];
            s0 += (dv[t].x * w0.x + dv[t].y * w0.y) + (dv[t].z * w0.z + dv[t].w * w0.w);
            s1 += (dv[t].x * w1.x + dv[t].y * w1.y) + (dv[t].z * w1.z + dv[t].w * w1.w);
            s2 += (dv[t].x * w2.x + dv[t].y * w2.y) + (dv[t].z * w2.z + dv[t].w * w2.w);
            s3 += (dv[t].x * w3.x + dv[t].y * w3.y) + (dv[t].z * w3.z + dv[t].w * w3.w);
        }
        s0 = cdf_group_sum(s0, LP); s1 = cdf_group_sum(s1, LP); s2 = cdf_group_sum(s2, LP); s3 = cdf_group_sum(s3, LP);
        if (live && l == 0) {
            float4 o = make_float4(s0, s1, s2, s3);
            float* dst = dx + (long long)m * 4;
            if (accumulate) {
                const float4 old = *(const float4*)dst;
                o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            }
            *(float4*)dst = o;
        }
    }
}

// partial[chunk][t*Cin + c][n] = sum_{m in chunk} x[m + off(t)][c] * dy[m][n] (c < Cin);  bsum[chunk][n] = sum dy[m][n]
// grid = nchunk; block = (256/LP) pixel-group lanes x LP channel lanes; 4*T float4 accumulators per thread.  Groups of PX consecutive
// pixels of a row per trip (PX = 4 when W % 4 == 0): one pixel decode and one K x (PX + K - 1) window per group instead of per pixel
// (the integer divisions and the nine border selects outnumbered the 144 FMAs of a pixel).
template <int K, int PX>
__global__ void __launch_bounds__(256) conv_cin4_wgrad_kernel(const float* x, const float* dy, int ldd, float* part, float* bsum, int B, int H,
                                                              int W, int Cin, int Cout, long long m_per_chunk) {
    constexpr int T = K * K, h = K / 2, WC = PX + K - 1;
    CDF_DYN_SMEM(smem);
    float4* red = (float4*)smem;                   // [PPB][LP] reused per accumulator row
    const int LP = Cout / 4, PPB = 256 / LP;
    const int l = threadIdx.x % LP, pl = threadIdx.x / LP;
    const int n = l * 4;
    const int M = B * H * W;
    const int m_lo = (int)(blockIdx.x * m_per_chunk);        // (a multiple of PX)
    int m_hi = m_lo + (int)m_per_chunk;
    if (m_hi > M) m_hi = M;
    float4 acc[T][4], bs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pl < PPB) {
        for (int m = m_lo + pl * PX; m < m_hi; m += PPB * PX) {
            const int px = m % W, py = (m / W) % H;
            float4 d[PX];
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                d[p] = *(const float4*)(dy + (long long)(m + p) * ldd + n);
                bs.x += d[p].x; bs.y += d[p].y; bs.z += d[p].z; bs.w += d[p].w;
            }
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int iy = py + ky - h;
                const bool rok = iy >= 0 && iy < H;
                float4 xw[WC];
#pragma unroll
                for (int c = 0; c < WC; ++c) {
                    const int ix = px + c - h;
                    const bool ok = rok && ix >= 0 && ix < W;
                    xw[c] = *(const float4*)(ok ? x + (long long)(m + (ky - h) * W + (c - h)) * 4 : cdf_zero_page);
                }
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int t = ky * K + kx;
#pragma unroll
                    for (int p = 0; p < PX; ++p) {
                        const float4 xv = xw[p + kx], dd = d[p];
                        acc[t][0].x = fmaf(xv.x, dd.x, acc[t][0].x); acc[t][0].y = fmaf(xv.x, dd.y, acc[t][0].y); acc[t][0].z = fmaf(xv.x, dd.z, acc[t][0].z); acc[t][0].w = fmaf(xv.x, dd.w, acc[t][0].w);
                        acc[t][1].x = fmaf(xv.y, dd.x, acc[t][1].x); acc[t][1].y = fmaf(xv.y, dd.y, acc[t][1].y); acc[t][1].z = fmaf(xv.y, dd.z, acc[t][1].z); acc[t][1].w = fmaf(xv.y, dd.w, acc[t][1].w);
                        acc[t][2].x = fmaf(xv.z, dd.x, acc[t][2].x); acc[t][2].y = fmaf(xv.z, dd.y, acc[t][2].y); acc[t][2].z = fmaf(xv.z, dd.z, acc[t][2].z); acc[t][2].w = fmaf(xv.z, dd.w, acc[t][2].w);
                        acc[t][3].x = fmaf(xv.w, dd.x, acc[t][3].x); acc[t][3].y = fmaf(xv.w, dd.y, acc[t][3].y); acc[t][3].z = fmaf(xv.w, dd.z, acc[t][3].z); acc[t][3].w = fmaf(xv.w, dd.w, acc[t][3].w);
                    }
                }
            }
        }
    }
    // fold the pixel lanes through LDS, one accumulator row at a time (static indices only: acc[] stays in registers)
    float* prow = part + (long long)blockIdx.x * T * Cin * Cout;
    auto fold = [&](const float4& v, float* dst) {
        __syncthreads();
        if (pl < PPB) red[pl * LP + l] = v;
        __syncthreads();
        if (pl == 0) {
            float4 s = red[l];
            for (int p = 1; p < PPB; ++p) {
                const float4 r = red[p * LP + l];
                s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
            }
            *(float4*)(dst + n) = s;
        }
    };
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < Cin) fold(acc[t][c], prow + (long long)(t * Cin + c) * Cout);       // (block-uniform)
    if (bsum) fold(bs, bsum + (long long)blockIdx.x * Cout);
}

// dst[t][c][n] = c < Cin ? w[(n*Cin + c)*KK + t] : 0   (w in the PyTorch layout [Cout][Cin][k][k]); n >= Cout zero
__global__ void pack_cin4_kernel(const float* w, float* dst, int ldw, int Cout, int Cin, int KK) {
    const int n_all = KK * 4 * ldw;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_all; i += gridDim.x * blockDim.x) {
        const int n = i % ldw, tc = i / ldw, c = tc & 3, t = tc >> 2;
        dst[i] = (c < Cin && n < Cout) ? w[((long long)n * Cin + c) * KK + t] : 0.f;
    }
}

extern "C" int cdf_pack_cin4(const float* w, float* dst, int ldw, int Cout, int Cin, int k, void* stream) {
    CDF_REQUIRE(w && dst && Cin >= 1 && Cin <= 4 && ldw >= Cout && (k == 1 || k == 3), "cdf_pack_cin4: bad args");
    const int n = k * k * 4 * ldw;
    CDF_LAUNCH(pack_cin4_kernel, dim3(cdf_cdiv(n, 256)), dim3(256), 0, CDF_S, w, dst, ldw, Cout, Cin, k * k);
    return cdf_check_launch("pack_cin4");
}

// ================================================================================================
template <int K>
static int launch_cin4_fwd(const Cin4Args& a, hipStream_t s) {
    const int LP = a.Cout / 4, PPB = 256 / LP;
    const long long MG = (long long)a.B * a.H * a.W;
    long long grid = (MG + PPB - 1) / PPB;
    if (grid > 4096) grid = 4096;
    const size_t lds = (size_t)K * K * 4 * LP * sizeof(float4);
    CDF_LAUNCH((conv_cin4_fwd_kernel<K, 1>), dim3((unsigned)grid), dim3(256), lds, s, a);
    return cdf_check_launch("conv_cin4_fwd");
}

static int cin4_check(const char* who, int Cout, int k, int ldw) {
    CDF_REQUIRE(Cout % 4 == 0 && Cout >= 4 && Cout <= 256 && 256 % (Cout / 4) == 0, "%s: Cout must be 4 * (a power of two <= 64), got %d", who, Cout);
    CDF_REQUIRE(k == 1 || k == 3, "%s: kernel size 1 or 3", who);
    CDF_REQUIRE(ldw % 4 == 0 && ldw >= Cout, "%s: bad weight pitch", who);
    return CDF_OK;
}

extern "C" int cdf_conv_cin4_fwd(const float* x, const float* w, int ldw, const float* bias, float* y, int ldy, float* pre, int ldp,
                                 void* y_hi, void* y_lo, int ld_ys, int B, int H, int W, int Cout, int k, int act, void* stream) {
    CDF_REQUIRE(x && w && (y || y_hi), "cdf_conv_cin4_fwd: null pointer");
    int rc = cin4_check("cdf_conv_cin4_fwd", Cout, k, ldw);
    if (rc) return rc;
    CDF_REQUIRE((!y || ldy % 4 == 0) && (!pre || ldp % 4 == 0) && (!y_hi || ld_ys % 4 == 0), "cdf_conv_cin4_fwd: pitches must be multiples of 4");
    Cin4Args a{x, w, bias, y, pre, (unsigned short*)y_hi, (unsigned short*)y_lo, ldw, ldy, ldp, ld_ys, B, H, W, Cout, k, act};
    return k == 1 ? launch_cin4_fwd<1>(a, CDF_S) : launch_cin4_fwd<3>(a, CDF_S);
}

extern "C" int cdf_conv_cin4_dgrad(const float* dy, int ldd, const float* w, int ldw, float* dx, int B, int H, int W, int Cout, int k,
                                   int accumulate, void* stream) {
    CDF_REQUIRE(dy && w && dx && ldd % 4 == 0 && ldd >= Cout, "cdf_conv_cin4_dgrad: bad args");
    int rc = cin4_check("cdf_conv_cin4_dgrad", Cout, k, ldw);
    if (rc) return rc;
    const int LP = Cout / 4, PPB = 256 / LP;
    const long long M = (long long)B * H * W;
    long long grid = (M + PPB - 1) / PPB;
    if (grid > 4096) grid = 4096;
    const size_t lds = (size_t)k * k * 4 * LP * sizeof(float4);
    if (k == 1) CDF_LAUNCH((conv_cin4_dgrad_kernel<1>), dim3((unsigned)grid), dim3(256), lds, CDF_S, dy, ldd, w, ldw, dx, B, H, W, Cout, accumulate);
    else CDF_LAUNCH((conv_cin4_dgrad_kernel<3>), dim3((unsigned)grid), dim3(256), lds, CDF_S, dy, ldd, w, ldw, dx, B, H, W, Cout, accumulate);
    return cdf_check_launch("conv_cin4_dgrad");
}

// ---- data gradient of a 3 x 3 conv with <= 4 input channels, second stage --------------------------------------------------
// dx[p][c] = sum_{ky,kx} sum_co dy[p - (ky-1, kx-1)][co] W[co][c][ky][kx].  As an implicit GEMM that is K = 9 Cout deep with 3 useful
// output columns (0.4 ms at 128 x 128 x 128 channels: the 128-wide dy rows are gathered nine times).  Instead the sum over co runs
// FIRST, per pixel, for all 27 (c, ky, kx) at once: z[q][c*9 + ky*3 + kx] = sum_co dy[q][co] W[co][c][ky][kx] is a plain 1 x 1 GEMM
// whose weight matrix [Cout][Cin 9] IS the parameter in its PyTorch layout (dy read once), and this kernel adds the nine shifted
// 3-vectors: dx[p][c] = sum_{ky,kx} z[p - (ky-1, kx-1)][c*9 + ky*3 + kx] (zero outside the image).  z is 28 floats per pixel.
// One block = a 16 x 16 pixel tile of one image: the 18 x 18 halo of z rows (ldz floats each, contiguous per image row) comes in as
// float4s, the nine shifted 3-vectors are gathered from LDS (per-lane 4-byte global gathers 112 B apart took 0.21 ms; this 0.02).
#define TS_T 16
__global__ void __launch_bounds__(256) tapsum3_kernel(const float* z, int ldz, float* dx, int H, int W, int Cin, int accumulate, int tiles_x,
                                                      int tiles_y) {
    CDF_DYN_SMEM(smem_raw);
    float* sz = (float*)smem_raw;                         // [(TS_T+2)^2][ldz]
    const int tile = blockIdx.x, tx = tile % tiles_x, t2 = tile / tiles_x, ty = t2 % tiles_y, b = t2 / tiles_y;
    const int y0 = ty * TS_T - 1, x0 = tx * TS_T - 1;     // halo origin
    const int v4 = ldz / 4, nvec = (TS_T + 2) * (TS_T + 2) * v4;
    const float* zb = z + (size_t)b * H * W * ldz;
    for (int i = threadIdx.x; i < nvec; i += 256) {
        const int p = i / v4, k = i - p * v4;
        const int py = y0 + p / (TS_T + 2), px = x0 + p % (TS_T + 2);
        const bool ok = py >= 0 && py < H && px >= 0 && px < W;
        const float4 v = *(const float4*)(zb + ((size_t)(ok ? py : 0) * W + (ok ? px : 0)) * ldz + k * 4);
        *(float4*)(sz + (size_t)p * ldz + k * 4) = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int ly = threadIdx.x / TS_T, lx = threadIdx.x % TS_T;
    const int y = ty * TS_T + ly, x = tx * TS_T + lx;
    if (y >= H || x >= W) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            // source pixel (y - (ky-1), x - (kx-1)) = halo cell (ly + 2 - ky, lx + 2 - kx)
            const float* zp = sz + (size_t)((ly + 2 - ky) * (TS_T + 2) + (lx + 2 - kx)) * ldz + ky * 3 + kx;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < Cin) acc[c] += zp[c * 9];
        }
    float4* o = (float4*)(dx + (((size_t)b * H + y) * W + x) * 4);
    float4 r = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (accumulate) { const float4 old = *o; r.x += old.x; r.y += old.y; r.z += old.z; r.w += old.w; }
    *o = r;
}

extern "C" int cdf_conv_cin4_tapsum3(const float* z, int ldz, float* dx, int B, int H, int W, int Cin, int accumulate, void* stream) {
    CDF_REQUIRE(z && dx && B > 0 && H > 0 && W > 0 && Cin >= 1 && Cin <= 4 && ldz >= 9 * Cin && ldz % 4 == 0 && ldz <= 64 &&
                ((((uintptr_t)dx) | ((uintptr_t)z)) & 15) == 0,
                "cdf_conv_cin4_tapsum3: 1..4 input channels, z rows hold 9 Cin values with a pitch % 4 == 0, dx is [B,H,W,4], both 16-byte aligned");
    const int tiles_x = cdf_cdiv(W, TS_T), tiles_y = cdf_cdiv(H, TS_T);
    const size_t lds = (size_t)(TS_T + 2) * (TS_T + 2) * ldz * sizeof(float);
    CDF_LAUNCH(tapsum3_kernel, dim3((unsigned)(tiles_x * tiles_y * B)), dim3(256), lds, CDF_S, z, ldz, dx, H, W, Cin, accumulate, tiles_x, tiles_y);
    return cdf_check_launch("conv_cin4_tapsum3");
}

extern "C" int cdf_conv_cin4_nchunk(long long M) {
    long long n = M / 512;
    if (n < 1) n = 1;
    if (n > 1024) n = 1024;
    return (int)n;
}

// part: nchunk * taps*Cin * Cout floats, bsum (nullable): nchunk * Cout floats; reduce with cdf_unpack_reduce
extern "C" int cdf_conv_cin4_wgrad(const float* x, const float* dy, int ldd, float* part, float* bsum, int B, int H, int W, int Cin, int Cout,
                                   int k, void* stream) {
    CDF_REQUIRE(x && dy && part && ldd % 4 == 0 && ldd >= Cout && Cin >= 1 && Cin <= 4, "cdf_conv_cin4_wgrad: bad args");
    int rc = cin4_check("cdf_conv_cin4_wgrad", Cout, k, Cout);
    if (rc) return rc;
    const long long M = (long long)B * H * W;
    const int nchunk = cdf_conv_cin4_nchunk(M);
    const int PX = W % 4 == 0 ? 4 : 1;
    const long long mpc = ((M + nchunk - 1) / nchunk + PX - 1) / PX * PX;                  // chunk edges on group edges
    const size_t lds = (size_t)256 * sizeof(float4);
    if (k == 1 && PX == 4) CDF_LAUNCH((conv_cin4_wgrad_kernel<1, 4>), dim3(nchunk), dim3(256), lds, CDF_S, x, dy, ldd, part, bsum, B, H, W, Cin, Cout, mpc);
    else if (k == 1) CDF_LAUNCH((conv_cin4_wgrad_kernel<1, 1>), dim3(nchunk), dim3(256), lds, CDF_S, x, dy, ldd, part, bsum, B, H, W, Cin, Cout, mpc);
    else if (PX == 4) CDF_LAUNCH((conv_cin4_wgrad_kernel<3, 4>), dim3(nchunk), dim3(256), lds, CDF_S, x, dy, ldd, part, bsum, B, H, W, Cin, Cout, mpc);
    else CDF_LAUNCH((conv_cin4_wgrad_kernel<3, 1>), dim3(nchunk), dim3(256), lds, CDF_S, x, dy, ldd, part, bsum, B, H, W, Cin, Cout, mpc);
    return cdf_check_launch("conv_cin4_wgrad");
}

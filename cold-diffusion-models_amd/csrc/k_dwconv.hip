// k_dwconv.hip — depthwise 7x7 convolution of the ConvNeXt block on NHWC feature maps.
//   reference: nn.Conv2d(dim, dim, 7, padding=3, groups=dim) + time-embedding bias
//   deblurring_diffusion_pytorch.py:145,157-162
// HBM-bound (49 MAC per element): each thread produces a 4-pixel x 4-channel strip so that one
// 10x7 window of float4 loads feeds 196 float4 FMAs; lanes run along channels (coalesced 16 B).
// Weights are packed [49][Cp] (cdf_pack_weight with R=1), Cp = C rounded up to 4, zero padded.
#include <type_traits>
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
//
// Per layer: 8 B/element of HBM traffic against 49 FMA/element -- on MI355X the two floors are about equal
// (134 MB in + 134 MB out = 54 us, 1.6 G FMA = 42 us at 128x128x64, batch 32), so neither may be wasted, and the
// loads must not be latency-bound (a register-window version with 10 dependent-ish loads per row ran at 4x the
// floor).  Stencil through LDS:
//   block = 256 threads = 32 channels (8 float4 lanes, 128 B coalesced) x 32 thread tiles of 4 x 2 output pixels,
//   i.e. a TBW x TBH = 256-pixel output tile (32 x 8, or 16 x 16 for narrow images);
//   the (TBW+6) x (TBH+6) input halo goes to LDS with ~17 independent, unconditional float4 loads per thread
//   (clamped address + select), 2.1x the tile's own bytes and L2/MALL-resident for the neighbours;
//   each thread then slides a 10-wide register window over 8 halo rows: 80 ds_read_b128 + 49 weight reads per 8
//   outputs.  Row pitch = (TBW+6) pixels + 64 B: the two thread rows inside a 16-lane ds_read_b128 group land 32
//   banks apart (conflict-free).  68 KB + 6 KB of weights: two blocks per CU overlap one's loads with the other's FMAs.
// grid = (tiles_x * tiles_y * B, ceil(C4 / 8)).
// Round 3 (PMC on MI355X, 64 channels at 128 x 128: VALU active 55 % of the SIMD time -- the kernel is VALU-bound, a wave64 VALU
// instruction occupies its SIMD for 4 cycles and only v_pk_fma_f32 reaches the 157 TFLOP/s vector peak):
//   * the halo loads are unconditional from clamped addresses and the out-of-image select happens when the values go to LDS
//     (a select right behind the load parked the wave on s_waitcnt before the other loads were even requested);
//   * the two (halo row, output row) pairs whose kernel row falls outside 0..6 are peeled off instead of multiplying by zero
//     weights behind a per-component select: 448 v_cndmask and 1/8 of the FMAs per thread gone.
// (A block walking several tiles with the next halo requested under the current tile's FMAs was built and measured: slower,
//  218 VGPRs and no gain from the overlap -- the waves do not wait for HBM, they wait for the vector ALU.)
// BF (bf16 activation storage): x and res are bf16 tensors (pitches in bf16 elements): the halo comes in as 8-byte quads and is
// widened when it goes to LDS; YB: y is bf16 too (rounded to nearest even when it leaves) -- YB = false with BF = true is the one place
// where the bf16 stream hands a gradient to fp32 tensors (the data gradient entering the image-side block).  Weights, biases and the
// arithmetic stay fp32.
template <int TBW, int TBH, bool BF, bool YB = BF>
__global__ void __launch_bounds__(256, 2) dwconv7_kernel(const void* x, int ldx, const float* w, int ldw, const float* bias,
                                                         const float* sbias, int ld_sbias, void* y, int ldy, int B, int H,
                                                         int W, int C4, int flip, int accumulate, const void* res, int ldr,
                                                         void* y_hi, void* y_lo, int ld_ys) {
    typedef typename cdf_quad<BF>::raw raw_t;
    typedef typename cdf_quad<BF>::elem elem_t;
    typedef typename cdf_quad<YB>::elem yelem_t;
    constexpr int HW_ = TBW + 6, HH_ = TBH + 6;              // halo extent
    constexpr int RP = HW_ * 8 + 4;                          // row pitch in float4 (pixels x 8 channel quads + 64 B)
    constexpr int TX = TBW / 4, TY = TBH / 2;                // thread tiles
    static_assert(TX * TY == 32, "256 output pixels per block");
    CDF_DYN_SMEM(smem_raw);
    float4* halo = (float4*)smem_raw;                        // [HH_][RP]
    float4* wl = halo + HH_ * RP;                            // [49][8]

    const int tid = threadIdx.x;
    const int tiles_w = (W + TBW - 1) / TBW, tiles_h = (H + TBH - 1) / TBH;
    // XCD-aware tile order: the (TBW+6) x (TBH+6) halo is 2.1x the tile, i.e. half of what a block reads is shared with its
    // neighbours.  In dispatch order the neighbours sit on other XCDs and every L2 fetched the shared rows from HBM again
    // (rocprofv3 FETCH_SIZE: 2.3x the tensor); an XCD now walks a contiguous run of tiles of one channel group (its 64 resident
    // blocks = one 128 x 128 image).
    const int vid = cdf_xcd_order(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    // bf16 tensors: a block's 32 channels are HALF a 128-byte line of a 64-channel pixel, and a half-line read costs the whole line
    // (profiles/round3_fetch_half_lines.md) -- PMC: 3.0x the tensor fetched.  The channel group is therefore the FAST index there: the
    // blocks that read the two halves of a line are neighbours in an XCD's run and the second one finds the line in that XCD's L2.
    int t = BF ? vid / (int)gridDim.y : vid % (int)gridDim.x;
    const int cq0 = (BF ? vid % (int)gridDim.y : vid / (int)gridDim.x) * 8;
    const int bx = t % tiles_w;
    t /= tiles_w;
    const int by = t % tiles_h, b = t / tiles_h;
    const int X0 = bx * TBW, Y0 = by * TBH;

    for (int i = tid; i < DW_TAPS * 8; i += 256) {
        const int tp = i >> 3, l = i & 7;
        const int tap = flip ? DW_TAPS - 1 - tp : tp;
        wl[i] = (cq0 + l) < C4 ? *(const float4*)(w + (long long)tap * ldw + (cq0 + l) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // Index arithmetic of the 17 halo loads without 32-bit multiplies or 64-bit VALU math (round 3: integer multiplies run at a quarter
    // of the FMA rate and were, with the per-load divisions, more vector work than the 784 packed FMAs): element offsets inside one
    // image are 24-bit x 24-bit products (host: H W pitch < 2^30), the image base is a scalar, and (hy, hx) of slot k follow from
    // slot k - 1 by adding 32 pixels.
    const elem_t* xb = (const elem_t*)x + (long long)b * H * W * ldx;
    constexpr int NHALO = HH_ * HW_ * 8, NIT = (NHALO + 255) / 256;
    constexpr int STEP_Y = 32 / HW_, STEP_X = 32 % HW_;      // 32 pixels further in the [HH_][HW_] halo
    const int l_ = tid & 7;
    const unsigned lc4 = (unsigned)((cq0 + l_) * 4);
    const bool lok = (cq0 + l_) < C4;
    // Round 6: the halo comes in through range-checked buffer loads over the image (cdf_buf): an element outside the image (or a channel quad
    // past C4) asks for offset CDF_BUF_OOB and gets zeros without a memory access -- no coordinate clamps, no per-component selects when the
    // values go to LDS, 32-bit offsets against a scalar base (the global-load form spent as many vector instructions on those as on the
    // 784 packed FMAs).
    const cdf_buf xr = cdf_make_buf(xb, (unsigned)H * (unsigned)W * (unsigned)ldx * (unsigned)sizeof(elem_t));
    raw_t hv[NIT];
    {
        int hy = 0, hx = tid >> 3;                           // (tid >> 3 < 32 <= HW_)
        if (hx >= HW_) { hx -= HW_; ++hy; }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {                      // every load requested before anything is used
            const int iy = Y0 + hy - 3, ix = X0 + hx - 3;
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && lok && hy < HH_;
            const unsigned off = (__umul24(__umul24((unsigned)iy, (unsigned)W) + (unsigned)ix, (unsigned)ldx) + lc4) * (unsigned)sizeof(elem_t);
            hv[k] = cdf_buf_ld(xr, ok ? off : CDF_BUF_OOB, (const raw_t*)nullptr);
            hy += STEP_Y;
            hx += STEP_X;
            if (hx >= HW_) { hx -= HW_; ++hy; }
        }
    }
    {
        int hy = 0, hx = tid >> 3;
        if (hx >= HW_) { hx -= HW_; ++hy; }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            if (hy < HH_) halo[__umul24((unsigned)hy, (unsigned)RP) + hx * 8 + l_] = cdf_quad_cvt(hv[k]);
            hy += STEP_Y;
            hx += STEP_X;
            if (hx >= HW_) { hx -= HW_; ++hy; }
        }
    }
    __syncthreads();

    // lanes: channel quad (8) fastest, then thread row (TY), then thread column (TX)
    const int l8 = tid & 7, ty = (tid >> 3) % TY, tx = (tid >> 3) / TY;
    const int cq = cq0 + l8;
    const int x0 = tx * 4, y0 = ty * 2;                      // inside the tile
    float4 acc[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[o][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    // halo row y0 + r feeds output row o through kernel row ky = r - o: r = 0 only o = 0, r = 7 only o = 1, r = 1..6 both.
    // Kernel row ky is therefore used twice, at r = ky (output row 0) and r = ky + 1 (output row 1): it is read from LDS ONCE and
    // kept for the next step (two register sets alternating, the loop walks two halo rows per trip so that no set is ever copied) --
    // 49 weight reads per thread instead of 98, 129 ds_read_b128 against 784 packed FMAs where 178 had the LDS pipe (4 cycles per
    // read, one pipe for the four SIMDs) as busy as the vector ALUs (round 6).
    auto load_w = [&](float4 (&wv)[DW_K], int ky) {
        const float4* wrow = wl + ky * DW_K * 8 + l8;
#pragma unroll
        for (int kx = 0; kx < DW_K; ++kx) wv[kx] = wrow[kx * 8];
    };
    auto row_step = [&](int r, const float4 (&w0)[DW_K], const float4 (&w1)[DW_K], bool do0, bool do1) {   // w0 / w1: kernel rows r / r - 1
        float4 win[10];
        const float4* hrow = halo + (y0 + r) * RP + x0 * 8 + l8;
#pragma unroll
        for (int q = 0; q < 10; ++q) win[q] = hrow[q * 8];
#pragma unroll
        for (int kx = 0; kx < DW_K; ++kx) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (do0) f4_fma(acc[0][j], win[kx + j], w0[kx]);       // (compile-time per call site)
                if (do1) f4_fma(acc[1][j], win[kx + j], w1[kx]);
            }
        }
    };
    float4 wA[DW_K], wB[DW_K];
    load_w(wA, 0);
    row_step(0, wA, wA, true, false);
    // rolled on purpose: unrolled, hipcc hoists all window + weight reads to the top and spills
#pragma unroll 1
    for (int r = 1; r < 7; r += 2) {
        load_w(wB, r);
        row_step(r, wB, wA, true, true);
        load_w(wA, r + 1);
        row_step(r + 1, wA, wB, true, true);
    }
    row_step(7, wA, wA, false, true);

    // outputs (and the fused `+= y`, `+ res` operands) through buffer resources over the image too: a pixel past the image edge or a channel
    // quad past C4 is one select on the offset, the store is dropped by the range check.  Every operand is requested before the first store
    // (stores count in vmcnt on gfx9: a load behind a store waits for its acknowledgement).  Requested AHEAD of the FMAs instead (64 more
    // live registers) the launch itself gains another 3 % and the training step LOSES 1.2 % -- the GEMMs around it slow down by more than
    // this kernel gains (profiles/round6_dwconv_buffer_ab.txt; measured three times in one call, unexplained: not kept)
    typedef typename cdf_quad<YB>::raw yraw_t;
    const int c = (cq < C4 ? cq : 0) * 4;
    const unsigned img_px = (unsigned)H * (unsigned)W;
    const cdf_buf yr = cdf_make_buf((const yelem_t*)y + (long long)b * H * W * ldy, img_px * (unsigned)ldy * (unsigned)sizeof(yelem_t));
    const cdf_buf rr = cdf_make_buf(res ? (const elem_t*)res + (long long)b * H * W * ldr : (const elem_t*)x, res ? img_px * (unsigned)ldr * (unsigned)sizeof(elem_t) : 0u);
    // (fp32 y only) the output also as bf16 hi / lo planes -- the GEMM operand form its consumer would otherwise produce with cdf_split_bf16
    const bool planes = !YB && y_hi != nullptr;
    const cdf_buf hr = cdf_make_buf(planes ? (const unsigned short*)y_hi + (long long)b * H * W * ld_ys : (const unsigned short*)x, planes ? img_px * (unsigned)ld_ys * 2u : 0u);
    const cdf_buf lr = cdf_make_buf(planes ? (const unsigned short*)y_lo + (long long)b * H * W * ld_ys : (const unsigned short*)x, planes ? img_px * (unsigned)ld_ys * 2u : 0u);
    unsigned offy[2][4], offr[2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int oy = Y0 + y0 + o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ox = X0 + x0 + j;
            const bool ok = oy < H && ox < W && cq < C4;
            const unsigned pix = __umul24((unsigned)oy, (unsigned)W) + (unsigned)ox;
            offy[o][j] = ok ? (__umul24(pix, (unsigned)ldy) + (unsigned)c) * (unsigned)sizeof(yelem_t) : CDF_BUF_OOB;
            offr[o][j] = ok ? (__umul24(pix, (unsigned)ldr) + (unsigned)c) * (unsigned)sizeof(elem_t) : CDF_BUF_OOB;
        }
    }
    // (one register array when y and res have the same storage type -- the engine passes `+= y` or `+ res`, never both; if both do
    //  come, y is read after the FMAs instead)
    constexpr bool SAME = sizeof(raw_t) == sizeof(yraw_t);
    raw_t opv[2][4];                                         // res, or (SAME, no res) the old y
    yraw_t oldv[2][4];
    if (res) {
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int j = 0; j < 4; ++j) opv[o][j] = cdf_buf_ld(rr, offr[o][j], (const raw_t*)nullptr);
    } else if (SAME && accumulate) {
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int j = 0; j < 4; ++j) opv[o][j] = cdf_buf_ld(yr, offy[o][j], (const raw_t*)nullptr);
    }
    if constexpr (!SAME) {
        if (accumulate) {
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int j = 0; j < 4; ++j) oldv[o][j] = cdf_buf_ld(yr, offy[o][j], (const yraw_t*)nullptr);
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
    // OLD: 0 none, 1 the old y sits in opv (same storage type, no res), 2 in oldv -- one straight-line copy per case (a run-time choice
    // between the two arrays would put both into scratch memory)
    auto finish = [&](auto old_tag) {
        constexpr int OLD = decltype(old_tag)::value;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4 v = make_float4(acc[o][j].x + add.x, acc[o][j].y + add.y, acc[o][j].z + add.z, acc[o][j].w + add.w);
                if constexpr (OLD == 1) {
                    const float4 old = cdf_quad_cvt(opv[o][j]);
                    v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
                } else if constexpr (OLD == 2) {
                    const float4 old = cdf_quad_cvt(oldv[o][j]);
                    v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
                    if (res) {
                        const float4 rv = cdf_quad_cvt(opv[o][j]);
                        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                    }
                } else if (res) {                            // fused residual (e.g. dx = dy + conv^T(dh))
                    const float4 rv = cdf_quad_cvt(opv[o][j]);
                    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                }
                cdf_buf_st(yr, offy[o][j], cdf_quad_raw(v, (const yraw_t*)nullptr));
                if constexpr (!YB) {
                    if (planes) {
                        uint2 h2, l2;
                        cdf_split4(v.x, v.y, v.z, v.w, h2, l2);
                        const int oy = Y0 + y0 + o, ox = X0 + x0 + j;
                        const unsigned offp = (oy < H && ox < W && cq < C4) ? (__umul24(__umul24((unsigned)oy, (unsigned)W) + (unsigned)ox, (unsigned)ld_ys) + (unsigned)c) * 2u : CDF_BUF_OOB;
                        cdf_buf_st(hr, offp, h2);
                        cdf_buf_st(lr, offp, l2);
                    }
                }
            }
        }
    };
    if (!accumulate) {
        finish(std::integral_constant<int, 0>{});
    } else if (SAME && !res) {
        finish(std::integral_constant<int, 1>{});
    } else {
        if constexpr (SAME) {                                // both operands with one storage type: y is read here, behind the FMAs
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int j = 0; j < 4; ++j) oldv[o][j] = cdf_buf_ld(yr, offy[o][j], (const yraw_t*)nullptr);
        }
        finish(std::integral_constant<int, 2>{});
    }
}

// Weight gradient partials: part[((b*nchunk + chunk)*50 + tap)*C + c], tap 49 = sum of dy (bias gradients).
//   dw[c][ky][kx] = sum_{b,y,x} x[b, y+ky-3, x+kx-3, c] * dy[b, y, x, c]
// Round 3 (rewritten).  The round-2 kernel had every wave fetch its own dy strips and its own x windows straight from global
// memory: 520 float4 load instructions per wave, rocprofv3 FETCH_SIZE 3.6x the two tensors at 128 x 128 x 64, 1.6 TB/s.  Now a block
// owns (image, chunk of rows, 32 channels) and walks it in 32 x 4-pixel tiles: the tile of dy and the (32+6) x (4+6) halo of x go to
// LDS ONCE (coalesced 128-byte pixels, all of a thread's loads in flight together), and thread (channel quad, g) accumulates one
// (kernel row ky, tile row r) pair -- g = 7 r + ky, 28 of the 32 thread groups busy -- over the 32 pixels of that row: 14 + 8 LDS
// reads per 56 float4 FMAs, seven accumulators that stay in registers across every tile of the chunk.  The four tile rows are folded
// through LDS at the end; grid = (ceil(C/32), nchunk, B) in the XCD-aware order (consecutive chunks share 6 of their 10 halo rows).
#define DWG_TW 32
#define DWG_TR 4
template <bool BF>       // BF: x and dy are bf16 tensors (bf16 activation storage); partial sums and everything downstream stay fp32
__global__ void __launch_bounds__(256, 2) dwconv7_wgrad_partial_kernel(const void* x, int ldx, const void* dy, int lddy,
                                                                      float* part, int H, int W, int C, int rows_per_chunk) {
    typedef typename cdf_quad<BF>::raw raw_t;
    constexpr int HWX = DWG_TW + 6, HRX = DWG_TR + 6;        // halo extent of x
    constexpr int RPX = HWX * 8 + 8;                         // row pitches in float4: adjacent rows 32 banks apart (two thread groups of a
    constexpr int RPD = DWG_TW * 8 + 8;                      // 16-lane ds_read_b128 group read adjacent rows)
    __shared__ float4 xs[HRX * RPX];                         // 49.9 KB
    __shared__ float4 ds[DWG_TR * RPD];                      // 16.9 KB   (+ reused for the final fold)
    const int tid = threadIdx.x;
    const int vid = cdf_xcd_order(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
    const int bid_c = vid % (int)gridDim.x, bid_chunk = (vid / (int)gridDim.x) % (int)gridDim.y, b = vid / (int)(gridDim.x * gridDim.y);
    const int C4 = (C + 3) >> 2, cq0 = bid_c * 8;
    const int ya = bid_chunk * rows_per_chunk;
    int yb = ya + rows_per_chunk;
    if (yb > H) yb = H;
    const int l8 = tid & 7, g = tid >> 3;
    const bool busy = g < 7 * DWG_TR;
    const int r = busy ? g / 7 : 0, ky = busy ? g - 7 * r : 0;
    typedef typename cdf_quad<BF>::elem elem_t;
    const elem_t* xb = (const elem_t*)x + (long long)b * H * W * ldx;
    const elem_t* db = (const elem_t*)dy + (long long)b * H * W * lddy;
    const bool qok = (cq0 + l8) < C4;
    const unsigned lc4 = (unsigned)((cq0 + l8) * 4);
    // Round 6: both operands through range-checked buffer resources over the image (cdf_buf, as in dwconv7_kernel): an element outside the
    // image / the chunk / the channel count asks for CDF_BUF_OOB and arrives as zeros -- no clamps, no selects at the LDS store, 32-bit
    // offsets (24 x 24-bit products: the host checks H W pitch < 2^30) instead of 64-bit lane addresses.
    const cdf_buf xr = cdf_make_buf(xb, (unsigned)H * (unsigned)W * (unsigned)ldx * (unsigned)sizeof(elem_t));
    const cdf_buf dr = cdf_make_buf(db, (unsigned)H * (unsigned)W * (unsigned)lddy * (unsigned)sizeof(elem_t));

    float4 acc[DW_K], dsum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < DW_K; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);

    constexpr int NX = HRX * HWX, NXI = (NX * 8 + 255) / 256;          // halo float4 per thread
    constexpr int ND = DWG_TR * DWG_TW, NDI = ND * 8 / 256;
    const int tiles_w = (W + DWG_TW - 1) / DWG_TW;
    for (int y0 = ya; y0 < yb; y0 += DWG_TR) {
        for (int tx = 0; tx < tiles_w; ++tx) {
            const int X0 = tx * DWG_TW;
            raw_t hx[NXI], hd[NDI];
#pragma unroll
            for (int k = 0; k < NXI; ++k) {                  // every load requested before anything is used
                const int p = (tid >> 3) + 32 * k;
                const int hy = p / HWX, hxx = p - hy * HWX;
                const int iy = y0 + hy - 3, ix = X0 + hxx - 3;
                const bool ok = qok && p < NX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const unsigned off = (__umul24(__umul24((unsigned)iy, (unsigned)W) + (unsigned)ix, (unsigned)ldx) + lc4) * (unsigned)sizeof(elem_t);
                hx[k] = cdf_buf_ld(xr, ok ? off : CDF_BUF_OOB, (const raw_t*)nullptr);
            }
#pragma unroll
            for (int k = 0; k < NDI; ++k) {
                const int p = (tid >> 3) + 32 * k;
                const int ty = p / DWG_TW, px = p - ty * DWG_TW;
                const int iy = y0 + ty, ix = X0 + px;
                const bool ok = qok && iy < yb && ix < W;    // rows past the chunk belong to the next block
                const unsigned off = (__umul24(__umul24((unsigned)iy, (unsigned)W) + (unsigned)ix, (unsigned)lddy) + lc4) * (unsigned)sizeof(elem_t);
                hd[k] = cdf_buf_ld(dr, ok ? off : CDF_BUF_OOB, (const raw_t*)nullptr);
            }
#pragma unroll
            for (int k = 0; k < NXI; ++k) {
                const int p = (tid >> 3) + 32 * k;
                const int hy = p / HWX, hxx = p - hy * HWX;
                if (p < NX) xs[hy * RPX + hxx * 8 + l8] = cdf_quad_cvt(hx[k]);
            }
#pragma unroll
            for (int k = 0; k < NDI; ++k) {
                const int p = (tid >> 3) + 32 * k;
                const int ty = p / DWG_TW, px = p - ty * DWG_TW;
                ds[ty * RPD + px * 8 + l8] = cdf_quad_cvt(hd[k]);
            }
            __syncthreads();
            if (busy) {
                const float4* xrow = xs + (r + ky) * RPX + l8;
                const float4* drow = ds + r * RPD + l8;
#pragma unroll 1
                for (int p0 = 0; p0 < DWG_TW; p0 += 8) {     // rolled: 4 strips of 8 pixels (22 LDS reads, 56 float4 FMAs each)
                    float4 d[8], win[14];
#pragma unroll
                    for (int j = 0; j < 8; ++j) d[j] = drow[(p0 + j) * 8];
#pragma unroll
                    for (int q = 0; q < 14; ++q) win[q] = xrow[(p0 + q) * 8];
#pragma unroll
                    for (int kx = 0; kx < DW_K; ++kx)
#pragma unroll
                        for (int j = 0; j < 8; ++j) f4_fma(acc[kx], win[kx + j], d[j]);
                    if (ky == 0) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { dsum.x += d[j].x; dsum.y += d[j].y; dsum.z += d[j].z; dsum.w += d[j].w; }
                    }
                }
            }
            __syncthreads();
        }
    }
    // fold the DWG_TR tile rows: fold[r][ky * 7 + kx | 49][l8] through LDS (xs is free now), then threads (tap, l8) write the 50 x 32 slab
    float4* fold = xs;                                       // [DWG_TR][50][8]
    if (busy) {
#pragma unroll
        for (int kx = 0; kx < DW_K; ++kx) fold[(r * 50 + ky * DW_K + kx) * 8 + l8] = acc[kx];
        if (ky == 0) fold[(r * 50 + DW_TAPS) * 8 + l8] = dsum;
    }
    __syncthreads();
    float* dst = part + (((long long)b * gridDim.y + bid_chunk) * (DW_TAPS + 1)) * C;
    for (int i = tid; i < 50 * 8; i += 256) {
        const int tap = i >> 3, q = i & 7;
        float4 v = fold[(0 * 50 + tap) * 8 + q];
#pragma unroll
        for (int rr = 1; rr < DWG_TR; ++rr) {
            const float4 u = fold[(rr * 50 + tap) * 8 + q];
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        const int c = (cq0 + q) * 4;
        if (cq0 + q >= C4) continue;
        float* pp = dst + (long long)tap * C + c;
        if (c + 3 < C && (C & 3) == 0) {
            *(float4*)pp = v;
        } else {
            if (c + 0 < C) pp[0] = v.x;
            if (c + 1 < C) pp[1] = v.y;
            if (c + 2 < C) pp[2] = v.z;
            if (c + 3 < C) pp[3] = v.w;
        }
    }
}

// Narrow images (W < 32: the 16 x 16 level, where half of a 32-pixel tile would be padding): the round-2 form, straight from global memory.
// Same two floors as the forward kernel (read x and dy once, 49 FMA per element).  grid = (ceil(C/64), nchunk, B),
// block 256 = 4 waves.  Wave w owns kernel rows ky = 2w, 2w+1 (wave 3: ky = 6 and the dy sum): 14 taps x 4 channels
// of accumulators per lane.  Lanes: 16 channel-quads (float4, 256 B coalesced) x 4 strips of 8 pixels in flight;
// per strip a wave loads the dy strip (8 float4) and one 14-wide x window per kernel row, all unconditionally.
// The 4 strip slots are folded with two cross-lane adds at the end.
template <bool BF>
__global__ void __launch_bounds__(256) dwconv7_wgrad_partial_narrow_kernel(const void* x, int ldx, const void* dy, int lddy,
                                                                   float* part, int H, int W, int C, int rows_per_chunk) {
    constexpr int TW = 8;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: branches on it are uniform
    const int l16 = lane & 15, slot = lane >> 4;
    // XCD-aware chunk order: a chunk of rows_per_chunk (= 4 at 128 rows) image rows reads 6 halo rows of x besides its own, shared
    // with the chunks above and below; consecutive chunks of an image now run on one XCD and find them in its L2 (FETCH_SIZE was
    // 2.5x the two tensors).
    const int vid = cdf_xcd_order(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
    const int bid_c = vid % (int)gridDim.x, bid_chunk = (vid / (int)gridDim.x) % (int)gridDim.y, b = vid / (int)(gridDim.x * gridDim.y);
    const int c = bid_c * 64 + l16 * 4;
    const int C4r = (C + 3) & ~3;
    const bool cv = c < C4r;
    const int cc = cv ? c : 0;
    const int y0 = bid_chunk * rows_per_chunk;
    int y1 = y0 + rows_per_chunk;
    if (y1 > H) y1 = H;
    const int strips_w = (W + TW - 1) / TW, nstrips = (y1 - y0) * strips_w;
    const int ky0 = 2 * wave, nky = wave == 3 ? 1 : 2;

    float4 acc[2][DW_K], dsum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < DW_K; ++k) acc[r][k] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int s = slot; s < nstrips; s += 4) {
        const int yy = y0 + s / strips_w, xs = (s % strips_w) * TW;
        float4 d[TW];
        const long long drow = (((long long)b * H + yy) * W) * lddy + cc;
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int ix = xs + j;
            const float4 v = cdf_quad_cvt(cdf_quad_ld<BF>(dy, drow + (long long)(ix < W ? ix : W - 1) * lddy));
            d[j] = (cv && ix < W) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r >= nky) break;                               // wave-uniform
            const int iy = yy + ky0 + r - 3;
            const bool rowok = iy >= 0 && iy < H;
            const long long row = (((long long)b * H + (iy < 0 ? 0 : (iy >= H ? H - 1 : iy))) * W) * ldx + cc;
            float4 win[TW + 6];
#pragma unroll
            for (int q = 0; q < TW + 6; ++q) {
                const int ix = xs + q - 3;
                const float4 v = cdf_quad_cvt(cdf_quad_ld<BF>(x, row + (long long)(ix < 0 ? 0 : (ix >= W ? W - 1 : ix)) * ldx));
                win[q] = (rowok && ix >= 0 && ix < W) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int kx = 0; kx < DW_K; ++kx)
#pragma unroll
                for (int j = 0; j < TW; ++j) f4_fma(acc[r][kx], win[kx + j], d[j]);
        }
#pragma unroll
        for (int j = 0; j < TW; ++j) { dsum.x += d[j].x; dsum.y += d[j].y; dsum.z += d[j].z; dsum.w += d[j].w; }   // (only wave 3's is stored)
    }
    // fold the 4 strip slots (lane bits 4, 5), then slot 0 writes 16 lanes x float4 = 64 channels per tap
    auto fold = [&](float4& v) {
        v.x += __shfl_xor(v.x, 16); v.y += __shfl_xor(v.y, 16); v.z += __shfl_xor(v.z, 16); v.w += __shfl_xor(v.w, 16);
        v.x += __shfl_xor(v.x, 32); v.y += __shfl_xor(v.y, 32); v.z += __shfl_xor(v.z, 32); v.w += __shfl_xor(v.w, 32);
    };
    float* dst = part + (((long long)b * gridDim.y + bid_chunk) * (DW_TAPS + 1)) * C;
    auto put = [&](int tap, const float4& v) {
        if (slot != 0 || !cv) return;
        float* p = dst + (long long)tap * C + c;
        if (c + 3 < C && (C & 3) == 0) {
            *(float4*)p = v;
        } else {
            if (c + 0 < C) p[0] = v.x;
            if (c + 1 < C) p[1] = v.y;
            if (c + 2 < C) p[2] = v.z;
            if (c + 3 < C) p[3] = v.w;
        }
    };
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int kx = 0; kx < DW_K; ++kx) {
            fold(acc[r][kx]);
            if (r < nky) put((ky0 + r) * DW_K + kx, acc[r][kx]);
        }
    fold(dsum);
    if (wave == 3) put(DW_TAPS, dsum);
}

// dw[c*49 + tap] (+)= sum_{b,chunk} part ; dbias[c] (+)= sum part[..][49] ; dsb[b][c] = sum_chunk part[b][..][49]
// grid = (ceil(C/64), 50 taps); block 1024 = 16 partial lanes x 64 channels (the lanes split the b*nchunk partials,
// two independent loads in flight each)
__global__ void __launch_bounds__(1024) dwconv7_wgrad_final_kernel(const float* part, int B, int nchunk, int C, float* dw, float* dbias,
                                                                  float* dsb, int ld_dsb, int accumulate) {
    __shared__ float red[16][64];
    const int l = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + l, t = blockIdx.y;
    const int n = B * nchunk;
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        int i = rl;
        for (; i + 16 < n; i += 32) {
            s0 += part[((long long)i * (DW_TAPS + 1) + t) * C + c];
            s1 += part[((long long)(i + 16) * (DW_TAPS + 1) + t) * C + c];
        }
        if (i < n) s0 += part[((long long)i * (DW_TAPS + 1) + t) * C + c];
    }
    red[rl][l] = s0 + s1;
    __syncthreads();
    if (rl == 0 && c < C) {
        float tot = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) tot += red[r][l];
        if (t < DW_TAPS) {
            float* dst = dw + (long long)c * DW_TAPS + t;
            *dst = accumulate ? *dst + tot : tot;
        } else if (dbias) {
            dbias[c] = accumulate ? dbias[c] + tot : tot;
        }
    }
    if (t == DW_TAPS && dsb) {          // per-sample time-bias gradient: sum over the chunks of each sample
        __syncthreads();
        for (int b = rl; b < B; b += 16) {
            if (c < C) {
                float sb = 0.f;
                for (int k = 0; k < nchunk; ++k) sb += part[(((long long)b * nchunk + k) * (DW_TAPS + 1) + DW_TAPS) * C + c];
                dsb[(long long)b * ld_dsb + c] = sb;
            }
        }
    }
}

template <int TBW, int TBH, bool BF, bool YB = BF>
static int launch_dwconv7(const void* x, int ldx, const float* w, int ldw, const float* bias, const float* sbias, int ld_sbias, void* y,
                          int ldy, int B, int H, int W, int C4, int flip, int accumulate, const void* res, int ldr, hipStream_t s,
                          void* y_hi = nullptr, void* y_lo = nullptr, int ld_ys = 0) {
    constexpr size_t lds = ((size_t)(TBH + 6) * ((TBW + 6) * 8 + 4) + DW_TAPS * 8) * sizeof(float4);
#ifndef CDF_EMU
    static CdfDeviceLatch attr_done;
    if (attr_done.first()) {
        (void)hipFuncSetAttribute((const void*)dwconv7_kernel<TBW, TBH, BF, YB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
#endif
    const long long tiles = (long long)B * cdf_cdiv(H, TBH) * cdf_cdiv(W, TBW);
    CDF_LAUNCH((dwconv7_kernel<TBW, TBH, BF, YB>), dim3((unsigned)tiles, cdf_cdiv(C4, 8)), dim3(256), lds, s, x, ldx, w, ldw, bias, sbias, ld_sbias, y,
               ldy, B, H, W, C4, flip, accumulate, res, ldr, y_hi, y_lo, ld_ys);
    return cdf_check_launch("dwconv7");
}

// ================================================================================================
// io_bf16 != 0: x, y and res are bf16 tensors (pitches in bf16 elements, 8-byte aligned); io_bf16 == 2: x and res bf16, y fp32; w, bias,
// sbias stay fp32
static int dwconv7_entry(const void* x, int ldx, const float* w, int ldw, const float* bias, const float* sbias,
                              int ld_sbias, void* y, int ldy, int B, int H, int W, int C, int flip, int accumulate,
                              const void* res, int ldr, int io_bf16, void* y_hi, void* y_lo, int ld_ys, void* stream) {
    const uintptr_t amask = io_bf16 ? 7 : 15, ymask = io_bf16 == 1 ? 7 : 15;
    CDF_REQUIRE(io_bf16 >= 0 && io_bf16 <= 2, "cdf_dwconv7_io: io_bf16 must be 0 (fp32), 1 (bf16) or 2 (bf16 x / res, fp32 y)");
    CDF_REQUIRE(!res || (ldr % 4 == 0 && (((uintptr_t)res) & amask) == 0), "cdf_dwconv7: residual must be 16B aligned (bf16: 8B) with pitch %% 4 == 0");
    CDF_REQUIRE(x && w && y, "cdf_dwconv7: null pointer");
    CDF_REQUIRE((y_hi != nullptr) == (y_lo != nullptr) && (!y_hi || (io_bf16 == 0 && ld_ys % 4 == 0 && ld_ys >= ((C + 3) & ~3) && ((((uintptr_t)y_hi) | ((uintptr_t)y_lo)) & 7) == 0 && ld_ys < (1 << 24))),
                "cdf_dwconv7_planes: both planes or neither, fp32 y only, 8-byte aligned, pitch %% 4 == 0");
    CDF_REQUIRE((((uintptr_t)x) & amask) == 0 && (((uintptr_t)y) & ymask) == 0, "cdf_dwconv7: x / y must be 16B aligned (bf16: 8B)");
    const int Cp = (C + 3) & ~3;
    CDF_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ldw % 4 == 0 && ldx >= Cp && ldy >= Cp && ldw >= Cp, "cdf_dwconv7: pitches must be multiples of 4 and >= roundup4(C)");
    CDF_REQUIRE(!bias || (C % 4 == 0), "cdf_dwconv7: bias with C %% 4 != 0 needs a padded bias (pass a padded vector and C rounded up)");
    {   // the kernel's per-image element offsets are 24 x 24-bit products kept in 32 bits
        const long long ldmax = ldx > ldy ? (ldx > ldr ? ldx : ldr) : (ldy > ldr ? ldy : ldr);
        CDF_REQUIRE((long long)H * W < (1 << 24) && ldmax < (1 << 24) && (long long)H * W * ldmax < (1LL << 30),
                    "cdf_dwconv7: image of %d x %d pixels at pitch %lld is beyond the kernel's 32-bit per-image offsets", H, W, ldmax);
    }
    if (io_bf16 == 2) {
        if (W <= 16) return launch_dwconv7<16, 16, true, false>(x, ldx, w, ldw, bias, sbias, ld_sbias, y, ldy, B, H, W, Cp / 4, flip, accumulate, res, ldr, CDF_S);
        return launch_dwconv7<32, 8, true, false>(x, ldx, w, ldw, bias, sbias, ld_sbias, y, ldy, B, H, W, Cp / 4, flip, accumulate, res, ldr, CDF_S);
    }
    if (io_bf16) {
        if (W <= 16) return launch_dwconv7<16, 16, true>(x, ldx, w, ldw, bias, sbias, ld_sbias, y, ldy, B, H, W, Cp / 4, flip, accumulate, res, ldr, CDF_S);
        return launch_dwconv7<32, 8, true>(x, ldx, w, ldw, bias, sbias, ld_sbias, y, ldy, B, H, W, Cp / 4, flip, accumulate, res, ldr, CDF_S);
    }
    if (W <= 16) return launch_dwconv7<16, 16, false>(x, ldx, w, ldw, bias, sbias, ld_sbias, y, ldy, B, H, W, Cp / 4, flip, accumulate, res, ldr, CDF_S, y_hi, y_lo, ld_ys);
    return launch_dwconv7<32, 8, false>(x, ldx, w, ldw, bias, sbias, ld_sbias, y, ldy, B, H, W, Cp / 4, flip, accumulate, res, ldr, CDF_S, y_hi, y_lo, ld_ys);
}
extern "C" int cdf_dwconv7_io(const void* x, int ldx, const float* w, int ldw, const float* bias, const float* sbias,
                              int ld_sbias, void* y, int ldy, int B, int H, int W, int C, int flip, int accumulate,
                              const void* res, int ldr, int io_bf16, void* stream) {
    return dwconv7_entry(x, ldx, w, ldw, bias, sbias, ld_sbias, y, ldy, B, H, W, C, flip, accumulate, res, ldr, io_bf16, nullptr, nullptr, 0, stream);
}
// fp32 tensors, and y ALSO as bf16 hi / lo planes (pitch ld_ys): the data-gradient pass hands its result to a GEMM in that form
extern "C" int cdf_dwconv7_planes(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* sbias, int ld_sbias, float* y, int ldy,
                                  int B, int H, int W, int C, int flip, int accumulate, const float* res, int ldr, void* y_hi, void* y_lo, int ld_ys,
                                  void* stream) {
    return dwconv7_entry(x, ldx, w, ldw, bias, sbias, ld_sbias, y, ldy, B, H, W, C, flip, accumulate, res, ldr, 0, y_hi, y_lo, ld_ys, stream);
}

extern "C" int cdf_dwconv7(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* sbias,
                           int ld_sbias, float* y, int ldy, int B, int H, int W, int C, int flip, int accumulate,
                           const float* res, int ldr, void* stream) {
    return cdf_dwconv7_io(x, ldx, w, ldw, bias, sbias, ld_sbias, y, ldy, B, H, W, C, flip, accumulate, res, ldr, 0, stream);
}

extern "C" int cdf_dwconv7_wgrad_nchunk(int H) {
    int n = H / 4;
    if (n < 1) n = 1;
    if (n > 32) n = 32;
    return n;
}

// ws >= B * nchunk * 50 * C floats; dw in the PyTorch layout [C][1][7][7]; dsb [B][ld_dsb] (overwritten)
// io_bf16 != 0: x and dy are bf16 tensors (pitches in bf16 elements, 8-byte aligned)
extern "C" int cdf_dwconv7_wgrad_io(const void* x, int ldx, const void* dy, int lddy, float* dw, float* dbias,
                                    float* dsb, int ld_dsb, float* ws, int B, int H, int W, int C, int accumulate, int io_bf16,
                                    void* stream) {
    CDF_REQUIRE(x && dy && dw && ws, "cdf_dwconv7_wgrad: null pointer");
    CDF_REQUIRE(ldx % 4 == 0 && lddy % 4 == 0 && ldx >= ((C + 3) & ~3) && lddy >= ((C + 3) & ~3) && ((((uintptr_t)x) | ((uintptr_t)dy)) & (io_bf16 ? 7 : 15)) == 0,
                "cdf_dwconv7_wgrad: pitches must be multiples of 4 and >= roundup4(C), pointers 16B aligned (bf16: 8B)");
    {   // the wide kernel's per-image element offsets are 24 x 24-bit products kept in 32 bits
        const long long ldmax = ldx > lddy ? ldx : lddy;
        CDF_REQUIRE(W < 32 || ((long long)H * W < (1 << 24) && ldmax < (1 << 24) && (long long)H * W * ldmax < (1LL << 30)),
                    "cdf_dwconv7_wgrad: image of %d x %d pixels at pitch %lld is beyond the kernel's 32-bit per-image offsets", H, W, ldmax);
    }
    const int nchunk = cdf_dwconv7_wgrad_nchunk(H), rpc = cdf_cdiv(H, nchunk);
    if (W >= 32) {
        if (io_bf16) CDF_LAUNCH(dwconv7_wgrad_partial_kernel<true>, dim3(cdf_cdiv(C, 32), nchunk, B), dim3(256), 0, CDF_S, x, ldx, dy, lddy, ws, H, W, C, rpc);
        else CDF_LAUNCH(dwconv7_wgrad_partial_kernel<false>, dim3(cdf_cdiv(C, 32), nchunk, B), dim3(256), 0, CDF_S, x, ldx, dy, lddy, ws, H, W, C, rpc);
    } else {
        if (io_bf16) CDF_LAUNCH(dwconv7_wgrad_partial_narrow_kernel<true>, dim3(cdf_cdiv(C, 64), nchunk, B), dim3(256), 0, CDF_S, x, ldx, dy, lddy, ws, H, W, C, rpc);
        else CDF_LAUNCH(dwconv7_wgrad_partial_narrow_kernel<false>, dim3(cdf_cdiv(C, 64), nchunk, B), dim3(256), 0, CDF_S, x, ldx, dy, lddy, ws, H, W, C, rpc);
    }
    CDF_LAUNCH(dwconv7_wgrad_final_kernel, dim3(cdf_cdiv(C, 64), DW_TAPS + 1), dim3(1024), 0, CDF_S, (const float*)ws, B, nchunk, C, dw, dbias, dsb, ld_dsb, accumulate);
    return cdf_check_launch("dwconv7_wgrad");
}

extern "C" int cdf_dwconv7_wgrad(const float* x, int ldx, const float* dy, int lddy, float* dw, float* dbias,
                                 float* dsb, int ld_dsb, float* ws, int B, int H, int W, int C, int accumulate,
                                 void* stream) {
    return cdf_dwconv7_wgrad_io(x, ldx, dy, lddy, dw, dbias, dsb, ld_dsb, ws, B, H, W, C, accumulate, 0, stream);
}

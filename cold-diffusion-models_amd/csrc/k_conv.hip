// k_conv.hip — dense convolutions / linear layers of the UNets as implicit GEMMs on the gfx950
// matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD).
//
// Replaces the ATen/MIOpen ops behind (reference file:line)
//   nn.Conv2d 3x3 / 1x1 / 4x4-s2, nn.ConvTranspose2d 4x4-s2, nn.Linear
//     deblurring_diffusion_pytorch.py:105-109,140-154,173-174,211-216,253 ; Model2.py:36-73,85-112,148-163
// and their autograd backward (dgrad = the same gather-GEMM with a mirrored tap table and the
// [tap][Cout][Cin] weight packing; wgrad = pixel-reduction GEMM with split-K).
//
// One gather-GEMM kernel covers every forward and data-gradient case:
//   Y[m, co] = epilogue( sum_{tap, ci} X[pix(m, tap), ci] * Wp[tap][ci][co] )
//   m = (b, qy, qx) over a per-phase output grid; pix = (qy*is + dy[tap], qx*is + dx[tap]),
//   zero outside the image; output pixel (qy*os + phase.oy, qx*os + phase.ox).
//   Regular convs are one phase; stride-2 transposed convs are 4 output-parity phases of 2x2 taps.
//
// Tiling (wave64): 256 threads = 4 waves, block tile BMxBN, wave tile WMxWN built from 32x32
// MFMA tiles, BK = 16 channels of one tap per main-loop step, operands staged through LDS
// (register prefetch of step i+1 while step i is on the matrix pipe, one barrier per step).
// The MFMA k index is free as long as A and B agree, so lane half h owns k = 8h..8h+7 of the
// chunk: its A fragment is two ds_read_b128 (row stride 20 floats -> conflict-free).
#include <atomic>
#include "cdf_common.h"
#include "cdf_epilogue.h"
#include "colddiff.h"

#define CDF_MAX_TAPS 16

struct ConvPhase {
    int oy, ox, ntaps;
    signed char dy[CDF_MAX_TAPS], dx[CDF_MAX_TAPS], wi[CDF_MAX_TAPS];
};

struct ConvArgs {
    const float* x;
    const float* w;
    float* y;
    const float* bias;   // [Cout] nullable
    const float* sbias;  // [B][ld_sbias] per-sample bias (time embedding) nullable
    const float* res;    // residual, added last, nullable
    float* pre;          // nullable: receives the pre-activation value
    const float* mul;    // nullable: epilogue multiplier source (activation-gradient fusion)
    int ldx, ldw, ldy, ld_sbias, ldr, ldp, ldm;
    int B, H, W, Cin, OH, OW, Cout, QH, QW, os, is;
    int act;         // 0 none, 1 GELU, 2 SiLU, 3 ReLU
    int mul_mode;    // 0 none, 1 v*=gelu'(mul), 2 v*=silu'(mul), 3 v*=mul
    int accumulate;  // y += v
    int nphase, vec;
    unsigned short* ys_hi;   // (unused by the fp32 kernels: always null)
    unsigned short* ys_lo;
    int ld_ys;
    int io_bf;               // CDF_IO_*_BF16 bits (cdf_epilogue.h): always 0 for the fp32 entry points
    long long x_bs, w_bs, y_bs;     // blockIdx.z = outer*batch2 + inner: outer batch strides (elements)
    long long x_bs2, w_bs2, y_bs2;  // inner batch strides (e.g. attention heads)
    int epi_follow;                 // batched launch whose y offsets are whole rows (y_bs % ldy == 0, y_bs2 % ldy == 0) and that has epilogue
                                    // operands: pre / mul / res are addressed like y (row offset added to the pixel index), not from row 0
    int batch2;
    ConvPhase ph[4];
};

__device__ __forceinline__ int cdf_xcd_swizzle(int bid, int nblk) {
    // bijective remap so that each XCD (bid % 8 round-robin dispatch) walks a contiguous tile range
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <int BM, int BN, int WM, int WN, bool BT>
__global__ void __launch_bounds__(256, 4) conv_igemm_kernel(ConvArgs a) {
    constexpr int BK = 16, AS = BK + 4;
    constexpr int MT = WM / 32, NT = WN / 32, WAVES_N = BN / WN;
    constexpr int APASS = BM / 64;
    constexpr int BVEC = BK * BN / 4;               // float4 per B tile
    constexpr int BPASS = (BVEC + 255) / 256;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
    constexpr int ASZ = BM * AS, BSZ = BK * BN;
    constexpr int EPI_ROWS = (BM / WM) * 32, EPI_SZ = EPI_ROWS * (BN + 8);      // one 32-row slab per wave row, see epilogue
    constexpr int SMEM = 2 * (ASZ + BSZ) > EPI_SZ ? 2 * (ASZ + BSZ) : EPI_SZ;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float (*As)[ASZ] = (float (*)[ASZ])smem;
    float (*Bs)[BSZ] = (float (*)[BSZ])(smem + 2 * ASZ);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int M = a.B * a.QH * a.QW;
    const int tiles_n = (a.Cout + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const int tile = cdf_xcd_swizzle(blockIdx.x, tiles_m * tiles_n);
    const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
    const ConvPhase& ph = a.ph[blockIdx.y];
    const int zo = blockIdx.z / a.batch2, zi = blockIdx.z - zo * a.batch2;
    const float* X = a.x + (long long)zo * a.x_bs + (long long)zi * a.x_bs2;
    const float* Wt = a.w + (long long)zo * a.w_bs + (long long)zi * a.w_bs2;
    float* Y = a.y + (long long)zo * a.y_bs + (long long)zi * a.y_bs2;
    long long pix_off = 0;
    if (a.epi_follow) {
        pix_off = ((long long)zo * a.y_bs + (long long)zi * a.y_bs2) / a.ldy;
        Y = a.y;
    }

    // ---- A-operand row bookkeeping ----------------------------------------------------------
    const int a_col = (tid & 3) * 4;
    int a_iy0[APASS], a_ix0[APASS];
    unsigned a_pix[APASS];                                   // pixel index of (b, iy0, ix0); the tap adds dy*W + dx
#pragma unroll
    for (int p = 0; p < APASS; ++p) {
        const int m = tile_m * BM + (tid >> 2) + 64 * p;
        if (m < M) {
            const int qx = m % a.QW, t2 = m / a.QW;
            a_iy0[p] = (t2 % a.QH) * a.is;
            a_ix0[p] = qx * a.is;
            a_pix[p] = (unsigned)(((t2 / a.QH) * a.H + a_iy0[p]) * a.W + a_ix0[p]);
        } else {
            a_iy0[p] = -(1 << 28);
            a_ix0[p] = 0;
            a_pix[p] = 0;
        }
    }
    const int nchunks = (a.Cin + BK - 1) / BK;
    const int niter = ph.ntaps * nchunks;

    // Tap table -> LDS once (a dynamic index into the by-value kernel argument compiles to global byte loads in front
    // of every chunk's tile loads); the next entry is fetched when the tap counter advances, an iteration ahead.
    // CDF_MAX_TAPS + 1 entries: the fetch one past the end is harmless.
    __shared__ int tap_lds[CDF_MAX_TAPS + 1];
    if (tid <= CDF_MAX_TAPS)
        tap_lds[tid] = tid < ph.ntaps ? (ph.dy[tid] & 0xFF) | ((ph.dx[tid] & 0xFF) << 8) | ((ph.wi[tid] & 0xFF) << 16) : 0;
    __syncthreads();
    int tap_cur = tap_lds[0];

    // Straight-line prefetch: every load is unconditional -- an element outside the image / channel range reads the
    // library's zero page instead (pointer select, no value select: hipcc turns "ok ? load : 0" back into a branch, and
    // a load inside a divergent branch waits for everything in flight first).
    f32x4_t ra[APASS], rb[BPASS];                            // (arrays of HIP float4 structs would live in scratch)
    int tap = 0, c0 = 0, issued = 0;                         // (tap, channel chunk) of the NEXT load
    auto load_global = [&]() {
        int tc = tap_cur;                                    // stays in a VGPR (see k_conv_sp.hip)
#ifndef CDF_EMU
        asm volatile("" : "+v"(tc));
#endif
        const int dy = (int)(signed char)(tc & 0xFF), dx = (int)(signed char)((tc >> 8) & 0xFF), wi = (tc >> 16) & 0xFF;
        const int tap_pix = dy * a.W + dx;
#pragma unroll
        for (int p = 0; p < APASS; ++p) {
            const unsigned iy = (unsigned)(a_iy0[p] + dy), ix = (unsigned)(a_ix0[p] + dx);
            const bool ok = iy < (unsigned)a.H && ix < (unsigned)a.W && (c0 + a_col) < a.Cin;
            const size_t off = (size_t)(a_pix[p] + (unsigned)tap_pix) * (unsigned)a.ldx + (unsigned)(c0 + a_col);
            ra[p] = *(const f32x4_t*)(ok ? X + off : cdf_zero_page);
        }
#pragma unroll
        for (int p = 0; p < BPASS; ++p) {
            const int idx = tid + 256 * p;
            if (!BT) {
                const int k = idx / (BN / 4), n4 = idx - k * (BN / 4);
                const int ci = c0 + k, co = tile_n * BN + n4 * 4;
                const bool ok = idx < BVEC && ci < a.Cin && co < a.Cout;
                const size_t off = (size_t)((unsigned)wi * (unsigned)a.Cin + (unsigned)ci) * (unsigned)a.ldw + (unsigned)co;
                rb[p] = *(const f32x4_t*)(ok ? Wt + off : cdf_zero_page);
            } else {
                // B given as [N][K] (K contiguous): read 4 consecutive k of one output column
                const int n = idx / (BK / 4), k4 = idx - n * (BK / 4);
                const int ci = c0 + k4 * 4, co = tile_n * BN + n;
                const bool ok = idx < BVEC && ci < a.Cin && co < a.Cout;
                const size_t off = (size_t)(unsigned)co * (unsigned)a.ldw + (unsigned)ci;
                rb[p] = *(const f32x4_t*)(ok ? Wt + off : cdf_zero_page);
            }
        }
        const bool more = issued + 1 < niter;                // block-uniform
        issued += more ? 1 : 0;
        const int c1 = c0 + BK;
        const bool wrap = c1 >= a.Cin;
        c0 = more ? (wrap ? 0 : c1) : c0;
        tap += (more && wrap) ? 1 : 0;
        tap_cur = tap_lds[tap];
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int p = 0; p < APASS; ++p) *(f32x4_t*)(&As[buf][((tid >> 2) + 64 * p) * AS + a_col]) = ra[p];
#pragma unroll
        for (int p = 0; p < BPASS; ++p) {
            const int idx = tid + 256 * p;
            if (idx < BVEC) {
                if (!BT) {
                    const int k = idx / (BN / 4), n4 = idx - k * (BN / 4);
                    *(f32x4_t*)(&Bs[buf][k * BN + n4 * 4]) = rb[p];
                } else {
                    const int n = idx / (BK / 4), k4 = idx - n * (BK / 4);
                    Bs[buf][(k4 * 4 + 0) * BN + n] = rb[p][0];
                    Bs[buf][(k4 * 4 + 1) * BN + n] = rb[p][1];
                    Bs[buf][(k4 * 4 + 2) * BN + n] = rb[p][2];
                    Bs[buf][(k4 * 4 + 3) * BN + n] = rb[p][3];
                }
            }
        }
    };

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int half = lane >> 5, l31 = lane & 31;
    if (niter > 0) {
        load_global();
        store_lds(0);
    }
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const int buf = it & 1;
        load_global();                                       // chunk it + 1 (past the end: the last chunk again, never stored)
        float af[MT][8], bf[NT][8];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const float* p = &As[buf][(wm * WM + i * 32 + l31) * AS + half * 8];
            const float4 v0 = *(const float4*)p, v1 = *(const float4*)(p + 4);
            af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w;
            af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int s = 0; s < 8; ++s) bf[j][s] = Bs[buf][(half * 8 + s) * BN + wn * WN + j * 32 + l31];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        if (it + 1 < niter) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: MT passes of one 32-row slab per wave through LDS, float4 rows out (cdf_epilogue.h) ----
    constexpr int CP = BN + 8;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if (i > 0) __syncthreads();          // previous pass fully read (the K loop itself ends with a barrier)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                smem[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * CP + wn * WN + j * 32 + l31] = acc[i][j][r];
        __syncthreads();
        cdf_epilogue_rows<BN, EPI_ROWS>(a, ph, Y, smem, tile_m * BM, tile_n * BN, M, tid,
                                        [i](int p) { return (p >> 5) * WM + i * 32 + (p & 31); }, pix_off);
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad: out[z][tap][ca][cb] = sum_{m in split z} XA[pixA(m,tap)][ca] * XB[pixB(m,tap)][cb]
// m = (b, qy, qx); pixA = (qy*sa + day, qx*sa + dax) in [HA,WA]; pixB likewise. Rows with an
// out-of-range pixel on either side contribute zero.
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* xa;
    const float* xb;
    float* out;
    int lda, ldb, ldo;
    int B, QH, QW;
    int HA, WA, sa, HB, WB, sb;
    int CA, CB;
    int ntaps;
    int nsplit, m_per_split;
    long long a_bs, b_bs, o_bs;  // batch strides (blockIdx.z / nsplit)
    int nbatch;
    float* bsum;                 // nullable: [nsplit][ldo] column sums of XB (bias gradient), written by tile_a == 0, tap == 0
    signed char day[CDF_MAX_TAPS], dax[CDF_MAX_TAPS], dby[CDF_MAX_TAPS], dbx[CDF_MAX_TAPS];
};

template <int BMC, int BNC, int WM, int WN>
__global__ void __launch_bounds__(256, 4) conv_wgrad_kernel(WgradArgs a) {
    constexpr int BK = 16;
    constexpr int MT = WM / 32, NT = WN / 32, WAVES_N = BNC / WN;
    constexpr int AVEC = BK * BMC / 4, BVEC = BK * BNC / 4;
    constexpr int APASS = (AVEC + 255) / 256, BPASS = (BVEC + 255) / 256;
    static_assert((BMC / WM) * (BNC / WN) == 4, "4 waves per block");
    __shared__ __attribute__((aligned(16))) float Xs[2][BK * BMC];
    __shared__ __attribute__((aligned(16))) float Ys[2][BK * BNC];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tiles_b = (a.CB + BNC - 1) / BNC;
    // XCD-aware block order (see cdf_wgrad_block in k_conv_sp.hip): the tiles and taps of one pixel range share an L2
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const int gx = gridDim.x, gy = gridDim.y;
        const int v = cdf_xcd_swizzle(bx + gx * (by + gy * bz), gx * gy * (int)gridDim.z);
        bx = v % gx;
        const int t2 = v / gx;
        by = t2 % gy;
        bz = t2 / gy;
    }
    const int tile_a = bx / tiles_b, tile_b = bx - tile_a * tiles_b;
    const int tap = by;
    const int batch = bz / a.nsplit, split = bz - batch * a.nsplit;
    const float* XA = a.xa + (long long)batch * a.a_bs;
    const float* XB = a.xb + (long long)batch * a.b_bs;
    const int M = a.B * a.QH * a.QW;
    const int m_lo = split * a.m_per_split;
    int m_hi = m_lo + a.m_per_split;
    if (m_hi > M) m_hi = M;
    const int niter = m_hi > m_lo ? (m_hi - m_lo + BK - 1) / BK : 0;
    const int day = a.day[tap], dax = a.dax[tap], dby = a.dby[tap], dbx = a.dbx[tap];

    // per-thread load slots: the pixel (b, qy, qx) of each slot is decoded once and then advanced
    // incrementally by BK rows per main-loop step (no integer divisions in the loop)
    int a_q[APASS][3], b_q[BPASS][3];   // qx, qy, b
    auto decode = [&](int m, int* q) {
        q[0] = m % a.QW;
        const int t2 = m / a.QW;
        q[1] = t2 % a.QH;
        q[2] = t2 / a.QH;
    };
    auto advance = [&](int* q) {
        q[0] += BK;
        while (q[0] >= a.QW) {
            q[0] -= a.QW;
            if (++q[1] >= a.QH) { q[1] = 0; ++q[2]; }
        }
    };
#pragma unroll
    for (int p = 0; p < APASS; ++p) decode(m_lo + (tid + 256 * p) / (BMC / 4), a_q[p]);
#pragma unroll
    for (int p = 0; p < BPASS; ++p) decode(m_lo + (tid + 256 * p) / (BNC / 4), b_q[p]);
    const bool do_bsum = a.bsum != nullptr && tile_a == 0 && tap == 0;
    float4 bs_acc[BPASS];
#pragma unroll
    for (int p = 0; p < BPASS; ++p) bs_acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);

    float4 ra[APASS], rb[BPASS];
    auto load_global = [&](int it) {
        const int m0 = m_lo + it * BK;
#pragma unroll
        for (int p = 0; p < APASS; ++p) {
            const int idx = tid + 256 * p;
            ra[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < AVEC) {
                const int k = idx / (BMC / 4), c4 = idx - k * (BMC / 4);
                const int m = m0 + k, ca = tile_a * BMC + c4 * 4;
                const int qx = a_q[p][0], qy = a_q[p][1], b = a_q[p][2];
                const int ay = qy * a.sa + day, ax = qx * a.sa + dax;
                const int by = qy * a.sb + dby, bx = qx * a.sb + dbx;
                const bool ok = m < m_hi && ca < a.CA && ay >= 0 && ay < a.HA && ax >= 0 && ax < a.WA && by >= 0 && by < a.HB && bx >= 0 && bx < a.WB;
                const float* ptr = ok ? XA + (((long long)b * a.HA + ay) * a.WA + ax) * a.lda + ca : XA;   // unconditional load, select after
                const float4 v = *(const float4*)ptr;
                ra[p] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
                advance(a_q[p]);
            }
        }
#pragma unroll
        for (int p = 0; p < BPASS; ++p) {
            const int idx = tid + 256 * p;
            rb[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < BVEC) {
                const int k = idx / (BNC / 4), c4 = idx - k * (BNC / 4);
                const int m = m0 + k, cb = tile_b * BNC + c4 * 4;
                const int qx = b_q[p][0], qy = b_q[p][1], b = b_q[p][2];
                const int by = qy * a.sb + dby, bx = qx * a.sb + dbx;
                const bool ok = m < m_hi && cb < a.CB && by >= 0 && by < a.HB && bx >= 0 && bx < a.WB;
                const float* ptr = ok ? XB + (((long long)b * a.HB + by) * a.WB + bx) * a.ldb + cb : XB;
                const float4 v = *(const float4*)ptr;
                rb[p] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
                advance(b_q[p]);
                if (do_bsum) {
                    bs_acc[p].x += rb[p].x; bs_acc[p].y += rb[p].y; bs_acc[p].z += rb[p].z; bs_acc[p].w += rb[p].w;
                }
            }
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int p = 0; p < APASS; ++p) {
            const int idx = tid + 256 * p;
            if (idx < AVEC) *(float4*)(&Xs[buf][idx * 4]) = ra[p];
        }
#pragma unroll
        for (int p = 0; p < BPASS; ++p) {
            const int idx = tid + 256 * p;
            if (idx < BVEC) *(float4*)(&Ys[buf][idx * 4]) = rb[p];
        }
    };

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int half = lane >> 5, l31 = lane & 31;
    if (niter > 0) {
        load_global(0);
        store_lds(0);
    }
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const int buf = it & 1;
        if (it + 1 < niter) load_global(it + 1);
        float af[MT][8], bf[NT][8];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int s = 0; s < 8; ++s) af[i][s] = Xs[buf][(half * 8 + s) * BMC + wm * WM + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int s = 0; s < 8; ++s) bf[j][s] = Ys[buf][(half * 8 + s) * BNC + wn * WN + j * 32 + l31];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        if (it + 1 < niter) store_lds(buf ^ 1);
        __syncthreads();
    }

    if (do_bsum) {
        // column sums of the XB rows this split visited: reduce the BK/… row lanes that share a column quad through LDS
        __syncthreads();
        float* red = &Ys[0][0];                    // reuse: [rows = 256*BPASS / (BNC/4)][BNC]
#pragma unroll
        for (int p = 0; p < BPASS; ++p) {
            const int idx = tid + 256 * p;
            if (idx < BVEC) *(float4*)(red + idx * 4) = bs_acc[p];
        }
        __syncthreads();
        for (int c = tid; c < BNC; c += 256) {
            float t = 0.f;
            for (int k = 0; k < BK; ++k) t += red[k * BNC + c];
            const int cb = tile_b * BNC + c;
            if (cb < a.ldo) a.bsum[(long long)(batch * a.nsplit + split) * a.ldo + cb] = cb < a.CB ? t : 0.f;
        }
    }
    // o_bs >= 0: [batch (stride o_bs)][split][tap][CA][ldo];  o_bs < 0: [split][batch][tap][CA][ldo] -- the per-batch results stay
    // separate and ONE cdf_unpack_reduce(T = nbatch * ntaps) sums the splits of all of them
    float* O = a.o_bs >= 0 ? a.out + (long long)batch * a.o_bs + ((long long)split * a.ntaps + tap) * a.CA * a.ldo
                           : a.out + (((long long)split * a.nbatch + batch) * a.ntaps + tap) * a.CA * a.ldo;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ca = tile_a * BMC + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (ca >= a.CA) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int cb = tile_b * BNC + wn * WN + j * 32 + l31;
                if (cb < a.ldo) O[(long long)ca * a.ldo + cb] = cb < a.CB ? acc[i][j][r] : 0.f;
            }
        }
}

// ------------------------------------------------------------------------------------------------
// weight (un)packing between the PyTorch parameter layouts and the GEMM "KN" layout
//   pack  : dst[t][r][c] = (c < C) ? src[c*s_c + r*s_r + t*s_t] : 0        (ldc = padded C)
//   unpack: g[c*s_c + r*s_r + t*s_t] (+)= scale * sum_z ws[z][t][r][c]
// ------------------------------------------------------------------------------------------------
__global__ void pack_weight_kernel(const float* src, float* dst, int T, int R, int C, int ldc, long long s_t,
                                   long long s_r, long long s_c) {
    const long long n = (long long)T * R * ldc;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ldc);
        const long long tr = i / ldc;
        const int r = (int)(tr % R), t = (int)(tr / R);
        dst[i] = c < C ? src[c * s_c + r * s_r + t * s_t] : 0.f;
    }
}
// block = SL slab lanes x 64 consecutive output elements (element index runs over [T][R][C], C fastest).  A lane adds
// slabs rl, rl + SL, ... in four independent chains (four loads in flight per lane): with 4 lanes x 2 chains the ~230-slab
// reductions of the 128 x 128-pixel layers were ~30 dependent round trips (80 us for 67 MB).  The summation tree is fixed
// by (SL, nsplit): deterministic.
// (bws / gb / bC / bld: optional second reduction folded into the same launch -- the bias-gradient partials [nsplit][bld] that the
// weight-gradient kernels produce next to their slabs: gb[c] (+)= sum_z bws[z][c].  The LAST blockIdx.y row of the grid does it.)
template <int SL>
__global__ void __launch_bounds__(64 * SL) unpack_reduce_kernel(const float* ws, float* g, int nsplit, int T, int R, int C, int ldc,
                                                                long long s_t, long long s_r, long long s_c, int accumulate,
                                                                const float* bws, float* gb, int bC, int bld) {
    __shared__ float red[SL][64];
    if (bws != nullptr && blockIdx.y == 1) {                 // the fused bias reduction: one 1 x 1 x bC "tensor" with unit strides
        ws = bws; g = gb; T = 1; R = 1; C = bC; ldc = bld; s_t = 0; s_r = 0; s_c = 1;
    }
    const long long n = (long long)T * R * C;
    const long long slab = (long long)T * R * ldc;
    const int l = threadIdx.x & 63, rl = threadIdx.x >> 6;
    for (long long base = (long long)blockIdx.x * 64; base < n; base += (long long)gridDim.x * 64) {
        const long long i = base + l;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int c = 0, r = 0, t = 0;
        if (i < n) {
            c = (int)(i % C);
            const long long tr = i / C;
            r = (int)(tr % R);
            t = (int)(tr / R);
            const float* p = ws + ((long long)t * R + r) * ldc + c;
            int z = rl;
            for (; z + 3 * SL < nsplit; z += 4 * SL) {
                s0 += p[z * slab];
                s1 += p[(z + SL) * slab];
                s2 += p[(z + 2 * SL) * slab];
                s3 += p[(z + 3 * SL) * slab];
            }
            if (z < nsplit) s0 += p[z * slab];
            if (z + SL < nsplit) s1 += p[(z + SL) * slab];
            if (z + 2 * SL < nsplit) s2 += p[(z + 2 * SL) * slab];
        }
        __syncthreads();
        red[rl][l] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (rl == 0 && i < n) {
            float tot = 0.f;
#pragma unroll
            for (int q = 0; q < SL; q += 4) tot += (red[q][l] + red[q + 1][l]) + (red[q + 2][l] + red[q + 3][l]);
            float* dst = g + c * s_c + r * s_r + t * s_t;
            *dst = accumulate ? *dst + tot : tot;
        }
    }
}

// Transposing form of the slab reduction for parameter layouts whose fast index is NOT the slab's (conv weights [Cout][Cin][kh][kw]:
// s_c = Cin kh kw).  The kernel above hands 64 consecutive c to a block: every result is a lone 4-byte read-modify-write
// s_c floats from its neighbours -- a 32-byte sector moved each way per 4 useful bytes, neighbours in (r, t) landing on other XCDs.
// Here a block owns a tile of 32 c x RJ r x T taps (J = RJ T <= 36 values per c): 32 c-lanes x 8 slab-lanes, every lane J independent
// loads per slab (128-byte rows over c), the 8 slab-lanes are folded through LDS in a fixed order (deterministic) and the tile leaves
// as runs of J consecutive floats per c (144 B for 3 x 3, the whole [c][t] block for transposed-conv weights).
template <int T_, int RJ>
__device__ __forceinline__ void unpack_tile_body(const float* ws, float* g, int nsplit, int R, int C, int ldc, long long s_t, long long s_r,
                                                 long long s_c, int accumulate, int tile, float* red, int order) {
    constexpr int J = T_ * RJ;
    const int cl = threadIdx.x & 31, zg = threadIdx.x >> 5;
    const int tiles_c = (C + 31) >> 5;
    int tr, tc;
    if (order == 2) {
        // r tiles fastest inside XCD-contiguous runs: the J-float runs of neighbouring r tiles continue each other in the parameter
        // (same c), so the partial 64-byte lines at their edges meet in ONE L2 instead of being merged in memory
        const int tiles_r = (R + RJ - 1) / RJ;
        const int vt = cdf_xcd_order(tile, tiles_c * tiles_r);
        tc = vt / tiles_r;
        tr = vt - tc * tiles_r;
    } else {
        tr = tile / tiles_c;
        tc = tile - tr * tiles_c;
    }
    const int c0 = tc * 32, r0 = tr * RJ;
    const long long slab = (long long)T_ * R * ldc;
    const int cc = c0 + cl < C ? c0 + cl : C - 1;
    int roff[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int r = r0 + j / T_, t = j % T_;
        roff[j] = (t * R + (r < R ? r : R - 1)) * ldc + cc;
    }
    float acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = 0.f;
    for (int z = zg; z < nsplit; z += 8) {
        const float* p = ws + (long long)z * slab;
        float v[J];
#pragma unroll
        for (int j = 0; j < J; ++j) v[j] = p[roff[j]];
#pragma unroll
        for (int j = 0; j < J; ++j) acc[j] += v[j];
    }
#pragma unroll
    for (int j = 0; j < J; ++j) red[(zg * J + j) * 33 + cl] = acc[j];
    __syncthreads();
    // all of a thread's read-modify-writes in flight together (unconditional loads from a clamped address, predicated stores)
    constexpr int NE = (32 * J + 255) / 256;
    float tot[NE], old[NE];
    float* dst[NE];
    bool ok[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = threadIdx.x + 256 * k;
        const int ee = e < 32 * J ? e : 0;
        const int c_l = ee / J, j = ee - c_l * J;
        tot[k] = 0.f;
#pragma unroll
        for (int q = 0; q < 8; q += 4)
            tot[k] += (red[(q * J + j) * 33 + c_l] + red[((q + 1) * J + j) * 33 + c_l]) + (red[((q + 2) * J + j) * 33 + c_l] + red[((q + 3) * J + j) * 33 + c_l]);
        const int c = c0 + c_l, r = r0 + j / T_, t = j % T_;
        ok[k] = e < 32 * J && c < C && r < R;
        dst[k] = ok[k] ? g + c * s_c + r * s_r + t * s_t : g;
    }
    if (accumulate) {
#pragma unroll
        for (int k = 0; k < NE; ++k) old[k] = *dst[k];
#pragma unroll
        for (int k = 0; k < NE; ++k) tot[k] += old[k];
    }
#pragma unroll
    for (int k = 0; k < NE; ++k)
        if (ok[k]) *dst[k] = tot[k];
}

template <int T_, int RJ>
__global__ void __launch_bounds__(256) unpack_reduce_tiled_kernel(const float* ws, float* g, int nsplit, int R, int C, int ldc, long long s_t,
                                                                   long long s_r, long long s_c, int accumulate, const float* bws, float* gb,
                                                                   int bC, int bld, int order) {
    __shared__ float red[8 * T_ * RJ * 33 > 8 * 33 ? 8 * T_ * RJ * 33 : 8 * 33];
    if (blockIdx.y == 1) {                                   // the fused bias reduction: a [1][1][bC] tensor, unit strides
        if ((int)blockIdx.x < (bC + 31) / 32) unpack_tile_body<1, 1>(bws, gb, nsplit, 1, bC, bld, 0, 0, 1, accumulate, blockIdx.x, red, 1);
        return;
    }
    unpack_tile_body<T_, RJ>(ws, g, nsplit, R, C, ldc, s_t, s_r, s_c, accumulate, blockIdx.x, red, order);
}

// column sums of a row-major matrix with pitch, two deterministic stages:
//   stage 1: part[(seg*nchunk + chunk)][c] = sum_{r in chunk of segment} x[r*ld + c]
//   stage 2: out[seg][c] (+)= sum_chunk part
// grid1 = (ceil(C/64), nseg, nchunk); block = 256 = 4 row-lanes x 64 channels
template <bool BF>                                          // BF: x is a bf16 tensor (ld in bf16 elements), widened as it is read
__global__ void colsum_partial_kernel(const void* xv, float* part, int rows_per_seg, int rows_per_chunk, int C, int ld) {
    typedef typename cdf_quad<BF>::elem elem_t;
    const elem_t* x = (const elem_t*)xv;
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int r0 = blockIdx.z * rows_per_chunk;
    int r1 = r0 + rows_per_chunk;
    if (r1 > rows_per_seg) r1 = rows_per_seg;
    const elem_t* p = x + (long long)blockIdx.y * rows_per_seg * ld;
    float s = 0.f;
    if (c < C) {
        // 8 independent rows in flight per lane (a one-load-per-trip loop crawled at 0.6 TB/s)
        elem_t t[8];
        int r = r0 + rl;
        for (; r + 28 < r1; r += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = p[(long long)(r + 4 * u) * ld + c];
            s += ((cdf_widen(t[0]) + cdf_widen(t[1])) + (cdf_widen(t[2]) + cdf_widen(t[3]))) +
                 ((cdf_widen(t[4]) + cdf_widen(t[5])) + (cdf_widen(t[6]) + cdf_widen(t[7])));
        }
        for (; r < r1; r += 4) s += cdf_widen(p[(long long)r * ld + c]);
    }
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        const int l = threadIdx.x;
        part[((long long)blockIdx.y * gridDim.z + blockIdx.z) * C + c] = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
    }
}
__global__ void __launch_bounds__(1024) colsum_final_kernel(const float* part, float* out, int nchunk, int C, int ldo, int accumulate) {
    __shared__ float red[16][64];               // 16 partial lanes x 64 columns
    const int l = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + l;
    float s = 0.f;
    if (c < C) {
        const float* p = part + (long long)blockIdx.y * nchunk * C + c;
        for (int k = rl; k < nchunk; k += 16) s += p[(long long)k * C];
    }
    red[rl][l] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][l];
        float* dst = out + (long long)blockIdx.y * ldo + c;
        *dst = accumulate ? *dst + t : t;
    }
}

// ================================================================================================
// C ABI
// ================================================================================================
static int check_feat(const char* who, const void* p, int ld, int C) {
    CDF_REQUIRE(p != nullptr, "%s: null tensor", who);
    CDF_REQUIRE((((uintptr_t)p) & 15) == 0, "%s: tensor not 16-byte aligned", who);
    CDF_REQUIRE(ld % 4 == 0 && ld >= C, "%s: pitch %d must be a multiple of 4 and >= C=%d", who, ld, C);
    return CDF_OK;
}

template <int BM, int BN, int WM, int WN>
static int launch_conv(const ConvArgs& a, int batch, bool bt, hipStream_t s) {
    const int M = a.B * a.QH * a.QW;
    const int tiles = cdf_cdiv(M, BM) * cdf_cdiv(a.Cout, BN);
    if (bt)
        CDF_LAUNCH((conv_igemm_kernel<BM, BN, WM, WN, true>), dim3(tiles, a.nphase, batch), dim3(256), 0, s, a);
    else
        CDF_LAUNCH((conv_igemm_kernel<BM, BN, WM, WN, false>), dim3(tiles, a.nphase, batch), dim3(256), 0, s, a);
    return cdf_check_launch("conv_igemm");
}

// Generic gather-GEMM.  taps: int array [nphase][1 + 2 + 3*ntaps_max]... see colddiff.h
extern "C" int cdf_conv_gemm_io(const float* x, int ldx, const float* w, int ldw, float* y, int ldy, int B, int H, int W,
                                int Cin, int OH, int OW, int Cout, int QH, int QW, int os, int is, int nphase,
                                const int* phase_desc, const float* bias, const float* sbias, int ld_sbias,
                                const void* res, int ldr, void* pre, int ldp, const void* mul, int ldm, int act,
                                int mul_mode, int accumulate, int b_trans, int batch, long long x_bs, long long w_bs,
                                long long y_bs, int batch2, long long x_bs2, long long w_bs2, long long y_bs2, int io_bf16, void* y_hi,
                                int ld_ys, void* stream);

extern "C" int cdf_conv_gemm(const float* x, int ldx, const float* w, int ldw, float* y, int ldy, int B, int H, int W,
                             int Cin, int OH, int OW, int Cout, int QH, int QW, int os, int is, int nphase,
                             const int* phase_desc, const float* bias, const float* sbias, int ld_sbias,
                             const float* res, int ldr, float* pre, int ldp, const float* mul, int ldm, int act,
                             int mul_mode, int accumulate, int b_trans, int batch, long long x_bs, long long w_bs,
                             long long y_bs, int batch2, long long x_bs2, long long w_bs2, long long y_bs2, void* stream) {
    return cdf_conv_gemm_io(x, ldx, w, ldw, y, ldy, B, H, W, Cin, OH, OW, Cout, QH, QW, os, is, nphase, phase_desc, bias, sbias, ld_sbias, res, ldr,
                            pre, ldp, mul, ldm, act, mul_mode, accumulate, b_trans, batch, x_bs, w_bs, y_bs, batch2, x_bs2, w_bs2, y_bs2, 0, nullptr, 0,
                            stream);
}

// ... with typed epilogue operands (io_bf16: CDF_IO_RES_BF16 | CDF_IO_PRE_BF16 | CDF_IO_MUL_BF16) and an optional bf16 output plane y_hi
// (pitch ld_ys; with it y may be NULL: bf16 activation storage -- the fp32 product's result enters the bf16 stream rounded once, no fp32
// copy).  A batched launch without a y needs the row-offset addressing (an epilogue operand, outputs that are whole rows of one tensor:
// ldy / y_bs / y_bs2 then only describe the row offsets).
extern "C" int cdf_conv_gemm_io(const float* x, int ldx, const float* w, int ldw, float* y, int ldy, int B, int H, int W,
                                int Cin, int OH, int OW, int Cout, int QH, int QW, int os, int is, int nphase,
                                const int* phase_desc, const float* bias, const float* sbias, int ld_sbias,
                                const void* res_, int ldr, void* pre_, int ldp, const void* mul_, int ldm, int act,
                                int mul_mode, int accumulate, int b_trans, int batch, long long x_bs, long long w_bs,
                                long long y_bs, int batch2, long long x_bs2, long long w_bs2, long long y_bs2, int io_bf16, void* y_hi,
                                int ld_ys, void* stream) {
    const float* res = (const float*)res_;
    float* pre = (float*)pre_;
    const float* mul = (const float*)mul_;
    int rc;
    CDF_REQUIRE((io_bf16 & ~7) == 0, "cdf_conv_gemm_io: io_bf16 has unknown bits (%d)", io_bf16);
    CDF_REQUIRE(y || (y_hi && !accumulate), "cdf_conv_gemm_io: no output (y NULL needs y_hi and no accumulate)");
    CDF_REQUIRE(!y_hi || (ld_ys % 4 == 0 && ld_ys >= Cout && Cout % 4 == 0 && (((uintptr_t)y_hi) & 7) == 0), "cdf_conv_gemm_io: the output plane needs Cout %% 4 == 0, ld_ys %% 4 == 0, 8-byte alignment");
    if ((rc = check_feat("cdf_conv_gemm(x)", x, ldx, Cin))) return rc;
    if (y && (rc = check_feat("cdf_conv_gemm(y)", y, 4, 0))) return rc;
    CDF_REQUIRE(w && (((uintptr_t)w) & 15) == 0 && ldw % 4 == 0, "cdf_conv_gemm: weights must be 16B aligned, ldw%%4==0");
    CDF_REQUIRE(ldy >= Cout, "cdf_conv_gemm: ldy < Cout");
    CDF_REQUIRE(nphase >= 1 && nphase <= 4 && phase_desc, "cdf_conv_gemm: nphase must be 1..4");
    CDF_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && QH > 0 && QW > 0 && batch >= 1 && batch2 >= 1, "cdf_conv_gemm: bad shape");
    CDF_REQUIRE(!b_trans || (Cin % 4 == 0), "cdf_conv_gemm: b_trans needs K %% 4 == 0");
    CDF_REQUIRE(!mul_mode || mul, "cdf_conv_gemm: mul_mode without mul tensor");
    ConvArgs a;
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.sbias = sbias; a.res = res; a.pre = pre; a.mul = mul;
    a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.ld_sbias = ld_sbias; a.ldr = ldr; a.ldp = ldp; a.ldm = ldm;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.OH = OH; a.OW = OW; a.Cout = Cout; a.QH = QH; a.QW = QW;
    a.os = os; a.is = is; a.act = act; a.mul_mode = mul_mode; a.accumulate = accumulate; a.nphase = nphase;
    a.x_bs = x_bs; a.w_bs = w_bs; a.y_bs = y_bs;
    a.x_bs2 = x_bs2; a.w_bs2 = w_bs2; a.y_bs2 = y_bs2; a.batch2 = batch2;
    a.vec = cdf_epi_vec_ok(Cout, y, ldy, bias, sbias, ld_sbias, res, ldr, pre, ldp, mul, ldm) && y_bs % 4 == 0 && y_bs2 % 4 == 0;
    a.epi_follow = 0;
    if ((long long)batch * batch2 > 1 && (res || pre || mul)) {
        CDF_REQUIRE(y_bs % ldy == 0 && y_bs2 % ldy == 0 && !sbias, "cdf_conv_gemm: batched launches with pre / mul / res operands need per-batch outputs that are whole rows "
                    "of y (y_bs and y_bs2 multiples of ldy) and no per-sample bias");
        a.epi_follow = 1;
    }
    a.ys_hi = (unsigned short*)y_hi; a.ys_lo = nullptr; a.ld_ys = ld_ys; a.io_bf = io_bf16;
    CDF_REQUIRE((!y_hi && !io_bf16) || a.vec, "cdf_conv_gemm_io: bf16 operands / the output plane need the vectorised epilogue (16-byte-aligned pointers, pitches %% 4, Cout %% 4)");
    CDF_REQUIRE(y || (long long)batch * batch2 == 1 || a.epi_follow, "cdf_conv_gemm_io: a batched launch without y needs an epilogue operand (row-offset addressing)");
    batch *= batch2;
    // phase_desc: per phase [oy, ox, ntaps, (dy, dx, wi) * ntaps]
    const int* pd = phase_desc;
    for (int p = 0; p < nphase; ++p) {
        a.ph[p].oy = pd[0]; a.ph[p].ox = pd[1]; a.ph[p].ntaps = pd[2];
        CDF_REQUIRE(pd[2] >= 0 && pd[2] <= CDF_MAX_TAPS, "cdf_conv_gemm: too many taps (%d)", pd[2]);
        for (int t = 0; t < pd[2]; ++t) {
            a.ph[p].dy[t] = (signed char)pd[3 + 3 * t];
            a.ph[p].dx[t] = (signed char)pd[4 + 3 * t];
            a.ph[p].wi[t] = (signed char)pd[5 + 3 * t];
        }
        pd += 3 + 3 * pd[2];
    }
    const bool bt = b_trans != 0;
    if (Cout <= 32) return launch_conv<128, 32, 32, 32>(a, batch, bt, CDF_S);
    if (Cout <= 64) return launch_conv<256, 64, 64, 64>(a, batch, bt, CDF_S);
    return launch_conv<128, 128, 64, 64>(a, batch, bt, CDF_S);
}

template <int BMC, int BNC, int WM, int WN>
static int launch_wgrad(const WgradArgs& a, int batch, hipStream_t s) {
    const int tiles = cdf_cdiv(a.CA, BMC) * cdf_cdiv(a.CB, BNC);
    CDF_LAUNCH((conv_wgrad_kernel<BMC, BNC, WM, WN>), dim3(tiles, a.ntaps, batch * a.nsplit), dim3(256), 0, s, a);
    return cdf_check_launch("conv_wgrad");
}

extern "C" int cdf_wgrad_nsplit(int M, int CA, int CB, int ntaps) {
    // One full wave of workgroups: the 128x128 wgrad kernel is resident 4x per CU (1024 slots on 256 CUs).
    // tiles * nsplit must not spill a handful of blocks into a second round (a straggler round costs a whole
    // block time), so round DOWN to fill at most one round; keep >= 256 pixels per split.
    auto tiles_of = [&](int c) { return c <= 64 ? 1 : cdf_cdiv(c, 128); };
    const int tiles = tiles_of(CA) * tiles_of(CB) * ntaps;
    int ns = 1024 / tiles;
    const int max_by_m = M / 256 > 0 ? M / 256 : 1;
    if (ns > max_by_m) ns = max_by_m;
    if (ns < 1) ns = 1;
    if (ns > 256) ns = 256;
    return ns;
}

extern "C" int cdf_conv_wgrad(const float* xa, int lda, const float* xb, int ldb, float* ws, int ldo, int B, int QH,
                              int QW, int HA, int WA, int sa, int HB, int WB, int sb, int CA, int CB, int ntaps,
                              const int* tap_desc, int nsplit, int batch, long long a_bs, long long b_bs,
                              long long o_bs, float* bsum, void* stream) {
    int rc;
    if ((rc = check_feat("cdf_conv_wgrad(xa)", xa, lda, CA))) return rc;
    if ((rc = check_feat("cdf_conv_wgrad(xb)", xb, ldb, CB))) return rc;
    CDF_REQUIRE(ws && ldo % 4 == 0 && ldo >= CB, "cdf_conv_wgrad: bad workspace pitch");
    CDF_REQUIRE(ntaps >= 1 && ntaps <= CDF_MAX_TAPS && tap_desc, "cdf_conv_wgrad: bad tap count");
    CDF_REQUIRE(nsplit >= 1 && batch >= 1, "cdf_conv_wgrad: bad split/batch");
    WgradArgs a;
    a.xa = xa; a.xb = xb; a.out = ws; a.lda = lda; a.ldb = ldb; a.ldo = ldo;
    a.B = B; a.QH = QH; a.QW = QW; a.HA = HA; a.WA = WA; a.sa = sa; a.HB = HB; a.WB = WB; a.sb = sb;
    a.CA = CA; a.CB = CB; a.ntaps = ntaps; a.nsplit = nsplit;
    const int M = B * QH * QW;
    a.m_per_split = cdf_cdiv(cdf_cdiv(M, nsplit), 16) * 16;
    a.a_bs = a_bs; a.b_bs = b_bs; a.o_bs = o_bs; a.bsum = bsum; a.nbatch = batch;
    for (int t = 0; t < ntaps; ++t) {
        a.day[t] = (signed char)tap_desc[4 * t + 0];
        a.dax[t] = (signed char)tap_desc[4 * t + 1];
        a.dby[t] = (signed char)tap_desc[4 * t + 2];
        a.dbx[t] = (signed char)tap_desc[4 * t + 3];
    }
    if (CA <= 32) {
        if (CB <= 32) return launch_wgrad<64, 64, 32, 32>(a, batch, CDF_S);
        return launch_wgrad<32, 128, 32, 32>(a, batch, CDF_S);
    }
    if (CB <= 32) return launch_wgrad<128, 32, 32, 32>(a, batch, CDF_S);
    if (CA <= 64 && CB <= 64) return launch_wgrad<64, 64, 32, 32>(a, batch, CDF_S);
    if (CA <= 64) return launch_wgrad<64, 128, 32, 64>(a, batch, CDF_S);
    if (CB <= 64) return launch_wgrad<128, 64, 64, 32>(a, batch, CDF_S);
    return launch_wgrad<128, 128, 64, 64>(a, batch, CDF_S);
}

static inline int ew_grid2(long long n) {
    long long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    return g < 1 ? 1 : (int)g;
}

// Every cached GEMM layout of a model in ONE launch (the weights change once per optimizer step; one launch per layout was ~160
// launches of 2-60 us per step).  Entry e: dst[t][r][c] = (c < C) ? src[c*s_c + r*s_r + t*s_t] : 0 over [T][R][ldc], written as fp32
// (kind 0) or as bf16 hi [/ lo] planes (kind 1; lo == NULL: hi only).  Block b works on entry `e` with first_block[e] <= b <
// first_block[e+1] (cdf_pack_blocks(T, R, ldc, s_t) blocks per entry), elements (b - first_block[e]) * 1024 ... + 1023 of it.
struct CdfPackEntry {
    const float* src;
    void* dst0;
    void* dst1;
    long long s_t, s_r, s_c;
    int T, R, C, ldc, kind, first_block;
};

__global__ void __launch_bounds__(256) pack_many_kernel(const CdfPackEntry* tab, int nentries) {
    int lo = 0, hi = nentries - 1;                          // (wave-uniform binary search over <= a few hundred entries)
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    const CdfPackEntry e = tab[lo];
    if (e.s_t == 1 && e.T > 1 && e.T <= 16) {
        // taps contiguous in the source (conv weights [..][kh][kw]): a thread takes (r, c) pairs and moves ALL T taps of each, so the
        // 4 T-byte runs of neighbouring lanes are consumed whole while they are in flight (one tap per pass fetched every 64-byte line
        // T times: the 56 M-parameter net took 0.7 ms per step).  Block = 1024 (r, c) pairs; cdf_pack_blocks() gives the block count.
        const long long n2 = (long long)e.R * e.ldc;
        const long long j0 = (long long)(b - e.first_block) * 1024 + threadIdx.x;
        const long long plane = (long long)e.R * e.ldc;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long j = j0 + 256 * k;
            if (j >= n2) break;
            const int c = (int)(j % e.ldc), r = (int)(j / e.ldc);
            const bool ok = c < e.C;
            const float* sp = e.src + (ok ? c * e.s_c + r * e.s_r : 0);
            float v[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) v[t] = sp[t < e.T ? t : 0];
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                if (t >= e.T) break;
                const float x = ok ? v[t] : 0.f;
                const long long i = t * plane + j;
                if (e.kind == 0) {
                    ((float*)e.dst0)[i] = x;
                } else {
                    const unsigned h = cdf_f2bf(x);
                    ((unsigned short*)e.dst0)[i] = (unsigned short)h;
                    if (e.dst1) ((unsigned short*)e.dst1)[i] = (unsigned short)cdf_f2bf(x - cdf_bf2f(h));
                }
            }
        }
        return;
    }
    const long long n = (long long)e.T * e.R * e.ldc;
    const long long i0 = (long long)(b - e.first_block) * 1024 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = i0 + 256 * k;
        if (i >= n) break;
        const int c = (int)(i % e.ldc);
        const long long tr = i / e.ldc;
        const int r = (int)(tr % e.R), t = (int)(tr / e.R);
        const float v = c < e.C ? e.src[c * e.s_c + r * e.s_r + t * e.s_t] : 0.f;
        if (e.kind == 0) {
            ((float*)e.dst0)[i] = v;
        } else {
            const unsigned h = cdf_f2bf(v);
            ((unsigned short*)e.dst0)[i] = (unsigned short)h;
            if (e.dst1) ((unsigned short*)e.dst1)[i] = (unsigned short)cdf_f2bf(v - cdf_bf2f(h));
        }
    }
}

extern "C" int cdf_pack_entry_bytes(void) { return (int)sizeof(CdfPackEntry); }
// blocks an entry of cdf_pack_many spans (first_block of the next entry = first_block + this)
extern "C" int cdf_pack_blocks(int T, int R, int ldc, long long s_t) {
    const long long n = (s_t == 1 && T > 1 && T <= 16) ? (long long)R * ldc : (long long)T * R * ldc;
    return (int)((n + 1023) / 1024);
}

// table: nentries CdfPackEntry records in DEVICE memory (first_block ascending, entry e spanning ceil(T R ldc / 1024) blocks)
extern "C" int cdf_pack_many(const void* table, int nentries, int nblocks, void* stream) {
    CDF_REQUIRE(table && nentries > 0 && nblocks > 0, "cdf_pack_many: bad args");
    CDF_LAUNCH(pack_many_kernel, dim3(nblocks), dim3(256), 0, CDF_S, (const CdfPackEntry*)table, nentries);
    return cdf_check_launch("pack_many");
}

extern "C" int cdf_pack_weight(const float* src, float* dst, int T, int R, int C, int ldc, long long s_t,
                               long long s_r, long long s_c, void* stream) {
    CDF_REQUIRE(src && dst && T > 0 && R > 0 && C > 0 && ldc >= C && ldc % 4 == 0, "cdf_pack_weight: bad args");
    CDF_LAUNCH(pack_weight_kernel, dim3(ew_grid2((long long)T * R * ldc)), dim3(256), 0, CDF_S, src, dst, T, R, C, ldc, s_t, s_r, s_c);
    return cdf_check_launch("pack_weight");
}

static int launch_unpack_reduce(const float* ws, float* g, int nsplit, int T, int R, int C, int ldc, long long s_t, long long s_r,
                                long long s_c, int accumulate, const float* bws, float* gb, int bC, int bld, int tiled, hipStream_t s) {
    const int tiled_mode = tiled ? 1 : 0;                    // (1: c tiles fastest; an r-tiles-fastest-per-XCD order measured no different and is gone)
    if (s_c != 1 && C >= 32 && (T == 1 || T == 9 || T == 16) && tiled_mode) {
        const int RJ = T == 9 ? 4 : (T == 16 ? 2 : 32);
        const long long tiles = (long long)((C + 31) / 32) * ((R + RJ - 1) / RJ);
        if (tiles < (1 << 30) && (!bws || (bC + 31) / 32 <= tiles)) {
            const dim3 tg((unsigned)tiles, bws ? 2 : 1);
            if (T == 9)
                CDF_LAUNCH((unpack_reduce_tiled_kernel<9, 4>), tg, dim3(256), 0, s, ws, g, nsplit, R, C, ldc, s_t, s_r, s_c, accumulate, bws, gb, bC, bld, tiled_mode);
            else if (T == 16)
                CDF_LAUNCH((unpack_reduce_tiled_kernel<16, 2>), tg, dim3(256), 0, s, ws, g, nsplit, R, C, ldc, s_t, s_r, s_c, accumulate, bws, gb, bC, bld, tiled_mode);
            else
                CDF_LAUNCH((unpack_reduce_tiled_kernel<1, 32>), tg, dim3(256), 0, s, ws, g, nsplit, R, C, ldc, s_t, s_r, s_c, accumulate, bws, gb, bC, bld, tiled_mode);
            return cdf_check_launch("unpack_reduce_tiled");
        }
    }
    const dim3 grid(ew_grid2((long long)T * R * C * 4), bws ? 2 : 1);
    if (nsplit >= 32)       // 16 slab lanes: every lane still has >= 2 slabs
        CDF_LAUNCH(unpack_reduce_kernel<16>, grid, dim3(1024), 0, s, ws, g, nsplit, T, R, C, ldc, s_t, s_r, s_c, accumulate, bws, gb, bC, bld);
    else
        CDF_LAUNCH(unpack_reduce_kernel<4>, grid, dim3(256), 0, s, ws, g, nsplit, T, R, C, ldc, s_t, s_r, s_c, accumulate, bws, gb, bC, bld);
    return cdf_check_launch("unpack_reduce");
}

extern "C" int cdf_unpack_reduce(const float* ws, float* g, int nsplit, int T, int R, int C, int ldc, long long s_t,
                                 long long s_r, long long s_c, int accumulate, int tiled, void* stream) {
    CDF_REQUIRE(ws && g && nsplit > 0 && T > 0 && R > 0 && C > 0 && ldc >= C, "cdf_unpack_reduce: bad args");
    return launch_unpack_reduce(ws, g, nsplit, T, R, C, ldc, s_t, s_r, s_c, accumulate, nullptr, nullptr, 0, 0, tiled, CDF_S);
}

extern "C" int cdf_unpack_reduce_bias(const float* ws, float* g, int nsplit, int T, int R, int C, int ldc, long long s_t,
                                      long long s_r, long long s_c, const float* bias_ws, float* gbias, int bias_ld, int accumulate,
                                      int tiled, void* stream) {
    CDF_REQUIRE(ws && g && bias_ws && gbias && nsplit > 0 && T > 0 && R > 0 && C > 0 && ldc >= C && bias_ld >= C, "cdf_unpack_reduce_bias: bad args");
    return launch_unpack_reduce(ws, g, nsplit, T, R, C, ldc, s_t, s_r, s_c, accumulate, bias_ws, gbias, C, bias_ld, tiled, CDF_S);
}

extern "C" int cdf_colsum_nchunk(int rows_per_seg) {
    int n = rows_per_seg / 512;
    if (n < 1) n = 1;
    if (n > 1024) n = 1024;
    return n;
}

// ws: >= nseg * cdf_colsum_nchunk(rows_per_seg) * C floats
extern "C" int cdf_colsum_io(const void* x, float* out, float* ws, int nseg, int rows_per_seg, int C, int ld, int ldo,
                             int accumulate, int x_bf16, void* stream) {
    CDF_REQUIRE(x && out && ws && nseg > 0 && rows_per_seg > 0 && C > 0 && ld >= C && ldo >= C, "cdf_colsum: bad args");
    const int nchunk = cdf_colsum_nchunk(rows_per_seg);
    const int rpc = cdf_cdiv(rows_per_seg, nchunk);
    if (x_bf16)
        CDF_LAUNCH(colsum_partial_kernel<true>, dim3(cdf_cdiv(C, 64), nseg, nchunk), dim3(256), 0, CDF_S, x, ws, rows_per_seg, rpc, C, ld);
    else
        CDF_LAUNCH(colsum_partial_kernel<false>, dim3(cdf_cdiv(C, 64), nseg, nchunk), dim3(256), 0, CDF_S, x, ws, rows_per_seg, rpc, C, ld);
    CDF_LAUNCH(colsum_final_kernel, dim3(cdf_cdiv(C, 64), nseg), dim3(1024), 0, CDF_S, (const float*)ws, out, nchunk, C, ldo, accumulate);
    return cdf_check_launch("colsum");
}
extern "C" int cdf_colsum(const float* x, float* out, float* ws, int nseg, int rows_per_seg, int C, int ld, int ldo,
                          int accumulate, void* stream) {
    return cdf_colsum_io(x, out, ws, nseg, rows_per_seg, C, ld, ldo, accumulate, 0, stream);
}

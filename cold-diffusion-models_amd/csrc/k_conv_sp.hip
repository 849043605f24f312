// k_conv_sp.hip — the dense-conv gather-GEMM on the bf16 matrix cores with SPLIT-PRECISION operands.
//
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of the exact-fp32 MFMA and has no TF32-like
// mode.  To stay inside the fp32 parity budget (1e-4 on UNet outputs) every fp32 operand is split
// into two bf16 terms, x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits), and a product
// is evaluated as  a*b ~= ah*bh + ah*bl + al*bh  with fp32 accumulation in the MFMA ("bf16x3",
// relative error ~2^-16 per product instead of bf16's 2^-8): 3 MFMAs at 16x the rate = 5.3x the fp32
// MFMA throughput.  split = 1 uses only the hi terms (plain bf16 operands, fp32 accumulate).
//
// Same tap-table / phase / epilogue semantics as conv_igemm_kernel (k_conv.hip); differences:
//   * weights arrive pre-split and pre-transposed: w_hi / w_lo are bf16 [tap][Cout][ldk] (K contiguous,
//     ldk = Cin rounded up to 32, zero padded) from cdf_pack_weight_bf16, so the B tile is a straight
//     16-byte copy into LDS;
//   * activations stay fp32 in HBM and are split while being written to LDS (VALU work hidden under MFMA);
//   * BK = 32, LDS rows are 40 bf16 (80 B) so that every ds_read_b128 fragment read is conflict-free.
#include "cdf_common.h"
#include "cdf_epilogue.h"
#include "colddiff.h"
#include <atomic>

#define CDF_MAX_TAPS 16

typedef short bf16x8_v __attribute__((ext_vector_type(8)));
typedef short bf16x4_v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_v __attribute__((ext_vector_type(4)));
#ifdef CDF_EMU
#define CDF_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
static inline bf16x4_v cdf_lds_read_tr16(const unsigned short* p) { return hipemu::ds_read_tr16_b64(p); }
#else
typedef __bf16 bf16x8_hw __attribute__((ext_vector_type(8)));
#define CDF_MFMA_BF16(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_hw, a), __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0)
// ds_read_b64_tr_b16: the 16 lanes of a group pass the addresses of a [4 rows][16 cols] bf16 block (lane t: row t >> 2,
// cols 4 (t & 3) .. +3, 8-byte aligned, any row pitch); lane t gets column t's 4 rows.
__device__ __forceinline__ bf16x4_v cdf_lds_read_tr16(const unsigned short* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_v*)p);
}
#endif

struct SpPhase {
    int oy, ox, ntaps;
    signed char dy[CDF_MAX_TAPS], dx[CDF_MAX_TAPS], wi[CDF_MAX_TAPS];
};

// Block tile BM x BN, WM x WN waves of (BM/WM) x (BN/WN): the whole tile goes through LDS in one pass (cdf_epilogue.h).
constexpr int CDF_SP_CPITCH = 136;
constexpr size_t CDF_SP_EPI_LDS = (size_t)128 * CDF_SP_CPITCH * sizeof(float);

template <int BM, int BN, int WM = 2, int WN = 2, class Args>
__device__ __forceinline__ void cdf_sp_epilogue(const Args& a, const SpPhase& ph, const f32x16_t (&acc)[BM / WM / 32][BN / WN / 32], float* cs,
                                                int tile_m, int tile_n, int M, int tid) {
    constexpr int CP = BN + 8, TM = BM / WM, TN = BN / WN;
    const int lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN, half = lane >> 5, l31 = lane & 31;
    // (the K loop ends with a barrier: every wave is done with the operand tiles)
#pragma unroll
    for (int i = 0; i < TM / 32; ++i)
#pragma unroll
        for (int j = 0; j < TN / 32; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                cs[(wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * CP + wn * TN + j * 32 + l31] = acc[i][j][r];
    __syncthreads();
    cdf_epilogue_rows<BN, BM, 64 * WM * WN>(a, ph, a.y, cs, tile_m * BM, tile_n * BN, M, tid, [](int p) { return p; });
}

struct SpArgs {
    const float* x;
    const unsigned short* w_hi;
    const unsigned short* w_lo;
    float* y;
    const float* bias;
    const float* sbias;
    const float* res;
    float* pre;
    const float* mul;
    int ldx, ldk, ldy, ld_sbias, ldr, ldp, ldm;
    int B, H, W, Cin, OH, OW, Cout, QH, QW, os, is;
    int act, mul_mode, accumulate, nphase, vec;
    unsigned short* ys_hi;         // nullable: bf16 hi / lo planes of the output (pitch ld_ys), written by the epilogue
    unsigned short* ys_lo;
    int ld_ys;
    int io_bf;                     // CDF_IO_*_BF16 bits (cdf_epilogue.h)
    SpPhase ph[4];
};


// In-kernel split of an activation quad, kept to ~4 VALU ops per element (the kernel is VALU-, not
// MFMA-bound): hi = x truncated to bf16 (the residual x - hi is exact in fp32 and lands in lo, so
// truncating hi costs nothing), lo = (x - hi) truncated to bf16: x = hi + lo + O(2^-16 |x|).
__device__ __forceinline__ unsigned cdf_pack_hi16(unsigned u0, unsigned u1) { return (u0 >> 16) | (u1 & 0xFFFF0000u); }
__device__ __forceinline__ void cdf_split4_trunc(const float4& v, uint2& hi, uint2& lo) {
    const unsigned u0 = __float_as_uint(v.x), u1 = __float_as_uint(v.y), u2 = __float_as_uint(v.z), u3 = __float_as_uint(v.w);
    hi.x = cdf_pack_hi16(u0, u1);
    hi.y = cdf_pack_hi16(u2, u3);
    const unsigned r0 = __float_as_uint(v.x - __uint_as_float(u0 & 0xFFFF0000u));
    const unsigned r1 = __float_as_uint(v.y - __uint_as_float(u1 & 0xFFFF0000u));
    const unsigned r2 = __float_as_uint(v.z - __uint_as_float(u2 & 0xFFFF0000u));
    const unsigned r3 = __float_as_uint(v.w - __uint_as_float(u3 & 0xFFFF0000u));
    lo.x = cdf_pack_hi16(r0, r1);
    lo.y = cdf_pack_hi16(r2, r3);
}

__device__ __forceinline__ int cdf_sp_swizzle(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// One algorithmic product a * b on the matrix cores.  NS = 3: split precision, al*bh + ah*bl + ah*bh (x = hi + lo bf16 terms,
// fp32 accumulate); NS = 1: single-pass bf16 operands (the hi planes only; al / bl are never read and their loads fold away).
template <int NS>
__device__ __forceinline__ void cdf_mma_sp(f32x16_t& acc, const bf16x8_v& ah, const bf16x8_v& al, const bf16x8_v& bh, const bf16x8_v& bl) {
    if constexpr (NS == 3) {
        acc = CDF_MFMA_BF16(al, bh, acc);
        acc = CDF_MFMA_BF16(ah, bl, acc);
    }
    acc = CDF_MFMA_BF16(ah, bh, acc);
}

// All products of one K chunk (two k16 steps) of a wave tile, TERM-MAJOR: consecutive MFMAs go to different accumulators
// (al*bh for every tile, then ah*bl, then ah*bh), so no instruction waits for the result of the one just issued; the
// summation order per accumulator is the same as in cdf_mma_sp.
template <int NS, int MT, int NT>
__device__ __forceinline__ void cdf_mma_tile(f32x16_t (&acc)[MT][NT], const bf16x8_v (&ah)[2][MT], const bf16x8_v (&al)[2][MT],
                                             const bf16x8_v (&bh)[2][NT], const bf16x8_v (&bl)[2][NT]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        if constexpr (NS == 3) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = CDF_MFMA_BF16(al[ks][i], bh[ks][j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = CDF_MFMA_BF16(ah[ks][i], bl[ks][j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = CDF_MFMA_BF16(ah[ks][i], bh[ks][j], acc[i][j]);
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) cdf_mma_sp<NS>(acc[i][j], ah[ks][i], al[ks][i], bh[ks][j], bl[ks][j]);
        }
    }
}

// 256 threads = 4 waves (2x2), block tile 128x128, wave tile 64x64 = 2x2 MFMA 32x32 tiles.
template <int SPLIT>
__global__ void __launch_bounds__(256, 2) conv_igemm_sp_kernel(SpArgs a) {
    constexpr int BM = 128, BN = 128, BK = 32, AS = 40;      // AS: LDS row stride in bf16 elements (80 B)
    constexpr int NPL = SPLIT == 1 ? 1 : 2;                  // operand planes (hi [, lo])
    constexpr int PLANE = BM * AS;                           // elements per plane (BM == BN)
    constexpr int STAGE = 2 * NPL * PLANE;                   // A planes then B planes
    CDF_DYN_SMEM(smem_raw);
    unsigned short* smem = (unsigned short*)smem_raw;        // [2 stages][STAGE]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int M = a.B * a.QH * a.QW;
    const int tiles_n = (a.Cout + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const int tile = cdf_sp_swizzle(blockIdx.x, tiles_m * tiles_n);
    const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
    const SpPhase& ph = a.ph[blockIdx.y];

    // A rows of this thread: row = (tid >> 3) + 32 p, float4 column (tid & 7)
    const int a_c4 = (tid & 7) * 4;
    int a_iy0[4], a_ix0[4];
    unsigned a_pix[4];          // pixel index of (b, iy0, ix0); the tap adds a wave-uniform dy*W + dx
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int m = tile_m * BM + (tid >> 3) + 32 * p;
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
    // B rows of this thread: n = brow + 64 p, 16-byte column (tid & 3); brow pairs rows R and R+4 inside each
    // 8-lane ds_write_b128 group (conflict-free stores, see conv_igemm_spx_kernel)
    const int b_q = tid & 3;
    const int bg8 = tid >> 3, brow = ((bg8 >> 2) << 3) + (bg8 & 3) + (((tid >> 2) & 1) << 2);
    const int nchunks = (a.Cin + BK - 1) / BK;
    const int niter = ph.ntaps * nchunks;

    // Loads are UNCONDITIONAL (a load inside a divergent branch makes hipcc wait vmcnt(0) per load and
    // serialises the whole prefetch): out-of-image / tail elements read a clamped, always-valid address and
    // are zeroed by a select when they are written to LDS.  Weight rows/columns outside the tile are clamped
    // too; they only feed output columns that are never stored.
    float4 ra[4];
    uint4 rbh0, rbh1, rbl0, rbl1;          // named registers (an array of HIP vector structs ends up in scratch here)
    rbl0 = rbl1 = make_uint4(0u, 0u, 0u, 0u);
    unsigned a_ok = 0;
    int b_row[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int n = tile_n * BN + brow + 64 * p;
        b_row[p] = n < a.Cout ? n : a.Cout - 1;
    }
    auto load_global = [&](int it) {
        const int tap = it / nchunks, c0 = (it - tap * nchunks) * BK;
        const int dy = ph.dy[tap], dx = ph.dx[tap], wi = ph.wi[tap];
        const int tap_pix = dy * a.W + dx;
        const bool cok = (c0 + a_c4) < a.Cin;
        const float* xc = a.x + c0 + a_c4;
        a_ok = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned iy = (unsigned)(a_iy0[p] + dy), ix = (unsigned)(a_ix0[p] + dx);     // unsigned compare folds the >= 0 test
            const bool ok = iy < (unsigned)a.H && ix < (unsigned)a.W && cok;
            const float* ptr = ok ? xc + (size_t)(a_pix[p] + (unsigned)tap_pix) * (unsigned)a.ldx : a.x;
            ra[p] = *(const float4*)ptr;
            a_ok |= (ok ? 1u : 0u) << p;
        }
        const long long off0 = ((long long)wi * a.Cout + b_row[0]) * a.ldk + c0 + b_q * 8;
        const long long off1 = ((long long)wi * a.Cout + b_row[1]) * a.ldk + c0 + b_q * 8;
        rbh0 = *(const uint4*)(a.w_hi + off0);
        rbh1 = *(const uint4*)(a.w_hi + off1);
        if (SPLIT > 1) {
            rbl0 = *(const uint4*)(a.w_lo + off0);
            rbl1 = *(const uint4*)(a.w_lo + off1);
        }
    };
    auto store_lds = [&](int buf) {
        unsigned short* st = smem + buf * STAGE;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            uint2 hi, lo;
            const float keep = ((a_ok >> p) & 1u) ? 1.0f : 0.0f;       // loaded values are finite: zeroing by multiplication
            const float4 v = make_float4(ra[p].x * keep, ra[p].y * keep, ra[p].z * keep, ra[p].w * keep);
            if (SPLIT > 1) {
                cdf_split4_trunc(v, hi, lo);
            } else {      // plain bf16 operands: round to nearest even
                hi.x = cdf_f2bf(v.x) | (cdf_f2bf(v.y) << 16);
                hi.y = cdf_f2bf(v.z) | (cdf_f2bf(v.w) << 16);
                lo = hi;
            }
            const int off = ((tid >> 3) + 32 * p) * AS + a_c4;
            *(uint2*)(st + off) = hi;
            if (SPLIT > 1) *(uint2*)(st + PLANE + off) = lo;
        }
        unsigned short* sb = st + NPL * PLANE;
        const int offb0 = brow * AS + b_q * 8, offb1 = (brow + 64) * AS + b_q * 8;
        *(uint4*)(sb + offb0) = rbh0;
        *(uint4*)(sb + offb1) = rbh1;
        if (SPLIT > 1) {
            *(uint4*)(sb + PLANE + offb0) = rbl0;
            *(uint4*)(sb + PLANE + offb1) = rbl1;
        }
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
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
        const unsigned short* sa = smem + buf * STAGE;
        const unsigned short* sb = sa + NPL * PLANE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int k0 = ks * 16 + half * 8;
            bf16x8_v ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int off = (wm * 64 + i * 32 + l31) * AS + k0;
                ah[i] = *(const bf16x8_v*)(sa + off);
                if (SPLIT > 1) al[i] = *(const bf16x8_v*)(sa + PLANE + off);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int off = (wn * 64 + j * 32 + l31) * AS + k0;
                bh[j] = *(const bf16x8_v*)(sb + off);
                if (SPLIT > 1) bl[j] = *(const bf16x8_v*)(sb + PLANE + off);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (SPLIT > 1) {
                        // small cross terms first, the dominant hi*hi term last
                        acc[i][j] = CDF_MFMA_BF16(al[i], bh[j], acc[i][j]);
                        acc[i][j] = CDF_MFMA_BF16(ah[i], bl[j], acc[i][j]);
                    }
                    acc[i][j] = CDF_MFMA_BF16(ah[i], bh[j], acc[i][j]);
                }
        }
        if (it + 1 < niter) store_lds(buf ^ 1);
        __syncthreads();
    }

    cdf_sp_epilogue<128, 128>(a, ph, acc, (float*)smem_raw, tile_m, tile_n, M, tid);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient on the bf16 matrix cores, split precision (same contract as conv_wgrad_kernel):
//   out[z][tap][ca][cb] = sum_{m in split z} XA[pixA(m,tap)][ca] * XB[pixB(m,tap)][cb]
// The contraction runs over PIXELS, which are the slow index of both NHWC operands, so the LDS tiles
// stay pixel-major ([32 px][128 ch] bf16 hi / lo, split while being stored) and each lane assembles
// its 8-pixel MFMA fragment from eight 16-bit LDS reads (32 consecutive channels per half-wave:
// conflict-free).  Tile 128 (ca) x 128 (cb), BK = 32 pixels, 4 waves of 64x64.
// ------------------------------------------------------------------------------------------------
struct SpWgradArgs {
    const float* xa;
    const float* xb;
    float* out;
    float* bsum;
    int lda, ldb, ldo;
    int B, QH, QW;
    int HA, WA, sa, HB, WB, sb;
    int CA, CB;
    int ntaps, nsplit, m_per_split, xcd_swizzle;
    signed char day[CDF_MAX_TAPS], dax[CDF_MAX_TAPS], dby[CDF_MAX_TAPS], dbx[CDF_MAX_TAPS];
};

// XCD-aware block order of the weight-gradient grids (tiles, taps, splits).  Workgroups go to the 8 XCDs round-robin in
// dispatch order, so the taps of one pixel range (next to each other in dispatch order) would land on 8 different L2s
// and each of them would fetch the same operand rows over the fabric: measured 4-6x the algorithmic bytes (rocprofv3
// FETCH_SIZE).  Re-numbered so that every XCD works through a CONTIGUOUS range of (tile, tap, split) ids: all tiles and
// taps of a pixel range run on one XCD at about the same time and share its L2.
__device__ __forceinline__ void cdf_wgrad_block(int enable, int& bx, int& by, int& bz) {
    bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (enable) {
        const int gx = gridDim.x, gy = gridDim.y;
        const int v = cdf_sp_swizzle(bx + gx * (by + gy * bz), gx * gy * (int)gridDim.z);
        bx = v % gx;
        const int t2 = v / gx;
        by = t2 % gy;
        bz = t2 / gy;
    }
}

__global__ void __launch_bounds__(256, 2) conv_wgrad_sp_kernel(SpWgradArgs a) {
    constexpr int BC = 128, BK = 32;
    constexpr int PLANE = BK * BC;                 // bf16 elements per plane
    constexpr int STAGE = 4 * PLANE;               // A hi, A lo, B hi, B lo
    CDF_DYN_SMEM(smem_raw);
    unsigned short* smem = (unsigned short*)smem_raw;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_b = (a.CB + BC - 1) / BC;
    int bx, by, bz;
    cdf_wgrad_block(a.xcd_swizzle, bx, by, bz);
    const int tile_a = bx / tiles_b, tile_b = bx - tile_a * tiles_b;
    const int tap = by, split = bz;
    const int M = a.B * a.QH * a.QW;
    const int m_lo = split * a.m_per_split;
    int m_hi = m_lo + a.m_per_split;
    if (m_hi > M) m_hi = M;
    const int niter = m_hi > m_lo ? (m_hi - m_lo + BK - 1) / BK : 0;
    const int day = a.day[tap], dax = a.dax[tap], dby = a.dby[tap], dbx = a.dbx[tap];

    // load slots: pixel k = (tid >> 5) + 8 p, channel quad c4 = tid & 31 (both operands)
    const int c4 = (tid & 31) * 4;
    int q[4][3];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int m = m_lo + (tid >> 5) + 8 * p;
        q[p][0] = m % a.QW;
        const int t2 = m / a.QW;
        q[p][1] = t2 % a.QH;
        q[p][2] = t2 / a.QH;
    }
    const bool do_bsum = a.bsum != nullptr && tile_a == 0 && tap == 0;
    float4 bs_acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) bs_acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int ca = tile_a * BC + c4, cb = tile_b * BC + c4;

    float4 ra[4], rb[4];
    const int ca_l = ca < a.CA ? ca : 0, cb_l = cb < a.CB ? cb : 0;      // clamped (always readable) channel offsets
    auto load_global = [&](int it) {
        const int m0 = m_lo + it * BK;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = m0 + (tid >> 5) + 8 * p;
            const int qx = q[p][0], qy = q[p][1], b = q[p][2];
            const unsigned ay = (unsigned)(qy * a.sa + day), ax = (unsigned)(qx * a.sa + dax);
            const unsigned by = (unsigned)(qy * a.sb + dby), bx = (unsigned)(qx * a.sb + dbx);
            const bool bok = m < m_hi && by < (unsigned)a.HB && bx < (unsigned)a.WB;
            const bool aok = bok && ay < (unsigned)a.HA && ax < (unsigned)a.WA && ca < a.CA;
            const bool bok2 = bok && cb < a.CB;
            // unconditional loads from clamped addresses, zero-select afterwards (no divergent branch around a load)
            const float* pa = aok ? a.xa + (((long long)b * a.HA + ay) * a.WA + ax) * a.lda + ca_l : a.xa;
            const float* pb = bok2 ? a.xb + (((long long)b * a.HB + by) * a.WB + bx) * a.ldb + cb_l : a.xb;
            const float4 va = *(const float4*)pa, vb = *(const float4*)pb;
            ra[p] = aok ? va : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[p] = bok2 ? vb : make_float4(0.f, 0.f, 0.f, 0.f);
            q[p][0] += BK;
            while (q[p][0] >= a.QW) {
                q[p][0] -= a.QW;
                if (++q[p][1] >= a.QH) { q[p][1] = 0; ++q[p][2]; }
            }
            if (do_bsum) {
                bs_acc[p].x += rb[p].x; bs_acc[p].y += rb[p].y; bs_acc[p].z += rb[p].z; bs_acc[p].w += rb[p].w;
            }
        }
    };
    auto store_lds = [&](int buf) {
        unsigned short* st = smem + buf * STAGE;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int off = ((tid >> 5) + 8 * p) * BC + c4;
            uint2 hi, lo;
            cdf_split4_trunc(ra[p], hi, lo);
            *(uint2*)(st + off) = hi;
            *(uint2*)(st + PLANE + off) = lo;
            cdf_split4_trunc(rb[p], hi, lo);
            *(uint2*)(st + 2 * PLANE + off) = hi;
            *(uint2*)(st + 3 * PLANE + off) = lo;
        }
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
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
        const unsigned short* st = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int k0 = ks * 16 + half * 8;
            bf16x8_v ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned short* pa = st + k0 * BC + wm * 64 + i * 32 + l31;
                const unsigned short* pb = st + 2 * PLANE + k0 * BC + wn * 64 + i * 32 + l31;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ah[i][e] = (short)pa[e * BC];
                    al[i][e] = (short)pa[PLANE + e * BC];
                    bh[i][e] = (short)pb[e * BC];
                    bl[i][e] = (short)pb[PLANE + e * BC];
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = CDF_MFMA_BF16(al[i], bh[j], acc[i][j]);
                    acc[i][j] = CDF_MFMA_BF16(ah[i], bl[j], acc[i][j]);
                    acc[i][j] = CDF_MFMA_BF16(ah[i], bh[j], acc[i][j]);
                }
        }
        if (it + 1 < niter) store_lds(buf ^ 1);
        __syncthreads();
    }

    if (do_bsum) {
        float* red = (float*)smem;                 // [32 px][128] floats = 16 KB (stage 0 is idle now)
#pragma unroll
        for (int p = 0; p < 4; ++p) *(float4*)(red + ((tid >> 5) + 8 * p) * BC + c4) = bs_acc[p];
        __syncthreads();
        for (int c = tid; c < BC; c += 256) {
            float t = 0.f;
            for (int k = 0; k < BK; ++k) t += red[k * BC + c];
            const int cc = tile_b * BC + c;
            if (cc < a.ldo) a.bsum[(long long)split * a.ldo + cc] = cc < a.CB ? t : 0.f;
        }
    }
    float* O = a.out + ((long long)split * a.ntaps + tap) * a.CA * a.ldo;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tile_a * BC + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row >= a.CA) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = tile_b * BC + wn * 64 + j * 32 + l31;
                if (col < a.ldo) O[(long long)row * a.ldo + col] = col < a.CB ? acc[i][j][r] : 0.f;
            }
        }
}

// ================================================================================================
// PRE-SPLIT operand variants.  Splitting inside the GEMM costs ~9 VALU instructions per MFMA and is
// repeated by every N tile and again by dgrad / wgrad, which makes the in-kernel-split kernels
// issue-slot bound (measured: MFMA pipe 33 % busy).  Here the activation has been split ONCE by
// split_bf16_kernel into bf16 hi / lo planes ([rows][ld] each, same bytes as the fp32 tensor) and the
// GEMM main loop is pure 16-byte copies global -> LDS plus MFMAs.  Out-of-image taps read from a
// caller-provided zero page, so there is no masking arithmetic at all.
// ================================================================================================
__global__ void split_bf16_kernel(const float* x, int ldx, unsigned short* hi, unsigned short* lo, int ldo, long long rows, int C4) {
    const long long n = rows * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const long long r = i / C4;
        const float4 v = *(const float4*)(x + r * ldx + c);
        uint2 h, l;
        cdf_split4(v.x, v.y, v.z, v.w, h, l);
        *(uint2*)(hi + r * ldo + c) = h;
        if (lo) *(uint2*)(lo + r * ldo + c) = l;
    }
}

// the way back (bf16 activation storage): a bf16 tensor widened to fp32 for a kernel that has no bf16-input form (exact conversion)
__global__ void widen_bf16_kernel(const unsigned short* x, int ldx, float* y, int ldy, long long rows, int C4) {
    const long long n = rows * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const long long r = i / C4;
        *(float4*)(y + r * ldy + c) = cdf_quad_cvt(*(const uint2*)(x + r * ldx + c));
    }
}

#define CDF_GLDS16_K(g, l) CDF_GLDS16(g, l)

struct SpxArgs {
    const unsigned short* x_hi;
    const unsigned short* x_lo;
    const unsigned short* zero;    // >= 16 zero bytes, 16-byte aligned
    const unsigned short* w_hi;
    const unsigned short* w_lo;
    float* y;
    const float* bias;
    const float* sbias;
    const float* res;
    float* pre;
    const float* mul;
    int ldx, ldk, ldy, ld_sbias, ldr, ldp, ldm;
    int B, H, W, Cin, OH, OW, Cout, QH, QW, os, is;
    int act, mul_mode, accumulate, nphase, vec;
    int taprot;                    // 1: a tile is one image row and the 9 taps are 3 row groups -> per-block row-group order (see kernel)
    int dephase;                   // 1: the two waves of a SIMD run half a K step apart (one reads fragments / issues DMA while the other multiplies)
    unsigned short* ys_hi;         // nullable: bf16 hi / lo planes of the output (pitch ld_ys), written by the epilogue
    unsigned short* ys_lo;
    int ld_ys;
    int ksplit;                    // > 1 (generic kernel, one phase): blockIdx.z takes ntaps / ksplit taps and writes its raw partial
    float* ks_ws;                  //      sums to ks_ws[z][m][ks_ld]; conv_splitk_finish_kernel adds them up and runs the epilogue
    int ks_ld;
    int io_bf;                     // CDF_IO_*_BF16 bits (cdf_epilogue.h): res / pre / mul are bf16 tensors (bf16 activation storage)
    SpPhase ph[4];
};

// what cdf_epilogue_rows reads, for a raw store of the accumulator tile (split-K partial sums): rows m of a [M][ldy] slab
struct RawEpiArgs {
    int Cout, vec, os, QH, QW, OH, OW, ldy, ldp, ldm, ldr, ld_sbias, ld_ys, act, mul_mode, accumulate, io_bf;
    const float* bias;
    const float* sbias;
    float* pre;
    const float* mul;
    const float* res;
    unsigned short* ys_hi;
    unsigned short* ys_lo;
};

template <int BM, int BN, int WM, int WN, int NSTAGE, int OCC = 512 / (64 * WM * WN), int NS = 3>
__global__ void __launch_bounds__(64 * WM * WN, OCC) conv_igemm_spx_kernel(SpxArgs a) {
    // Block tile BM x BN, WM x WN waves of (BM/WM) x (BN/WN), BK = 32, NSTAGE LDS stages.  Two shapes of the template are
    // used: 4 waves (2 x 2) on a 64/128 x 64/128 tile with 2 stages, two blocks per CU; and 8 waves (4 x 2) on a
    // 256 x 128 tile with 3 stages, one block per CU -- the same 8 waves per CU, but the DMA of chunk it+2 is in flight
    // while chunk it is multiplied (a global fetch takes longer than one chunk's MFMAs) and each B tile feeds twice the
    // MFMAs.  Operand tiles go global -> LDS by LDS-DMA
    // (CDF_GLDS16): the register-staged version spent as long in ds_write_b128 (13 LDS-path cycles per wave
    // instruction) as in the MFMAs.  DMA images are lane-linear, so a stage plane is [rows][64 B] without padding and
    // the bank spreading is an XOR swizzle applied on BOTH sides: the 16-byte column c of row r lives at column
    // c ^ ((r >> 2) & 3) -- the lane that fills LDS slot (r, c') fetches global column c' ^ ((r >> 2) & 3), the
    // fragment read of (r, c) goes to c ^ ((r >> 2) & 3).  With that the 16 rows of every ds_read_b128 lane group
    // (rows = r mod 4 classes x 4 distinct (r >> 2) & 3) cover all 64 banks exactly once.
    constexpr int BK = 32, RE = 32, NW = WM * WN, NTHR = 64 * NW;     // RE: row elements (64 bytes)
    constexpr int MT = BM / WM / 32, NT = BN / WN / 32;               // 32 x 32 MFMA tiles per wave
    constexpr int SA = BM / 16 / NW, SB = BN / 16 / NW;               // 16-row DMA segments per wave and plane
    static_assert(SA >= 1 && SB >= 1 && SA * NW * 16 == BM && SB * NW * 16 == BN, "tile must split into 16-row segments per wave");
    constexpr int PLANE_A = BM * RE, PLANE_B = BN * RE;
    constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_B;                  // A hi, A lo, B hi, B lo
    CDF_DYN_SMEM(smem_raw);
    unsigned short* smem = (unsigned short*)smem_raw;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int M = a.B * a.QH * a.QW;
    const int tiles_n = (a.Cout + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
    const int tile = cdf_sp_swizzle(blockIdx.x, tiles_m * tiles_n);
    const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
    const SpPhase& ph = a.ph[blockIdx.y];

    // DMA slots of this lane: wave w fills the 16-row segments w*SA + p of both A planes and w*SB + p of both B planes;
    // inside a segment lane l is row l >> 2, LDS column l & 3, i.e. global column (l & 3) ^ ((l >> 4) & 3).
    const int srow = lane >> 2;
    const int q8 = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    int a_iy0[SA], a_ix0[SA], b_row[SB];
    unsigned a_pix[SA];
#pragma unroll
    for (int p = 0; p < SA; ++p) {
        const int m = tile_m * BM + (wave * SA + p) * 16 + srow;
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
#pragma unroll
    for (int p = 0; p < SB; ++p) {
        const int n = tile_n * BN + (wave * SB + p) * 16 + srow;
        b_row[p] = n < a.Cout ? n : a.Cout - 1;
    }
    const int nchunks = (a.Cin + BK - 1) / BK;
    // split-K over the taps (small grids: see dispatch_gemm_bf16x): this block's share of the taps
    const int ntaps_blk = a.ksplit > 1 ? ph.ntaps / a.ksplit : ph.ntaps;
    const int tap_lo = a.ksplit > 1 ? (int)blockIdx.z * ntaps_blk : 0;
    const int niter = ntaps_blk * nchunks;

    // Tap table -> LDS once, behind the stages (a dynamic index into the by-value kernel argument compiles to
    // per-iteration global byte loads in front of the tile loads).  CDF_MAX_TAPS + 1 entries: reading one past the
    // end is harmless.
    // Row-group rotation (a.taprot: 3 x 3 taps as three groups of equal dy, tile = exactly one image row, so tile_m is the
    // global row index).  Input row r is needed by the three tiles r - dy, each in its group dy.  In the table's order every
    // tile would read it in a different third of its life and, with the ~64 co-resident tiles of an XCD streaming more than
    // the 4 MB L2 per third, each of the three reads came over the fabric (measured 3.4x the algorithmic bytes).  Here tile j
    // handles group dy in slot (j + dy) mod 3: the tiles of an XCD run in lockstep (same start, same work), so the three
    // readers of a row now read it at the same time and the L2 fetches it once.
    int* tap_lds = (int*)(smem + NSTAGE * STAGE);
    if (tid <= CDF_MAX_TAPS) {
        int src = tid + tap_lo;
        if (src > CDF_MAX_TAPS) src = CDF_MAX_TAPS;
        if (a.taprot && tid < 9) {
            const int slot = tid / 3, kx = tid - 3 * slot;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const int r = (tile_m + (int)ph.dy[3 * g]) % 3;      // (tile_m + dy >= -1)
                if ((r < 0 ? r + 3 : r) == slot) src = 3 * g + kx;
            }
        }
        tap_lds[tid] = src < ph.ntaps ? (ph.dy[src] & 0xFF) | ((ph.dx[src] & 0xFF) << 8) | ((ph.wi[src] & 0xFF) << 16) : 0;
    }
    CDF_LDS_BARRIER();

    // DMA source pointers of this lane, valid for the current tap and advanced by one K chunk per fetch.  The address
    // generation (bounds test, 64-bit multiply, zero-page select) runs once per TAP, not per chunk: per-chunk it was
    // 3.6 vector instructions per MFMA (PMC), all competing with the MFMAs for issue slots.  An element outside the
    // image fetches the zero page (pointer does not advance); all fetches are unconditional.
    const unsigned short* pa_hi[SA];
    const unsigned short* pa_lo[SA];
    const unsigned short* pb_hi[SB];
    const unsigned short* pb_lo[SB];
    int a_inc[SA];
    const bool ragged = (a.Cin & (BK - 1)) != 0;             // last chunk of a tap only partly inside the channel range
    auto retap = [&](int tap) {
        const int tc = tap_lds[tap];
        const int dy = (int)(signed char)(tc & 0xFF), dx = (int)(signed char)((tc >> 8) & 0xFF), wi = (tc >> 16) & 0xFF;
        const int tap_pix = dy * a.W + dx;
#pragma unroll
        for (int p = 0; p < SA; ++p) {
            const unsigned iy = (unsigned)(a_iy0[p] + dy), ix = (unsigned)(a_ix0[p] + dx);
            const bool ok = iy < (unsigned)a.H && ix < (unsigned)a.W && q8 < a.Cin;
            const size_t off = (size_t)(a_pix[p] + (unsigned)tap_pix) * (unsigned)a.ldx + (unsigned)q8;
            pa_hi[p] = ok ? a.x_hi + off : a.zero;
            pa_lo[p] = ok ? a.x_lo + off : a.zero;
            a_inc[p] = ok ? BK : 0;
        }
#pragma unroll
        for (int p = 0; p < SB; ++p) {
            const size_t woff = (size_t)((unsigned)wi * (unsigned)a.Cout + (unsigned)b_row[p]) * (unsigned)a.ldk + (unsigned)q8;
            pb_hi[p] = a.w_hi + woff;
            pb_lo[p] = a.w_lo + woff;
        }
    };
    // Past the last chunk the last one is simply fetched again into an idle stage (never read).
    int tap = 0, c0 = 0, issued = 0;                         // (tap, channel chunk) of the NEXT fetch
    retap(0);
    auto fetch = [&](int buf) {
        unsigned short* st = smem + buf * STAGE;
        const bool cok = !ragged || (c0 + q8) < a.Cin;       // (false only in the ragged last chunk of a tap)
#pragma unroll
        for (int p = 0; p < SA; ++p) {
            unsigned short* seg = st + (wave * SA + p) * 16 * RE;
            CDF_GLDS16(cok ? pa_hi[p] : a.zero, seg);
            if constexpr (NS == 3) CDF_GLDS16(cok ? pa_lo[p] : a.zero, seg + PLANE_A);
        }
#pragma unroll
        for (int p = 0; p < SB; ++p) {
            unsigned short* seg = st + 2 * PLANE_A + (wave * SB + p) * 16 * RE;
            CDF_GLDS16(pb_hi[p], seg);                       // (weights are zero padded along K to the chunk size)
            if constexpr (NS == 3) CDF_GLDS16(pb_lo[p], seg + PLANE_B);
        }
        const bool more = issued + 1 < niter;                // block-uniform
        issued += more ? 1 : 0;
        if (more) {
            c0 += BK;
            if (c0 >= a.Cin) {                               // next tap: block-uniform branch, no load inside
                c0 = 0;
                ++tap;
                retap(tap);
            } else {
#pragma unroll
                for (int p = 0; p < SA; ++p) {
                    pa_hi[p] += a_inc[p];
                    pa_lo[p] += a_inc[p];
                }
#pragma unroll
                for (int p = 0; p < SB; ++p) {
                    pb_hi[p] += BK;
                    pb_lo[p] += BK;
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
    const int sw = (l31 >> 2) & 3;                           // read-side swizzle (tile row offsets are multiples of 32)
    constexpr int PIECES = (NS == 3 ? 2 : 1) * (SA + SB);    // this wave's DMA instructions per chunk
    // chunk c lives in stage c % NSTAGE; NSTAGE - 1 chunks are in flight ahead of the one being multiplied
    int fbuf = 0;                                            // stage of the next fetch
    if (niter > 0) {
#pragma unroll
        for (int d = 0; d < NSTAGE - 1; ++d) {
            fetch(fbuf);
            fbuf = fbuf + 1 == NSTAGE ? 0 : fbuf + 1;
        }
    }
    CDF_WAIT_DMA_LEAVE((NSTAGE - 2) * PIECES);               // chunk 0 has landed; later ones may still be in flight
    CDF_LDS_BARRIER();
    int buf = 0;
    bf16x8_v ah[2][MT], al[2][MT], bh[2][NT], bl[2][NT];
    // De-phased waves (a.dephase, 8-wave tiles: waves 4..7 share their SIMDs with waves 0..3): a late wave multiplies the
    // fragments it read in the PREVIOUS step first, then issues its DMA and reads this step's fragments -- while one wave of a
    // SIMD is stalled issuing global_load_lds / reading LDS the other one feeds the matrix pipe (see conv_igemm_halo_kernel).
    const bool late = a.dephase != 0 && NW == 8 && wave >= 4;    // (wave-uniform)
    if (late) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) { ah[ks][i][e] = 0; al[ks][i][e] = 0; }
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) { bh[ks][j][e] = 0; bl[ks][j][e] = 0; }
        }
    }
    auto read_frags = [&](const unsigned short* sa, const unsigned short* sb) {
        // all fragment reads of the chunk are issued up front: the second k-step's LDS latency hides behind the first
        // k-step's MFMAs (the registers are there -- LDS, not VGPRs, limits the residency)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int kc = ((ks * 2 + half) ^ sw) * 8;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int off = (wm * (BM / WM) + i * 32 + l31) * RE + kc;
                ah[ks][i] = *(const bf16x8_v*)(sa + off);
                if constexpr (NS == 3) al[ks][i] = *(const bf16x8_v*)(sa + PLANE_A + off);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int off = (wn * (BN / WN) + j * 32 + l31) * RE + kc;
                bh[ks][j] = *(const bf16x8_v*)(sb + off);
                if constexpr (NS == 3) bl[ks][j] = *(const bf16x8_v*)(sb + PLANE_B + off);
            }
        }
    };
    auto mma_frags = [&]() {
        cdf_mma_tile<NS, MT, NT>(acc, ah, al, bh, bl);
    };
    for (int it = 0; it < niter; ++it) {
        if (late) {
            mma_frags();
            CDF_SCHED_FENCE();
        }
        fetch(fbuf);                                         // chunk it + NSTAGE - 1
        fbuf = fbuf + 1 == NSTAGE ? 0 : fbuf + 1;
        const unsigned short* sa = smem + buf * STAGE;
        const unsigned short* sb = sa + 2 * PLANE_A;
        buf = buf + 1 == NSTAGE ? 0 : buf + 1;
        read_frags(sa, sb);
        if (!late) mma_frags();
        CDF_WAIT_DMA_LEAVE((NSTAGE - 2) * PIECES);           // this wave's pieces of chunk it + 1 have landed ...
        CDF_LDS_BARRIER();                                   // ... and so have everybody else's; chunk it is fully consumed
    }
    if (late) mma_frags();                                   // the fragments of the last chunk
    CDF_WAIT_DMA_LEAVE(0);                                   // the tail fetches (never read) must not land in the epilogue tile
    CDF_LDS_BARRIER();

    if (a.ksplit > 1) {
        // raw partial sums of this tap share -> slab z (rows m, pitch ks_ld); bias / activation / residual ... run in the finish kernel
        RawEpiArgs r;
        r.Cout = a.Cout; r.vec = (a.Cout & 3) == 0 ? 1 : 0; r.os = 1; r.QH = 1; r.QW = 1; r.OH = 1; r.OW = 1; r.ldy = a.ks_ld;
        r.ldp = r.ldm = r.ldr = r.ld_sbias = r.ld_ys = 0; r.act = 0; r.mul_mode = 0; r.accumulate = 0;
        r.bias = nullptr; r.sbias = nullptr; r.pre = nullptr; r.mul = nullptr; r.res = nullptr; r.ys_hi = nullptr; r.ys_lo = nullptr; r.io_bf = 0;
        constexpr int CP = BN + 8, TM = BM / WM, TN = BN / WN;
        float* cs = (float*)smem_raw;
        const int half_ = lane >> 5, l31_ = lane & 31;
#pragma unroll
        for (int i = 0; i < TM / 32; ++i)
#pragma unroll
            for (int j = 0; j < TN / 32; ++j)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr)
                    cs[(wm * TM + i * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half_) * CP + wn * TN + j * 32 + l31_] = acc[i][j][rr];
        __syncthreads();
        cdf_epilogue_rows<BN, BM, 64 * WM * WN>(r, ph, a.ks_ws + (size_t)blockIdx.z * M * a.ks_ld, cs, tile_m * BM, tile_n * BN, M, tid,
                                                [](int p) { return p; });
        return;
    }
    cdf_sp_epilogue<BM, BN, WM, WN>(a, ph, acc, (float*)smem_raw, tile_m, tile_n, M, tid);
}

// Finish of a split-K launch: y = epilogue(sum_z ws[z][m][:]) for 16 x BN tiles (the epilogue of the GEMM itself: bias, per-sample
// bias, activation + pre-activation, gradient multiply, residual, accumulate, bf16 planes).  Small tiles and all slab loads of an
// element in flight at once: the tensors are a few hundred pixels, the kernel is pure latency.  grid = (row tiles x column tiles), block 256.
template <int BN>
__global__ void __launch_bounds__(256) conv_splitk_finish_kernel(SpxArgs a) {
    constexpr int BM = 16, CP = BN + 8;
    __shared__ __attribute__((aligned(16))) float cs[BM * CP];
    const int tid = threadIdx.x;
    const int M = a.B * a.QH * a.QW;
    const int tiles_n = (a.Cout + BN - 1) / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    constexpr int V = BN / 4;                                // float4 per tile row
    const size_t zs = (size_t)M * a.ks_ld;
    for (int e = tid; e < BM * V; e += 256) {
        const int r = e / V, c4 = (e - r * V) * 4;
        const int m = tile_m * BM + r, n = tile_n * BN + c4;
        const bool ok = m < M && n < a.ks_ld;
        const float* p = a.ks_ws + (ok ? (size_t)m * a.ks_ld + n : 0);
        float4 v[CDF_MAX_TAPS];
#pragma unroll
        for (int z = 0; z < CDF_MAX_TAPS; ++z) v[z] = *(const float4*)(p + (z < a.ksplit ? z : 0) * zs);      // unconditional, clamped
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int z = 0; z < CDF_MAX_TAPS; ++z)
            if (z < a.ksplit) { sum.x += v[z].x; sum.y += v[z].y; sum.z += v[z].z; sum.w += v[z].w; }
        *(float4*)(cs + r * CP + c4) = ok ? sum : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    cdf_epilogue_rows<BN, BM, 256>(a, a.ph[0], a.y, cs, tile_m * BM, tile_n * BN, M, tid, [](int p) { return p; });
}

// ================================================================================================
// 3 x 3 stride-1 convolutions with the INPUT TILE RESIDENT IN LDS ("halo" kernel).
//
// What bounds conv_igemm_spx_kernel is the operand DMA, not the matrix pipe (tools/ablate.py on MI355X, 512 -> 1024 at
// 16 x 16: 0.233 ms; without the DMA 0.131; MFMAs + barriers alone 0.125): every 32-channel K step brings 32 KB into LDS
// for 96 MFMAs, and only ~64 KB per CU are in flight against ~1.1 us of L2 / MALL latency.  Of those bytes half are the
// A tile -- and the nine taps of a 3 x 3 conv fetch the SAME pixels nine times, shifted.  Here the K loop runs channel
// chunk outermost, taps innermost: per 32-channel chunk the tile's pixels plus a one-pixel halo ((TH+2) x (W+2) rows of
// 64 B, both planes) are fetched ONCE, double buffered, and all nine taps read their A fragments out of that image at
// row offset dy (W+2) + dx.  Only the weights still stream per tap (3 stages).  DMA bytes per chunk, 128 x 128 tile:
// 9 x 32 KB -> 144 KB + 24..50 KB.
//
// Tile = TH = 128 / W full image rows (W = 16, 32, 64 or 128: one tile never straddles two images), so the tile's
// pixels are the contiguous range [128 tile_m, 128 tile_m + 128) of the flattened pixel index and the epilogue of the
// generic kernel applies unchanged.  Halo rows outside the image come from the zero page.  8 waves (4 x 2 of 32 x 64).
// Same XOR swizzle of the 16-byte column by (row >> 2) & 3 on both sides; a lane's 16 fragment rows are consecutive
// halo rows except at an image-row wrap (+2), where a 2-way bank conflict can occur.
// ================================================================================================
// how many of the halo segments requested in steps t, t-1, ... t-(n-1) (tap index modulo 9) fall on steps with a request (t' < ta)
constexpr int cdf_halo_parts(int t, int n, int ta) {
    int c = 0;
    for (int d = 0; d < n; ++d) c += ((t - d + 9) % 9) < ta ? 1 : 0;
    return c;
}

template <int W, int BN, int NB, int BM, int NS = 3>                    // NB weight stages: NB - 1 tap steps requested ahead; BM = 128 or 256 pixels
__global__ void __launch_bounds__(512, 1) conv_igemm_halo_kernel(SpxArgs a) {
    constexpr int WM = 4, WN = 2, NW = 8, BK = 32, RE = 32, MT = BM / WM / 32;
    constexpr int TH = BM / W, HW2 = W + 2, HR = (TH + 2) * HW2;          // halo rows (pixels)
    constexpr int NSEG = (HR + 15) / 16, HRP = NSEG * 16;                 // 16-row DMA segments
    constexpr int TA = (NSEG + NW - 1) / NW;                              // tap steps in which a wave fetches one A segment
    static_assert(NB >= 3 && NB <= 7 && TA <= 11 - NB && TA <= 12 - NB, "the next chunk's halo must be requested before the weights of its first tap");
    constexpr int NT = BN / WN / 32;                                      // 32 x 32 MFMA tiles per wave along N (M: 1)
    constexpr int SB = BN / 16 / NW;                                      // B segments per wave and plane (1 for BN = 128)
    static_assert(SB * NW * 16 == BN || BN == 64, "B tile must split into 16-row segments");
    constexpr int SBI = BN == 64 ? 1 : SB;                                // (BN = 64: waves 0..3 fetch a segment, 4..7 repeat them)
    constexpr int PLANE_A = HRP * RE, ABUF = 2 * PLANE_A;                 // (unsigned short units)
    constexpr int PLANE_B = BN * RE, BSTAGE = 2 * PLANE_B;
    CDF_DYN_SMEM(smem_raw);
    unsigned short* smem = (unsigned short*)smem_raw;
    unsigned short* const abuf0 = smem;
    unsigned short* const bst0 = smem + 2 * ABUF;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int M = a.B * a.QH * a.QW;
    const int tiles_n = (a.Cout + BN - 1) / BN, tiles_m = M / BM;
    const int tile = cdf_sp_swizzle(blockIdx.x, tiles_m * tiles_n);
    const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
    const SpPhase& ph = a.ph[0];
    const int tpi = a.H / TH;                                              // tiles per image
    const int img = tile_m / tpi, y0 = (tile_m - img * tpi) * TH;

    // (tap indices are compile-time constants in the unrolled loops below: ph.dy[t] etc. are scalar kernel-argument loads
    // hoisted out of the K loop -- an LDS tap table would put an lgkmcnt(0) wait between the fragment reads and the MFMAs)

    // ---- DMA sources.  A: segment g = wave + 8 q (q < TA; past NSEG the wave repeats segment g mod NSEG -- same bytes to
    // the same place, so that every wave issues the same number of DMA instructions per step and one s_waitcnt count fits all)
    const int srow = lane >> 2;
    const int q8 = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    const unsigned short* pa_hi[TA];
    const unsigned short* pa_lo[TA];
    int a_inc[TA], a_seg[TA];
#pragma unroll
    for (int q = 0; q < TA; ++q) {
        int g = wave + NW * q;
        if (g >= NSEG) g -= (g / NSEG) * NSEG;
        a_seg[q] = g;
        const int r = g * 16 + srow;
        const int hy = r / HW2, hx = r - hy * HW2;
        const int y = y0 - 1 + hy, x = hx - 1;
        const bool ok = r < HR && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)W;
        const size_t off = ((size_t)((img * a.H + y) * W + x)) * (unsigned)a.ldx + (unsigned)q8;
        pa_hi[q] = ok ? a.x_hi + off : a.zero;
        pa_lo[q] = ok ? a.x_lo + off : a.zero;
        a_inc[q] = ok ? BK : 0;
    }
    int b_row[SBI];
#pragma unroll
    for (int p = 0; p < SBI; ++p) {
        const int seg = BN == 64 ? (wave & 3) : wave * SB + p;
        const int n = tile_n * BN + seg * 16 + srow;
        b_row[p] = n < a.Cout ? n : a.Cout - 1;
    }
    const int nchunks = a.Cin / BK;

    auto fetch_a = [&](int q, int buf) {                     // segment a_seg[q] of the chunk the pointers stand at -> halo buffer buf
        unsigned short* seg = abuf0 + buf * ABUF + a_seg[q] * 16 * RE;
        CDF_GLDS16_K(pa_hi[q], seg);
        if constexpr (NS == 3) CDF_GLDS16_K(pa_lo[q], seg + PLANE_A);
    };
    auto advance_a = [&]() {
#pragma unroll
        for (int q = 0; q < TA; ++q) {
            pa_hi[q] += a_inc[q];
            pa_lo[q] += a_inc[q];
        }
    };
    auto fetch_b = [&](int c, int t, int stage) {            // weights of (chunk c, tap t) -> stage (= step % 3 = t % 3)
        if (c >= nchunks) c = nchunks - 1;                   // past the end: valid weights again, into an idle stage
        const int wi = ph.wi[t];
        unsigned short* st = bst0 + stage * BSTAGE;
#pragma unroll
        for (int p = 0; p < SBI; ++p) {
            const int seg = BN == 64 ? (wave & 3) : wave * SB + p;
            const size_t woff = (size_t)((unsigned)wi * (unsigned)a.Cout + (unsigned)b_row[p]) * (unsigned)a.ldk + (unsigned)(c * BK + q8);
            CDF_GLDS16_K(a.w_hi + woff, st + seg * 16 * RE);
            if constexpr (NS == 3) CDF_GLDS16_K(a.w_lo + woff, st + PLANE_B + seg * 16 * RE);
        }
    };
    constexpr int NPL = NS == 3 ? 2 : 1;                     // operand planes in flight (hi [, lo])
    constexpr int PB = NPL * SBI, PA = NPL;                  // DMA instructions per wave: one B step, one A segment

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int half = lane >> 5, l31 = lane & 31;
    // this lane's A fragment rows for tap (0, 0): pixels p = (BM/4) wm + 32 i + l31 of the tile
    int row0[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int pix = wm * (BM / WM) + i * 32 + l31;
        const int py = pix / W, px = pix - py * W;
        row0[i] = (py + 1) * HW2 + px + 1;
    }
    const int swb = (l31 >> 2) & 3;                          // B rows: tile-local, multiples of 32 apart

    // ---- prologue: halo of chunk 0, weights of steps 0 .. NB-2
#pragma unroll
    for (int q = 0; q < TA; ++q) fetch_a(q, 0);
    if (nchunks > 1) advance_a();                            // the pointers stand at the chunk requested next (the last one, at the end)
#pragma unroll
    for (int u = 0; u < NB - 1; ++u) fetch_b(u / 9, u % 9, u);  // (NB - 1 <= 9: all in chunk 0)
    int rd = 0;                                              // weight stage of the current step
    CDF_WAIT_DMA_LEAVE((NB - 2) * PB);                       // the halo and the weights of step 0 have landed
    CDF_LDS_BARRIER();
    bf16x8_v ah[2][MT], al[2][MT], bh[2][NT], bl[2][NT];
    const bool late = a.dephase != 0 && wave >= NW / 2;      // (wave-uniform)
    if (late) {                                              // first step of a late wave: multiplies zeros
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) { ah[ks][i][e] = 0; al[ks][i][e] = 0; }
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) { bh[ks][j][e] = 0; bl[ks][j][e] = 0; }
        }
    }
    auto mma_frags = [&]() { cdf_mma_tile<NS, MT, NT>(acc, ah, al, bh, bl); };
    for (int c = 0; c < nchunks; ++c) {
        const unsigned short* sa = abuf0 + (c & 1) * ABUF;
        // (during the last chunk its own halo is requested again, into the idle buffer: every step issues the same
        // number of DMA instructions, so the wait counts below are compile-time constants)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t < TA) fetch_a(t, (c + 1) & 1);
            if (t == TA - 1 && c + 2 < nchunks) advance_a();
            fetch_b(t + NB - 1 < 9 ? c : c + 1, (t + NB - 1) % 9, rd == 0 ? NB - 1 : rd - 1);   // step + NB-1 -> the stage read last step
            const int tapoff = (int)ph.dy[t] * HW2 + (int)ph.dx[t];
            const unsigned short* sb = bst0 + rd * BSTAGE;
            rd = rd + 1 == NB ? 0 : rd + 1;
            auto read_frags = [&]() {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const int row = row0[i] + tapoff;
                        const int off = row * RE + ((ks * 2 + half) ^ ((row >> 2) & 3)) * 8;
                        ah[ks][i] = *(const bf16x8_v*)(sa + off);
                        if constexpr (NS == 3) al[ks][i] = *(const bf16x8_v*)(sa + PLANE_A + off);
                    }
                    const int kc = ((ks * 2 + half) ^ swb) * 8;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int offb = (wn * (BN / WN) + j * 32 + l31) * RE + kc;
                        bh[ks][j] = *(const bf16x8_v*)(sb + offb);
                        if constexpr (NS == 3) bl[ks][j] = *(const bf16x8_v*)(sb + PLANE_B + offb);
                    }
                }
            };
            // De-phased waves (a.dephase): the block's waves 4..7 share their SIMDs with waves 0..3 and the step barrier keeps all
            // eight in lockstep, so fragment reads (LDS) and MFMAs (matrix pipe) of a SIMD's two waves used to happen one after the
            // other, never together.  Waves 4..7 therefore multiply the fragments they read in the PREVIOUS step first and read this
            // step's fragments afterwards: while one wave of a SIMD multiplies, the other one reads.
            if (late) {
                mma_frags();
                CDF_SCHED_FENCE();                           // (the reads overwrite the fragments just multiplied: hoisting them doubles the live set)
            }
            read_frags();
            if (!late) mma_frags();
            // the weights of step + 1 (requested NB - 2 steps ago) have landed -- and with them, in order, every halo segment
            // requested before them; still in flight: the weight requests of the last NB - 2 steps and the halo segments
            // requested in those steps (a compile-time count per tap index)
            switch (cdf_halo_parts(t, NB - 2, TA)) {
                case 0: CDF_WAIT_DMA_LEAVE((NB - 2) * PB); break;
                case 1: CDF_WAIT_DMA_LEAVE((NB - 2) * PB + PA); break;
                case 2: CDF_WAIT_DMA_LEAVE((NB - 2) * PB + 2 * PA); break;
                case 3: CDF_WAIT_DMA_LEAVE((NB - 2) * PB + 3 * PA); break;
                case 4: CDF_WAIT_DMA_LEAVE((NB - 2) * PB + 4 * PA); break;
                default: CDF_WAIT_DMA_LEAVE((NB - 2) * PB + 5 * PA); break;
            }
            CDF_LDS_BARRIER();
        }
    }
    if (late) mma_frags();                                   // the fragments of the last step
    CDF_WAIT_DMA_LEAVE(0);                                   // the tail requests (never read) must not land in the epilogue tile
    CDF_LDS_BARRIER();

    cdf_sp_epilogue<BM, BN, WM, WN>(a, ph, acc, (float*)smem_raw, tile_m, tile_n, M, tid);
}

// ================================================================================================
// 3 x 3 stride-1 convolutions with 64 / 128 input channels, 256-pixel tiles, the input rows of ONE TAP ROW resident in LDS, resident
// blocks with ONE operand stream over all the tiles of a CU ("row-halo stream" kernel).
//
// The halo kernel above keeps (TH + 2) x (W + 2) pixels per channel chunk; at 128-pixel width two such buffers leave room for the weight
// stages of a 128-wide N tile only with 128-pixel tiles, where it is no faster than the generic kernel.  This form shares the input
// across the three dx taps only: per (channel chunk, dy) it fetches the tile's TH rows shifted by dy with one pixel of halo left and
// right (TH x (W + 2) rows of 64 B, both planes, double buffered), and the three taps of that row read their A fragments at pixel offsets
// -1, 0, +1: 141 bytes of DMA per MFMA against 250 for the generic 256 x 128 tile (64 -> 128 at 128 x 128: 0.325 -> 0.298 ms as one block
// per tile, round 2).  Requires the taps in dy-major order (checked by the host).  Used for the > 64-channel outputs at 128-pixel width.
//
// Round 3: per 256-pixel tile of such a short-K layer (18 tap steps) a one-tile block spent ~24 us in its K loop, ~7 us before it (until
// the first rows and weights have arrived) and ~8 us after it (epilogue until the stores are acknowledged), one block per CU, nothing
// overlapped.  Here a block is resident and walks its tiles (tile j of block b = the XCD-aware index of b + j gridDim.x), and the operand
// pipeline runs on across the tile boundary: with THREE weight stages a tile's 9 NCH steps are a whole number of stage rotations and
// (NCH even) of row-buffer alternations, so the requests a one-tile loop wastes past its last step ARE the next tile's first rows and
// weights, landing in row buffer 0 and weight stages 0, 1 while the epilogue runs.  The epilogue goes in two passes of 128 rows through
// a staging tile that aliases only what is idle then -- row buffer 1, weight stage 2 and the tail of the LDS:
//     LDS:  rows 0 | weights 0 | weights 1 | rows 1 | weights 2 | ...        staging [128][BN + 8] floats from "rows 1" on
// K loop: tap row, chunk, dx, fully unrolled (the two half-line chunks of a 64-channel pixel in consecutive groups: a 32-channel chunk
// is half a 128-byte line, and half-line reads cost full lines); 8 waves (4 x 2 of 64 x 64), late waves de-phased as in the other kernels.
// -6 ... -9 % against the one-tile form (removed in round 4, profiles/round3_rowhalo_stream_ab.txt).  Round 4, measured and not kept:
// blocks walking CONTIGUOUS runs of tiles (profiles/round4_rowhalo_strips_ab.txt).
// ================================================================================================
template <int W, int BN, int NS = 3, int NCH = 2, int BM = 256>
__global__ void __launch_bounds__(512, 1) conv_igemm_rowhalo_stream_kernel(SpxArgs a) {
    constexpr int WM = BM / 64, WN = 8 / WM, NW = 8, BK = 32, RE = 32, MT = BM / WM / 32, NB = 3;
    static_assert(BM == 256 && WM * WN == 8 && MT == 2, "8 waves (4 x 2) of 64-row tiles");
    static_assert(NCH % 2 == 0, "an even number of tap-row groups per tile returns the pipeline to row buffer 0");
    constexpr int EROWS = BM / 2;                                                 // rows per epilogue pass
    constexpr int TH = BM / W, HW2 = W + 2, RH = TH * HW2;
    constexpr int NSEG = (RH + 15) / 16, HRP = NSEG * 16;
    constexpr int TAG = (NSEG + NW - 1) / NW;
    constexpr int NT = BN / WN / 32;
    constexpr int SB = BN / 16 / NW;
    static_assert(SB * NW * 16 == BN || BN == 64, "B tile must split into 16-row segments");
    constexpr int SBI = BN == 64 ? 1 : SB;
    constexpr int PLANE_A = HRP * RE, ABUF = 2 * PLANE_A;
    constexpr int PLANE_B = BN * RE, BSTAGE = 2 * PLANE_B;
    constexpr int OFF_A1 = ABUF + 2 * BSTAGE, OFF_B2 = OFF_A1 + ABUF;          // (elements) rows 0 | weights 0 | weights 1 | rows 1 | weights 2
    CDF_DYN_SMEM(smem_raw);
    unsigned short* smem = (unsigned short*)smem_raw;
    float* const cs = (float*)(smem + OFF_A1);                                   // epilogue staging: rows 1, weights 2 and the tail are idle then
    constexpr int CP = BN + 8;
    static_assert((size_t)OFF_A1 * 2 + (size_t)EROWS * CP * 4 <= 160 * 1024, "the staging tile must fit behind the live operand buffers");

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int M = a.B * a.QH * a.QW;
    const int tiles_n = (a.Cout + BN - 1) / BN, tiles_m = M / BM, ntiles = tiles_m * tiles_n;
    const SpPhase& ph = a.ph[0];
    const int tpi = a.H / TH;

    const int srow = lane >> 2;
    const int q8 = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
    int a_seg[TAG], a_ry[TAG], a_x[TAG];
#pragma unroll
    for (int q = 0; q < TAG; ++q) {
        int g = wave + NW * q;
        if (g >= NSEG) g -= (g / NSEG) * NSEG;
        a_seg[q] = g;
        const int r = g * 16 + srow;
        a_ry[q] = r < RH ? r / HW2 : -(1 << 20);
        a_x[q] = r - (r / HW2) * HW2 - 1;
    }
    // rows of (tile position (img, y0), chunk c, tap row offset dy) -> row buffer buf; img < 0: no such tile, zero page
    auto fetch_a = [&](int img, int y0, int c, int dy, int buf) {
        unsigned short* base = smem + (buf ? OFF_A1 : 0);
#pragma unroll
        for (int q = 0; q < TAG; ++q) {
            const int y = y0 + a_ry[q] + dy;
            const bool ok = img >= 0 && (unsigned)y < (unsigned)a.H && (unsigned)a_x[q] < (unsigned)W;
            const size_t off = ((size_t)(((ok ? img : 0) * a.H + (ok ? y : 0)) * W + (ok ? a_x[q] : 0))) * (unsigned)a.ldx + (unsigned)(c * BK + q8);
            unsigned short* seg = base + a_seg[q] * 16 * RE;
            CDF_GLDS16_K(ok ? a.x_hi + off : a.zero, seg);
            if constexpr (NS == 3) CDF_GLDS16_K(ok ? a.x_lo + off : a.zero, seg + PLANE_A);
        }
    };
    auto fetch_b = [&](int tile_n, int c, int wi, int stage) {       // weights of (N tile, chunk c, tap with weight index wi) -> stage
        unsigned short* st = smem + (stage == 2 ? OFF_B2 : ABUF + stage * BSTAGE);
#pragma unroll
        for (int p = 0; p < SBI; ++p) {
            const int seg = BN == 64 ? (wave & 3) : wave * SB + p;
            const int n = tile_n * BN + seg * 16 + srow;
            const int brow = n < a.Cout ? n : a.Cout - 1;
            const size_t woff = (size_t)((unsigned)wi * (unsigned)a.Cout + (unsigned)brow) * (unsigned)a.ldk + (unsigned)(c * BK + q8);
            CDF_GLDS16_K(a.w_hi + woff, st + seg * 16 * RE);
            if constexpr (NS == 3) CDF_GLDS16_K(a.w_lo + woff, st + PLANE_B + seg * 16 * RE);
        }
    };
    constexpr int NPL = NS == 3 ? 2 : 1;
    constexpr int PB = NPL * SBI, PAG = NPL * TAG;           // DMA instructions per wave: one weight step, one group of rows

    const int half = lane >> 5, l31 = lane & 31;
    int row0[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int pix = wm * (BM / WM) + i * 32 + l31;
        const int py = pix / W, px = pix - py * W;
        row0[i] = py * HW2 + px + 1;
    }
    const int swb = (l31 >> 2) & 3;
    const bool late = a.dephase != 0 && wave >= NW / 2;      // (wave-uniform)

    auto tile_pos = [&](int v, int& img, int& y0, int& tn, int& tm) {    // virtual block id -> tile (img < 0: past the last tile)
        if (v < ntiles) {
            const int tile = cdf_sp_swizzle(v, ntiles);
            tm = tile / tiles_n;
            tn = tile - tm * tiles_n;
            img = tm / tpi;
            y0 = (tm - img * tpi) * TH;
        } else {
            img = -1; y0 = 0; tn = 0; tm = 0;
        }
    };
    int img, y0, tile_n, tile_m;
    int v = blockIdx.x;
    tile_pos(v, img, y0, tile_n, tile_m);
    // ---- pipeline fill: rows of (chunk 0, tap row 0), weights of steps 0, 1 of the first tile
    fetch_a(img, y0, 0, ph.dy[0], 0);
    fetch_b(tile_n, 0, ph.wi[0], 0);
    fetch_b(tile_n, 0, ph.wi[1], 1);
    CDF_WAIT_DMA_LEAVE(PB);                                  // rows and the weights of step 0 have landed
    CDF_LDS_BARRIER();

    while (img >= 0) {
        int img_n, y0_n, tile_n_n, tile_m_n;
        tile_pos(v + gridDim.x, img_n, y0_n, tile_n_n, tile_m_n);
        f32x16_t acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        bf16x8_v ah[2][MT], al[2][MT], bh[2][NT], bl[2][NT];
        if (late) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ah[ks][i][e] = 0; al[ks][i][e] = 0; }
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { bh[ks][j][e] = 0; bl[ks][j][e] = 0; }
            }
        }
        auto mma_frags = [&]() { cdf_mma_tile<NS, MT, NT>(acc, ah, al, bh, bl); };
#pragma unroll
        for (int g = 0; g < 3; ++g) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
#pragma unroll
                for (int i3 = 0; i3 < 3; ++i3) {
                    const int t = 3 * g + i3;
                    const int step = (g * NCH + c) * 3 + i3;                     // 0 .. 9 NCH - 1
                    const int par = (g * NCH + c) & 1, rd = step % NB;
                    // requests: the weights two steps ahead, then (first step of a group) the next group's rows -- past this tile's last
                    // step / group they are the NEXT tile's first ones.  Weights first: loads complete in order, and the rows (from HBM)
                    // are not needed before the end of the group, the weights (from L2) at the end of the next step.
                    {
                        const int s2 = step + 2, gc2 = (s2 / 3) % (3 * NCH);     // group of the step two ahead (wraps into the next tile)
                        const bool over = s2 >= 9 * NCH;
                        const int g2 = gc2 / NCH, c2 = gc2 - g2 * NCH;
                        fetch_b(over ? tile_n_n : tile_n, c2, ph.wi[3 * g2 + s2 % 3], s2 % NB);
                    }
                    if (i3 == 0) {
                        const bool lastc = c + 1 == NCH, over = lastc && g == 2;
                        const int nc = lastc ? 0 : c + 1, ng = over ? 0 : (lastc ? g + 1 : g);
                        fetch_a(over ? img_n : img, over ? y0_n : y0, nc, ph.dy[3 * ng], par ^ 1);
                    }
                    // (the buffer bases as opaque scalars: as constants beyond the 64 KB reach of a ds_read immediate they made hipcc keep one
                    //  precomputed fragment address per (buffer, dx, fragment) live across the whole tile loop -- 58 VGPRs spilled)
                    int sa_e = par ? OFF_A1 : 0, sb_e = rd == 2 ? OFF_B2 : ABUF + rd * BSTAGE;
#ifndef CDF_EMU
                    asm volatile("" : "+s"(sa_e), "+s"(sb_e));
#endif
                    const unsigned short* sa = smem + sa_e;
                    const unsigned short* sb = smem + sb_e;
                    const int dx = ph.dx[t];
                    auto read_frags = [&]() {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                            for (int i = 0; i < MT; ++i) {
                                const int row = row0[i] + dx;
                                const int off = row * RE + ((ks * 2 + half) ^ ((row >> 2) & 3)) * 8;
                                ah[ks][i] = *(const bf16x8_v*)(sa + off);
                                if constexpr (NS == 3) al[ks][i] = *(const bf16x8_v*)(sa + PLANE_A + off);
                            }
                            const int kc = ((ks * 2 + half) ^ swb) * 8;
#pragma unroll
                            for (int j = 0; j < NT; ++j) {
                                const int offb = (wn * (BN / WN) + j * 32 + l31) * RE + kc;
                                bh[ks][j] = *(const bf16x8_v*)(sb + offb);
                                if constexpr (NS == 3) bl[ks][j] = *(const bf16x8_v*)(sb + PLANE_B + offb);
                            }
                        }
                    };
                    if (late) {
                        mma_frags();
                        CDF_SCHED_FENCE();
                    }
                    read_frags();
                    if (!late) mma_frags();
                    // the weights of step + 1 have landed (requested one step ago, before that step's row request); may still be in
                    // flight: this step's weights and the rows requested in this group's first step -- those only at the group's end not
                    if (i3 <= 1)
                        CDF_WAIT_DMA_LEAVE(PB + PAG);
                    else
                        CDF_WAIT_DMA_LEAVE(PB);
                    CDF_LDS_BARRIER();
                }
            }
        }
        if (late) mma_frags();
        // (the last barrier of the loop: every wave is done with this tile's rows and weights; in flight / landed: the next tile's
        // rows 0 and weights 0, 1 -- none of them under the staging tile)
#pragma unroll 1
        for (int hp = 0; hp < 2; ++hp) {
            if ((wm * 64) / EROWS == hp) {                   // this wave's 64 rows belong to pass hp
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            cs[(wm * 64 - hp * EROWS + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * CP + wn * (BN / WN) + j * 32 + l31] = acc[i][j][r];
            }
            CDF_LDS_BARRIER();                                   // (LDS traffic only: the stores of the previous pass keep draining)
            cdf_epilogue_rows<BN, EROWS, 512>(a, ph, a.y, cs, tile_m * BM + hp * EROWS, tile_n * BN, M, tid, [](int p) { return p; });
            CDF_LDS_BARRIER();
        }
        img = img_n; y0 = y0_n; tile_n = tile_n_n; tile_m = tile_m_n;
        v += gridDim.x;
    }
    CDF_WAIT_DMA_LEAVE(0);                                   // (the requests past the last tile fetched the zero page / weights: let them land)
}

// weight gradient with both operands pre-split ([pixels][ld] bf16 hi / lo planes)
struct SpxWgradArgs {
    const unsigned short* a_hi;
    const unsigned short* a_lo;
    const unsigned short* b_hi;
    const unsigned short* b_lo;
    const unsigned short* zero;
    float* out;
    float* bsum;
    int lda, ldb, ldo;
    int B, QH, QW;
    int HA, WA, sa, HB, WB, sb;
    int CA, CB;
    int ntaps, nsplit, m_per_split, xcd_swizzle;
    signed char day[CDF_MAX_TAPS], dax[CDF_MAX_TAPS], dby[CDF_MAX_TAPS], dbx[CDF_MAX_TAPS];
};

template <int NS>
__device__ __forceinline__ void cdf_bf16x8_accum(float* acc8, const u32x4_v& h, const u32x4_v& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if constexpr (NS == 3) {
            acc8[2 * e] += __uint_as_float(h[e] << 16) + __uint_as_float(l[e] << 16);
            acc8[2 * e + 1] += __uint_as_float(h[e] & 0xFFFF0000u) + __uint_as_float(l[e] & 0xFFFF0000u);
        } else {
            acc8[2 * e] += __uint_as_float(h[e] << 16);
            acc8[2 * e + 1] += __uint_as_float(h[e] & 0xFFFF0000u);
        }
    }
}

// Tile TA (ca) x TB (cb), each 64 or 128; 4 waves as 2 x 2 of (TA/2) x (TB/2); BK = 32 pixels.
// The contraction index (pixels) is the SLOW index of both NHWC operands, so the LDS tiles stay pixel-major,
// [32 px][T + 32] bf16 per plane, written with ds_write_b128 exactly as loaded.  The MFMA fragment (8
// consecutive pixels of one channel per lane) comes out of two ds_read_b64_tr_b16 -- gfx950's transposing LDS
// read: the 16 lanes of a group hand in the addresses of a [4 px][16 ch] block (lane t: pixel t>>2, channels
// 4(t&3)..+3) and lane t receives channel t's 4 pixels.  Pitch T+32 puts the 4 pixel rows of a group 16 banks
// apart and the second group of the 32-lane pass 8 banks further: conflict-free.

template <int T>
struct SpxWgradSlot {                  // one operand's share of a thread's loads for a 32-pixel chunk
    static constexpr int VPR = T / 8;              // uint4 per pixel row
    static constexpr int PASS = 32 * VPR / 256;    // T/64
    static constexpr int PPP = 256 / VPR;          // pixels per pass
    static constexpr int PITCH = T + 32;
};

// STACK2 (TA = 128 with CA <= 64): a 64-channel A operand would fill only half of the 128 MFMA rows, so the tile takes
// TWO taps -- rows 0..63 = tap 2*blockIdx.y, rows 64..127 = tap 2*blockIdx.y + 1 (all-zero when past the last tap).  The B
// rows are shared: valid when every tap reads B at the same offset (plain convolutions; checked by the host).
template <int TA, int TB, bool STACK2 = false, int NS = 3>
__global__ void __launch_bounds__(256, 2) conv_wgrad_spx_kernel(SpxWgradArgs a) {
    using SA = SpxWgradSlot<TA>;
    using SB = SpxWgradSlot<TB>;
    constexpr int BK = 32, MT = TA / 64, NT = TB / 64;
    constexpr int PLANE_A = BK * SA::PITCH, PLANE_B = BK * SB::PITCH;
    constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_B;
    CDF_DYN_SMEM(smem_raw);
    unsigned short* smem = (unsigned short*)smem_raw;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_b = (a.CB + TB - 1) / TB;
    int bx, by, bz;
    cdf_wgrad_block(a.xcd_swizzle, bx, by, bz);
    const int tile_a = bx / tiles_b, tile_b = bx - tile_a * tiles_b;
    const int tap = STACK2 ? 2 * by : by, split = bz;
    const int M = a.B * a.QH * a.QW;
    const int m_lo = split * a.m_per_split;
    int m_hi = m_lo + a.m_per_split;
    if (m_hi > M) m_hi = M;
    const int niter = m_hi > m_lo ? (m_hi - m_lo + BK - 1) / BK : 0;
    const int dby = a.dby[tap], dbx = a.dbx[tap];
    // load slots: operand X, pass p: pixel (tid / VPR) + PPP p of the chunk, 16-byte channel column (tid % VPR) * 8
    static_assert(!STACK2 || TA == 128, "tap stacking fills a 128-row A tile with two 64-channel taps");
    const int a_half = STACK2 ? ((tid % SA::VPR) >> 3) : 0;                 // which of the two stacked taps this lane loads
    const int tap_l = tap + a_half;
    const bool tap_ok = tap_l < a.ntaps;
    const int day = a.day[tap_ok ? tap_l : tap], dax = a.dax[tap_ok ? tap_l : tap];
    const int ca = STACK2 ? ((tid % SA::VPR) & 7) * 8 : tile_a * TA + (tid % SA::VPR) * 8;
    const int cb = tile_b * TB + (tid % SB::VPR) * 8;
    int qa[SA::PASS][3], qb[SB::PASS][3];
#pragma unroll
    for (int p = 0; p < SA::PASS; ++p) {
        const int m = m_lo + tid / SA::VPR + SA::PPP * p;
        qa[p][0] = m % a.QW;
        const int t2 = m / a.QW;
        qa[p][1] = t2 % a.QH;
        qa[p][2] = t2 / a.QH;
    }
#pragma unroll
    for (int p = 0; p < SB::PASS; ++p) {
        const int m = m_lo + tid / SB::VPR + SB::PPP * p;
        qb[p][0] = m % a.QW;
        const int t2 = m / a.QW;
        qb[p][1] = t2 % a.QH;
        qb[p][2] = t2 / a.QH;
    }
    const bool do_bsum = a.bsum != nullptr && tile_a == 0 && tap == 0;
    float bs_acc[SB::PASS][8];
#pragma unroll
    for (int p = 0; p < SB::PASS; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) bs_acc[p][e] = 0.f;

    // Everything below is straight-line code on purpose: a divergent branch or loop between the loads makes hipcc
    // wait for the loads already in flight before it (measured: the prefetch of a chunk degenerates into four
    // dependent round trips).  The (qx, qy, b) carry uses an exact float reciprocal: q + 0.5 is never a multiple
    // of the divisor and both stay tiny (q < QW + 32), so the truncation is exact.
    const float rcp_qw = 1.0f / (float)a.QW, rcp_qh = 1.0f / (float)a.QH;
    auto advance = [&](int* q) {
        const int x = q[0] + BK;
        const int cx = (int)(((float)x + 0.5f) * rcp_qw);
        q[0] = x - cx * a.QW;
        const int y = q[1] + cx;
        const int cy = (int)(((float)y + 0.5f) * rcp_qh);
        q[1] = y - cy * a.QH;
        q[2] += cy;
    };
    u32x4_v rah[SA::PASS], ral[SA::PASS], rbh[SB::PASS], rbl[SB::PASS];   // (arrays of HIP uint4 structs would live in scratch)
    auto load_global = [&](int it) {
        const int m0 = m_lo + it * BK;
#pragma unroll
        for (int p = 0; p < SA::PASS; ++p) {
            // an operand row is zero when its own tap falls outside its image: the product then vanishes whatever
            // the other side holds (and the B rows stay intact for the fused bias gradient)
            const int m = m0 + tid / SA::VPR + SA::PPP * p;
            const unsigned ay = (unsigned)(qa[p][1] * a.sa + day), ax = (unsigned)(qa[p][0] * a.sa + dax);
            const bool ok = m < m_hi && ay < (unsigned)a.HA && ax < (unsigned)a.WA && ca < a.CA && tap_ok;
            const unsigned pix = ((unsigned)qa[p][2] * (unsigned)a.HA + ay) * (unsigned)a.WA + ax;
            const size_t off = (size_t)pix * (unsigned)a.lda + (unsigned)ca;
            rah[p] = *(const u32x4_v*)(ok ? a.a_hi + off : a.zero);
            if constexpr (NS == 3) ral[p] = *(const u32x4_v*)(ok ? a.a_lo + off : a.zero);
            advance(qa[p]);
        }
#pragma unroll
        for (int p = 0; p < SB::PASS; ++p) {
            const int m = m0 + tid / SB::VPR + SB::PPP * p;
            const unsigned by = (unsigned)(qb[p][1] * a.sb + dby), bx = (unsigned)(qb[p][0] * a.sb + dbx);
            const bool ok = m < m_hi && by < (unsigned)a.HB && bx < (unsigned)a.WB && cb < a.CB;
            const unsigned pix = ((unsigned)qb[p][2] * (unsigned)a.HB + by) * (unsigned)a.WB + bx;
            const size_t off = (size_t)pix * (unsigned)a.ldb + (unsigned)cb;
            rbh[p] = *(const u32x4_v*)(ok ? a.b_hi + off : a.zero);
            if constexpr (NS == 3) rbl[p] = *(const u32x4_v*)(ok ? a.b_lo + off : a.zero);
            advance(qb[p]);
        }
    };
    auto store_lds = [&](int buf) {
        unsigned short* st = smem + buf * STAGE;
#pragma unroll
        for (int p = 0; p < SA::PASS; ++p) {
            const int so = (tid / SA::VPR + SA::PPP * p) * SA::PITCH + (tid % SA::VPR) * 8;
            *(u32x4_v*)(st + so) = rah[p];
            if constexpr (NS == 3) *(u32x4_v*)(st + PLANE_A + so) = ral[p];
        }
#pragma unroll
        for (int p = 0; p < SB::PASS; ++p) {
            const int so = (tid / SB::VPR + SB::PPP * p) * SB::PITCH + (tid % SB::VPR) * 8;
            *(u32x4_v*)(st + 2 * PLANE_A + so) = rbh[p];
            if constexpr (NS == 3) *(u32x4_v*)(st + 2 * PLANE_A + PLANE_B + so) = rbl[p];
            if (do_bsum) cdf_bf16x8_accum<NS>(bs_acc[p], rbh[p], rbl[p]);     // here the loads have landed anyway
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
    // transposing-read lane geometry: group g = lane >> 4 -> channel block 16 (g & 1), pixel block 8 (g >> 1)
    const int t16 = lane & 15, g16 = lane >> 4;
    const int tr_row = (g16 >> 1) * 8 + (t16 >> 2), tr_col = (g16 & 1) * 16 + (t16 & 3) * 4;
    const int tra = tr_row * SA::PITCH + wm * (TA / 2) + tr_col;
    const int trb = tr_row * SB::PITCH + wn * (TB / 2) + tr_col;
    if (niter > 0) {
        load_global(0);
        store_lds(0);
    }
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const int buf = it & 1;
        if (it + 1 < niter) load_global(it + 1);
        const unsigned short* sa = smem + buf * STAGE;
        const unsigned short* sb = sa + 2 * PLANE_A;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_v ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const unsigned short* pa = sa + tra + ks * 16 * SA::PITCH + i * 32;
                const bf16x4_v h0 = cdf_lds_read_tr16(pa), h1 = cdf_lds_read_tr16(pa + 4 * SA::PITCH);
                ah[i] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                if constexpr (NS == 3) {
                    const bf16x4_v l0 = cdf_lds_read_tr16(pa + PLANE_A), l1 = cdf_lds_read_tr16(pa + PLANE_A + 4 * SA::PITCH);
                    al[i] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const unsigned short* pb = sb + trb + ks * 16 * SB::PITCH + j * 32;
                const bf16x4_v h0 = cdf_lds_read_tr16(pb), h1 = cdf_lds_read_tr16(pb + 4 * SB::PITCH);
                bh[j] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                if constexpr (NS == 3) {
                    const bf16x4_v l0 = cdf_lds_read_tr16(pb + PLANE_B), l1 = cdf_lds_read_tr16(pb + PLANE_B + 4 * SB::PITCH);
                    bl[j] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    cdf_mma_sp<NS>(acc[i][j], ah[i], al[i], bh[j], bl[j]);
                }
        }
        if (it + 1 < niter) store_lds(buf ^ 1);
        __syncthreads();
    }

    float* red = (float*)smem_raw;
    if (do_bsum) {                                 // [32 px][TB] partial column sums -> one row
#pragma unroll
        for (int p = 0; p < SB::PASS; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(tid / SB::VPR + SB::PPP * p) * TB + (tid % SB::VPR) * 8 + e] = bs_acc[p][e];
        __syncthreads();
        for (int c = tid; c < TB; c += 256) {
            float t = 0.f;
            for (int k = 0; k < BK; ++k) t += red[k * TB + c];
            const int cc = tile_b * TB + c;
            if (cc < a.ldo) a.bsum[(long long)split * a.ldo + cc] = cc < a.CB ? t : 0.f;
        }
        __syncthreads();
    }
    // accumulators -> LDS [TA][TB + 8] -> float4 rows of the split's partial-sum slab (see cdf_epilogue.h)
    constexpr int CP = TB + 8;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                red[(wm * (TA / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * CP + wn * (TB / 2) + j * 32 + l31] = acc[i][j][r];
    __syncthreads();
    float* O = a.out + ((long long)split * a.ntaps + tap) * a.CA * a.ldo;
    constexpr int TPR = TB / 4, RPS = 256 / TPR;
    const int c4 = (tid % TPR) * 4, col = tile_b * TB + c4;
    if (col < a.ldo) {
        for (int r = tid / TPR; r < TA; r += RPS) {
            int row = tile_a * TA + r;
            if (STACK2) {                          // tile row r = (stacked tap r >> 6, channel r & 63): the next tap's slab follows
                if (tap + (r >> 6) >= a.ntaps || (r & 63) >= a.CA) continue;
                row = (r >> 6) * a.CA + (r & 63);
            } else if (row >= a.CA) break;
            float4 v = *(const float4*)(red + r * CP + c4);
            if (col + 3 >= a.CB) {                 // zero the pitch padding (ldo % 4 == 0 keeps the store in bounds)
                if (col + 0 >= a.CB) v.x = 0.f;
                if (col + 1 >= a.CB) v.y = 0.f;
                if (col + 2 >= a.CB) v.z = 0.f;
                v.w = 0.f;
            }
            *(float4*)(O + (long long)row * a.ldo + col) = v;
        }
    }
}

// ================================================================================================
// Weight gradient of 3 x 3 stride-1 "same" convolutions, one block per ROW OF TAPS (dy fixed; dx = -1, 0, +1).
//
// conv_wgrad_spx_kernel gives every tap its own block, and each of the nine loads the same dY tile and a one-pixel-shifted
// X tile: like the forward GEMM it is bound by the bytes its waves have to push through the vector-memory
// path per MFMA.  Here a 32-pixel chunk (always inside one image row, or two rows of a 16-pixel-wide image) brings in dY ONCE
// and X ONCE with a pixel of halo on either side ([34 or 36 px][TA]), and the three dx taps read their X fragments from that
// tile at pixel offsets 0, 1, 2: a third of the loads (and of the per-chunk address arithmetic) per MFMA.  Pixel addresses are
// linear in the chunk index (no per-tap decode); the row / image borders are a per-lane mask.  8 waves (32 x TB/WB tiles, three
// accumulator sets), one block per CU; grid (tiles, 3 tap rows, splits) in the XCD-aware order of cdf_wgrad_block.
// ================================================================================================
template <int TA, int TB, int NS = 3>
__global__ void __launch_bounds__(512, 1) conv_wgrad_row3_kernel(SpxWgradArgs a) {
    constexpr int BK = 32, NTHR = 512;
    constexpr int WA_ = TA / 32, WB_ = 8 / WA_, TNW = TB / WB_, NT = TNW / 32;
    static_assert(NT >= 1 && NT * 32 == TNW, "wave tile along B must be a multiple of 32 channels");
    constexpr int NRAP = 36;                                   // halo rows: 34 (W >= 32) or 2 x 18 (W = 16)
    constexpr int PITCH_A = TA + 32, PITCH_B = TB + 32;
    constexpr int PLANE_A = NRAP * PITCH_A, PLANE_B = BK * PITCH_B;
    constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_B;
    constexpr int VPR_A = TA / 8, RPP_A = NTHR / VPR_A, PASS_A = (NRAP + RPP_A - 1) / RPP_A;
    constexpr int VPR_B = TB / 8;                              // (512 / VPR_B >= 32 rows: one pass)
    CDF_DYN_SMEM(smem_raw);
    unsigned short* smem = (unsigned short*)smem_raw;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave / WB_, wb = wave % WB_;
    const int tiles_b = (a.CB + TB - 1) / TB;
    int bx, by, bz;
    cdf_wgrad_block(a.xcd_swizzle, bx, by, bz);
    const int tile_a = bx / tiles_b, tile_b = bx - tile_a * tiles_b;
    const int grp = by, split = bz;                            // tap row: taps 3 grp .. 3 grp + 2 share day
    const int W = a.QW, H = a.QH;
    const int M = a.B * H * W;
    const int m_lo = split * a.m_per_split;
    int m_hi = m_lo + a.m_per_split;
    if (m_hi > M) m_hi = M;
    const int niter = m_hi > m_lo ? (m_hi - m_lo) / BK : 0;    // (M and m_per_split are multiples of 32)
    const int dy = a.day[3 * grp];
    const int cw = W < 32 ? W : 32, rps = cw + 2;              // chunk row width, halo rows per image row
    const int nra = (32 / cw) * rps;

    // ---- load slots.  A: halo row r = (sub-row, xr): pixel (y + sub + dy, x0 + xr - 1); B: chunk pixel pb
    int a_r[PASS_A], a_sub[PASS_A], a_xr[PASS_A];
    const int ca = tile_a * TA + (tid % VPR_A) * 8;
#pragma unroll
    for (int p = 0; p < PASS_A; ++p) {
        a_r[p] = tid / VPR_A + RPP_A * p;
        a_sub[p] = a_r[p] / rps;
        a_xr[p] = a_r[p] - a_sub[p] * rps;
    }
    const int pb = tid / VPR_B;
    const int cb = tile_b * TB + (tid % VPR_B) * 8;
    const bool b_lane = pb < BK;
    // chunk position (wave-uniform): x0 = first pixel's column, yc = its image row
    int x0 = m_lo % W, yc = (m_lo / W) % H;
    const bool do_bsum = a.bsum != nullptr && tile_a == 0 && grp == 0;
    float bs_acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bs_acc[e] = 0.f;

    u32x4_v rah[PASS_A], ral[PASS_A], rbh, rbl;
    auto load_global = [&](int it) {
        const int m0 = m_lo + it * BK;
#pragma unroll
        for (int p = 0; p < PASS_A; ++p) {
            const unsigned ax = (unsigned)(x0 + a_xr[p] - 1), ay = (unsigned)(yc + a_sub[p] + dy);
            const bool ok = a_r[p] < nra && ax < (unsigned)W && ay < (unsigned)H && ca < a.CA;
            const long long pix = (long long)m0 + (a_sub[p] + dy) * W + a_xr[p] - 1;
            const size_t off = (size_t)(ok ? pix : 0) * (unsigned)a.lda + (unsigned)ca;
            rah[p] = *(const u32x4_v*)(ok ? a.a_hi + off : a.zero);
            if constexpr (NS == 3) ral[p] = *(const u32x4_v*)(ok ? a.a_lo + off : a.zero);
        }
        {
            const bool ok = b_lane && cb < a.CB;
            const size_t off = (size_t)(m0 + (b_lane ? pb : 0)) * (unsigned)a.ldb + (unsigned)cb;
            rbh = *(const u32x4_v*)(ok ? a.b_hi + off : a.zero);
            if constexpr (NS == 3) rbl = *(const u32x4_v*)(ok ? a.b_lo + off : a.zero);
        }
        // next chunk (uniform scalars, selects only): 32 pixels further -- inside the row, to the next row(s), to the next image
        const int nx = x0 + (W < 32 ? 0 : 32);
        const int wrap = nx >= W ? 1 : 0;
        x0 = wrap ? 0 : nx;
        yc += (W < 32 ? 32 / W : 0) + wrap;
        yc = yc >= H ? yc - H : yc;
    };
    auto store_lds = [&](int buf) {
        unsigned short* st = smem + buf * STAGE;
#pragma unroll
        for (int p = 0; p < PASS_A; ++p) {
            if (a_r[p] < NRAP) {
                const int so = a_r[p] * PITCH_A + (tid % VPR_A) * 8;
                *(u32x4_v*)(st + so) = rah[p];
                if constexpr (NS == 3) *(u32x4_v*)(st + PLANE_A + so) = ral[p];
            }
        }
        if (b_lane) {
            const int so = pb * PITCH_B + (tid % VPR_B) * 8;
            *(u32x4_v*)(st + 2 * PLANE_A + so) = rbh;
            if constexpr (NS == 3) *(u32x4_v*)(st + 2 * PLANE_A + PLANE_B + so) = rbl;
            if (do_bsum) cdf_bf16x8_accum<NS>(bs_acc, rbh, rbl);
        }
    };

    f32x16_t acc[3][NT];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int half = lane >> 5, l31 = lane & 31;
    const int t16 = lane & 15, g16 = lane >> 4;
    const int tr_row = (g16 >> 1) * 8 + (t16 >> 2), tr_col = (g16 & 1) * 16 + (t16 & 3) * 4;
    const int trb = tr_row * PITCH_B + wb * TNW + tr_col;
    // halo row of chunk pixel p for tap dx: p + 1 + dx (+ 2 from the second image row of a 16-wide chunk on)
    int tra[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) tra[i] = (tr_row + 1 + (int)a.dax[3 * grp + i]) * PITCH_A + wa * 32 + tr_col;
    const int ks_skip = W < 32 ? 2 * PITCH_A : 0;              // k-step 1 = pixels 16..31 = the second row when W = 16

    if (niter > 0) {
        load_global(0);
        store_lds(0);
    }
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const int buf = it & 1;
        if (it + 1 < niter) load_global(it + 1);
        const unsigned short* sa = smem + buf * STAGE;
        const unsigned short* sb = sa + 2 * PLANE_A;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_v bh[NT], bl[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const unsigned short* q = sb + trb + ks * 16 * PITCH_B + j * 32;
                const bf16x4_v h0 = cdf_lds_read_tr16(q), h1 = cdf_lds_read_tr16(q + 4 * PITCH_B);
                bh[j] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                if constexpr (NS == 3) {
                    const bf16x4_v l0 = cdf_lds_read_tr16(q + PLANE_B), l1 = cdf_lds_read_tr16(q + PLANE_B + 4 * PITCH_B);
                    bl[j] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const unsigned short* q = sa + tra[i] + ks * (16 * PITCH_A + ks_skip);
                const bf16x4_v h0 = cdf_lds_read_tr16(q), h1 = cdf_lds_read_tr16(q + 4 * PITCH_A);
                const bf16x8_v ah = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                bf16x8_v al;
                if constexpr (NS == 3) {
                    const bf16x4_v l0 = cdf_lds_read_tr16(q + PLANE_A), l1 = cdf_lds_read_tr16(q + PLANE_A + 4 * PITCH_A);
                    al = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    cdf_mma_sp<NS>(acc[i][j], ah, al, bh[j], bl[j]);
                }
            }
        }
        if (it + 1 < niter) store_lds(buf ^ 1);
        __syncthreads();
    }

    float* red = (float*)smem_raw;
    if (do_bsum) {                                 // [32 px][TB] partial column sums -> one row
        if (b_lane) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[pb * TB + (tid % VPR_B) * 8 + e] = bs_acc[e];
        }
        __syncthreads();
        for (int c = tid; c < TB; c += NTHR) {
            float t = 0.f;
            for (int k = 0; k < BK; ++k) t += red[k * TB + c];
            const int cc = tile_b * TB + c;
            if (cc < a.ldo) a.bsum[(long long)split * a.ldo + cc] = cc < a.CB ? t : 0.f;
        }
        __syncthreads();
    }
    // accumulators of one tap -> LDS [TA][TB + 8] -> float4 rows of that tap's slab; three times
    constexpr int CP = TB + 8, TPR = TB / 4, RPS = NTHR / TPR;
    const int c4 = (tid % TPR) * 4, col = tile_b * TB + c4;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                red[(wa * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * CP + wb * TNW + j * 32 + l31] = acc[i][j][r];
        __syncthreads();
        float* O = a.out + ((long long)split * a.ntaps + 3 * grp + i) * a.CA * a.ldo;
        if (col < a.ldo) {
            for (int r = tid / TPR; r < TA; r += RPS) {
                const int row = tile_a * TA + r;
                if (row >= a.CA) break;
                float4 v = *(const float4*)(red + r * CP + c4);
                if (col + 3 >= a.CB) {             // zero the pitch padding (ldo % 4 == 0 keeps the store in bounds)
                    if (col + 0 >= a.CB) v.x = 0.f;
                    if (col + 1 >= a.CB) v.y = 0.f;
                    if (col + 2 >= a.CB) v.z = 0.f;
                    v.w = 0.f;
                }
                *(float4*)(O + (long long)row * a.ldo + col) = v;
            }
        }
        __syncthreads();
    }
}

// ================================================================================================
// LinearAttention forward, k | v projection + context in ONE pass (round 2).
//
// kv = xn . Wkv^T is a K = dim <= 128 GEMM onto 256 channels: as a convolution launch it writes 1 KB per pixel, and the context
// pass (softmax over the pixels of k, then k~^T v per head) reads it all back.  Here a block keeps a 128-pixel x 256-channel
// tile -- ALL of k | v of its pixels -- : in-kernel-split operands exactly as conv_igemm_sp_kernel (xn fp32 -> bf16 hi / lo while it
// is staged, weights pre-split), accumulators -> an LDS staging tile [128][256] that aliases the operand stages, from which
//   * the tile leaves as full 1 KB rows (kv is still needed by the backward pass), and
//   * each head's context partial is updated on the spot (online softmax: running column max m, acc = acc * exp(m_old - m) +
//     exp(k - m)^T v on the fp32 matrix cores, running column sums), two waves per head, 64 pixels each.
// A block walks a contiguous run of tiles of ONE image and writes one partial per head at the end; cdf_linattn_finalize
// (k_attn.hip) folds the partials of an image.  k and v are never re-read: 537 -> 0 MB per micro-batch at 128 x 128.
// 8 waves: GEMM wave (wm = w / 4: pixel half, wn = w % 4: channel quarter), context wave (head w / 2, pixel half w % 2).
// ================================================================================================
struct KvCtxArgs {
    const float* xn;
    const unsigned short* w_hi;
    const unsigned short* w_lo;
    float* kv;
    float* max_part;      // [B][P][HD]
    float* ctx_part;      // [B][P][heads][32][32]
    float* sum_part;      // [B][P][HD]
    int ldx, ldk, ldkv;
    int n, dim, P, tiles_per_block;
};

// BK = 64 when dim % 64 == 0 (a dim = 64 tile is ONE chunk: all of the next tile's operands travel during the current tile's store /
// context phase), else 32.  One LDS operand stage (the next chunk waits in registers), aliased by the staging tile.
template <int SPLIT, int BK>
__global__ void __launch_bounds__(512, 1) linattn_kvctx_kernel(KvCtxArgs a) {
    constexpr int BM = 128, BN = 256, AS = BK + 8;           // AS: LDS row stride in bf16 elements
    constexpr int NPL = SPLIT == 1 ? 1 : 2;
    constexpr int PLANE_A = BM * AS, PLANE_B = BN * AS;
    constexpr int SP = BN + 8;                               // staging row pitch (floats)
    constexpr int HD = 128, LD = 32;
    constexpr int AV = BK / 4, AQ = BM * AV / 512;           // float4 per A row, A loads per thread
    constexpr int BV = BK / 8, BQ = BN * BV / 512;           // uint4 per B row and plane, B loads per thread and plane
    CDF_DYN_SMEM(smem_raw);
    unsigned short* smem = (unsigned short*)smem_raw;        // operand stage [A planes | B planes] ...
    float* stg = (float*)smem_raw;                           // ... aliased by the [128][SP] fp32 staging tile
    float* sstat = (float*)(smem_raw + (size_t)BM * SP * sizeof(float));     // [2 pixel halves][128] column maxima of k

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y, p = blockIdx.x;
    const int tiles = a.n / BM;
    const int t_lo = p * a.tiles_per_block;
    int t_hi = t_lo + a.tiles_per_block;
    if (t_hi > tiles) t_hi = tiles;
    const int nch = a.dim / BK;
    const float* xb = a.xn + (size_t)b * a.n * a.ldx;
    float* kvb = a.kv + (size_t)b * a.n * a.ldkv;

    const int a_row = tid / AV, a_c4 = (tid % AV) * 4;       // + (512 / AV) rows per further load
    const int b_row = tid / BV, b_q = tid % BV;
    f32x4_t ra[AQ];
    u32x4_v rbh[BQ], rbl[BQ];
#pragma unroll
    for (int q = 0; q < BQ; ++q) rbl[q] = u32x4_v{0u, 0u, 0u, 0u};
    auto load_chunk = [&](int tile, int c) {
        const float* xa = xb + (size_t)tile * BM * a.ldx + c * BK + a_c4;
#pragma unroll
        for (int q = 0; q < AQ; ++q) ra[q] = *(const f32x4_t*)(xa + (size_t)(a_row + (512 / AV) * q) * a.ldx);
#pragma unroll
        for (int q = 0; q < BQ; ++q) {
            const size_t off = (size_t)(b_row + (512 / BV) * q) * a.ldk + c * BK + b_q * 8;
            rbh[q] = *(const u32x4_v*)(a.w_hi + off);
            if (SPLIT > 1) rbl[q] = *(const u32x4_v*)(a.w_lo + off);
        }
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            uint2 hi, lo;
            const float4 v = make_float4(ra[q].x, ra[q].y, ra[q].z, ra[q].w);
            if (SPLIT > 1) {
                cdf_split4_trunc(v, hi, lo);
            } else {
                hi.x = cdf_f2bf(v.x) | (cdf_f2bf(v.y) << 16);
                hi.y = cdf_f2bf(v.z) | (cdf_f2bf(v.w) << 16);
                lo = hi;
            }
            const int off = (a_row + (512 / AV) * q) * AS + a_c4;
            *(uint2*)(smem + off) = hi;
            if (SPLIT > 1) *(uint2*)(smem + PLANE_A + off) = lo;
        }
        unsigned short* sb = smem + NPL * PLANE_A;
#pragma unroll
        for (int q = 0; q < BQ; ++q) {
            const int off = (b_row + (512 / BV) * q) * AS + b_q * 8;
            *(u32x4_v*)(sb + off) = rbh[q];
            if (SPLIT > 1) *(u32x4_v*)(sb + PLANE_B + off) = rbl[q];
        }
    };

    // ---- context state: waves 0-3 own one head each (all 128 pixels of a tile); waves 4-7 stream the tile out meanwhile
    const bool ctx_wave = wave < 4;
    const int ch = wave & 3;
    f32x16_t cacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) cacc[r] = 0.f;
    float m_run = -3.0e38f, psum = 0.f;                      // lane (i = l31): column d = i of this head (both pixel parities hold m_run)

    if (t_lo < t_hi) load_chunk(t_lo, 0);
    for (int tile = t_lo; tile < t_hi; ++tile) {
        f32x16_t acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int c = 0; c < nch; ++c) {
            if (c > 0) __syncthreads();                       // every wave is done with the previous chunk's fragments
            store_lds();
            __syncthreads();
            // what is needed next travels during the MFMAs (and, for the last chunk, during the store / context phase)
            if (c + 1 < nch) load_chunk(tile, c + 1);
            else if (tile + 1 < t_hi) load_chunk(tile + 1, 0);
            const unsigned short* sa = smem;
            const unsigned short* sb = sa + NPL * PLANE_A;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int k0 = ks * 16 + half * 8;
                bf16x8_v ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int off = (wm * 64 + i * 32 + l31) * AS + k0;
                    ah[i] = *(const bf16x8_v*)(sa + off);
                    if (SPLIT > 1) al[i] = *(const bf16x8_v*)(sa + PLANE_A + off);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int off = (wn * 64 + j * 32 + l31) * AS + k0;
                    bh[j] = *(const bf16x8_v*)(sb + off);
                    if (SPLIT > 1) bl[j] = *(const bf16x8_v*)(sb + PLANE_B + off);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (SPLIT > 1) {
                            acc[i][j] = CDF_MFMA_BF16(al[i], bh[j], acc[i][j]);
                            acc[i][j] = CDF_MFMA_BF16(ah[i], bl[j], acc[i][j]);
                        }
                        acc[i][j] = CDF_MFMA_BF16(ah[i], bh[j], acc[i][j]);
                    }
            }
        }
        __syncthreads();                                      // the operand stage is free: it becomes the staging tile
        // ---- accumulators -> staging tile [pixel][channel]; the k waves (channel quarters 0, 1) leave the column maxima of their
        //      64 pixels next to it
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stg[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * SP + wn * 64 + j * 32 + l31] = acc[i][j][r];
        if (wn < 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float m = -3.0e38f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[i][j][r]);
                m = fmaxf(m, __shfl_xor(m, 32));
                if (half == 0) sstat[wm * HD + wn * 64 + j * 32 + l31] = m;
            }
        }
        __syncthreads();
        if (!ctx_wave) {
            // ---- k | v rows out: 64 float4 = one 1 KB row per wave instruction (waves 4-7: the stores' back-pressure stalls nobody else)
            float* dst = kvb + (size_t)tile * BM * a.ldkv;
#pragma unroll 4
            for (int e = tid - 256; e < BM * (BN / 4); e += 256) {
                const int px = e >> 6, c4 = (e & 63) * 4;
                *(float4*)(dst + (size_t)px * a.ldkv + c4) = *(const float4*)(stg + px * SP + c4);
            }
        } else {
            // ---- context of head ch: acc = acc * exp(m_old - m) + exp(k - m)^T v over the tile's 128 pixels
            const float* kcol = stg + half * SP + ch * LD + l31;                       // pixel 2 s + half, column d = l31
            const float* vcol = kcol + HD;
            const float m_new = fmaxf(m_run, fmaxf(sstat[ch * LD + l31], sstat[HD + ch * LD + l31]));
            const float f = expf(m_run - m_new);                                       // (first tile: exp(-inf) = 0 on zero accumulators)
            // accumulator row of register r is d = (r & 3) + 8 (r >> 2) + 4 half: its factor lives in lane d
#pragma unroll
            for (int r = 0; r < 16; ++r) cacc[r] *= __shfl(f, (r & 3) + 8 * (r >> 2) + 4 * half);
            psum *= f;
            m_run = m_new;
#pragma unroll 8
            for (int sx = 0; sx < 64; ++sx) {
                const float pk = expf(kcol[2 * sx * SP] - m_new);
                psum += pk;
                cacc = __builtin_amdgcn_mfma_f32_32x32x2f32(pk, vcol[2 * sx * SP], cacc, 0, 0, 0);
            }
        }
        __syncthreads();                                      // the staging tile (and sstat) are rewritten by the next trip
    }
    // ---- one partial per (block, head)
    psum += __shfl_xor(psum, 32);
    float* fold = stg;                                        // [4 heads][32][32], then [4][32] sums
    if (ctx_wave) {
#pragma unroll
        for (int r = 0; r < 16; ++r) fold[(ch * LD + (r & 3) + 8 * (r >> 2) + 4 * half) * LD + l31] = cacc[r];
        if (half == 0) fold[4 * LD * LD + ch * LD + l31] = psum;
    }
    __syncthreads();
    const size_t pb = (size_t)b * a.P + p;
    for (int e = tid; e < 4 * LD * LD; e += 512) a.ctx_part[pb * 4 * (LD * LD) + e] = t_lo < t_hi ? fold[e] : 0.f;
    if (tid < HD) a.sum_part[pb * HD + tid] = t_lo < t_hi ? fold[4 * LD * LD + tid] : 0.f;
    if (ctx_wave && half == 0) a.max_part[pb * HD + ch * LD + l31] = m_run;
}

__global__ void pack_weight_bf16_kernel(const float* src, unsigned short* dst_hi, unsigned short* dst_lo, int T, int R, int C,
                                        int ldc, long long s_t, long long s_r, long long s_c) {
    const long long n = (long long)T * R * ldc;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ldc);
        const long long tr = i / ldc;
        const int r = (int)(tr % R), t = (int)(tr / R);
        const float v = c < C ? src[c * s_c + r * s_r + t * s_t] : 0.f;
        const unsigned h = cdf_f2bf(v);
        dst_hi[i] = (unsigned short)h;
        if (dst_lo) dst_lo[i] = (unsigned short)cdf_f2bf(v - cdf_bf2f(h));
    }
}

// ================================================================================================
// blocks per image of cdf_linattn_kvctx (= partials per image and head for cdf_linattn_finalize)
extern "C" int cdf_linattn_kvctx_parts(int B, int n, int slots) {      // slots: target block count per launch (<= 0: the default, 512)
    const int tiles = n / 128;
    if (tiles < 1 || B < 1) return 0;
    int P = (slots > 0 ? slots : 512) / B;                        // default ~2 blocks per CU queued: a block's last tile overlaps another's start
    if (P < 1) P = 1;
    if (P > tiles) P = tiles;
    const int tpb = (tiles + P - 1) / P;
    return (tiles + tpb - 1) / tpb;
}

extern "C" int cdf_linattn_kvctx(const float* xn, int ldx, const void* w_hi, const void* w_lo, int ldk, float* kv, int ldkv, float* ws, int B,
                                 int n, int dim, int heads, int slots, void* stream) {
    CDF_REQUIRE(xn && w_hi && kv && ws && B > 0, "cdf_linattn_kvctx: null pointer");
    CDF_REQUIRE(heads == 4 && n >= 128 && n % 128 == 0 && dim >= 32 && dim % 32 == 0 && dim <= 512,
                "cdf_linattn_kvctx: 4 heads, n %% 128 == 0, dim a multiple of 32 (<= 512); got heads=%d n=%d dim=%d", heads, n, dim);
    CDF_REQUIRE(ldx % 4 == 0 && ldx >= dim && ldk % 8 == 0 && ldk >= dim && ldkv % 4 == 0 && ldkv >= 256 &&
                ((((uintptr_t)xn) | ((uintptr_t)w_hi) | ((uintptr_t)w_lo) | ((uintptr_t)kv)) & 15) == 0,
                "cdf_linattn_kvctx: pitches (xn % 4, weights % 8, kv % 4 and >= 256) / 16-byte alignment");
    const int P = cdf_linattn_kvctx_parts(B, n, slots), tiles = n / 128, HD = 128;
    KvCtxArgs a;
    a.xn = xn; a.w_hi = (const unsigned short*)w_hi; a.w_lo = (const unsigned short*)w_lo; a.kv = kv;
    a.max_part = ws;
    a.ctx_part = ws + (size_t)B * P * HD;
    a.sum_part = a.ctx_part + (size_t)B * P * heads * 1024;
    a.ldx = ldx; a.ldk = ldk; a.ldkv = ldkv; a.n = n; a.dim = dim; a.P = P;
    a.tiles_per_block = (tiles + P - 1) / P;
    const size_t lds = (size_t)128 * (256 + 8) * sizeof(float) + 8 * 32 * sizeof(float);
#ifndef CDF_EMU
    static CdfDeviceLatch attr_done;
    if (attr_done.first()) {
        (void)hipFuncSetAttribute((const void*)linattn_kvctx_kernel<1, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)linattn_kvctx_kernel<3, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)linattn_kvctx_kernel<1, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)linattn_kvctx_kernel<3, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
#endif
    if (dim % 64 == 0) {
        if (w_lo) CDF_LAUNCH((linattn_kvctx_kernel<3, 64>), dim3(P, B), dim3(512), lds, CDF_S, a);
        else CDF_LAUNCH((linattn_kvctx_kernel<1, 64>), dim3(P, B), dim3(512), lds, CDF_S, a);
    } else {
        if (w_lo) CDF_LAUNCH((linattn_kvctx_kernel<3, 32>), dim3(P, B), dim3(512), lds, CDF_S, a);
        else CDF_LAUNCH((linattn_kvctx_kernel<1, 32>), dim3(P, B), dim3(512), lds, CDF_S, a);
    }
    return cdf_check_launch("linattn_kvctx");
}

extern "C" int cdf_pack_weight_bf16(const float* src, void* dst_hi, void* dst_lo, int T, int R, int C, int ldc, long long s_t,
                                    long long s_r, long long s_c, void* stream) {
    CDF_REQUIRE(src && dst_hi && T > 0 && R > 0 && C > 0 && ldc >= C && ldc % 32 == 0, "cdf_pack_weight_bf16: bad args (ldc must be a multiple of 32)");
    long long g = ((long long)T * R * ldc + 255) / 256;
    if (g > 4096) g = 4096;
    CDF_LAUNCH(pack_weight_bf16_kernel, dim3((int)g), dim3(256), 0, CDF_S, src, (unsigned short*)dst_hi, (unsigned short*)dst_lo, T, R, C, ldc, s_t, s_r, s_c);
    return cdf_check_launch("pack_weight_bf16");
}

extern "C" int cdf_conv_gemm_bf16(const float* x, int ldx, const void* w_hi, const void* w_lo, int ldk, float* y, int ldy, int B,
                                  int H, int W, int Cin, int OH, int OW, int Cout, int QH, int QW, int os, int is, int nphase,
                                  const int* phase_desc, const float* bias, const float* sbias, int ld_sbias, const float* res,
                                  int ldr, float* pre, int ldp, const float* mul, int ldm, int act, int mul_mode, int accumulate,
                                  int split, void* stream) {
    CDF_REQUIRE(x && w_hi && y && (((uintptr_t)x) & 15) == 0 && ldx % 4 == 0 && ldx >= Cin, "cdf_conv_gemm_bf16: bad x");
    CDF_REQUIRE((split == 1 || split == 3) && (split == 1 || w_lo), "cdf_conv_gemm_bf16: split must be 1 (bf16) or 3 (hi/lo bf16x3)");
    CDF_REQUIRE((((uintptr_t)w_hi) & 15) == 0 && ldk % 32 == 0 && ldk >= Cin, "cdf_conv_gemm_bf16: weights must be 16B aligned with ldk %% 32 == 0");
    CDF_REQUIRE(nphase >= 1 && nphase <= 4 && phase_desc && ldy >= Cout, "cdf_conv_gemm_bf16: bad geometry");
    CDF_REQUIRE(!mul_mode || mul, "cdf_conv_gemm_bf16: mul_mode without mul tensor");
    SpArgs a;
    a.x = x; a.w_hi = (const unsigned short*)w_hi; a.w_lo = (const unsigned short*)w_lo; a.y = y;
    a.bias = bias; a.sbias = sbias; a.res = res; a.pre = pre; a.mul = mul;
    a.ldx = ldx; a.ldk = ldk; a.ldy = ldy; a.ld_sbias = ld_sbias; a.ldr = ldr; a.ldp = ldp; a.ldm = ldm;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.OH = OH; a.OW = OW; a.Cout = Cout; a.QH = QH; a.QW = QW; a.os = os; a.is = is;
    a.act = act; a.mul_mode = mul_mode; a.accumulate = accumulate; a.nphase = nphase;
    a.vec = cdf_epi_vec_ok(Cout, y, ldy, bias, sbias, ld_sbias, res, ldr, pre, ldp, mul, ldm);
    a.ys_hi = nullptr; a.ys_lo = nullptr; a.ld_ys = 0; a.io_bf = 0;
    const int* pd = phase_desc;
    for (int p = 0; p < nphase; ++p) {
        a.ph[p].oy = pd[0]; a.ph[p].ox = pd[1]; a.ph[p].ntaps = pd[2];
        CDF_REQUIRE(pd[2] >= 0 && pd[2] <= CDF_MAX_TAPS, "cdf_conv_gemm_bf16: too many taps (%d)", pd[2]);
        for (int t = 0; t < pd[2]; ++t) {
            a.ph[p].dy[t] = (signed char)pd[3 + 3 * t];
            a.ph[p].dx[t] = (signed char)pd[4 + 3 * t];
            a.ph[p].wi[t] = (signed char)pd[5 + 3 * t];
        }
        pd += 3 + 3 * pd[2];
    }
    const int M = B * QH * QW;
    const int tiles = cdf_cdiv(M, 128) * cdf_cdiv(Cout, 128);
#ifndef CDF_EMU
    static CdfDeviceLatch attr_done;
    if (attr_done.first()) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_sp_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_igemm_sp_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
#endif
    if (split == 1) {
        const size_t lds = CDF_SP_EPI_LDS;                 // operand stages (40 KB) < epilogue tile
        CDF_LAUNCH((conv_igemm_sp_kernel<1>), dim3(tiles, nphase), dim3(256), lds, CDF_S, a);
    } else {
        const size_t lds = (size_t)2 * 2 * 2 * 128 * 40 * sizeof(unsigned short);   // 80 KB >= CDF_SP_EPI_LDS
        CDF_LAUNCH((conv_igemm_sp_kernel<3>), dim3(tiles, nphase), dim3(256), lds, CDF_S, a);
    }
    return cdf_check_launch("conv_igemm_sp");
}

extern "C" int cdf_conv_wgrad_bf16(const float* xa, int lda, const float* xb, int ldb, float* ws, int ldo, int B, int QH, int QW,
                                   int HA, int WA, int sa, int HB, int WB, int sb, int CA, int CB, int ntaps, const int* tap_desc,
                                   int nsplit, float* bsum, void* stream) {
    CDF_REQUIRE(xa && xb && ws && (((uintptr_t)xa) & 15) == 0 && (((uintptr_t)xb) & 15) == 0, "cdf_conv_wgrad_bf16: null / unaligned operand");
    CDF_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && lda >= CA && ldb >= CB && ldo % 4 == 0 && ldo >= CB, "cdf_conv_wgrad_bf16: bad pitch");
    CDF_REQUIRE(ntaps >= 1 && ntaps <= CDF_MAX_TAPS && tap_desc && nsplit >= 1, "cdf_conv_wgrad_bf16: bad tap / split count");
    SpWgradArgs a;
    a.xa = xa; a.xb = xb; a.out = ws; a.bsum = bsum; a.lda = lda; a.ldb = ldb; a.ldo = ldo;
    a.B = B; a.QH = QH; a.QW = QW; a.HA = HA; a.WA = WA; a.sa = sa; a.HB = HB; a.WB = WB; a.sb = sb;
    a.CA = CA; a.CB = CB; a.ntaps = ntaps; a.nsplit = nsplit; a.xcd_swizzle = 1;
    const int M = B * QH * QW;
    a.m_per_split = cdf_cdiv(cdf_cdiv(M, nsplit), 32) * 32;
    for (int t = 0; t < ntaps; ++t) {
        a.day[t] = (signed char)tap_desc[4 * t + 0];
        a.dax[t] = (signed char)tap_desc[4 * t + 1];
        a.dby[t] = (signed char)tap_desc[4 * t + 2];
        a.dbx[t] = (signed char)tap_desc[4 * t + 3];
    }
#ifndef CDF_EMU
    static CdfDeviceLatch attr_done;
    if (attr_done.first()) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_sp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
#endif
    const size_t lds = (size_t)2 * 4 * 32 * 128 * sizeof(unsigned short);
    const int tiles = cdf_cdiv(CA, 128) * cdf_cdiv(CB, 128);
    CDF_LAUNCH(conv_wgrad_sp_kernel, dim3(tiles, ntaps, nsplit), dim3(256), lds, CDF_S, a);
    return cdf_check_launch("conv_wgrad_sp");
}

// ---- pre-split operand entry points ---------------------------------------------------------------------
extern "C" int cdf_split_bf16(const float* x, int ldx, void* hi, void* lo, int ldo, long long rows, int C, void* stream) {
    CDF_REQUIRE(x && hi && rows > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldo % 8 == 0 && ldo >= C, "cdf_split_bf16: bad args (C %% 4, ldo %% 8)");
    long long g = (rows * (C / 4) + 255) / 256;
    if (g > 8192) g = 8192;
    CDF_LAUNCH(split_bf16_kernel, dim3((int)g), dim3(256), 0, CDF_S, x, ldx, (unsigned short*)hi, (unsigned short*)lo, ldo, rows, C / 4);
    return cdf_check_launch("split_bf16");
}

extern "C" int cdf_bf16_to_f32(const void* x, int ldx, float* y, int ldy, long long rows, int C, void* stream) {
    CDF_REQUIRE(x && y && rows > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C && (((uintptr_t)x) & 7) == 0 &&
                (((uintptr_t)y) & 15) == 0, "cdf_bf16_to_f32: bad args (C %% 4, pitches %% 4, x 8-byte / y 16-byte aligned)");
    long long g = (rows * (C / 4) + 255) / 256;
    if (g > 8192) g = 8192;
    CDF_LAUNCH(widen_bf16_kernel, dim3((int)g), dim3(256), 0, CDF_S, (const unsigned short*)x, ldx, y, ldy, rows, C / 4);
    return cdf_check_launch("bf16_to_f32");
}

static int fill_phases(SpPhase* ph, int nphase, const int* pd, const char* who) {
    for (int p = 0; p < nphase; ++p) {
        ph[p].oy = pd[0]; ph[p].ox = pd[1]; ph[p].ntaps = pd[2];
        CDF_REQUIRE(pd[2] >= 0 && pd[2] <= CDF_MAX_TAPS, "%s: too many taps (%d)", who, pd[2]);
        for (int t = 0; t < pd[2]; ++t) {
            ph[p].dy[t] = (signed char)pd[3 + 3 * t];
            ph[p].dx[t] = (signed char)pd[4 + 3 * t];
            ph[p].wi[t] = (signed char)pd[5 + 3 * t];
        }
        pd += 3 + 3 * pd[2];
    }
    return CDF_OK;
}

// ---- tuning: an explicit, optional argument of the GEMM entry points (include/colddiff.h: cdf_gemm_tuning) -----------------------------
// No mutable process-wide state: a NULL pointer means these defaults, anything else is read once per call.  The choices only select
// between kernels / tile shapes that compute the same sums (fp32 summation order aside).
static const cdf_gemm_tuning kTuneDefault = {(int)sizeof(cdf_gemm_tuning), 0, 0, 0, 1, 1, 1, 47, 1, 0, 1, 1, 1, 1, 1, 0};
extern "C" int cdf_gemm_tuning_default(cdf_gemm_tuning* t) {
    CDF_REQUIRE(t, "cdf_gemm_tuning_default: null pointer");
    *t = kTuneDefault;
    return 0;
}
static const cdf_gemm_tuning* cdf_tune(const cdf_gemm_tuning* t) { return (t && t->size == (int)sizeof(cdf_gemm_tuning)) ? t : &kTuneDefault; }
static bool cdf_tune_ok(const cdf_gemm_tuning* t) {
    if (!t) return true;
    const bool bm_ok = t->tile_bm == 0 || t->tile_bm == 64 || t->tile_bm == 128 || (t->tile_bm == 256 && (t->tile_bn == 0 || t->tile_bn == 128));
    const bool bn_ok = t->tile_bn == 0 || t->tile_bn == 64 || t->tile_bn == 128;
    return t->size == (int)sizeof(cdf_gemm_tuning) && bm_ok && bn_ok && (t->max_bm == 0 || t->max_bm == 128 || t->max_bm == 256) &&
           (t->halo_bm == 0 || t->halo_bm == 128 || t->halo_bm == 256) && t->halo >= 0 && t->halo <= 127 && t->halo_min_tiles >= 0 && t->resident_reserve >= 0 && t->resident_reserve <= 248 && (t->rowhalo_stream == 0 || t->rowhalo_stream == 1);
}
#define CDF_TUNE_CHECK(t, who)                                                                                                          \
    CDF_REQUIRE(cdf_tune_ok(t), who ": bad cdf_gemm_tuning (size %d, expected %d; tile_bm 0/64/128/256 (256 with tile_bn 0/128), tile_bn 0/64/128, " \
                                    "max_bm 0/128/256, halo_bm 0/128/256, halo 0..127, rowhalo_stream 0/1, resident_reserve 0..248): start from cdf_gemm_tuning_default",            \
                (t) ? (t)->size : 0, (int)sizeof(cdf_gemm_tuning))

template <int NS, int BM, int BN, int WM, int WN, int NSTAGE, int OCC = 512 / (64 * WM * WN)>
static int launch_igemm_spx(const SpxArgs& a, int M, hipStream_t s) {
    constexpr size_t stages = (size_t)NSTAGE * 2 * (BM + BN) * 32 * sizeof(unsigned short) + (CDF_MAX_TAPS + 1) * sizeof(int);
    constexpr size_t epi = (size_t)BM * (BN + 8) * sizeof(float);
    constexpr size_t lds = stages > epi ? stages : epi;      // 128 x 128 x 2 stages: 68 KB (epilogue tile), two blocks per CU;
                                                             // 256 x 128 x 3 stages: 144 KB, one block per CU
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
#ifndef CDF_EMU
    static CdfDeviceLatch attr_done;
    if (attr_done.first()) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_spx_kernel<BM, BN, WM, WN, NSTAGE, OCC, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
#endif
    const int tiles = cdf_cdiv(M, BM) * cdf_cdiv(a.Cout, BN);
    CDF_LAUNCH((conv_igemm_spx_kernel<BM, BN, WM, WN, NSTAGE, OCC, NS>), dim3(tiles, a.nphase, a.ksplit > 1 ? a.ksplit : 1), dim3(64 * WM * WN), lds, s, a);
    if (a.ksplit > 1) {
        const int ftiles = cdf_cdiv(M, 16) * cdf_cdiv(a.Cout, BN);
        if (BN == 64) CDF_LAUNCH((conv_splitk_finish_kernel<64>), dim3(ftiles), dim3(256), 0, s, a);
        else CDF_LAUNCH((conv_splitk_finish_kernel<128>), dim3(ftiles), dim3(256), 0, s, a);
    }
    return cdf_check_launch("conv_igemm_spx");
}

static int cdf_num_cus() {                                     // CUs of the current device (blocks of the resident kernels), a multiple of 8 XCDs
#ifdef CDF_EMU
    return 8;
#else
    static int n[64] = {0};                                  // per device ordinal (a process may drive several devices)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!n[dev]) {
        int cus = 0;
        n[dev] = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= 8) ? cus / 8 * 8 : 256;
    }
    return n[dev];
#endif
}

template <int NS, int W, int BN, int BM>
static int launch_igemm_halo(const SpxArgs& a, int M, hipStream_t s) {
    // weight stages: as many as fit next to the two halo buffers
    constexpr int TH = BM / W, HR = (TH + 2) * (W + 2), HRP = (HR + 15) / 16 * 16;
    constexpr size_t abytes = (size_t)2 * 2 * HRP * 64, bstage = (size_t)2 * BN * 64;
    constexpr int NBfit = (int)((160 * 1024 - 64 - abytes) / bstage);
    constexpr int NB = NBfit > 6 ? 6 : NBfit;
    static_assert(NB >= 3, "halo tile leaves no room for three weight stages");
    constexpr size_t stages = abytes + (size_t)NB * bstage + 16 * sizeof(int);
    constexpr size_t epi = (size_t)BM * (BN + 8) * sizeof(float);
    constexpr size_t lds = stages > epi ? stages : epi;
    static_assert(lds <= 160 * 1024, "halo tile does not fit the LDS");
#ifndef CDF_EMU
    static CdfDeviceLatch attr_done;
    if (attr_done.first()) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_halo_kernel<W, BN, NB, BM, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
#endif
    const int tiles = (M / BM) * cdf_cdiv(a.Cout, BN);
    CDF_LAUNCH((conv_igemm_halo_kernel<W, BN, NB, BM, NS>), dim3(tiles), dim3(512), lds, s, a);
    return cdf_check_launch("conv_igemm_halo");
}

template <int NS, int W, int BN>
static int launch_igemm_rowhalo_stream(const SpxArgs& a, int M, hipStream_t s, int reserve) {
    constexpr int TH = 256 / W, RH = TH * (W + 2), HRP = (RH + 15) / 16 * 16;
    constexpr size_t st_a = (size_t)2 * HRP * 64, st_b = (size_t)2 * BN * 64;
    constexpr size_t lds_s = (st_a + 2 * st_b) + ((st_a + st_b) > (size_t)128 * (BN + 8) * 4 ? (st_a + st_b) : (size_t)128 * (BN + 8) * 4);
    static_assert(lds_s <= 160 * 1024, "streaming row-halo tile does not fit the LDS");
#ifndef CDF_EMU
    static CdfDeviceLatch attr_done;
    if (attr_done.first()) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_rowhalo_stream_kernel<W, BN, NS, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_igemm_rowhalo_stream_kernel<W, BN, NS, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
#endif
    const int tiles = (M / 256) * cdf_cdiv(a.Cout, BN);
    // resident blocks: one per CU -- minus the CUs the caller keeps free for kernels that run concurrently (cdf_gemm_tuning.resident_reserve:
    // the collectives of a multi-rank gradient exchange; a resident block that finds its CU taken would run its fixed share of the tiles
    // after everybody else), in whole XCD rounds
#ifdef CDF_EMU
    int ncu = cdf_num_cus() - reserve;      // (the simulator's 8 "CUs" are not XCD rounds: a reserve really shrinks the grid there, so the CPU
    if (ncu < 1) ncu = 1;                   //  suite walks several tiles per resident block -- down to ONE block taking every tile)
#else
    int ncu = cdf_num_cus() - (reserve + 7) / 8 * 8;
    if (ncu < 8) ncu = 8;
#endif
    const int grid = tiles < ncu ? tiles : ncu;
    if (a.Cin == 64)
        CDF_LAUNCH((conv_igemm_rowhalo_stream_kernel<W, BN, NS, 2>), dim3(grid), dim3(512), lds_s, s, a);
    else
        CDF_LAUNCH((conv_igemm_rowhalo_stream_kernel<W, BN, NS, 4>), dim3(grid), dim3(512), lds_s, s, a);
    return cdf_check_launch("conv_igemm_rowhalo_stream");
}

// Split-K factor of the generic pre-split GEMM for grids far below one 64-row tile per CU: the smallest divisor of the tap count
// that brings the launch to >= 192 blocks (else the largest); 1 = no split.
static int spx_ksplit(int M, int Cout, int nphase, int ntaps, const cdf_gemm_tuning& T) {
    if (!T.splitk || !T.deep || nphase != 1 || (ntaps != 9 && ntaps != 16)) return 1;
    const bool n64 = Cout <= 64;
    const long long tiles128 = (long long)cdf_cdiv(M, 128) * cdf_cdiv(Cout, n64 ? 64 : 128);
    const long long tiles64 = (long long)cdf_cdiv(M, 64) * cdf_cdiv(Cout, n64 ? 64 : 128);
    if (tiles128 >= 384 || tiles64 > 128) return 1;
    for (int ks = 2; ks <= ntaps; ++ks)
        if (ntaps % ks == 0 && tiles64 * ks >= 192) return ks;
    return ntaps;
}
extern "C" int cdf_conv_gemm_bf16x_ksplit(int M, int Cout, int nphase, int ntaps, const cdf_gemm_tuning* tune) { return spx_ksplit(M, Cout, nphase, ntaps, *cdf_tune(tune)); }

template <int NS>
static int dispatch_gemm_bf16x(SpxArgs& a, int B, int H, int W, int Cin, int Cout, int QH, int QW, int os, int is, int nphase, long long ks_ws_floats,
                               const cdf_gemm_tuning& T, hipStream_t s) {
    // Tile choice: 64-wide N for Cout <= 64 (no half-empty MFMA columns); 64-row M tiles when 128-row tiles would
    // leave most of the 256 CUs x 2 resident blocks idle (deep, small-image layers: M = 8192 at 16 x 16); the 8-wave
    // 256 x 128 tile (3 stages, one block per CU) when it still gives every CU at least ~2 tiles.
    const int M = B * QH * QW;
    const bool n64 = T.tile_bn ? T.tile_bn == 64 : Cout <= 64;
    const long long tiles128 = (long long)cdf_cdiv(M, 128) * cdf_cdiv(Cout, n64 ? 64 : 128) * nphase;
    bool m64 = tiles128 < 384;
    bool m256 = !n64 && tiles128 >= 1024 && T.max_bm != 128;
    if (T.tile_bm) { m64 = T.tile_bm == 64; m256 = T.tile_bm == 256 && !n64; }
    // 3 x 3, stride 1, three groups of equal dy covering three consecutive rows: candidates for the row-group rotation
    bool is3x3 = nphase == 1 && is == 1 && os == 1 && a.ph[0].ntaps == 9 && a.ph[0].oy == 0 && a.ph[0].ox == 0;
    if (is3x3) {
        int seen = 0;
        for (int g = 0; g < 3; ++g) {
            const int dy = a.ph[0].dy[3 * g];
            is3x3 = is3x3 && a.ph[0].dy[3 * g + 1] == dy && a.ph[0].dy[3 * g + 2] == dy && dy >= -1 && dy <= 1;
            seen |= 1 << (dy + 1);
        }
        is3x3 = is3x3 && seen == 7;
    }
    const bool rot_ok = is3x3;
    // (measured at 128 x 128 images: 128 -> 64 channels 0.40 -> 0.37 ms with the rotation on its 128 x 64 tiles; for 64 -> 128
    // the two-row 256 x 128 tile without rotation stays ahead of one-row tiles with it, 0.405 vs 0.414 ms, so the tile
    // choice is not bent towards one-row tiles)
    const int bm = m256 ? 256 : (m64 ? 64 : 128);
    a.taprot = rot_ok && QW == bm;
    a.dephase = T.dephase;
    // 3 x 3 stride-1 layers whose rows tile into 128-pixel strips: input tile resident in LDS (conv_igemm_halo_kernel)
    if (T.halo && is3x3 && !T.tile_bm && QW == W && QH == H && Cin % 32 == 0 && Cin >= 64 && M % 128 == 0) {
        const bool n64_in = n64;
        int dxs = 0;
        for (int t = 0; t < 9; ++t) dxs |= 1 << (a.ph[0].dx[t] + 1);
        const bool dx_ok = dxs == 7;                         // (is3x3: three groups of equal dy in {-1, 0, 1})
        // Small grids (sampling batches, the 16 x 16 level): when 128-wide N tiles leave a third of the CUs without a block, 64-wide
        // ones double the block count -- every block is then half as long, and the launch is one block's latency either way
        // (1024 -> 512 channels at 16 x 16 pixels, 16 images: 128 tiles for 256 CUs).
        const bool n64 = n64_in || (T.small_n64 && !T.tile_bn && Cout > 64 && Cout % 64 == 0 && (long long)(M / 128) * cdf_cdiv(Cout, 128) < 176);
        const long long tiles = (long long)(M / 128) * cdf_cdiv(Cout, n64 ? 64 : 128);
        // row-halo kernel: 256-pixel tiles, input shared by the dx taps only.  Bit 32 (default): the > 64-channel outputs at
        // 128-pixel width, where it beats the generic 256 x 128 kernel (64 -> 128: 0.325 -> 0.298 ms); bit 64: wherever it applies
        // (at 64 pixels the halo kernel's 256-pixel tile stays ahead, 0.240 vs 0.252 ms)
        if (dx_ok && M % 256 == 0 && (T.rowhalo_stream & 1) && (Cin == 64 || Cin == 128) &&
            ((T.halo & 64) || ((T.halo & 32) && W == 128 && !n64 && (long long)(M / 256) * cdf_cdiv(Cout, 128) >= 256))) {
#define CDF_ROWHALO_CASE(WW)                                                                                           \
    if (W == WW && H % (256 / WW) == 0)                                                                                \
        return n64 ? launch_igemm_rowhalo_stream<NS, WW, 64>(a, M, s, T.resident_reserve) : launch_igemm_rowhalo_stream<NS, WW, 128>(a, M, s, T.resident_reserve);
            CDF_ROWHALO_CASE(128) CDF_ROWHALO_CASE(64) CDF_ROWHALO_CASE(32) CDF_ROWHALO_CASE(16)
#undef CDF_ROWHALO_CASE
        }
        if (dx_ok && tiles >= (T.halo_min_tiles > 0 ? T.halo_min_tiles : 1)) {
#define CDF_HALO_CASE(WW)                                                                                              \
    if (W == WW && (T.halo & (WW / 16)) && H % (128 / WW) == 0 && (WW < 128 || n64 || (T.halo & 16))) {          \
        /* 256-pixel tiles (half the weight bytes per MFMA) when they still give every CU a tile and fit the LDS      \
           (at 128-pixel width only next to 64-wide weight stages) */                                                  \
        if ((WW <= 64 || n64) && T.halo_bm != 128 && H % (256 / WW) == 0 && M % 256 == 0 &&                          \
            (T.halo_bm == 256 || (long long)(M / 256) * cdf_cdiv(Cout, n64 ? 64 : 128) >= 256))                     \
            return n64 ? launch_igemm_halo<NS, WW, 64, 256>(a, M, s)                                                    \
                       : launch_igemm_halo<NS, WW, 128, WW <= 64 ? 256 : 128>(a, M, s);                                 \
        return n64 ? launch_igemm_halo<NS, WW, 64, 128>(a, M, s) : launch_igemm_halo<NS, WW, 128, 128>(a, M, s);        \
    }
            CDF_HALO_CASE(128) CDF_HALO_CASE(64) CDF_HALO_CASE(32) CDF_HALO_CASE(16)
#undef CDF_HALO_CASE
        }
    }
    if (m256) return launch_igemm_spx<NS, 256, 128, 4, 2, 3>(a, M, s);
    // Grids that do not even give every CU one 64-row tile (the 4 x 4 / 8 x 8-pixel levels of the 32 x 32 configurations, small
    // sampling batches): a block's life is its K loop, and with two stages every step waited out a whole DMA round trip (144 steps
    // of 1.5 us for 512 -> 1024 channels at 4 x 4 pixels).  Six stages, one block per CU: five chunks in flight per block.
    const long long tiles64 = (long long)cdf_cdiv(M, 64) * cdf_cdiv(Cout, n64 ? 64 : 128) * nphase;
    if (m64 && tiles64 <= 256 && T.deep) {
        // ... and when even that leaves most CUs without a block, the taps are shared out over blockIdx.z (split-K, partial sums through
        // the caller's workspace, conv_splitk_finish_kernel adds them up and runs the epilogue)
        const int ks = spx_ksplit(M, Cout, nphase, a.ph[0].ntaps, T);
        if (ks > 1 && a.ks_ws && ks_ws_floats >= (long long)ks * M * ((Cout + 3) / 4 * 4)) {
            a.ksplit = ks;
            a.ks_ld = (Cout + 3) / 4 * 4;
            a.taprot = 0;
        } else {
            a.ksplit = 1;
        }
        if (n64) return launch_igemm_spx<NS, 64, 64, 2, 2, 6, 1>(a, M, s);
        return launch_igemm_spx<NS, 64, 128, 2, 2, 6, 1>(a, M, s);
    }
    if (n64) return m64 ? launch_igemm_spx<NS, 64, 64, 2, 2, 2>(a, M, s) : launch_igemm_spx<NS, 128, 64, 2, 2, 2>(a, M, s);
    if (m64) return launch_igemm_spx<NS, 64, 128, 2, 2, 2>(a, M, s);
    return launch_igemm_spx<NS, 128, 128, 2, 2, 2>(a, M, s);
}

extern "C" int cdf_conv_gemm_bf16x_io(const void* x_hi, const void* x_lo, int ldx, const void* zero, const void* w_hi, const void* w_lo,
                                      int ldk, float* y, int ldy, int B, int H, int W, int Cin, int OH, int OW, int Cout, int QH, int QW,
                                      int os, int is, int nphase, const int* phase_desc, const float* bias, const float* sbias,
                                      int ld_sbias, const void* res, int ldr, void* pre, int ldp, const void* mul, int ldm, int act,
                                      int mul_mode, int accumulate, int io_bf16, void* y_hi, void* y_lo, int ld_ys, float* ws,
                                      long long ws_floats, const cdf_gemm_tuning* tune, void* stream);

extern "C" int cdf_conv_gemm_bf16x(const void* x_hi, const void* x_lo, int ldx, const void* zero, const void* w_hi, const void* w_lo,
                                   int ldk, float* y, int ldy, int B, int H, int W, int Cin, int OH, int OW, int Cout, int QH, int QW,
                                   int os, int is, int nphase, const int* phase_desc, const float* bias, const float* sbias,
                                   int ld_sbias, const float* res, int ldr, float* pre, int ldp, const float* mul, int ldm, int act,
                                   int mul_mode, int accumulate, void* y_hi, void* y_lo, int ld_ys, float* ws, long long ws_floats,
                                   const cdf_gemm_tuning* tune, void* stream) {
    return cdf_conv_gemm_bf16x_io(x_hi, x_lo, ldx, zero, w_hi, w_lo, ldk, y, ldy, B, H, W, Cin, OH, OW, Cout, QH, QW, os, is, nphase, phase_desc,
                                  bias, sbias, ld_sbias, res, ldr, pre, ldp, mul, ldm, act, mul_mode, accumulate, 0, y_hi, y_lo, ld_ys, ws,
                                  ws_floats, tune, stream);
}

// ... with typed epilogue operands (io_bf16: CDF_IO_RES_BF16 | CDF_IO_PRE_BF16 | CDF_IO_MUL_BF16 -- that operand is ONE bf16 plane with its
// pitch in bf16 elements): the bf16-activation-storage engine, where every feature map between kernels is a bf16 tensor.
extern "C" int cdf_conv_gemm_bf16x_io(const void* x_hi, const void* x_lo, int ldx, const void* zero, const void* w_hi, const void* w_lo,
                                      int ldk, float* y, int ldy, int B, int H, int W, int Cin, int OH, int OW, int Cout, int QH, int QW,
                                      int os, int is, int nphase, const int* phase_desc, const float* bias, const float* sbias,
                                      int ld_sbias, const void* res_, int ldr, void* pre_, int ldp, const void* mul_, int ldm, int act,
                                      int mul_mode, int accumulate, int io_bf16, void* y_hi, void* y_lo, int ld_ys, float* ws,
                                      long long ws_floats, const cdf_gemm_tuning* tune, void* stream) {
    const float* res = (const float*)res_;
    float* pre = (float*)pre_;
    const float* mul = (const float*)mul_;
    CDF_REQUIRE((io_bf16 & ~15) == 0, "cdf_conv_gemm_bf16x_io: io_bf16 has unknown bits (%d)", io_bf16);
    CDF_REQUIRE(!(io_bf16 & CDF_IO_PRE_GRAD) || (pre_ && (act == 1 || act == 2)), "cdf_conv_gemm_bf16x_io: CDF_IO_PRE_GRAD needs a pre tensor and act = GELU / SiLU");
    CDF_REQUIRE(x_hi && zero && w_hi && (y || (y_hi && !accumulate)), "cdf_conv_gemm_bf16x: null pointer");
    CDF_TUNE_CHECK(tune, "cdf_conv_gemm_bf16x");
    CDF_REQUIRE(!ws || (((uintptr_t)ws) & 15) == 0, "cdf_conv_gemm_bf16x: the split-K workspace must be 16-byte aligned");
    CDF_REQUIRE((x_lo != nullptr) == (w_lo != nullptr), "cdf_conv_gemm_bf16x: pass both lo planes (split precision, 3 MFMAs per product) or neither (single-pass bf16)");
    CDF_REQUIRE((!y_hi && !y_lo) || (y_hi && ld_ys % 4 == 0 && ld_ys >= Cout && Cout % 4 == 0 && ((((uintptr_t)y_hi) | ((uintptr_t)y_lo)) & 7) == 0),
                "cdf_conv_gemm_bf16x: output planes need Cout %% 4 == 0, ld_ys %% 4 == 0, 8-byte alignment (y_lo optional: hi-only planes)");
    CDF_REQUIRE(((((uintptr_t)x_hi) | ((uintptr_t)x_lo) | ((uintptr_t)zero) | ((uintptr_t)w_hi) | ((uintptr_t)w_lo)) & 15) == 0, "cdf_conv_gemm_bf16x: operands must be 16B aligned");
    CDF_REQUIRE(ldx % 8 == 0 && Cin % 8 == 0 && ldx >= Cin && ldk % 32 == 0 && ldk >= Cin, "cdf_conv_gemm_bf16x: Cin and pitches must be multiples of 8 (ldk of 32)");
    CDF_REQUIRE(nphase >= 1 && nphase <= 4 && phase_desc && (!y || ldy >= Cout), "cdf_conv_gemm_bf16x: bad geometry");
    CDF_REQUIRE(!mul_mode || mul, "cdf_conv_gemm_bf16x: mul_mode without mul tensor");
    SpxArgs a;
    a.x_hi = (const unsigned short*)x_hi; a.x_lo = (const unsigned short*)x_lo; a.zero = (const unsigned short*)zero;
    a.w_hi = (const unsigned short*)w_hi; a.w_lo = (const unsigned short*)w_lo; a.y = y;
    a.bias = bias; a.sbias = sbias; a.res = res; a.pre = pre; a.mul = mul;
    a.ldx = ldx; a.ldk = ldk; a.ldy = ldy; a.ld_sbias = ld_sbias; a.ldr = ldr; a.ldp = ldp; a.ldm = ldm;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.OH = OH; a.OW = OW; a.Cout = Cout; a.QH = QH; a.QW = QW; a.os = os; a.is = is;
    a.act = act; a.mul_mode = mul_mode; a.accumulate = accumulate; a.nphase = nphase;
    a.vec = cdf_epi_vec_ok(Cout, y, ldy, bias, sbias, ld_sbias, res, ldr, pre, ldp, mul, ldm);
    a.ys_hi = (unsigned short*)y_hi; a.ys_lo = (unsigned short*)y_lo; a.ld_ys = ld_ys;
    a.io_bf = io_bf16;
    CDF_REQUIRE(!y_hi || a.vec, "cdf_conv_gemm_bf16x: split output planes need the vectorised epilogue (aligned pointers, pitches %% 4)");
    CDF_REQUIRE(!(io_bf16 & 7) || a.vec, "cdf_conv_gemm_bf16x_io: bf16 epilogue operands need the vectorised epilogue (16-byte-aligned pointers, pitches %% 4, Cout %% 4)");
    int rc = fill_phases(a.ph, nphase, phase_desc, "cdf_conv_gemm_bf16x");
    if (rc) return rc;
    a.ksplit = 1; a.ks_ws = ws; a.ks_ld = 0;
    return x_lo ? dispatch_gemm_bf16x<3>(a, B, H, W, Cin, Cout, QH, QW, os, is, nphase, ws ? ws_floats : 0, *cdf_tune(tune), CDF_S)
                : dispatch_gemm_bf16x<1>(a, B, H, W, Cin, Cout, QH, QW, os, is, nphase, ws ? ws_floats : 0, *cdf_tune(tune), CDF_S);
}

template <int NS, int TA, int TB, bool STACK2 = false>
static int launch_wgrad_spx(const SpxWgradArgs& a, hipStream_t s) {
    constexpr size_t stage = (size_t)2 * 32 * ((TA + 32) + (TB + 32)) * sizeof(unsigned short);
    constexpr size_t epi = (size_t)TA * (TB + 8) * sizeof(float);
    constexpr size_t lds = 2 * stage > epi ? 2 * stage : epi;
#ifndef CDF_EMU
    static CdfDeviceLatch attr_done;
    if (attr_done.first()) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_spx_kernel<TA, TB, STACK2, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
#endif
    const int tiles = (STACK2 ? 1 : cdf_cdiv(a.CA, TA)) * cdf_cdiv(a.CB, TB);
    CDF_LAUNCH((conv_wgrad_spx_kernel<TA, TB, STACK2, NS>), dim3(tiles, STACK2 ? cdf_cdiv(a.ntaps, 2) : a.ntaps, a.nsplit), dim3(256), lds, s, a);
    return cdf_check_launch("conv_wgrad_spx");
}

// 1 if cdf_conv_wgrad_bf16x takes the row-of-taps kernel for this geometry (the caller sizes the split count by it:
// 3 tap blocks per tile and one block per CU instead of 9 (or 5) and two)
extern "C" int cdf_conv_wgrad_bf16x_is_row3(int QH, int QW, int CA, int CB, int ntaps, int same_size_3x3, const cdf_gemm_tuning* tune) {
    return cdf_tune(tune)->wgrad_row3 && same_size_3x3 && ntaps == 9 && (QW == 16 || QW == 32 || QW == 64 || QW == 128) && (QH * QW) % 32 == 0 &&
           (QW >= 32 || QH % (32 / QW) == 0) && !(CA <= 64 && CB <= 64);
}

template <int NS, int TA, int TB>
static int launch_wgrad_row3(const SpxWgradArgs& a, hipStream_t s) {
    constexpr size_t stage = (size_t)2 * (36 * (TA + 32) + 32 * (TB + 32)) * sizeof(unsigned short);
    constexpr size_t epi = (size_t)TA * (TB + 8) * sizeof(float);
    constexpr size_t lds = 2 * stage > epi ? 2 * stage : epi;
#ifndef CDF_EMU
    static CdfDeviceLatch attr_done;
    if (attr_done.first()) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_row3_kernel<TA, TB, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
#endif
    const int tiles = cdf_cdiv(a.CA, TA) * cdf_cdiv(a.CB, TB);
    CDF_LAUNCH((conv_wgrad_row3_kernel<TA, TB, NS>), dim3(tiles, 3, a.nsplit), dim3(512), lds, s, a);
    return cdf_check_launch("conv_wgrad_row3");
}

template <int NS>
static int dispatch_wgrad_bf16x(SpxWgradArgs& a, int QH, int QW, int HA, int WA, int sa, int HB, int WB, int sb, int CA, int CB, int ntaps,
                                const cdf_gemm_tuning& T, hipStream_t s) {
    // 3 x 3 stride-1 "same" convolutions (X shifted per tap, dY read in place): one block per row of taps
    if (T.wgrad_row3 && ntaps == 9 && sa == 1 && sb == 1 && HA == QH && WA == QW && HB == QH && WB == QW &&
        (QW == 16 || QW == 32 || QW == 64 || QW == 128) && (QH * QW) % 32 == 0 && (QW >= 32 || QH % (32 / QW) == 0) && !(CA <= 64 && CB <= 64)) {
        bool ok = true;
        for (int g = 0; g < 3 && ok; ++g) {
            int seen = 0;
            for (int i = 0; i < 3; ++i) {
                const int t = 3 * g + i;
                ok = ok && a.day[t] == a.day[3 * g] && a.dby[t] == 0 && a.dbx[t] == 0 && a.dax[t] >= -1 && a.dax[t] <= 1 && a.day[t] >= -1 && a.day[t] <= 1;
                seen |= 1 << (a.dax[t] + 1);
            }
            ok = ok && seen == 7;
        }
        if (ok) {
            if (CA <= 64) return launch_wgrad_row3<NS, 64, 128>(a, s);
            if (CB <= 64) return launch_wgrad_row3<NS, 128, 64>(a, s);
            return launch_wgrad_row3<NS, 128, 128>(a, s);
        }
    }
    // thin layers get 64-wide tiles so that no half of a tile multiplies padding
    if (CA <= 64 && CB <= 64) return launch_wgrad_spx<NS, 64, 64>(a, s);
    if (CA <= 64) {
        bool same_b = ntaps >= 2;                  // two taps can share the B rows only if B is read at one offset
        for (int t = 1; t < ntaps; ++t) same_b = same_b && a.dby[t] == a.dby[0] && a.dbx[t] == a.dbx[0];
        if (same_b && T.wgrad_stack) return launch_wgrad_spx<NS, 128, 128, true>(a, s);
        return launch_wgrad_spx<NS, 64, 128>(a, s);
    }
    if (CB <= 64) return launch_wgrad_spx<NS, 128, 64>(a, s);
    return launch_wgrad_spx<NS, 128, 128>(a, s);
}

extern "C" int cdf_conv_wgrad_bf16x(const void* a_hi, const void* a_lo, int lda, const void* b_hi, const void* b_lo, int ldb,
                                    const void* zero, float* ws, int ldo, int B, int QH, int QW, int HA, int WA, int sa, int HB, int WB,
                                    int sb, int CA, int CB, int ntaps, const int* tap_desc, int nsplit, float* bsum, const cdf_gemm_tuning* tune,
                                    void* stream) {
    CDF_REQUIRE(a_hi && b_hi && zero && ws, "cdf_conv_wgrad_bf16x: null pointer");
    CDF_TUNE_CHECK(tune, "cdf_conv_wgrad_bf16x");
    CDF_REQUIRE((a_lo != nullptr) == (b_lo != nullptr), "cdf_conv_wgrad_bf16x: pass both lo planes (split precision) or neither (single-pass bf16)");
    CDF_REQUIRE(((((uintptr_t)a_hi) | ((uintptr_t)a_lo) | ((uintptr_t)b_hi) | ((uintptr_t)b_lo) | ((uintptr_t)zero)) & 15) == 0, "cdf_conv_wgrad_bf16x: operands must be 16B aligned");
    CDF_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && CA % 8 == 0 && CB % 8 == 0 && lda >= CA && ldb >= CB && ldo % 4 == 0 && ldo >= CB, "cdf_conv_wgrad_bf16x: channels / pitches must be multiples of 8");
    CDF_REQUIRE(ntaps >= 1 && ntaps <= CDF_MAX_TAPS && tap_desc && nsplit >= 1, "cdf_conv_wgrad_bf16x: bad tap / split count");
    SpxWgradArgs a;
    a.a_hi = (const unsigned short*)a_hi; a.a_lo = (const unsigned short*)a_lo; a.b_hi = (const unsigned short*)b_hi;
    a.b_lo = (const unsigned short*)b_lo; a.zero = (const unsigned short*)zero; a.out = ws; a.bsum = bsum;
    a.lda = lda; a.ldb = ldb; a.ldo = ldo;
    a.B = B; a.QH = QH; a.QW = QW; a.HA = HA; a.WA = WA; a.sa = sa; a.HB = HB; a.WB = WB; a.sb = sb;
    a.CA = CA; a.CB = CB; a.ntaps = ntaps; a.nsplit = nsplit; a.xcd_swizzle = cdf_tune(tune)->wgrad_swizzle;
    const int M = B * QH * QW;
    a.m_per_split = cdf_cdiv(cdf_cdiv(M, nsplit), 32) * 32;
    for (int t = 0; t < ntaps; ++t) {
        a.day[t] = (signed char)tap_desc[4 * t + 0];
        a.dax[t] = (signed char)tap_desc[4 * t + 1];
        a.dby[t] = (signed char)tap_desc[4 * t + 2];
        a.dbx[t] = (signed char)tap_desc[4 * t + 3];
    }
    return a_lo ? dispatch_wgrad_bf16x<3>(a, QH, QW, HA, WA, sa, HB, WB, sb, CA, CB, ntaps, *cdf_tune(tune), CDF_S)
                : dispatch_wgrad_bf16x<1>(a, QH, QW, HA, WA, sa, HB, WB, sb, CA, CB, ntaps, *cdf_tune(tune), CDF_S);
}

// cdf_conv_sp.h -- what the translation units of the split-precision / bf16 GEMM family share (round 6: k_conv_sp.hip was one 2600-line
// unit of 104 kernel instantiations, 3 minutes to compile; now one unit per kernel family, compiled in parallel):
//   k_conv_sp.hip         in-kernel-split GEMM + weight gradient, operand split / widen / weight packing, the k | v + context kernel
//   k_conv_spx.hip        generic pre-split gather-GEMM (LDS-DMA), split-K finish, the dispatcher and the C entry points of the pre-split GEMM
//   k_conv_halo.hip       3 x 3 stride-1 GEMM with the input tile + halo resident in LDS
//   k_conv_rowhalo.hip    resident row-halo stream kernel (64 / 128 input channels at 128-pixel width)
//   k_conv_wgrad_spx.hip  pre-split weight gradients (per-tap and row-of-taps kernels), their dispatcher and C entry points
#pragma once
#include "cdf_common.h"
#include "cdf_epilogue.h"
#include "colddiff.h"

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

// BFF: the kernel's epilogue family (cdf_epilogue.h: ids 1..6 for fp32 tensors, 7..11 for bf16 activation storage = the NS == 1 kernels)
template <int BM, int BN, int WM = 2, int WN = 2, bool BFF = false, class Args>
__device__ __forceinline__ void cdf_sp_epilogue(const Args& a, const SpPhase& ph, const f32x16_t (&acc)[BM / WM / 32][BN / WN / 32], float* cs,
                                                int tile_m, int tile_n, int M, int tid) {
    constexpr int CP = BN + 8, TM = BM / WM, TN = BN / WN, NTHR = 64 * WM * WN;
    const int lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN, half = lane >> 5, l31 = lane & 31;
    // specialised straight-line form (block-uniform choice): the fused operand loads of the whole tile are issued BEFORE the
    // accumulators go through LDS
    const bool fast = cdf_epi_tile_ok<BM, BN>(a, M) && cdf_epi_family_ok(a.epi, BFF);
    // (the K loop ends with a barrier: every wave is done with the operand tiles)
    auto dump = [&]() {
#pragma unroll
        for (int i = 0; i < TM / 32; ++i)
#pragma unroll
            for (int j = 0; j < TN / 32; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    cs[(wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * CP + wn * TN + j * 32 + l31] = acc[i][j][r];
    };
    if (a.epi == CDF_EPI_LNBWD) {                            // (block-uniform; the host guarantees whole tiles and Cout == BN)
        cdf_epi_lnbwd<BM, BN, NTHR>(a, cs, tile_m, tid, dump);
        return;
    }
    if (fast) {
        const long long trow = (long long)tile_m * BM;
        cdf_epi_dispatch<BFF>(a.epi, [&](auto spec) {
            using E = cdf_epi_fast<BN, BM, NTHR, decltype(spec)>;
            f32x4_t q[E::NR], bs[2];
            cdf_epi_load_bias<BN, NTHR>(a, bs, trow, tile_n * BN, tid);
            E::load(a, q, trow, tile_n * BN, tid);
            dump();
            CDF_LDS_BARRIER();                               // (LDS only: the operand loads stay in flight)
            E::template finish<false>(a, q, bs, cs, trow, tile_n * BN, tid);
        });
        return;
    }
    dump();
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
    int epi;                       // id of the specialised epilogue (cdf_epi_select; 0: the generic run-time-selected form)
    const float* ln_x;             // epi == CDF_EPI_LNBWD: LayerNorm input h (pitch ld_lnx), its statistics [M], the partial-sum output [tiles][2][Cout]
    const float* ln_mean;
    const float* ln_rstd;
    float* ln_part;
    int ld_lnx;
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
    int epi;                       // id of the specialised epilogue (cdf_epi_select; 0: the generic run-time-selected form)
    const float* ln_x;             // epi == CDF_EPI_LNBWD: LayerNorm input h (pitch ld_lnx), its statistics [M], the partial-sum output [tiles][2][Cout]
    const float* ln_mean;
    const float* ln_rstd;
    float* ln_part;
    int ld_lnx;
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

// ---- tuning: an explicit, optional argument of the GEMM entry points (include/colddiff.h: cdf_gemm_tuning) -----------------------------
// No mutable process-wide state: a NULL pointer means these defaults, anything else is read once per call.  The choices only select
// between kernels / tile shapes that compute the same sums (fp32 summation order aside).
static const cdf_gemm_tuning kTuneDefault = {(int)sizeof(cdf_gemm_tuning), 0, 0, 0, 1, 1, 1, 47, 1, 0, 1, 1, 1, 1, 1, 0, 1};
static inline const cdf_gemm_tuning* cdf_tune(const cdf_gemm_tuning* t) { return (t && t->size == (int)sizeof(cdf_gemm_tuning)) ? t : &kTuneDefault; }
static inline bool cdf_tune_ok(const cdf_gemm_tuning* t) {
    if (!t) return true;
    const bool bm_ok = t->tile_bm == 0 || t->tile_bm == 64 || t->tile_bm == 128 || (t->tile_bm == 256 && (t->tile_bn == 0 || t->tile_bn == 128));
    const bool bn_ok = t->tile_bn == 0 || t->tile_bn == 64 || t->tile_bn == 128;
    return t->size == (int)sizeof(cdf_gemm_tuning) && bm_ok && bn_ok && (t->max_bm == 0 || t->max_bm == 128 || t->max_bm == 256) &&
           (t->halo_bm == 0 || t->halo_bm == 128 || t->halo_bm == 256) && t->halo >= 0 && t->halo <= 127 && t->halo_min_tiles >= 0 && t->resident_reserve >= 0 && t->resident_reserve <= 248 && (t->rowhalo_stream == 0 || t->rowhalo_stream == 1) && (t->epilogue == 0 || t->epilogue == 1);
}
#define CDF_TUNE_CHECK(t, who)                                                                                                          \
    CDF_REQUIRE(cdf_tune_ok(t), who ": bad cdf_gemm_tuning (size %d, expected %d; tile_bm 0/64/128/256 (256 with tile_bn 0/128), tile_bn 0/64/128, " \
                                    "max_bm 0/128/256, halo_bm 0/128/256, halo 0..127, rowhalo_stream 0/1, resident_reserve 0..248, epilogue 0/1): start from cdf_gemm_tuning_default",            \
                (t) ? (t)->size : 0, (int)sizeof(cdf_gemm_tuning))

static inline int cdf_num_cus() {                                     // CUs of the current device (blocks of the resident kernels), a multiple of 8 XCDs
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


// ---- cross-unit launchers (one plain function per kernel family; the template dispatch lives next to the kernels) --------------------------------
// halo kernel: W in {16, 32, 64, 128}, n64: 64-wide N tiles, bm: 128 or 256 pixel rows per tile (the caller has checked that the geometry fits)
int cdf_launch_igemm_halo(int ns, int W, bool n64, int bm, const SpxArgs& a, int M, hipStream_t s);
// resident row-halo stream kernel; returns CDF_E_UNSUPPORTED when this build has no instance for W (the caller then falls through)
int cdf_launch_igemm_rowhalo(int ns, int W, bool n64, const SpxArgs& a, int M, hipStream_t s, int reserve);

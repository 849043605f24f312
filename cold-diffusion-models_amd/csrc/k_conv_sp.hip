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
#include "cdf_conv_sp.h"
#include <atomic>

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

    cdf_sp_epilogue<128, 128, 2, 2, SPLIT == 1>(a, ph, acc, (float*)smem_raw, tile_m, tile_n, M, tid);
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
    a.epi = cdf_epi_select(a);
    a.ln_x = nullptr; a.ln_mean = nullptr; a.ln_rstd = nullptr; a.ln_part = nullptr; a.ld_lnx = 0;
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


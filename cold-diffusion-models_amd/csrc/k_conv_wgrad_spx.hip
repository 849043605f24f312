#include "cdf_conv_sp.h"

// weight gradient with both operands pre-split ([pixels][ld] bf16 hi / lo planes)

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

    // Register-staged operand chunks, TWO ahead of the one being multiplied (round 6; round 3 measured this form at +2.6 ... +10.5 % on the
    // 128 x 128-pixel layers and dropped it for +0.2 % of the step of its day): with one chunk ahead the loads had one chunk's MFMAs to
    // land -- 18 per wave for a 64-channel side, 0.25 us against ~1.5 us of HBM latency, and a third of that in `bf16` mode -- so the two
    // blocks of a CU spent most of their time waiting for each other's loads.  Set s holds chunk c with c % 2 == s; chunk it + 1 goes to
    // LDS between the two k-steps of chunk it, its registers are refilled with chunk it + 3 right behind, and the barrier orders LDS only.
    // (128 x 64 tiles in split precision: the second register set takes the kernel from 125 to 156 VGPRs, i.e. from two resident blocks per
    //  CU to one -- measured -2.5 % -- so that instantiation keeps ONE chunk ahead.)
    constexpr bool X2 = !(TA == 128 && TB == 64 && NS == 3);
    struct Regs {
        u32x4_v ah[PASS_A], al[PASS_A], bh, bl;
    };
    Regs r0, r1;
    auto load_global = [&](Regs& r, int it) {
        const int m0 = m_lo + it * BK;
#pragma unroll
        for (int p = 0; p < PASS_A; ++p) {
            const unsigned ax = (unsigned)(x0 + a_xr[p] - 1), ay = (unsigned)(yc + a_sub[p] + dy);
            const bool ok = a_r[p] < nra && ax < (unsigned)W && ay < (unsigned)H && ca < a.CA;
            const long long pix = (long long)m0 + (a_sub[p] + dy) * W + a_xr[p] - 1;
            const size_t off = (size_t)(ok ? pix : 0) * (unsigned)a.lda + (unsigned)ca;
            r.ah[p] = *(const u32x4_v*)(ok ? a.a_hi + off : a.zero);
            if constexpr (NS == 3) r.al[p] = *(const u32x4_v*)(ok ? a.a_lo + off : a.zero);
        }
        {
            const bool ok = b_lane && cb < a.CB;
            const size_t off = (size_t)(m0 + (b_lane ? pb : 0)) * (unsigned)a.ldb + (unsigned)cb;
            r.bh = *(const u32x4_v*)(ok ? a.b_hi + off : a.zero);
            if constexpr (NS == 3) r.bl = *(const u32x4_v*)(ok ? a.b_lo + off : a.zero);
        }
        // next chunk (uniform scalars, selects only): 32 pixels further -- inside the row, to the next row(s), to the next image
        const int nx = x0 + (W < 32 ? 0 : 32);
        const int wrap = nx >= W ? 1 : 0;
        x0 = wrap ? 0 : nx;
        yc += (W < 32 ? 32 / W : 0) + wrap;
        yc = yc >= H ? yc - H : yc;
    };
    auto store_lds = [&](const Regs& r, int buf) {
        unsigned short* st = smem + buf * STAGE;
#pragma unroll
        for (int p = 0; p < PASS_A; ++p) {
            if (a_r[p] < NRAP) {
                const int so = a_r[p] * PITCH_A + (tid % VPR_A) * 8;
                *(u32x4_v*)(st + so) = r.ah[p];
                if constexpr (NS == 3) *(u32x4_v*)(st + PLANE_A + so) = r.al[p];
            }
        }
        if (b_lane) {
            const int so = pb * PITCH_B + (tid % VPR_B) * 8;
            *(u32x4_v*)(st + 2 * PLANE_A + so) = r.bh;
            if constexpr (NS == 3) *(u32x4_v*)(st + 2 * PLANE_A + PLANE_B + so) = r.bl;
            if (do_bsum) cdf_bf16x8_accum<NS>(bs_acc, r.bh, r.bl);
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

    auto mma_ks = [&](const unsigned short* sa, const unsigned short* sb, int ks) {
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
    };
    if constexpr (X2) {
        // one chunk: multiply chunk `it` (stage it & 1); `nxt` holds chunk it + 1 (-> the other stage, then refilled with chunk it + 3)
        auto step = [&](int it, Regs& nxt) {
            const unsigned short* sa = smem + (it & 1) * STAGE;
            const unsigned short* sb = sa + 2 * PLANE_A;
            mma_ks(sa, sb, 0);
            if (it + 1 < niter) store_lds(nxt, (it + 1) & 1);    // (that stage was last read in chunk it - 1: every wave is past the barrier that ended it)
            if (it + 3 < niter) load_global(nxt, it + 3);
            mma_ks(sa, sb, 1);
            CDF_LDS_BARRIER();                                   // (LDS traffic only: the chunks in flight stay in flight)
        };
        if (niter > 0) {
            load_global(r0, 0);
            store_lds(r0, 0);
            if (niter > 1) load_global(r1, 1);                   // set 1: odd chunks
            if (niter > 2) load_global(r0, 2);                   // set 0: even chunks
        }
        __syncthreads();
        for (int it = 0; it < niter; it += 2) {
            step(it, r1);
            if (it + 1 < niter) step(it + 1, r0);
        }
    } else {
        if (niter > 0) {
            load_global(r0, 0);
            store_lds(r0, 0);
        }
        __syncthreads();
        for (int it = 0; it < niter; ++it) {
            if (it + 1 < niter) load_global(r0, it + 1);
            const unsigned short* sa = smem + (it & 1) * STAGE;
            const unsigned short* sb = sa + 2 * PLANE_A;
            mma_ks(sa, sb, 0);
            mma_ks(sa, sb, 1);
            if (it + 1 < niter) store_lds(r0, (it + 1) & 1);
            __syncthreads();
        }
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


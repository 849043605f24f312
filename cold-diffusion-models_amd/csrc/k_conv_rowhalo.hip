#include "cdf_conv_sp.h"

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
    const bool fast_epi = cdf_epi_tile_ok<BM, BN>(a, M) && cdf_epi_family_ok(a.epi, NS == 1);
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
        // The lane's fragment rows, through an opaque register once per tile: every fragment address of the unrolled K loop derives from
        // them and is tile-invariant, so hipcc kept ~80 precomputed addresses live across the whole tile loop -- EPILOGUE included, where
        // the specialised form wants those registers for its operand prefetch (with them it spilled 289 VGPRs, the accumulators among them).
        // Recomputed per tile (a few dozen adds) they are dead after the last K step.
        int row0t[MT], l31t = l31;
#pragma unroll
        for (int i = 0; i < MT; ++i) row0t[i] = row0[i];
#ifndef CDF_EMU
#pragma unroll
        for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(row0t[i]));
        asm volatile("" : "+v"(l31t));
#endif
        const int swbt = (l31t >> 2) & 3;
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
                                const int row = row0t[i] + dx;
                                const int off = row * RE + ((ks * 2 + half) ^ ((row >> 2) & 3)) * 8;
                                ah[ks][i] = *(const bf16x8_v*)(sa + off);
                                if constexpr (NS == 3) al[ks][i] = *(const bf16x8_v*)(sa + PLANE_A + off);
                            }
                            const int kc = ((ks * 2 + half) ^ swbt) * 8;
#pragma unroll
                            for (int j = 0; j < NT; ++j) {
                                const int offb = (wn * (BN / WN) + j * 32 + l31t) * RE + kc;
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
        auto dump = [&](int hp) {                            // the accumulators of the waves whose 64 rows belong to pass hp -> staging tile
            if ((wm * 64) / EROWS == hp) {
                // (one opaque base per call, the 64 element offsets as ds_write immediates: with three call sites hipcc otherwise precomputes
                //  all 64 addresses outside the tile loop and spills them)
                int base = (wm * 64 - hp * EROWS + 4 * half) * CP + wn * (BN / WN) + l31;
#ifndef CDF_EMU
                asm volatile("" : "+v"(base));
#endif
                float* cb = cs + base;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            cb[(i * 32 + (r & 3) + 8 * (r >> 2)) * CP + j * 32] = acc[i][j][r];
            }
        };
        if (fast_epi) {
            // specialised straight-line epilogue (block-uniform).  The fused operand loads of pass 0 are issued before its dump; pass 1's
            // operand of row k is requested inside pass 0 as soon as row k's register is free, in front of that row's stores (two full
            // register arrays next to the accumulators do not fit 256 VGPRs at BN = 128) -- a wait for a loaded operand then needs only
            // OLDER stores acknowledged, never the ones just issued (stores count in vmcnt on gfx9).
            const long long trow = (long long)tile_m * BM;
            cdf_epi_dispatch<NS == 1>(a.epi, [&](auto spec) {
                using E = cdf_epi_fast<BN, EROWS, 512, decltype(spec)>;
                f32x4_t q[E::NR], bs[2];
                cdf_epi_load_bias<BN, 512>(a, bs, trow, tile_n * BN, tid);
                E::load(a, q, trow, tile_n * BN, tid);
                dump(0);
                CDF_LDS_BARRIER();
                E::template finish<true>(a, q, bs, cs, trow, tile_n * BN, tid, trow + EROWS);       // (refills q for pass 1)
                CDF_LDS_BARRIER();
                dump(1);
                CDF_LDS_BARRIER();
                E::template finish<false>(a, q, bs, cs, trow + EROWS, tile_n * BN, tid);
                CDF_LDS_BARRIER();
            });
        } else {
#pragma unroll 1
            for (int hp = 0; hp < 2; ++hp) {
                dump(hp);
                CDF_LDS_BARRIER();                               // (LDS traffic only: the stores of the previous pass keep draining)
                cdf_epilogue_rows<BN, EROWS, 512>(a, ph, a.y, cs, tile_m * BM + hp * EROWS, tile_n * BN, M, tid, [](int p) { return p; });
                CDF_LDS_BARRIER();
            }
        }
        img = img_n; y0 = y0_n; tile_n = tile_n_n; tile_m = tile_m_n;
        v += gridDim.x;
    }
    CDF_WAIT_DMA_LEAVE(0);                                   // (the requests past the last tile fetched the zero page / weights: let them land)
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


// Widths with an instance: 128 (the default dispatch: the > 64-channel outputs at 128-pixel width) in the device build; the host SIMT-simulator
// build also carries 64 / 32 / 16 (cdf_gemm_tuning.halo bit 64: the CPU suite drives the resident kernel at widths the simulator finishes in seconds).
template <int NS>
static int launch_rowhalo_ns(int W, bool n64, const SpxArgs& a, int M, hipStream_t s, int reserve) {
#define CDF_ROWHALO_CASE(WW) \
    if (W == WW) return n64 ? launch_igemm_rowhalo_stream<NS, WW, 64>(a, M, s, reserve) : launch_igemm_rowhalo_stream<NS, WW, 128>(a, M, s, reserve);
    CDF_ROWHALO_CASE(128)
#ifdef CDF_EMU
    CDF_ROWHALO_CASE(64) CDF_ROWHALO_CASE(32) CDF_ROWHALO_CASE(16)
#endif
#undef CDF_ROWHALO_CASE
    return CDF_E_UNSUPPORTED;
}

int cdf_launch_igemm_rowhalo(int ns, int W, bool n64, const SpxArgs& a, int M, hipStream_t s, int reserve) {
    return ns == 3 ? launch_rowhalo_ns<3>(W, n64, a, M, s, reserve) : launch_rowhalo_ns<1>(W, n64, a, M, s, reserve);
}

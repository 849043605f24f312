#include "cdf_conv_sp.h"

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

    cdf_sp_epilogue<BM, BN, WM, WN, NS == 1>(a, ph, acc, (float*)smem_raw, tile_m, tile_n, M, tid);
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


template <int NS>
static int launch_halo_ns(int W, bool n64, int bm, const SpxArgs& a, int M, hipStream_t s) {
#define CDF_HALO_CASE(WW)                                                                                              \
    if (W == WW) {                                                                                                     \
        if (bm == 256)                                                                                                 \
            return n64 ? launch_igemm_halo<NS, WW, 64, 256>(a, M, s)                                                    \
                       : launch_igemm_halo<NS, WW, 128, WW <= 64 ? 256 : 128>(a, M, s);                                 \
        return n64 ? launch_igemm_halo<NS, WW, 64, 128>(a, M, s) : launch_igemm_halo<NS, WW, 128, 128>(a, M, s);        \
    }
    CDF_HALO_CASE(128) CDF_HALO_CASE(64) CDF_HALO_CASE(32) CDF_HALO_CASE(16)
#undef CDF_HALO_CASE
    cdf_set_error("conv_igemm_halo: no kernel for image width %d", W);
    return CDF_E_UNSUPPORTED;
}

int cdf_launch_igemm_halo(int ns, int W, bool n64, int bm, const SpxArgs& a, int M, hipStream_t s) {
    return ns == 3 ? launch_halo_ns<3>(W, n64, bm, a, M, s) : launch_halo_ns<1>(W, n64, bm, a, M, s);
}

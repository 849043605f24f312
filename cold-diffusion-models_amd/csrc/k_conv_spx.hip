// k_conv_spx.hip -- the generic pre-split gather-GEMM on the bf16 matrix cores (operands already split into bf16 hi / lo planes, tiles brought
// in by LDS-DMA), its split-K finish, and the dispatcher + C entry points of the pre-split GEMM family (see cdf_conv_sp.h for the other units).
#include "cdf_conv_sp.h"

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
    cdf_sp_epilogue<BM, BN, WM, WN, NS == 1>(a, ph, acc, (float*)smem_raw, tile_m, tile_n, M, tid);
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

extern "C" int cdf_gemm_tuning_default(cdf_gemm_tuning* t) {
    CDF_REQUIRE(t, "cdf_gemm_tuning_default: null pointer");
    *t = kTuneDefault;
    return 0;
}

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
                               const cdf_gemm_tuning& T, hipStream_t s, int* bm_out = nullptr) {
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
            if ((W == 128 || W == 64 || W == 32 || W == 16) && H % (256 / W) == 0) {
                const int rc = cdf_launch_igemm_rowhalo(NS, W, n64, a, M, s, T.resident_reserve);
                if (rc != CDF_E_UNSUPPORTED) return rc;      // (no instance for this width in this build: the halo / generic kernels below take it)
            }
        }
        if (dx_ok && tiles >= (T.halo_min_tiles > 0 ? T.halo_min_tiles : 1)) {
            if ((W == 128 || W == 64 || W == 32 || W == 16) && (T.halo & (W / 16)) && H % (128 / W) == 0 && (W < 128 || n64 || (T.halo & 16))) {
                // 256-pixel tiles (half the weight bytes per MFMA) when they still give every CU a tile and fit the LDS
                // (at 128-pixel width only next to 64-wide weight stages)
                const bool bm256 = (W <= 64 || n64) && T.halo_bm != 128 && H % (256 / W) == 0 && M % 256 == 0 &&
                                   (T.halo_bm == 256 || (long long)(M / 256) * cdf_cdiv(Cout, n64 ? 64 : 128) >= 256);
                if (bm_out) *bm_out = bm256 ? 256 : 128;
                return cdf_launch_igemm_halo(NS, W, n64, bm256 ? 256 : 128, a, M, s);
            }
        }
    }
    if (bm_out) *bm_out = bm;                               // (the generic kernels below take the row tile chosen above)
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
    a.epi = cdf_tune(tune)->epilogue ? cdf_epi_select(a) : 0;
    a.ln_x = nullptr; a.ln_mean = nullptr; a.ln_rstd = nullptr; a.ln_part = nullptr; a.ld_lnx = 0;
    CDF_REQUIRE(!y_hi || a.vec, "cdf_conv_gemm_bf16x: split output planes need the vectorised epilogue (aligned pointers, pitches %% 4)");
    CDF_REQUIRE(!(io_bf16 & 7) || a.vec, "cdf_conv_gemm_bf16x_io: bf16 epilogue operands need the vectorised epilogue (16-byte-aligned pointers, pitches %% 4, Cout %% 4)");
    int rc = fill_phases(a.ph, nphase, phase_desc, "cdf_conv_gemm_bf16x");
    if (rc) return rc;
    a.ksplit = 1; a.ks_ws = ws; a.ks_ld = 0;
    return x_lo ? dispatch_gemm_bf16x<3>(a, B, H, W, Cin, Cout, QH, QW, os, is, nphase, ws ? ws_floats : 0, *cdf_tune(tune), CDF_S)
                : dispatch_gemm_bf16x<1>(a, B, H, W, Cin, Cout, QH, QW, os, is, nphase, ws ? ws_floats : 0, *cdf_tune(tune), CDF_S);
}


// Data gradient of a 3 x 3 stride-1 "same" convolution whose INPUT was a channel LayerNorm's output, with that LayerNorm's backward applied in
// the epilogue (cdf_epilogue.h: cdf_epi_lnbwd): dh = LayerNorm'(h; mean, rstd, g)[conv_dgrad(dy)], dg / db (+)= the parameter gradients.
// Needs one N tile to hold every channel (Cout = the LayerNorm's width = 64 or 128) and whole row tiles (M % 256 == 0);
// cdf_conv_gemm_bf16x_lnbwd_ok tells.  part: >= (M / 64) * 2 * Cout floats of scratch.
extern "C" int cdf_conv_gemm_bf16x_lnbwd_ok(int B, int H, int W, int Cin, int Cout, int nphase, int ntaps) {
    const long long M = (long long)B * H * W;
    return (Cout == 64 || Cout == 128) && nphase == 1 && ntaps == 9 && M % 256 == 0 && M < (1ll << 31) && Cin % 32 == 0 && Cin >= 64;
}
extern "C" int cdf_conv_gemm_bf16x_lnbwd(const void* x_hi, const void* x_lo, int ldx, const void* zero, const void* w_hi, const void* w_lo, int ldk,
                                         int B, int H, int W, int Cin, int Cout, const int* phase_desc, const float* ln_x, int ld_lnx,
                                         const float* ln_mean, const float* ln_rstd, const float* ln_g, float* dh, int lddh, float* dg, float* db,
                                         float* part, const cdf_gemm_tuning* tune, void* stream) {
    CDF_REQUIRE(x_hi && zero && w_hi && phase_desc && ln_x && ln_mean && ln_rstd && ln_g && dh && dg && db && part, "cdf_conv_gemm_bf16x_lnbwd: null pointer");
    CDF_TUNE_CHECK(tune, "cdf_conv_gemm_bf16x_lnbwd");
    CDF_REQUIRE((x_lo != nullptr) == (w_lo != nullptr), "cdf_conv_gemm_bf16x_lnbwd: pass both lo planes or neither");
    CDF_REQUIRE(cdf_conv_gemm_bf16x_lnbwd_ok(B, H, W, Cin, Cout, 1, phase_desc[2]), "cdf_conv_gemm_bf16x_lnbwd: needs a 3 x 3 stride-1 layer with 64 or 128 output channels, B*H*W %% 256 == 0");
    CDF_REQUIRE(((((uintptr_t)x_hi) | ((uintptr_t)x_lo) | ((uintptr_t)zero) | ((uintptr_t)w_hi) | ((uintptr_t)w_lo) | ((uintptr_t)ln_x) | ((uintptr_t)ln_g) | ((uintptr_t)dh)) & 15) == 0 &&
                ldx % 8 == 0 && ldx >= Cin && ldk % 32 == 0 && ldk >= Cin && ld_lnx % 4 == 0 && ld_lnx >= Cout && lddh % 4 == 0 && lddh >= Cout,
                "cdf_conv_gemm_bf16x_lnbwd: alignment / pitches");
    SpxArgs a;
    a.x_hi = (const unsigned short*)x_hi; a.x_lo = (const unsigned short*)x_lo; a.zero = (const unsigned short*)zero;
    a.w_hi = (const unsigned short*)w_hi; a.w_lo = (const unsigned short*)w_lo; a.y = dh;
    a.bias = ln_g; a.sbias = nullptr; a.res = nullptr; a.pre = nullptr; a.mul = nullptr;
    a.ldx = ldx; a.ldk = ldk; a.ldy = lddh; a.ld_sbias = 0; a.ldr = 0; a.ldp = 0; a.ldm = 0;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.OH = H; a.OW = W; a.Cout = Cout; a.QH = H; a.QW = W; a.os = 1; a.is = 1;
    a.act = 0; a.mul_mode = 0; a.accumulate = 0; a.nphase = 1; a.vec = 1;
    a.ys_hi = nullptr; a.ys_lo = nullptr; a.ld_ys = 0; a.io_bf = 0;
    a.epi = CDF_EPI_LNBWD;
    a.ln_x = ln_x; a.ln_mean = ln_mean; a.ln_rstd = ln_rstd; a.ln_part = part; a.ld_lnx = ld_lnx;
    int rc = fill_phases(a.ph, 1, phase_desc, "cdf_conv_gemm_bf16x_lnbwd");
    if (rc) return rc;
    a.ksplit = 1; a.ks_ws = nullptr; a.ks_ld = 0;
    // every kernel but the resident two-pass one runs the whole-tile epilogue this form lives in; N tiles as wide as the layer
    cdf_gemm_tuning T = *cdf_tune(tune);
    T.rowhalo_stream = 0; T.small_n64 = 0; T.tile_bn = 0; T.splitk = 0;
    int bm = 0;
    rc = x_lo ? dispatch_gemm_bf16x<3>(a, B, H, W, Cin, Cout, H, W, 1, 1, 1, 0, T, CDF_S, &bm) : dispatch_gemm_bf16x<1>(a, B, H, W, Cin, Cout, H, W, 1, 1, 1, 0, T, CDF_S, &bm);
    if (rc) return rc;
    CDF_REQUIRE(bm == 64 || bm == 128 || bm == 256, "cdf_conv_gemm_bf16x_lnbwd: the dispatcher reported no row tile");
    return cdf_norm_param_reduce(part, (int)((long long)B * H * W / bm), Cout, dg, db, 1, stream);
}

// Shared GEMM epilogue: accumulator tile -> LDS transpose -> float4 rows.
//
// The MFMA 32x32 accumulator layout gives each lane ONE channel of 16 scattered pixels, i.e. dword stores
// of two 128-byte pieces per instruction.  Measured on MI355X that pattern writes at ~1 TB/s and was 40 % of
// the whole conv kernel on the 128x128-pixel layers.  The tile is therefore transposed through LDS (free once
// the K loop is over): [rows][BN + 8] fp32 (pitch BN+8: the two half-waves of a ds_write_b32 land 32 banks
// apart), after which every lane owns 4 consecutive channels of one pixel: float4 loads/stores of the output
// and of every fused operand (bias, per-sample bias, pre-activation, activation-gradient source, residual),
// BN/4 lanes = one contiguous pixel row.
// Args::ys_hi / ys_lo / ld_ys (optional): the stored values again as bf16 hi / lo planes (operand split fused into the producer).
// Args::io_bf (CDF_IO_*_BF16 bits): res / pre / mul are bf16 tensors (bf16 activation storage; needs the vector layout, checked on the host).
// Args::vec (host-computed, cdf_epi_vec_ok) = all pitches % 4 == 0, Cout % 4 == 0, 16-byte-aligned pointers;
// otherwise the same code runs with per-element accesses.
#pragma once
#include "cdf_common.h"

static inline int cdf_epi_vec_ok(int Cout, const float* y, int ldy, const float* bias, const float* sbias, int ld_sbias, const float* res,
                                 int ldr, const float* pre, int ldp, const float* mul, int ldm) {
    const uintptr_t ptrs = (uintptr_t)y | (uintptr_t)bias | (uintptr_t)sbias | (uintptr_t)res | (uintptr_t)pre | (uintptr_t)mul;
    const int pitches = ldy | (sbias ? ld_sbias : 0) | (res ? ldr : 0) | (pre ? ldp : 0) | (mul ? ldm : 0);
    return (Cout % 4 == 0 && (ptrs & 15) == 0 && (pitches & 3) == 0) ? 1 : 0;
}

__device__ __forceinline__ void cdf_ld4(float* v, const float* p, int n, bool vec) {
    if (vec) {
        const float4 t = *(const float4*)p;
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = e < n ? p[e] : 0.f;
    }
}

__device__ __forceinline__ void cdf_st4(float* p, const float* v, int n, bool vec) {
    if (vec) {
        *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < n) p[e] = v[e];
    }
}

// One pass over RP transposed rows held in cs ([RP][BN + 8]); rowmap(p) = row of pass-row p inside the block tile.
// NTHR threads (default 256).  Contains no barrier (callers sync around it).
// pix_off (batched launches whose per-batch outputs are whole rows of ONE output tensor): added to the output pixel index, so
// that Y, pre, mul and res are all addressed from their own, un-offset bases.
template <int BN, int RP, int NTHR = 256, class Args, class Phase, class RowMap>
__device__ __forceinline__ void cdf_epilogue_rows(const Args& a, const Phase& ph, float* Y, const float* cs, int m_base, int n_base, int M,
                                                  int tid, RowMap rowmap, long long pix_off = 0) {
    constexpr int CP = BN + 8, TPR = BN / 4, RPS = NTHR / TPR;
    const int c4 = (tid % TPR) * 4, co = n_base + c4;
    if (co >= a.Cout) return;
    const int nval = a.Cout - co < 4 ? a.Cout - co : 4;
    const bool vec = a.vec != 0;
    const bool direct = (a.os == 1 && a.QH == a.OH && a.QW == a.OW);
    const int qhw = a.QH * a.QW;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) cdf_ld4(bv, a.bias + co, nval, vec);
    // Element offsets of the row in the five tensors the epilogue may touch, carried from row to row: the pixel index moves by a small
    // step (RPS for the kernels whose tile pixels are a contiguous range), so offset += step * pitch is one 32 x 32 -> 64-bit multiply-add
    // (or, with a constant step, a loop-invariant the compiler hoists) instead of a 64 x 32-bit product per tensor and row -- the GELU /
    // residual epilogues of the short-K layers are VALU-bound and these products were a third of their vector work.
    long long prev = 0, o_pre = co, o_mul = co, o_res = co, o_y = co, o_ys = co;
    bool first = true;
    for (int p = tid / TPR; p < RP; p += RPS) {
        const int m = m_base + rowmap(p);
        if (m >= M) continue;
        long long opix;
        int b;
        if (direct) {
            opix = m;
            b = m / qhw;
        } else {
            const int qx = m % a.QW, t2 = m / a.QW;
            const int qy = t2 % a.QH;
            b = t2 / a.QH;
            opix = ((long long)b * a.OH + qy * a.os + ph.oy) * a.OW + qx * a.os + ph.ox;
        }
        opix += pix_off;
        if (first) {
            o_pre += opix * a.ldp; o_mul += opix * a.ldm; o_res += opix * a.ldr; o_y += opix * a.ldy; o_ys += opix * a.ld_ys;
            first = false;
        } else {
            const int d = (int)(opix - prev);
            o_pre += (long long)d * a.ldp; o_mul += (long long)d * a.ldm; o_res += (long long)d * a.ldr; o_y += (long long)d * a.ldy;
            o_ys += (long long)d * a.ld_ys;
        }
        prev = opix;
        const float4 t = *(const float4*)(cs + p * CP + c4);
        float v[4] = {t.x + bv[0], t.y + bv[1], t.z + bv[2], t.w + bv[3]};
        float u[4];
        if (a.sbias) {
            cdf_ld4(u, a.sbias + (long long)b * a.ld_sbias + co, nval, vec);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += u[e];
        }
        if (a.pre && (a.io_bf & CDF_IO_PRE_GRAD)) {
            // `pre` receives act'(v): one erf / exp evaluation serves GELU and its derivative, and the data-gradient epilogue of the
            // backward pass multiplies by a loaded value instead of evaluating them again (host: act is GELU or SiLU here)
            float gr[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (a.act == 1) {
                    v[e] = cdf_gelu_both(v[e], gr[e]);
                } else {
                    gr[e] = cdf_silu_grad(v[e]);
                    v[e] = cdf_silu(v[e]);
                }
            }
            if (a.io_bf & CDF_IO_PRE_BF16) cdf_st4_bf(a.pre, o_pre, gr);
            else cdf_st4(a.pre + o_pre, gr, nval, vec);
        } else {
        if (a.pre) {
            if (a.io_bf & CDF_IO_PRE_BF16) cdf_st4_bf(a.pre, o_pre, v);
            else cdf_st4(a.pre + o_pre, v, nval, vec);
        }
        if (a.act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = cdf_gelu(v[e]);
        } else if (a.act == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = cdf_silu(v[e]);
        } else if (a.act == 3) {                             // ReLU (the FID InceptionV3's BasicConv2d, Fid/inception.py)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f); // relu
        }
        }
        if (a.mul_mode) {
            if (a.io_bf & CDF_IO_MUL_BF16) cdf_ld4_bf(u, a.mul, o_mul);
            else cdf_ld4(u, a.mul + o_mul, nval, vec);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= (a.mul_mode == 1 ? cdf_gelu_grad(u[e]) : (a.mul_mode == 2 ? cdf_silu_grad(u[e]) : u[e]));
        }
        if (a.res) {
            if (a.io_bf & CDF_IO_RES_BF16) cdf_ld4_bf(u, a.res, o_res);
            else cdf_ld4(u, a.res + o_res, nval, vec);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += u[e];
        }
        if (Y) {                                             // (null: only the split planes below are wanted)
            float* dst = Y + o_y;
            if (a.accumulate) {
                cdf_ld4(u, dst, nval, vec);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += u[e];
            }
            cdf_st4(dst, v, nval, vec);
        }
        if (a.ys_hi && vec)                                  // the consumer GEMMs' bf16 hi / lo planes of the same values
            cdf_split_store4(a.ys_hi + o_ys, a.ys_lo ? a.ys_lo + o_ys : nullptr, v);
    }
}

// ================================================================================================
// Specialised epilogues (round 6).  cdf_epilogue_rows above selects every variant per ROW at run time; compiled, that is a rolled loop of
// ~1100 instructions and 118 branches in which every fused operand load (residual, GELU' source, accumulate) sits inside a branch --
// hipcc waits vmcnt(0) right behind such a load, and on gfx9 stores count in vmcnt too, so each row waited for its own load AND for the
// previous row's stores to be acknowledged: 8-16 dependent HBM round trips per thread and tile (ISA of round 5, VERDICT r5 Weak 6).
// Here the operation list is a template parameter (EpiSpec), the pass is straight-line code, and it is split in two halves:
//   load()   issues every operand load of the pass -- before the accumulators go through LDS, so they fly under the ds_writes and the barrier,
//   finish() reads the transposed rows, applies bias / activation / multiply / residual and stores; stores come after all loads in
//            program order, so no wait on a load ever waits for a store.
// Preconditions (checked by cdf_epi_select on the host + cdf_epi_tile_ok in the kernel, else the generic form runs): the vector layout
// (Args::vec), output pixel == GEMM row (os == 1, QH == OH, QW == OW), whole tiles (M % rows-per-block-tile == 0, Cout % BN == 0: no thread
// leaves early, the code has no divergent branch at all), one image per tile
// when there is a per-sample bias.  Same arithmetic in the same order as cdf_epilogue_rows: bit-identical results.
// ================================================================================================
typedef unsigned cdf_u32x2 __attribute__((ext_vector_type(2)));
#ifdef CDF_EMU
#define CDF_OPAQUE_V(x) ((void)0)
#else
#define CDF_OPAQUE_V(x) asm volatile("" : "+v"(x))
#endif

// ACT: 0 none, 1 GELU, 2 SiLU.  PRE: 0 none, 1 pre-activation stored, 2 act'(pre-activation) stored.  MUL: 0 none, 1 x GELU'(mul), 2 x SiLU'(mul),
// 3 x mul.  RES: + residual.  ACC: + previous contents of Y.  HASY: fp32 output written.  YS: 0 no planes, 1 bf16 hi plane, 2 hi + lo planes.
// IO: CDF_IO_{RES,PRE,MUL}_BF16 bits (that operand is a bf16 tensor).
template <int ACT_, int PRE_, int MUL_, bool RES_, bool ACC_, bool HASY_, int YS_, int IO_>
struct EpiSpec {
    static constexpr int ACT = ACT_, PRE = PRE_, MUL = MUL_, YS = YS_, IO = IO_;
    static constexpr bool RES = RES_, ACC = ACC_, HASY = HASY_;
};
// the instantiated list; ids are what cdf_epi_select returns (0 = generic).  1..6: fp32 tensors (split-precision mode), 7..11: bf16 activation storage
typedef EpiSpec<0, 0, 0, false, false, true, 0, 0> EpiPlain;                                  // 1  [bias] -> Y
typedef EpiSpec<0, 0, 0, true, false, true, 0, 0> EpiRes;                                     // 2  [bias] + residual -> Y
typedef EpiSpec<1, 1, 0, false, false, false, 2, 0> EpiGeluPrePlanes;                         // 3  bias, pre stored, GELU -> hi / lo planes
typedef EpiSpec<1, 0, 0, false, false, false, 2, 0> EpiGeluPlanes;                            // 4  bias, GELU -> hi / lo planes (no-grad forward)
typedef EpiSpec<0, 0, 1, false, false, false, 2, 0> EpiMulGeluPlanes;                         // 5  x GELU'(pre) -> hi / lo planes
typedef EpiSpec<0, 0, 0, false, true, true, 0, 0> EpiAcc;                                     // 6  Y += .
typedef EpiSpec<0, 0, 0, false, false, false, 1, 0> EpiBfPlain;                               // 7  [bias] -> bf16
typedef EpiSpec<0, 0, 0, true, false, false, 1, CDF_IO_RES_BF16> EpiBfRes;                    // 8  [bias] + bf16 residual -> bf16
typedef EpiSpec<1, 1, 0, false, false, false, 1, CDF_IO_PRE_BF16> EpiBfGeluPre;               // 9  bias, bf16 pre stored, GELU -> bf16
typedef EpiSpec<1, 0, 0, false, false, false, 1, 0> EpiBfGelu;                                // 10 bias, GELU -> bf16
typedef EpiSpec<0, 0, 1, false, false, false, 1, CDF_IO_MUL_BF16> EpiBfMulGelu;               // 11 x GELU'(bf16 pre) -> bf16
#define CDF_EPI_NSPEC 11

// host side: the id of the specialised epilogue that computes exactly what these arguments ask for, or 0
template <class Args>
static inline int cdf_epi_select(const Args& a) {
    if (!a.vec || a.os != 1 || a.QH != a.OH || a.QW != a.OW) return 0;
    const bool pre = a.pre != nullptr, mul = a.mul_mode != 0, res = a.res != nullptr, acc = a.accumulate != 0, y = a.y != nullptr;
    const int ys = a.ys_hi ? (a.ys_lo ? 2 : 1) : 0, io = a.io_bf;
    if (io & CDF_IO_PRE_GRAD) return 0;
    auto is = [&](int ACT, int PRE, int MUL, bool RES, bool ACC, bool HASY, int YS, int IO) {
        return a.act == ACT && (pre ? 1 : 0) == PRE && a.mul_mode == MUL && res == RES && acc == ACC && y == HASY && ys == YS &&
               (io & ((PRE ? CDF_IO_PRE_BF16 : 0) | (MUL ? CDF_IO_MUL_BF16 : 0) | (RES ? CDF_IO_RES_BF16 : 0))) == IO;
    };
    (void)mul;
    if (is(0, 0, 0, false, false, true, 0, 0)) return 1;
    if (is(0, 0, 0, true, false, true, 0, 0)) return 2;
    if (is(1, 1, 0, false, false, false, 2, 0)) return 3;
    if (is(1, 0, 0, false, false, false, 2, 0)) return 4;
    if (is(0, 0, 1, false, false, false, 2, 0)) return 5;
    if (is(0, 0, 0, false, true, true, 0, 0)) return 6;
    if (is(0, 0, 0, false, false, false, 1, 0)) return 7;
    if (is(0, 0, 0, true, false, false, 1, CDF_IO_RES_BF16)) return 8;
    if (is(1, 1, 0, false, false, false, 1, CDF_IO_PRE_BF16)) return 9;
    if (is(1, 0, 0, false, false, false, 1, 0)) return 10;
    if (is(0, 0, 1, false, false, false, 1, CDF_IO_MUL_BF16)) return 11;
    return 0;
}

// kernel side (block-uniform): may this block tile of TILE_ROWS GEMM rows take the specialised form?
template <int TILE_ROWS, int BN, class Args>
__device__ __forceinline__ bool cdf_epi_tile_ok(const Args& a, int M) {
    return a.epi != 0 && M % TILE_ROWS == 0 && a.Cout % BN == 0 && (a.sbias == nullptr || (a.QH * a.QW) % TILE_ROWS == 0);
}

template <bool BF>
__device__ __forceinline__ f32x4_t cdf_ld_raw4(const void* base, unsigned off) {      // off in elements of the tensor's own type; a bf16 quad sits in lanes 0, 1
    if constexpr (BF) {
        const cdf_u32x2 t = *(const cdf_u32x2*)((const unsigned short*)base + off);
        f32x4_t r = {__uint_as_float(t[0]), __uint_as_float(t[1]), 0.f, 0.f};
        return r;
    } else {
        return *(const f32x4_t*)((const float*)base + off);
    }
}
template <bool BF>
__device__ __forceinline__ f32x4_t cdf_cvt_raw4(const f32x4_t& r) {                   // the loaded quad as four floats
    if constexpr (BF) {
        const unsigned x = __float_as_uint(r[0]), y = __float_as_uint(r[1]);
        f32x4_t v = {__uint_as_float(x << 16), __uint_as_float(x & 0xFFFF0000u), __uint_as_float(y << 16), __uint_as_float(y & 0xFFFF0000u)};
        return v;
    } else {
        return r;
    }
}
template <bool BF>
__device__ __forceinline__ void cdf_st_raw4(void* base, unsigned off, const f32x4_t& v) {                // bf16: round to nearest even
    if constexpr (BF) {
        cdf_u32x2 h;
        h[0] = cdf_pack2bf(v[0], v[1]);
        h[1] = cdf_pack2bf(v[2], v[3]);
        *(cdf_u32x2*)((unsigned short*)base + off) = h;
    } else {
        *(f32x4_t*)((float*)base + off) = v;
    }
}

// bs[0] = bias quad, bs[1] = per-sample bias quad of the image row0 lies in (one image per tile: cdf_epi_tile_ok) -- unconditional loads from
// clamped addresses (zeros where the tensor is absent), issued with the operand loads BEFORE any store of the tile: a load whose value is
// awaited behind stores waits for every one of them (vmcnt counts in order).
template <int BN, int NTHR, class Args>
__device__ __forceinline__ void cdf_epi_load_bias(const Args& a, f32x4_t (&bs)[2], long long row0, int n_base, int tid) {
    int c4 = (tid % (BN / 4)) * 4;
    CDF_OPAQUE_V(c4);
    const int co = n_base + c4;
    const long long b = a.sbias ? row0 / (a.QH * a.QW) : 0;
    const float* pb = a.bias ? a.bias + co : (const float*)cdf_zero_page;
    const float* ps = a.sbias ? a.sbias + b * a.ld_sbias + co : (const float*)cdf_zero_page;
    bs[0] = *(const f32x4_t*)pb;
    bs[1] = *(const f32x4_t*)ps;
}

// One pass over RP transposed rows ([RP][BN + 8] floats in LDS) by NTHR threads: thread t owns channel quad t % (BN/4) of rows t / (BN/4) + k RPS.
// q[NR]: the ONE fused operand a spec loads per row (residual, multiply source or the output's previous contents: never two of them in
// the instantiated list), raw -- the register array has the same type for every spec, so load and finish can be dispatched separately
// with the accumulator dump and the barrier between them written once.
template <int BN, int RP, int NTHR, class S>
struct cdf_epi_fast {
    static constexpr int CP = BN + 8, TPR = BN / 4, RPS = NTHR / TPR, NR = RP / RPS;
    static_assert(NR >= 1 && NR * RPS == RP, "rows of a pass must divide among the threads");
    static_assert((S::RES ? 1 : 0) + (S::MUL ? 1 : 0) + (S::ACC ? 1 : 0) <= 1, "one loaded operand per row");
    static constexpr bool RES_BF = (S::IO & CDF_IO_RES_BF16) != 0, MUL_BF = (S::IO & CDF_IO_MUL_BF16) != 0, PRE_BF = (S::IO & CDF_IO_PRE_BF16) != 0;
    // Tensors are addressed as (block-uniform row base: 64-bit, scalar) + (this thread's 32-bit element offset): a pass spans
    // RP x pitch elements, far below 2^32.  row0 = first GEMM row (= output pixel) of the pass.
    // the fused operand of pass row k (row0 = first GEMM row of the pass, co = channel, p0 = the thread's first row)
    template <class Args>
    __device__ __forceinline__ static f32x4_t load_row(const Args& a, long long row0, int co, int p0, int k) {
        if constexpr (S::RES) {
            const void* base = RES_BF ? (const void*)((const unsigned short*)a.res + row0 * a.ldr) : (const void*)((const float*)a.res + row0 * a.ldr);
            return cdf_ld_raw4<RES_BF>(base, (unsigned)(p0 + k * RPS) * (unsigned)a.ldr + (unsigned)co);
        } else if constexpr (S::MUL != 0) {
            const void* base = MUL_BF ? (const void*)((const unsigned short*)a.mul + row0 * a.ldm) : (const void*)((const float*)a.mul + row0 * a.ldm);
            return cdf_ld_raw4<MUL_BF>(base, (unsigned)(p0 + k * RPS) * (unsigned)a.ldm + (unsigned)co);
        } else {
            return *(const f32x4_t*)(a.y + row0 * a.ldy + (unsigned)(p0 + k * RPS) * (unsigned)a.ldy + (unsigned)co);
        }
    }
    static constexpr bool LOADS = S::RES || S::MUL != 0 || S::ACC;
    template <class Args>
    __device__ __forceinline__ static void load(const Args& a, f32x4_t (&q)[NR], long long row0, int n_base, int tid) {
        if constexpr (LOADS) {
            int c4 = (tid % TPR) * 4, p0 = tid / TPR;
            CDF_OPAQUE_V(c4);                                // (see finish: nothing of a thread's address arithmetic may be hoisted out of a tile loop)
            CDF_OPAQUE_V(p0);
            const int co = n_base + c4;
#pragma unroll
            for (int k = 0; k < NR; ++k) q[k] = load_row(a, row0, co, p0, k);
        }
    }
    // NEXT: as soon as row k's operand has been consumed its register is refilled with the operand of row k of the pass starting at
    // next_row0 -- issued before row k's stores, so a later wait for it needs at most the OLDER stores acknowledged, and one register
    // array serves both passes of a two-pass epilogue.
    template <bool NEXT = false, class Args>
    __device__ __forceinline__ static void finish(const Args& a, f32x4_t (&q)[NR], const f32x4_t (&bs)[2], const float* cs, long long row0, int n_base, int tid,
                                                  long long next_row0 = 0) {
        // In a resident kernel this code sits inside the tile loop and every offset below is tile-invariant: hipcc hoists them all -- for every
        // spec of the switch, every row, every tensor -- out of the loop and spills (hundreds of VGPRs: even the accumulators went to scratch).
        // The thread's two indices therefore pass through an opaque register here, per call.
        int c4 = (tid % TPR) * 4, p0 = tid / TPR;
        CDF_OPAQUE_V(c4);
        CDF_OPAQUE_V(p0);
        const int co = n_base + c4;
        // (bias, then the per-sample bias, are added to the accumulator as in the generic form: the same roundings in the same order)
        finish_rows<NEXT>(a, q, cs, row0, co, c4, p0, bs[0], bs[1], a.sbias != nullptr, next_row0);
    }
    template <bool NEXT, class Args>
    __device__ __forceinline__ static void finish_rows(const Args& a, f32x4_t (&q)[NR], const float* cs, long long row0, int co, int c4, int p0,
                                                       const f32x4_t& bv, const f32x4_t& sb, const bool has_sb, long long next_row0) {
        float* const ybase = S::HASY ? a.y + row0 * a.ldy : nullptr;
        void* const pbase = S::PRE ? (PRE_BF ? (void*)((unsigned short*)a.pre + row0 * a.ldp) : (void*)((float*)a.pre + row0 * a.ldp)) : nullptr;
        unsigned short* const hbase = S::YS ? a.ys_hi + row0 * a.ld_ys : nullptr;
        unsigned short* const lbase = S::YS == 2 ? a.ys_lo + row0 * a.ld_ys : nullptr;
        // One row at a time, the NEXT row's LDS read in flight while this one is worked on, and a scheduling fence per row: left alone,
        // hipcc hoists every read and address of the unrolled pass to the top and spills (265 VGPRs in the 256 x 128 resident kernel).
        f32x4_t nxt = *(const f32x4_t*)(cs + (unsigned)p0 * CP + c4);
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const unsigned p = (unsigned)(p0 + k * RPS);
            f32x4_t v = nxt;
            if (k + 1 < NR) nxt = *(const f32x4_t*)(cs + (p + RPS) * CP + c4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bv[e];
            if (has_sb) {                                    // (block-uniform)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += sb[e];
            }
            if constexpr (S::PRE == 1) cdf_st_raw4<PRE_BF>(pbase, p * (unsigned)a.ldp + (unsigned)co, v);
            if constexpr (S::ACT == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = cdf_gelu(v[e]);
            } else if constexpr (S::ACT == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = cdf_silu(v[e]);
            }
            if constexpr (S::MUL != 0) {
                const f32x4_t u = cdf_cvt_raw4<MUL_BF>(q[k]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= (S::MUL == 1 ? cdf_gelu_grad(u[e]) : (S::MUL == 2 ? cdf_silu_grad(u[e]) : u[e]));
            }
            if constexpr (S::RES) {
                const f32x4_t u = cdf_cvt_raw4<RES_BF>(q[k]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += u[e];
            }
            if constexpr (S::ACC) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += q[k][e];
            }
            if constexpr (NEXT && LOADS) q[k] = load_row(a, next_row0, co, p0, k);
            if constexpr (S::HASY) *(f32x4_t*)(ybase + p * (unsigned)a.ldy + (unsigned)co) = v;
            if constexpr (S::YS != 0) {
                uint2 h, l;
                cdf_split4(v[0], v[1], v[2], v[3], h, l);
                const unsigned o = p * (unsigned)a.ld_ys + (unsigned)co;
                *(uint2*)(hbase + o) = h;
                if constexpr (S::YS == 2) *(uint2*)(lbase + o) = l;
            }
            CDF_SCHED_FENCE();
        }
    }
};

// run f(Spec{}) for the spec with this id (a block-uniform switch; f is a generic lambda: `[&](auto spec) { using S = decltype(spec); ... }`)
template <bool BF_FAMILY, class F>
__device__ __forceinline__ void cdf_epi_dispatch(int id, F&& f) {
    if constexpr (!BF_FAMILY) {
        switch (id) {
            case 1: f(EpiPlain{}); break;
            case 2: f(EpiRes{}); break;
            case 3: f(EpiGeluPrePlanes{}); break;
            case 4: f(EpiGeluPlanes{}); break;
            case 5: f(EpiMulGeluPlanes{}); break;
            default: f(EpiAcc{}); break;
        }
    } else {
        switch (id) {
            case 7: f(EpiBfPlain{}); break;
            case 8: f(EpiBfRes{}); break;
            case 9: f(EpiBfGeluPre{}); break;
            case 10: f(EpiBfGelu{}); break;
            default: f(EpiBfMulGelu{}); break;
        }
    }
}
// ids 1..6 belong to the fp32-tensor family, 7..11 to the bf16-storage family
__device__ __forceinline__ bool cdf_epi_family_ok(int id, bool bf_family) { return bf_family ? (id >= 7 && id <= 11) : (id >= 1 && id <= 6); }

// Use: ONE block-uniform switch around the whole epilogue of a tile,
//     cdf_epi_dispatch<BFF>(a.epi, [&](auto spec) { using E = cdf_epi_fast<BN, RP, NTHR, decltype(spec)>; f32x4_t q[E::NR], bs[2];
//                                                  cdf_epi_load_bias<BN, NTHR>(...); E::load(...); <dump, barrier>; E::finish(...); });
// (load and finish behind two separate switches leave the compiler's wait-count bookkeeping with a join between them: it then waits
//  vmcnt(0) -- every store included -- at the first use of a prefetched operand.)

// ================================================================================================
// Channel-LayerNorm backward inside the data-gradient epilogue (round 6; epilogue id 12, fp32 tensors).
// conv1 of a ConvNeXt block reads LayerNorm(h): its data gradient dy = d(LN output) went to HBM (512 B per pixel at 128 channels) only to be
// read back by layernorm_c_bwd next to h.  Where ONE N tile holds every channel of its pixels (Cout == BN: 64 or 128 channels) the row of
// the staging tile IS a pixel's whole dy, so the epilogue applies the LayerNorm backward itself (deblurring_diffusion_pytorch.py:111-121):
//     xh = (h - mean) rstd,  gy = dy g,  dh = rstd (gy - mean_c(gy) - xh mean_c(gy xh)),   dg += dy xh,  db += dy  (summed over the tile)
// -- the same expressions, in the same order per pixel, as layernorm_c_bwd_kernel; the channel sums run over the BN / 4 lanes of a row
// (wave shuffles), the parameter-gradient partials leave as ln_part[tile][2][BN] for cdf_norm_param_reduce.  h, mean and rstd are
// requested before the accumulators go through LDS.  Args: ln_x / ld_lnx (h), ln_mean, ln_rstd ([M]), bias = the LayerNorm gain g (a data
// gradient has no bias of its own), y / ldy = dh, ln_part.
// ================================================================================================
#define CDF_EPI_LNBWD 12
template <int BM, int BN, int NTHR, class Args, class Dump>
__device__ __forceinline__ void cdf_epi_lnbwd(const Args& a, float* cs, int tile_m, int tid, Dump&& dump) {
    constexpr int CP = BN + 8, TPR = BN / 4, RPS = NTHR / TPR, NR = BM / RPS;
    static_assert(NR >= 1 && NR * RPS == BM && (TPR & (TPR - 1)) == 0 && TPR <= 64, "rows divide among the threads; a row is a power-of-two lane group");
    int c4 = (tid % TPR) * 4, p0 = tid / TPR;
    CDF_OPAQUE_V(c4);
    CDF_OPAQUE_V(p0);
    const long long row0 = (long long)tile_m * BM;
    const f32x4_t g = *(const f32x4_t*)(a.bias + c4);
    const float* xb = a.ln_x + row0 * a.ld_lnx;
    const float* mb = a.ln_mean + row0;
    const float* rb = a.ln_rstd + row0;
    f32x4_t x[NR];
    float mean[NR], rstd[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const unsigned r = (unsigned)(p0 + k * RPS);
        x[k] = *(const f32x4_t*)(xb + r * (unsigned)a.ld_lnx + (unsigned)c4);
        mean[k] = mb[r];
        rstd[k] = rb[r];
    }
    dump();
    CDF_LDS_BARRIER();                                       // (LDS only: the operand loads stay in flight)
    float* const ybase = a.y + row0 * a.ldy;
    const float inv_c = 1.0f / (float)BN;
    f32x4_t dg = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f};
    f32x4_t nxt = *(const f32x4_t*)(cs + (unsigned)p0 * CP + c4);
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const unsigned p = (unsigned)(p0 + k * RPS);
        const f32x4_t d = nxt;
        if (k + 1 < NR) nxt = *(const f32x4_t*)(cs + (p + RPS) * CP + c4);
        f32x4_t xh, gy, o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xh[e] = (x[k][e] - mean[k]) * rstd[k];
            dg[e] += d[e] * xh[e];
            db[e] += d[e];
            gy[e] = d[e] * g[e];
        }
        float s1 = (gy[0] + gy[1]) + (gy[2] + gy[3]);
        float s2 = (gy[0] * xh[0] + gy[1] * xh[1]) + (gy[2] * xh[2] + gy[3] * xh[3]);
        s1 = cdf_group_sum(s1, TPR) * inv_c;
        s2 = cdf_group_sum(s2, TPR) * inv_c;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rstd[k] * (gy[e] - s1 - xh[e] * s2);
        *(f32x4_t*)(ybase + p * (unsigned)a.ldy + (unsigned)c4) = o;
        CDF_SCHED_FENCE();
    }
    // parameter-gradient partials of the tile: the RPS row groups folded through LDS in a fixed order (deterministic)
    CDF_LDS_BARRIER();                                       // every thread is done with the staging rows
    float* red = cs;                                         // [RPS][2][BN]
    *(f32x4_t*)(red + ((unsigned)p0 * 2 + 0) * BN + c4) = dg;
    *(f32x4_t*)(red + ((unsigned)p0 * 2 + 1) * BN + c4) = db;
    CDF_LDS_BARRIER();
    for (int e = tid; e < 2 * BN; e += NTHR) {
        float t = 0.f;
#pragma unroll 4
        for (int r = 0; r < RPS; ++r) t += red[r * 2 * BN + e];
        a.ln_part[(long long)tile_m * 2 * BN + e] = t;
    }
}

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

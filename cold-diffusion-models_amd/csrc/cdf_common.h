// cdf_common.h — shared definitions for the colddiff gfx950 kernels.
//
// Two build modes:
//   * default  : hipcc --offload-arch=gfx950 (the product; libcolddiff_hip.so)
//   * CDF_EMU  : host clang++ build against tests/emu/hipemu.h, a fiber-based SIMT
//                simulator used ONLY by the CPU test-suite to check kernel indexing
//                (wave64 shuffles, MFMA fragment layouts, LDS tiling) without a GPU.
//                It is test infrastructure: nothing in the Python package loads it.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef CDF_EMU
#include "hipemu.h"
#else
#include <hip/hip_runtime.h>
#define CDF_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__)
#define CDF_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains every global load in flight
// (vmcnt(0)), which would serialise a register prefetch that is meant to stay in flight across the barrier.
#define CDF_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// LDS-DMA (global_load_lds_dwordx4): every lane fetches 16 bytes from its own global address; the wave's 64 pieces land
// at lds_base + 16 * lane (lds_base wave-uniform).  No VGPR staging, no ds_write (a ds_write_b128 costs 13 LDS-path
// cycles per wave -- ~80 B/clk/CU).  hipcc does not count these loads: wait with CDF_WAIT_DMA() and pass a barrier
// before reading the data.
#define CDF_GLDS16(gptr, lds_base)                                                                            \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                    \
                                     (__attribute__((address_space(3))) void*)(lds_base), 16, 0, 0)
#define CDF_WAIT_DMA() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// wait until at most N (a compile-time constant) of this wave's DMA / global loads are still outstanding
#define CDF_WAIT_DMA_LEAVE(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
// nothing is scheduled across this point (compile-time only, no instruction)
#define CDF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// s_waitcnt lgkmcnt(0) as a real instruction the compiler's wait-count bookkeeping sees (vmcnt / expcnt left at their maxima)
#define CDF_WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xC07F)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
#endif
// One-time-PER-DEVICE latch for hipFuncSetAttribute(MaxDynamicSharedMemorySize): a process may drive several devices, and a function
// attribute belongs to the device that was current when it was set.  Unsynchronised on purpose -- a race only repeats an idempotent call.
#ifndef CDF_EMU
struct CdfDeviceLatch {
    unsigned long long seen = 0;                             // bit d: set on device d (ordinals >= 64 always repeat the call)
    bool first() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
        if (seen & (1ull << d)) return false;
        seen |= 1ull << d;
        return true;
    }
};
#endif
// Order the LDS accesses of ONE wave against each other (a wave's LDS operations execute in order on the hardware: only the compiler
// must not move them; the fiber simulator really has to let the other lanes catch up).
#ifdef CDF_EMU
#define CDF_WAVE_SYNC() hipemu::wave_barrier()
#else
#define CDF_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

// 256 zero bytes: out-of-range operand elements are loaded from here, so that loads never sit behind a branch.
// (one copy per translation unit; the contents never change)
#ifdef CDF_EMU
static const float cdf_zero_page[64] = {0};
#else
static __device__ float cdf_zero_page[64];     // (not const: a constant-address-space pointer would turn the selected loads into flat loads)
#endif

// ---- bf16 split helpers ----------------------------------------------------------------------
__device__ __forceinline__ unsigned cdf_f2bf(float x) {        // round-to-nearest-even bf16 (finite inputs)
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float cdf_bf2f(unsigned h) { return __uint_as_float(h << 16); }
// two floats -> packed bf16 pair (a in the low half), round to nearest even.  gfx950 has the conversion in hardware
// (v_cvt_pk_bf16_f32: one instruction per pair instead of ~8 integer ops); the emulator build keeps the integer form.
#ifndef CDF_HWCVT
#define CDF_HWCVT 1
#endif
__device__ __forceinline__ unsigned cdf_pack2bf(float a, float b) {
#if defined(CDF_EMU) || !CDF_HWCVT
    return cdf_f2bf(a) | (cdf_f2bf(b) << 16);
#else
    typedef __bf16 bf16x2_cvt_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_cvt_t __attribute__((ext_vector_type(2)));
    const f32x2_cvt_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_cvt_t));
#endif
}
// x = hi + lo (+ O(2^-16 |x|)): the two bf16 planes the split-precision GEMMs multiply
__device__ __forceinline__ void cdf_split4(float v0, float v1, float v2, float v3, uint2& hi, uint2& lo) {
    hi.x = cdf_pack2bf(v0, v1);
    hi.y = cdf_pack2bf(v2, v3);
    lo.x = cdf_pack2bf(v0 - __uint_as_float(hi.x << 16), v1 - __uint_as_float(hi.x & 0xFFFF0000u));
    lo.y = cdf_pack2bf(v2 - __uint_as_float(hi.y << 16), v3 - __uint_as_float(hi.y & 0xFFFF0000u));
}
__device__ __forceinline__ void cdf_split_store4(unsigned short* hi, unsigned short* lo, const float* v) {
    uint2 h, l;
    cdf_split4(v[0], v[1], v[2], v[3], h, l);
    *(uint2*)hi = h;
    if (lo) *(uint2*)lo = l;          // lo == nullptr: hi-only planes (single-pass bf16 operands)
}

// ---- activation storage type ---------------------------------------------------------------------
// "bf16" arithmetic mode with bf16 ACTIVATION STORAGE (BASELINE configs 3 / 5): feature maps between kernels are ONE bf16 plane
// [pixels][ld] (the GEMMs' "hi" plane IS the tensor); accumulation, statistics and parameters stay fp32.  The HBM-bound kernels are
// instantiated for both storage types: a 4-channel quad is a float4 (16 B) or four bf16 (8 B) in memory and a float4 in registers.
// Offsets are in ELEMENTS of the tensor's own type.
template <bool BF> struct cdf_quad { typedef float4 raw; typedef float elem; };
template <> struct cdf_quad<true> { typedef uint2 raw; typedef unsigned short elem; };
__device__ __forceinline__ float cdf_widen(float v) { return v; }                                        // one element of either type as a float
__device__ __forceinline__ float cdf_widen(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ float4 cdf_quad_cvt(const float4& r) { return r; }
__device__ __forceinline__ float4 cdf_quad_cvt(const uint2& r) {
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xFFFF0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xFFFF0000u));
}
// (the offset keeps its own integer type: a scalar base + 32-bit unsigned offset stays a 32-bit address computation)
template <bool BF, class Off>
__device__ __forceinline__ typename cdf_quad<BF>::raw cdf_quad_ld(const void* base, Off off) {           // the raw load (convert where the value is used)
    if constexpr (BF) return *(const uint2*)((const unsigned short*)base + off);
    else return *(const float4*)((const float*)base + off);
}
template <bool BF, class Off>
__device__ __forceinline__ void cdf_quad_st(void* base, Off off, const float4& v) {                      // round to nearest even
    if constexpr (BF) {
        uint2 h;
        h.x = cdf_pack2bf(v.x, v.y);
        h.y = cdf_pack2bf(v.z, v.w);
        *(uint2*)((unsigned short*)base + off) = h;
    } else {
        *(float4*)((float*)base + off) = v;
    }
}
// ---- range-checked loads (buffer instructions) ------------------------------------------------------
// A raw buffer resource over [base, base + bytes), base wave-uniform: a load whose byte offset falls outside returns zeros WITHOUT touching
// memory.  An out-of-image element of a stencil then costs one select on its offset (CDF_BUF_OOB) instead of clamps on both coordinates, a
// 64-bit address per lane and a select per loaded component.  The simulator build checks the range by hand.
#define CDF_BUF_OOB 0xFFFFFFFFu
#ifdef CDF_EMU
struct cdf_buf { const char* base; unsigned bytes; };
static inline cdf_buf cdf_make_buf(const void* p, unsigned bytes) { return cdf_buf{(const char*)p, bytes}; }
static inline float4 cdf_buf_ld(const cdf_buf& b, unsigned off, const float4*) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned long long)off + 16 <= b.bytes) v = *(const float4*)(b.base + off);
    return v;
}
static inline uint2 cdf_buf_ld(const cdf_buf& b, unsigned off, const uint2*) {
    uint2 v = make_uint2(0u, 0u);
    if ((unsigned long long)off + 8 <= b.bytes) v = *(const uint2*)(b.base + off);
    return v;
}
static inline void cdf_buf_st(const cdf_buf& b, unsigned off, const float4& v) {
    if ((unsigned long long)off + 16 <= b.bytes) *(float4*)(b.base + off) = v;
}
static inline void cdf_buf_st(const cdf_buf& b, unsigned off, const uint2& v) {
    if ((unsigned long long)off + 8 <= b.bytes) *(uint2*)(b.base + off) = v;
}
#else
typedef __amdgpu_buffer_rsrc_t cdf_buf;
__device__ __forceinline__ cdf_buf cdf_make_buf(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);     // (dword 3: raw buffer, 32-bit data format)
}
__device__ __forceinline__ float4 cdf_buf_ld(const cdf_buf& b, unsigned off, const float4*) {
    typedef int i4_t __attribute__((ext_vector_type(4)));
    const i4_t v = __builtin_amdgcn_raw_buffer_load_b128(b, (int)off, 0, 0);
    return make_float4(__int_as_float(v[0]), __int_as_float(v[1]), __int_as_float(v[2]), __int_as_float(v[3]));
}
__device__ __forceinline__ uint2 cdf_buf_ld(const cdf_buf& b, unsigned off, const uint2*) {
    typedef int i2_t __attribute__((ext_vector_type(2)));
    const i2_t v = __builtin_amdgcn_raw_buffer_load_b64(b, (int)off, 0, 0);
    return make_uint2((unsigned)v[0], (unsigned)v[1]);
}
__device__ __forceinline__ void cdf_buf_st(const cdf_buf& b, unsigned off, const float4& v) {        // (out of range: dropped)
    typedef int i4_t __attribute__((ext_vector_type(4)));
    const i4_t t = {__float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(t, b, (int)off, 0, 0);
}
__device__ __forceinline__ void cdf_buf_st(const cdf_buf& b, unsigned off, const uint2& v) {
    typedef int i2_t __attribute__((ext_vector_type(2)));
    const i2_t t = {(int)v.x, (int)v.y};
    __builtin_amdgcn_raw_buffer_store_b64(t, b, (int)off, 0, 0);
}
#endif
// a float4 in the tensor's storage type (bf16: rounded to nearest even), for the buffer stores
__device__ __forceinline__ float4 cdf_quad_raw(const float4& v, const float4*) { return v; }
__device__ __forceinline__ uint2 cdf_quad_raw(const float4& v, const uint2*) { return make_uint2(cdf_pack2bf(v.x, v.y), cdf_pack2bf(v.z, v.w)); }
// runtime-selected forms for the GEMM epilogues (the flag is a kernel argument: a scalar branch)
__device__ __forceinline__ void cdf_ld4_bf(float* v, const void* base, long long off) {
    const float4 t = cdf_quad_cvt(*(const uint2*)((const unsigned short*)base + off));
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void cdf_st4_bf(void* base, long long off, const float* v) {
    uint2 h;
    h.x = cdf_pack2bf(v[0], v[1]);
    h.y = cdf_pack2bf(v[2], v[3]);
    *(uint2*)((unsigned short*)base + off) = h;
}
#define CDF_IO_RES_BF16 1    /* epilogue operand types (io_bf16 bit mask of the *_io GEMM entry points): residual read as bf16 */
#define CDF_IO_PRE_BF16 2    /* pre-activation written as bf16 */
#define CDF_IO_MUL_BF16 4    /* activation-gradient source read as bf16 */
#define CDF_IO_PRE_GRAD 8    /* `pre` receives act'(v) instead of v (act = GELU / SiLU): the backward pass then multiplies by the stored
                                value (mul_mode 3) instead of evaluating erf / exp again in the data-gradient epilogue */

// ---- status codes (returned by every extern "C" entry point) -------------------------
#define CDF_OK 0
#define CDF_E_INVALID (-1)      // bad argument (shape, alignment, null pointer)
#define CDF_E_UNSUPPORTED (-2)  // valid request this build has no kernel for
#define CDF_E_LAUNCH (-3)       // hipLaunch / runtime error

#define CDF_WAVE 64
// C-ABI entry points take the stream as an opaque `void* stream`
#define CDF_S ((hipStream_t)stream)

void cdf_set_error(const char* fmt, ...);
int cdf_check_launch(const char* what);

#define CDF_REQUIRE(cond, ...)                  \
    do {                                        \
        if (!(cond)) {                          \
            cdf_set_error(__VA_ARGS__);         \
            return CDF_E_INVALID;               \
        }                                       \
    } while (0)

static inline int cdf_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// XCD-aware work order.  Workgroups are handed to the 8 XCDs (each with its own 4 MB L2) round-robin in dispatch order, so
// work items that are neighbours in memory -- adjacent image tiles with a shared halo, the taps of one pixel range -- land on 8
// different L2s and each fetches the shared bytes from HBM again.  Re-numbered so that every XCD walks a CONTIGUOUS range of
// work ids: bid -> the bid-th item of XCD (bid & 7)'s range.
__device__ __forceinline__ int cdf_xcd_order(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// ---- device helpers ---------------------------------------------------------------------
__device__ __forceinline__ float cdf_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float cdf_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// sum over aligned sub-groups of `width` lanes (width power of two <= 64)
__device__ __forceinline__ float cdf_group_sum(float v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// counter-based dropout mask (Model2.py:94,124 nn.Dropout; the mask stream is this hash, not torch's Philox): element idx of a
// [rows][C] map (idx = row * C + c) is kept iff cdf_hash32(seed, idx) >= p * 2^32; kept values are scaled by 1 / (1 - p)
__device__ __forceinline__ unsigned cdf_hash32(unsigned long long seed, unsigned long long idx) {
    unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (unsigned)(z >> 32);
}

// exact (erf) GELU, matches torch.nn.GELU(approximate='none')
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute), branch-free, ~14 instructions; also hands back
// exp(-z^2), which the GELU derivative needs anyway.  libm's erff is two divergent paths of ~35 instructions each:
// 64 of them per thread made the GELU epilogue a third of the 128x128-pixel conv kernels.  GELU itself is evaluated
// as 0.5 x (1 + erf(x / sqrt 2)) like the reference (DEBLUR:133-135 -> F.gelu); the absolute error is <= 0.75e-7 |x|.
__device__ __forceinline__ float cdf_erf_fast(float z, float& ez2) {
    const float az = fabsf(z);
#ifdef CDF_EMU
    const float t = 1.0f / fmaf(0.3275911f, az, 1.0f);
    ez2 = expf(-az * az);
#else
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));     // 1 ulp; the formula itself is 1.5e-7
    ez2 = __expf(-az * az);                                                 // v_exp_f32 path, ~2 ulp
#endif
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = fmaf(-p * t, ez2, 1.0f);
    return copysignf(r, z);
}
__device__ __forceinline__ float cdf_gelu(float x) {
    float e;
    return 0.5f * x * (1.0f + cdf_erf_fast(x * 0.70710678118654752440f, e));
}
__device__ __forceinline__ float cdf_gelu_grad(float x) {
    float e;                                                  // exp(-x^2 / 2)
    const float cdf = 0.5f * (1.0f + cdf_erf_fast(x * 0.70710678118654752440f, e));
    return cdf + x * (0.39894228040143267794f * e);
}
// GELU(x) and GELU'(x) from ONE erf / exp evaluation (the same expressions as cdf_gelu and cdf_gelu_grad: bit-identical values)
__device__ __forceinline__ float cdf_gelu_both(float x, float& grad) {
    float e;
    const float er = cdf_erf_fast(x * 0.70710678118654752440f, e);
    const float cdf = 0.5f * (1.0f + er);
    grad = cdf + x * (0.39894228040143267794f * e);
    return 0.5f * x * (1.0f + er);
}
__device__ __forceinline__ float cdf_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float cdf_silu(float x) { return x * cdf_sigmoid(x); }
__device__ __forceinline__ float cdf_silu_grad(float x) {
    const float s = cdf_sigmoid(x);
    return s * (1.0f + x * (1.0f - s));
}

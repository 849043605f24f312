// hipemu.h — TEST INFRASTRUCTURE ONLY.
//
// A minimal single-process SIMT simulator: every HIP thread of a workgroup is a ucontext
// fiber, workgroups run one after another, __syncthreads()/wave shuffles/MFMA are
// rendezvous points between fibers.  It lets the CPU test-suite execute the *same kernel
// source* that hipcc compiles for gfx950 (64-wide wavefronts, MFMA fragment layouts as
// documented for CDNA4) on tiny shapes, so indexing bugs are found without a GPU.
// It is never built into, or loaded by, the colddiff package.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef short bf16x8_t __attribute__((ext_vector_type(8)));

namespace hipemu {

// x86-64 stack switch (callee-saved registers only; no signal-mask syscalls like swapcontext)
extern "C" void hipemu_switch(void** save_sp, void* load_sp);

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = false;
    dim3 tid;
};

struct State {
    std::vector<Fiber> fibers;
    void* sched_sp = nullptr;
    int cur = 0;
    int nthreads = 0;
    std::function<void()> body;
    // block barrier
    int bar_count = 0;
    long bar_gen = 0;
    // per-wave rendezvous
    int wave_count[64];
    long wave_gen[64];
    float wave_a[64][64 * 8];
    float wave_b[64][64 * 8];
    unsigned long long wave_u[64][64];
    unsigned char* dyn_smem = nullptr;
    size_t dyn_cap = 0;
    long progress = 0;
};
extern State g;
extern dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

static const size_t kStack = 256 * 1024;

inline void yield() { hipemu_switch(&g.fibers[g.cur].sp, g.sched_sp); }

inline int lin_tid() {
    return (int)(g_threadIdx.x + g_blockDim.x * (g_threadIdx.y + g_blockDim.y * g_threadIdx.z));
}
inline int lane_id() { return lin_tid() & 63; }
inline int wave_id() { return lin_tid() >> 6; }
inline int wave_width(int w) {
    int rem = g.nthreads - w * 64;
    return rem < 64 ? rem : 64;
}

inline void block_barrier() {
    long my = g.bar_gen;
    g.progress++;
    if (++g.bar_count == g.nthreads) {
        g.bar_count = 0;
        g.bar_gen++;
    } else {
        while (g.bar_gen == my) yield();
    }
}
inline void wave_barrier() {
    int w = wave_id();
    long my = g.wave_gen[w];
    g.progress++;
    if (++g.wave_count[w] == wave_width(w)) {
        g.wave_count[w] = 0;
        g.wave_gen[w]++;
    } else {
        while (g.wave_gen[w] == my) yield();
    }
}

void trampoline();
void run_grid(dim3 grid, dim3 block, size_t shmem);

template <typename... KArgs, typename... Args>
void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
    g.body = [=]() { kernel(args...); };
    run_grid(grid, block, shmem);
}

template <typename T>
inline T shfl_from(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "shfl type too wide");
    int w = wave_id(), l = lane_id();
    unsigned long long raw = 0;
    memcpy(&raw, &v, sizeof(T));
    g.wave_u[w][l] = raw;
    wave_barrier();
    int ww = wave_width(w);
    if (src_lane < 0 || src_lane >= ww) src_lane = l;
    unsigned long long got = g.wave_u[w][src_lane];
    wave_barrier();
    T r;
    memcpy(&r, &got, sizeof(T));
    return r;
}

// gfx950 ds_read_b64_tr_b16: within each 16-lane group the lanes hand in the addresses of a [4 rows][16 cols]
// block of 16-bit elements (lane t: row t >> 2, cols 4 (t & 3) .. +3) and lane t receives column t's 4 rows.
// (semantics pinned on hardware with a probe kernel: see DESIGN.md, wgrad section)
typedef short emu_s4 __attribute__((ext_vector_type(4)));
inline emu_s4 ds_read_tr16_b64(const unsigned short* p) {
    int w = wave_id(), l = lane_id();
    g.wave_u[w][l] = (unsigned long long)(uintptr_t)p;
    wave_barrier();
    int grp = l & ~15, t = l & 15;
    emu_s4 r;
    for (int j = 0; j < 4; ++j) {
        const unsigned short* q = (const unsigned short*)(uintptr_t)g.wave_u[w][grp + j * 4 + (t >> 2)];
        r[j] = (short)q[t & 3];
    }
    wave_barrier();
    return r;
}

}  // namespace hipemu

#define threadIdx hipemu::g_threadIdx
#define blockIdx hipemu::g_blockIdx
#define blockDim hipemu::g_blockDim
#define gridDim hipemu::g_gridDim

#define CDF_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch(kernel, dim3(grid), dim3(block), (size_t)(shmem), stream, __VA_ARGS__)
#define CDF_DYN_SMEM(name) unsigned char* name = hipemu::g.dyn_smem
#define CDF_LDS_BARRIER() hipemu::block_barrier()
#define CDF_GLDS16(gptr, lds_base) memcpy((unsigned char*)(lds_base) + 16 * hipemu::lane_id(), (const void*)(gptr), 16)
#define CDF_WAIT_DMA() ((void)0)
#define CDF_WAIT_DMA_LEAVE(N) ((void)0)
#define CDF_SCHED_FENCE() ((void)0)
#define CDF_WAIT_LDS() ((void)0)

static inline void __syncthreads() { hipemu::block_barrier(); }

template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    (void)width;
    return hipemu::shfl_from(v, hipemu::lane_id() ^ mask);
}
template <typename T>
static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int l = hipemu::lane_id();
    int src = l + (int)d;
    if ((l & ~(width - 1)) != (src & ~(width - 1))) src = l;
    return hipemu::shfl_from(v, src);
}
template <typename T>
static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int l = hipemu::lane_id();
    int src = l - (int)d;
    if (src < 0 || (l & ~(width - 1)) != (src & ~(width - 1))) src = l;
    return hipemu::shfl_from(v, src);
}
template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
    int l = hipemu::lane_id();
    return hipemu::shfl_from(v, (l & ~(width - 1)) + (src & (width - 1)));
}
static inline int __builtin_amdgcn_readfirstlane(int v) { return hipemu::shfl_from(v, 0); }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }    // v_mul_u32_u24: low 32 bits of a 24 x 24-bit product

// ---- MFMA (fragment layouts per CDNA4 ISA; see cdna_hip_programming.md §3) ------------------
// 32x32x2 f32: A lane l -> A[i=l&31][k=l>>5], B lane l -> B[k=l>>5][j=l&31],
// C/D reg r, lane l -> row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31.
static inline f32x16_t emu_mfma_f32_32x32x2f32(float a, float b, f32x16_t c, int, int, int) {
    int w = hipemu::wave_id(), l = hipemu::lane_id();
    hipemu::g.wave_a[w][l] = a;
    hipemu::g.wave_b[w][l] = b;
    hipemu::wave_barrier();
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(hipemu::g.wave_a[w][k * 32 + row], hipemu::g.wave_b[w][k * 32 + col], acc);
        c[r] = acc;
    }
    hipemu::wave_barrier();
    return c;
}
// 16x16x4 f32: A lane l -> A[i=l&15][k=l>>4], B lane l -> B[k=l>>4][j=l&15],
// C/D reg r -> row=(l>>4)*4+r, col=l&15.
static inline f32x4_t emu_mfma_f32_16x16x4f32(float a, float b, f32x4_t c, int, int, int) {
    int w = hipemu::wave_id(), l = hipemu::lane_id();
    hipemu::g.wave_a[w][l] = a;
    hipemu::g.wave_b[w][l] = b;
    hipemu::wave_barrier();
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(hipemu::g.wave_a[w][k * 16 + row], hipemu::g.wave_b[w][k * 16 + col], acc);
        c[r] = acc;
    }
    hipemu::wave_barrier();
    return c;
}
static inline float emu_bf16_to_f32(short s) {
    unsigned u = ((unsigned)(unsigned short)s) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// 32x32x16 bf16: A lane l holds A[i=l&31][k=8*(l>>5)+j], j=0..7; B likewise B[k][j=l&31].
static inline f32x16_t emu_mfma_f32_32x32x16_bf16(bf16x8_t a, bf16x8_t b, f32x16_t c, int, int, int) {
    int w = hipemu::wave_id(), l = hipemu::lane_id();
    for (int j = 0; j < 8; ++j) {
        hipemu::g.wave_a[w][l * 8 + j] = emu_bf16_to_f32(a[j]);
        hipemu::g.wave_b[w][l * 8 + j] = emu_bf16_to_f32(b[j]);
    }
    hipemu::wave_barrier();
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c[r];
        for (int k = 0; k < 16; ++k) {
            int src_hi = k >> 3, j = k & 7;
            acc += hipemu::g.wave_a[w][(src_hi * 32 + row) * 8 + j] * hipemu::g.wave_b[w][(src_hi * 32 + col) * 8 + j];
        }
        c[r] = acc;
    }
    hipemu::wave_barrier();
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32 emu_mfma_f32_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emu_mfma_f32_16x16x4f32
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 emu_mfma_f32_32x32x16_bf16

// ---- atomics / math / runtime shims -----------------------------------------------------------
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

static inline float __fdividef(float a, float b) { return a / b; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float2int_rz(float f) { return (int)f; }

static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memmove(d, s, n); return hipSuccess; }
#define hipMemcpyDeviceToDevice 3

/* colddiff.h — C ABI of libcolddiff_hip.so, the MI355X (gfx950) kernel library behind the
 * cold-diffusion training + sampling hot path.
 *
 * The reference (arpitbansal297/Cold-Diffusion-Models) has no FFI: its boundary is the Python
 * class API (Unet / Model / GaussianDiffusion / Trainer).  The functions below are what the
 * host-side mirror of that API (cold-diffusion-models_amd/colddiff) binds through ctypes; every
 * group cites the reference code (file:line under /root/reference) whose ATen op chain it
 * replaces.  See INTEGRATION.md for the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C: pointers are device pointers (hipMalloc'ed / torch CUDA storage), sizes are ints.
 *   - `void* stream` is a hipStream_t; work is enqueued asynchronously on it.
 *   - return 0 on success, <0 on error (CDF_E_*); cdf_last_error() gives the thread-local text.
 *     No function throws, aborts, allocates or frees device memory, or calls hipSetDevice.
 *   - images  : NCHW fp32 [B,C,H,W] (the public GaussianDiffusion tensors)
 *   - features: NHWC fp32 with an explicit pixel pitch `ld*` (elements between consecutive
 *               pixels), so a tensor may be a channel slice of a wider concat buffer.
 *   - packed conv weights: [tap][Cin][Cout] fp32 ("KN" layout, produced by cdf_pack_weight).
 */
#ifndef COLDDIFF_H
#define COLDDIFF_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an exported signature changes incompatibly (rounds 1-3 all answered 1 while arguments were added: `tiled`,
 * `onepass`, `slots`, `tune`).  cdf_abi_version() returns the value the LIBRARY was built with; a binding compares it with the
 * header it was generated from before the first call (colddiff/_lib.py does) -- a mismatched pair would read shifted arguments. */
#define CDF_ABI_VERSION 6

#define CDF_E_INVALID (-1)
#define CDF_E_UNSUPPORTED (-2)
#define CDF_E_LAUNCH (-3)

/* ---- library identity / errors ------------------------------------------------------------ */
const char* cdf_last_error(void);
int cdf_abi_version(void);
int cdf_is_device_build(void);

/* ---- degradation operators D(x,t) ------------------------------------------------------------
 * Blur: replaces the nn.Conv2d(groups=C, padding_mode=circular|reflect) stack of
 * deblurring_diffusion_pytorch.py:351-361 applied sequentially in q_sample (:927-960) and in the
 * Algorithm-2 sampler (:436-451).  taps = the stacked `gaussian_kernels.{i}.weight` tensors
 * [nsteps][C][k][k].  Applies steps step_lo..hi(b) where hi(b) = t[b] (t != NULL) or step_hi.
 *   y    : state after step hi(b); if img != NULL instead y = img - D_hi + D_{hi-1}  (Alg. 2)
 *   snap : optional, state after step hi(b)-1
 *   collapse_step : -1, or the step after which the plane is replaced by its mean (discrete)
 *   quantise      : 8-bit truncation of y (discrete q_sample, :954-958)
 * The whole chain runs with the plane resident in LDS (one HBM read + one write). */
size_t cdf_blur_lds_bytes(int H, int W, int k);
int cdf_blur_chain(const float* x, float* y, float* snap, const float* img, const float* taps, const int64_t* t,
                   int B, int C, int H, int W, int k, int step_lo, int step_hi, int pad_mode, int collapse_step,
                   int quantise, void* stream);
/* one blur step from global memory (any plane size, per-step kernel size); x != y */
/* Separable form of cdf_blur_chain for rank-one (Gaussian g (x) g) kernels: taps1d = [nsteps][C][2][k], per step and
 * channel the 1-D factor along y followed by the factor along x.  2k instead of k*k FMAs per pixel and step; same
 * arguments and semantics otherwise (results equal cdf_blur_chain up to fp32 rounding of the taps' outer product).
 * In the Alg. 2 form (img != NULL) y must not alias x or img. */
size_t cdf_blur_sep_lds_bytes(int H, int W);
int cdf_blur_chain_sep(const float* x, float* y, float* snap, const float* img, const float* taps1d, const int64_t* t,
                       int B, int C, int H, int W, int k, int step_lo, int step_hi, int pad_mode, int collapse_step,
                       int quantise, void* stream);
int cdf_blur_step(const float* x, float* y, const float* taps, int B, int C, int H, int W, int k, int pad_mode,
                  void* stream);
int cdf_plane_mean(float* x, int planes, int HW, void* stream);

/* Gaussian-mask fade (defading_diffusion_gaussian.py:496-535, :405-420): x <- masks[i] * x for
 * i = step_lo..hi(b), sequential products; optional per-sample crop offsets (Random_* routines). */
int cdf_mask_chain(const float* x, float* y, float* snap, const float* img, const float* masks, const int64_t* t,
                   const int64_t* off_y, const int64_t* off_x, int B, int C, int H, int W, int MH, int MW,
                   int step_lo, int step_hi, int quantise, void* stream);

/* Pixelation (resolution_diffusion_pytorch.py:354-385): F.interpolate(size=sizes[i], mode) then
 * F.interpolate(size=H, 'nearest-exact'), compositionally for i = step_lo..hi(b).
 * mode 0 area, 1 bilinear, 2 bicubic; sizes is a device int array. */
int cdf_pixelate_chain(const float* x, float* y, float* snap, const float* img, const int* sizes, const int64_t* t,
                       int B, int C, int H, int step_lo, int step_hi, int mode, void* stream);

/* Algorithm-2 combine  out = img - d_t + d_tm1  (deblurring_diffusion_pytorch.py:451) */
int cdf_x0_step_down(const float* img, const float* d_t, const float* d_tm1, float* out, long long n, void* stream);

/* Gaussian-noise forward process and reverse step (denoising_diffusion_pytorch.py:517-522,
 * :342-375, :383-434); ca/cb = sqrt_alphas_cumprod / sqrt_one_minus_alphas_cumprod tables. */
int cdf_noise_qsample(const float* x0, const float* eps, const float* ca, const float* cb, const int64_t* t,
                      float* out, int B, long long per_sample, void* stream);
int cdf_noise_step(const float* img, const float* x1, const float* noise, const float* ca, const float* cb, int t,
                   int est_noise, float* out, long long n, void* stream);

/* Per-pixel blend of two images by mask tables alphas / one_minus [T][H*W] ("defading generation":
 * defading-generation-diffusion-pytorch/defading_diffusion_pytorch/defading_diffusion_pytorch.py:543-548 q_sample;
 * :386-419 / :428-457 the reverse step with the second image held fixed).  NCHW images, t per sample / per call. */
int cdf_blend_qsample(const float* x1, const float* x2, const float* alphas, const float* one_minus, const int64_t* t, float* out,
                      int B, int C, long long HW, void* stream);
int cdf_blend_step(const float* img, const float* x1, const float* x2, const float* alphas, const float* one_minus, int t, float* out,
                   long long HW, long long n, void* stream);

/* L1 / L2 training loss (deblurring_diffusion_pytorch.py:966-971): out[0] = mean|x-y| or
 * mean (x-y)^2; backward writes d loss / d y scaled by gout[0]. partial: >= 1024 floats. */
int cdf_loss_fwd(const float* x, const float* y, float* out, float* partial, long long n, int l2, void* stream);
int cdf_loss_bwd(const float* x, const float* y, const float* gout, float* gy, long long n, int l2, void* stream);

/* NCHW image <-> NHWC feature map (pitch ld); `add` (NCHW, nullable) is the Unet residual=True */
int cdf_nchw_to_nhwc(const float* x, float* y, int B, int C, int HW, int ldy, void* stream);
int cdf_nhwc_to_nchw(const float* x, float* y, const float* add, int B, int C, int HW, int ldx, void* stream);

/* ---- dense convolutions / linear layers as MFMA implicit GEMM ---------------------------------
 * Replaces nn.Conv2d (3x3, 1x1, 4x4 stride 2), nn.ConvTranspose2d (4x4 stride 2) and nn.Linear
 * forward and backward: deblurring_diffusion_pytorch.py:105-109,140-154,173-174,211-216,253 and
 * Model2.py:36-73,85-112,148-163.
 *
 * cdf_conv_gemm:  Y[m,co] = epi( sum_{tap,ci} X[pix(m,tap),ci] * Wp[wi(tap)][ci][co] )
 *   m = (b,qy,qx) over a per-phase QHxQW grid; input pixel (qy*is+dy, qx*is+dx), zero outside HxW;
 *   output pixel (qy*os+oy, qx*os+ox) of an OHxOW map.  phase_desc = per phase
 *   [oy, ox, ntaps, (dy, dx, wi) x ntaps] (ints, host memory).  Wp is [tap][Cin][ldw] (cdf_pack_weight)
 *   or, with b_trans, a plain [Cout][ldw>=Cin] matrix (one tap).  Epilogue, in order:
 *   v = acc + bias[co] + sbias[b][co]; pre = v; v = act(v) (1 GELU, 2 SiLU, 3 ReLU);
 *   v *= {1: gelu'(mul), 2: silu'(mul), 3: mul}; v += res; accumulate ? y += v : y = v.
 *   batch * batch2 independent GEMMs run in one launch (blockIdx.z = outer*batch2 + inner) with element
 *   strides x_bs / w_bs / y_bs (outer) and x_bs2 / w_bs2 / y_bs2 (inner, e.g. attention heads). */
int cdf_conv_gemm(const float* x, int ldx, const float* w, int ldw, float* y, int ldy, int B, int H, int W, int Cin,
                  int OH, int OW, int Cout, int QH, int QW, int os, int is, int nphase, const int* phase_desc,
                  const float* bias, const float* sbias, int ld_sbias, const float* res, int ldr, float* pre, int ldp,
                  const float* mul, int ldm, int act, int mul_mode, int accumulate, int b_trans, int batch,
                  long long x_bs, long long w_bs, long long y_bs, int batch2, long long x_bs2, long long w_bs2,
                  long long y_bs2, void* stream);

/* cdf_conv_wgrad: ws[z][tap][ca][cb] = sum_{m in split z} XA[pixA(m,tap)][ca] * XB[pixB(m,tap)][cb]
 *   m = (b,qy,qx); pixA = (qy*sa+day, qx*sa+dax) in HAxWA, pixB likewise; tap_desc = (day,dax,dby,dbx) x ntaps.
 *   ws holds nsplit (x batch) slabs of [ntaps][CA][ldo] -- [batch (stride o_bs)][split][tap], or with o_bs < 0 [split][batch][tap], so that
 *   one cdf_unpack_reduce(T = batch * ntaps) reduces the splits of every batch entry; cdf_unpack_reduce sums the slabs into the
 *   parameter-gradient tensor in its PyTorch layout.  bsum (nullable, [nsplit*batch][ldo]) receives the
 *   per-split column sums of XB's rows as they stream through (the bias gradient when XB = dY with
 *   sb = 1 and a zero tap offset, so that every row is visited exactly once); reduce it with
 *   cdf_unpack_reduce(T=1, R=1). */
int cdf_wgrad_nsplit(int M, int CA, int CB, int ntaps);
int cdf_conv_wgrad(const float* xa, int lda, const float* xb, int ldb, float* ws, int ldo, int B, int QH, int QW,
                   int HA, int WA, int sa, int HB, int WB, int sb, int CA, int CB, int ntaps, const int* tap_desc,
                   int nsplit, int batch, long long a_bs, long long b_bs, long long o_bs, float* bsum, void* stream);

/* ---- split-precision bf16 MFMA path of the same gather-GEMM (csrc/k_conv_sp.hip) ---------------------
 * Same geometry / epilogue contract as cdf_conv_gemm (no batch / b_trans).  Operands are fp32 in
 * HBM; x is split on the fly into bf16 hi + lo, the weights arrive pre-split from
 * cdf_pack_weight_bf16 as bf16 [tap][Cout][ldk] planes (K contiguous, ldk = Cin rounded up to 32):
 *   split = 3 : a*b ~= ah*bh + ah*bl + al*bh   (fp32-grade parity, 5.3x the fp32-MFMA rate)
 *   split = 1 : plain bf16 operands (w_lo may be NULL), fp32 accumulate. */
int cdf_pack_weight_bf16(const float* src, void* dst_hi, void* dst_lo, int T, int R, int C, int ldc, long long s_t,
                         long long s_r, long long s_c, void* stream);
int cdf_conv_gemm_bf16(const float* x, int ldx, const void* w_hi, const void* w_lo, int ldk, float* y, int ldy, int B, int H,
                       int W, int Cin, int OH, int OW, int Cout, int QH, int QW, int os, int is, int nphase,
                       const int* phase_desc, const float* bias, const float* sbias, int ld_sbias, const float* res, int ldr,
                       float* pre, int ldp, const float* mul, int ldm, int act, int mul_mode, int accumulate, int split,
                       void* stream);
/* Weight gradient on the same split-precision bf16 MFMA path (contract of cdf_conv_wgrad, no batch):
 * both operands are fp32 NHWC activations, split into bf16 hi/lo while staged in LDS. */
int cdf_conv_wgrad_bf16(const float* xa, int lda, const float* xb, int ldb, float* ws, int ldo, int B, int QH, int QW, int HA,
                        int WA, int sa, int HB, int WB, int sb, int CA, int CB, int ntaps, const int* tap_desc, int nsplit,
                        float* bsum, void* stream);
/* RE-ENTRANCY.  Every entry point is a pure function of its arguments: the library keeps NO mutable process-wide state that changes results (the only
 * statics are caches keyed by the CURRENT DEVICE ordinal: one-time hipFuncSetAttribute(MaxDynamicSharedMemorySize) latches and the CU
 * count the resident kernels size their grids by -- idempotent, a process may drive several devices).  What used to be process-wide tuning
 * setters is an explicit, OPTIONAL argument of the pre-split GEMM entry points: `const cdf_gemm_tuning* tune`, NULL = the defaults
 * below.  The fields only choose between kernels / tile shapes that compute the same sums (fp32 summation order aside), so two
 * models in one process -- or forward and backward threads -- can use different settings.  Fill a struct with
 * cdf_gemm_tuning_default() and change fields; `size` must stay sizeof(cdf_gemm_tuning) (checked: CDF_E_INVALID otherwise). */
typedef struct cdf_gemm_tuning {
    int size;            /* sizeof(cdf_gemm_tuning) */
    int tile_bm;         /* 0: automatic; 64 / 128 / 256: force the row tile of the generic gather-GEMM (256 only with tile_bn 128) */
    int tile_bn;         /* 0: automatic; 64 / 128: force its column tile */
    int max_bm;          /* 0: none; 128: the automatic choice never takes the 256 x 128 tile */
    int dephase;         /* 1: waves 4..7 of the 8-wave tiles multiply the previous K step's fragments first, read afterwards */
    int deep;            /* 1: grids of <= 256 64-row tiles run with six DMA stages, one block per CU */
    int splitk;          /* 1: ... and share the taps out over block groups when a workspace is given (the 4 x 4 / 8 x 8 levels of the 32 x 32
                            configurations: MNIST config +8..16 %, CIFAR-10 config +2.5 %; nothing at 128 x 128); 0: never */
    int halo;            /* 47: bit mask of the LDS-resident-input kernel over the image width 16 (1), 32 (2), 64 (4), 128 (8); 16 = at
                            width 128 also for > 64 output channels; 32 = row-halo form (256-pixel tiles, input shared by the dx taps of
                            a row) for the > 64-channel outputs at width 128; 64 = row-halo form wherever it applies; 0 = never */
    int halo_min_tiles;  /* 1: smallest tile count (128 pixels x BN) the LDS-resident form is used for */
    int halo_bm;         /* 0: automatic (256 pixels where every CU still gets a tile); 128 / 256 */
    int small_n64;       /* 1: 64-wide N tiles in the LDS-resident form when 128-wide ones would give < ~2/3 of the CUs a block */
    int wgrad_stack;     /* 1: two taps per 128-row tile in cdf_conv_wgrad_bf16x when CA <= 64 < CB */
    int wgrad_swizzle;   /* 1: XCD-aware block order of cdf_conv_wgrad_bf16x (the taps of a pixel range share one XCD's L2) */
    int wgrad_row3;      /* 1: weight gradients of 3 x 3 stride-1 same-size convolutions by one block per ROW of taps */
    int rowhalo_stream;  /* 1: the row-halo GEMM with 64 / 128 input channels runs as resident blocks with one operand stream over all the
                            tiles of a CU (the next tile's first rows and weights arrive under the current tile's epilogue); 0: those layers take the LDS-resident-input
                            / generic kernels (the one-tile row-halo kernel of rounds 2-3 is gone) */
    int resident_reserve;/* 0: CUs the resident kernels leave free (rounded up to whole rounds of the 8 XCDs).  Multi-rank training sets it: the
                            collective kernels of the gradient exchange run concurrently with backward and need CUs of their own -- a resident
                            block that finds its CU taken would run its fixed share of the tiles after everybody else */
    int epilogue;        /* 1: the template-specialised straight-line epilogues where one matches the call (csrc/cdf_epilogue.h: operand loads
                            issued before any store of the tile); 0: always the generic run-time-selected form.  Same arithmetic in the same
                            order: bit-identical results (tests/test_kernels.py) -- the switch exists for that test and for A/B timing */
} cdf_gemm_tuning;
int cdf_gemm_tuning_default(cdf_gemm_tuning* t);

/* Pre-split operand variants: an activation that feeds several GEMMs (forward, data gradient, weight
 * gradient, every N tile) is split ONCE into bf16 hi / lo planes [rows][ld] by cdf_split_bf16; the GEMMs
 * then only copy and multiply.  Every lo pointer is OPTIONAL: with x_lo == w_lo == NULL (a_lo == b_lo == NULL for the weight
 * gradient) the operands are single bf16 values and each product is ONE MFMA ("bf16" arithmetic mode, fp32 accumulate) instead of
 * the three of split precision; producers given lo == NULL / y_lo == NULL write the hi plane only.  `zero` = any 16-byte-aligned device buffer of >= 16 zero bytes (out-of-image
 * taps load from it).  Channel counts and pitches must be multiples of 8. */
int cdf_split_bf16(const float* x, int ldx, void* hi, void* lo, int ldo, long long rows, int C, void* stream);
/* cdf_conv_gemm_bf16x: y_hi / y_lo (nullable, pitch ld_ys) additionally receive the stored output split into bf16 hi / lo planes,
 * i.e. cdf_split_bf16 fused into the producer (needs Cout % 4 == 0 and the aligned / pitched layout of the vector epilogue).
 * With the planes given, y itself may be NULL (no fp32 copy is written; not with accumulate).  Same for cdf_layernorm_c_fwd. */
/* ws / ws_floats (nullable): workspace for split-K launches.  A grid far below one tile per CU (M = B*QH*QW of a few hundred pixels: the
 * 4 x 4 / 8 x 8 levels of the 32 x 32 configurations) shares the taps out over ks = cdf_conv_gemm_bf16x_ksplit(M, Cout, nphase, ntaps of
 * phase 0) block groups whose partial sums go through ws (>= ks * M * roundup4(Cout) floats, 16-byte aligned) and a finish kernel that
 * runs the epilogue; without a (large enough) workspace, or when the query returns 1 (always with tune->splitk = 0), the launch is the
 * plain one.  The library never allocates. */
int cdf_conv_gemm_bf16x_ksplit(int M, int Cout, int nphase, int ntaps, const cdf_gemm_tuning* tune);
int cdf_conv_gemm_bf16x(const void* x_hi, const void* x_lo, int ldx, const void* zero, const void* w_hi, const void* w_lo, int ldk,
                        float* y, int ldy, int B, int H, int W, int Cin, int OH, int OW, int Cout, int QH, int QW, int os, int is,
                        int nphase, const int* phase_desc, const float* bias, const float* sbias, int ld_sbias, const float* res,
                        int ldr, float* pre, int ldp, const float* mul, int ldm, int act, int mul_mode, int accumulate,
                        void* y_hi, void* y_lo, int ld_ys, float* ws, long long ws_floats, const cdf_gemm_tuning* tune, void* stream);
/* bf16 ACTIVATION STORAGE (the "bf16" mode BASELINE configs 3 / 5 name: bf16 operands AND bf16 tensors, fp32 accumulate / master weights /
 * statistics).  A feature map between kernels is ONE bf16 plane [pixels][ld] -- what the GEMMs above call the "hi" plane IS the tensor
 * (x_lo = w_lo = y_lo = NULL, y = NULL) -- and the *_io entry points below take the remaining operands in that type too.  Pitches of a
 * bf16 operand are in bf16 elements; bf16 pointers need 8-byte alignment (GEMM operand planes 16).
 * cdf_conv_gemm_bf16x_io: cdf_conv_gemm_bf16x with typed epilogue operands; io_bf16 = bit mask
 *   1 (CDF_IO_RES_BF16): res is bf16;  2 (CDF_IO_PRE_BF16): pre is written as bf16;  4 (CDF_IO_MUL_BF16): mul is bf16;
 *   8 (CDF_IO_PRE_GRAD, any storage type): pre receives act'(v) instead of v (act 1 / 2) -- the derivative comes out of the same erf / exp
 *   evaluation as the activation, and the backward data gradient multiplies by it (mul_mode 3) instead of evaluating it again.
 * cdf_bf16_to_f32: y[r][c] = float(x[r][c]) (exact) for the few kernels that have no bf16-input form; the other direction is
 *   cdf_split_bf16 with lo = NULL (round to nearest even). */
int cdf_conv_gemm_bf16x_io(const void* x_hi, const void* x_lo, int ldx, const void* zero, const void* w_hi, const void* w_lo, int ldk,
                           float* y, int ldy, int B, int H, int W, int Cin, int OH, int OW, int Cout, int QH, int QW, int os, int is,
                           int nphase, const int* phase_desc, const float* bias, const float* sbias, int ld_sbias, const void* res,
                           int ldr, void* pre, int ldp, const void* mul, int ldm, int act, int mul_mode, int accumulate, int io_bf16,
                           void* y_hi, void* y_lo, int ld_ys, float* ws, long long ws_floats, const cdf_gemm_tuning* tune, void* stream);
int cdf_bf16_to_f32(const void* x, int ldx, float* y, int ldy, long long rows, int C, void* stream);
/* cdf_conv_gemm_bf16x_lnbwd (round 6): the data gradient of a 3 x 3 stride-1 convolution that read a channel LayerNorm's output, with that
 * LayerNorm's backward applied in the epilogue -- dh = LN'(h; mean, rstd, g)[conv_dgrad(dy)], dg / db += the parameter gradients -- so the
 * gradient with respect to the LayerNorm output never goes to HBM (deblurring_diffusion_pytorch.py:111-121, 149-151).  x = the dy planes,
 * w = the data-gradient packing of the weight, Cin = dy's channels, Cout = the LayerNorm's width; ln_x = h (pitch ld_lnx), ln_mean / ln_rstd [B*H*W].
 * cdf_conv_gemm_bf16x_lnbwd_ok: 1 if the geometry qualifies (one N tile holds every channel: Cout 64 or 128; whole row tiles:
 * B*H*W % 256 == 0; nine taps).  part: >= (B*H*W / 64) * 2 * Cout floats.  Same expressions per pixel as cdf_layernorm_c_bwd. */
int cdf_conv_gemm_bf16x_lnbwd_ok(int B, int H, int W, int Cin, int Cout, int nphase, int ntaps);
int cdf_conv_gemm_bf16x_lnbwd(const void* x_hi, const void* x_lo, int ldx, const void* zero, const void* w_hi, const void* w_lo, int ldk,
                              int B, int H, int W, int Cin, int Cout, const int* phase_desc, const float* ln_x, int ld_lnx,
                              const float* ln_mean, const float* ln_rstd, const float* ln_g, float* dh, int lddh, float* dg, float* db,
                              float* part, const cdf_gemm_tuning* tune, void* stream);
/* cdf_conv_gemm_io: the exact-fp32 GEMM (cdf_conv_gemm) with typed epilogue operands (same io_bf16 bits) and an optional bf16 output plane
 * y_hi (pitch ld_ys; y may then be NULL): how the fp32 inside of the linear-attention block reads the bf16 stream as its residual and
 * writes its result into it.  A batched launch without y needs an epilogue operand (outputs that are whole rows of one tensor). */
int cdf_conv_gemm_io(const float* x, int ldx, const float* w, int ldw, float* y, int ldy, int B, int H, int W, int Cin,
                     int OH, int OW, int Cout, int QH, int QW, int os, int is, int nphase, const int* phase_desc,
                     const float* bias, const float* sbias, int ld_sbias, const void* res, int ldr, void* pre, int ldp,
                     const void* mul, int ldm, int act, int mul_mode, int accumulate, int b_trans, int batch,
                     long long x_bs, long long w_bs, long long y_bs, int batch2, long long x_bs2, long long w_bs2,
                     long long y_bs2, int io_bf16, void* y_hi, int ld_ys, void* stream);
/* cdf_conv_wgrad_bf16x_is_row3 tells the caller whether a geometry takes the row-of-taps kernel (3 tap blocks per tile, one 512-thread
 * block per CU) so that it can size nsplit. */
int cdf_conv_wgrad_bf16x_is_row3(int QH, int QW, int CA, int CB, int ntaps, int same_size_3x3, const cdf_gemm_tuning* tune);
int cdf_conv_wgrad_bf16x(const void* a_hi, const void* a_lo, int lda, const void* b_hi, const void* b_lo, int ldb, const void* zero,
                         float* ws, int ldo, int B, int QH, int QW, int HA, int WA, int sa, int HB, int WB, int sb, int CA, int CB,
                         int ntaps, const int* tap_desc, int nsplit, float* bsum, const cdf_gemm_tuning* tune, void* stream);

/* Device-side input pipeline (replaces Dataset_Aug1 / Dataset + DataLoader, deblurring_diffusion_pytorch.py:983-1026, 1094-1096).
 * cache: [N][S][S][C] uint8 (NHWC) images already resized to S = int(1.12 image_size) (the deterministic Resize of the reference's
 * transform chain, applied once when the cache is built).  out[b][c][y][x] (NCHW fp32, [B][C][H][W]) =
 * float(cache[idx[b]][oy[b] + y][ox[b] + (flip[b] ? W-1-x : x)][c]) / 255 * 2 - 1  -- RandomCrop / CenterCrop, RandomHorizontalFlip,
 * ToTensor and Lambda(t * 2 - 1) with their exact arithmetic.  idx: int64 [B] (device), oy / ox / flip: int32 [B] (device);
 * offsets must satisfy oy + H <= S, ox + W <= S (not checked on the device). */
int cdf_augment_batch(const void* cache, long long N, int S, int C, const long long* idx, const int* oy, const int* ox,
                      const int* flip, float* out, int B, int H, int W, void* stream);
/* The same over a rectangular cache [N][SH][SW][C] with RandomCrop's `padding=pad` border (constant fill 0, i.e. -1 after the
 * conversion): (oy, ox) address the (SH + 2 pad) x (SW + 2 pad) padded image.  Replaces resolution_diffusion_pytorch.py:817-831
 * (Dataset_Aug2: Resize(s), RandomCrop(s, padding=4)) and defading_diffusion_gaussian.py:579-599 (DatasetCifar10). */
int cdf_augment_batch_pad(const void* cache, long long N, int SH, int SW, int C, int pad, const long long* idx, const int* oy,
                          const int* ox, const int* flip, float* out, int B, int H, int W, void* stream);

/* The non-GEMM layers of the FID feature extractor (deblurring-diffusion-pytorch/Fid/inception.py:16-328; its BasicConv2d layers are the
 * conv GEMM entry points with act = 3 (ReLU) and BatchNorm folded into weight / bias).  Feature maps are NHWC fp32 with a pixel pitch.
 *   cdf_pool2d: k x k window, `stride`, zero `pad` (OH = (H + 2 pad - k) / stride + 1); mode 0 = max (nn.MaxPool2d(3, 2), inception.py:91, 100;
 *               F.max_pool2d(x, 3, 1, 1), inception.py:323), mode 1 = average over the taps inside the image
 *               (F.avg_pool2d(..., count_include_pad=False), inception.py:214, 243, 282).  C, ldx, ldy multiples of 4.
 *   cdf_global_avgpool: y[b][c] = mean over HW pixels (nn.AdaptiveAvgPool2d((1, 1)), inception.py:122).
 *   cdf_resize_bilinear_nhwc: x NCHW [B][C][H][W] -> y NHWC [B][OH][OW] (pitch ldy), F.interpolate(size, mode='bilinear',
 *               align_corners=False) followed by mul * v + add (inception.py:146-153: 2 x - 1). */
int cdf_pool2d(const float* x, int ldx, float* y, int ldy, int B, int H, int W, int C, int k, int stride, int pad, int mode, void* stream);
int cdf_global_avgpool(const float* x, int ldx, float* y, int ldy, int B, int HW, int C, void* stream);
int cdf_resize_bilinear_nhwc(const float* x, float* y, int ldy, int B, int C, int H, int W, int OH, int OW, float mul, float add, void* stream);

/* Metric step after sampling (deblurring_diffusion_pytorch.py:1677-1702): SSIM as pytorch_msssim.ssim computes it (11-tap Gaussian
 * window sigma 1.5 given by the caller as 11 HOST floats, "valid" filtering along H then W, C1 = (0.01 L)^2, C2 = (0.03 L)^2).
 * x, y: [planes][H][W] fp32 (planes = B * C); partial[plane][tile] (cdf_ssim_tiles(H, W) tiles of 32 x 32 valid positions per plane)
 * receives the SUM of the SSIM map over the tile; the caller divides the per-plane sums by (H-10)(W-10) and averages. */
int cdf_ssim_tiles(int H, int W);
int cdf_ssim_partial(const float* x, const float* y, float* partial, int planes, int H, int W, const float* window11, float C1,
                     float C2, void* stream);

/* parameter layout <-> GEMM layout: dst[t][r][c] = src[c*s_c + r*s_r + t*s_t] (c >= C zero-filled up to ldc);
 * g[c*s_c + r*s_r + t*s_t] (+)= sum_z ws[z][t][r][c] */
int cdf_pack_weight(const float* src, float* dst, int T, int R, int C, int ldc, long long s_t, long long s_r,
                    long long s_c, void* stream);
/* cdf_pack_many: every cached layout in one launch.  table = nentries records in device memory, each
 *   { const float* src; void* dst0; void* dst1; long long s_t, s_r, s_c; int T, R, C, ldc, kind, first_block; }   (cdf_pack_entry_bytes() bytes)
 * kind 0: cdf_pack_weight into dst0 (fp32); kind 1: cdf_pack_weight_bf16 into dst0 (hi) / dst1 (lo, nullable).  first_block ascending; entry e
 * owns cdf_pack_blocks(T, R, ldc, s_t) consecutive blocks (ceil(T*R*ldc / 1024), or ceil(R*ldc / 1024) when the taps are contiguous in
 * the source -- s_t == 1, 1 < T <= 16 -- and a thread moves all T taps of an (r, c) pair); nblocks = their total. */
int cdf_pack_entry_bytes(void);
int cdf_pack_blocks(int T, int R, int ldc, long long s_t);
int cdf_pack_many(const void* table, int nentries, int nblocks, void* stream);
int cdf_unpack_reduce(const float* ws, float* g, int nsplit, int T, int R, int C, int ldc, long long s_t,
                      long long s_r, long long s_c, int accumulate, int tiled, void* stream);
/* the same with the bias-gradient reduction of the same weight-gradient launch folded in (one launch instead of two):
 * gbias[c] (+)= sum_z bias_ws[z * bias_ld + c], c < C (the `bsum` partials of cdf_conv_wgrad*). */
int cdf_unpack_reduce_bias(const float* ws, float* g, int nsplit, int T, int R, int C, int ldc, long long s_t, long long s_r,
                           long long s_c, const float* bias_ws, float* gbias, int bias_ld, int accumulate, int tiled, void* stream);
/* `tiled` (both entry points): 1 (what the package passes) = parameter layouts whose fast index is not the slab's (s_c != 1, T in {1, 9, 16})
 * are reduced by the LDS-tiled transposing kernel (contiguous runs per output channel instead of lone 4-byte read-modify-writes);
 * 0 = always the element-wise kernel.  Same sums either way up to fp32 summation order. */

/* out[seg][c] (+)= sum over the rows of segment seg of x[r*ld + c]  (bias / time-bias gradients);
 * ws >= nseg * cdf_colsum_nchunk(rows_per_seg) * C floats */
int cdf_colsum_nchunk(int rows_per_seg);
int cdf_colsum(const float* x, float* out, float* ws, int nseg, int rows_per_seg, int C, int ld, int ldo,
               int accumulate, void* stream);
/* cdf_colsum_io: x_bf16 != 0 reads x as a bf16 tensor (ld in bf16 elements; the bias gradient of a transposed convolution on the bf16 stream) */
int cdf_colsum_io(const void* x, float* out, float* ws, int nseg, int rows_per_seg, int C, int ld, int ldo,
                  int accumulate, int x_bf16, void* stream);

/* ---- normalisation ------------------------------------------------------------------------------
 * channel LayerNorm (deblurring_diffusion_pytorch.py:111-121): per-pixel over C, biased variance,
 * y = (x-mean)/sqrt(var+eps)*g+b; mean/rstd [M] are saved for backward (nullable in inference).
 * backward also produces dg/db; part >= cdf_layernorm_blocks(M,C)*2*C floats of scratch. */
int cdf_layernorm_blocks(long long M, int C);
int cdf_layernorm_c_fwd(const float* x, int ldx, float* y, int ldy, const float* g, const float* b, float* mean,
                        float* rstd, long long M, int C, float eps, void* y_hi, void* y_lo, int ld_ys, void* stream);
/* add (nullable, pitch ldadd): dx = grad + add -- the residual branch of Residual(PreNorm(..)) added in the same pass instead of a
 * copy + accumulate; accumulate_dx = 1 is the same with add = dx (the two exclude each other). */
int cdf_norm_param_reduce(const float* part, int nblocks, int C, float* dg, float* db, int accumulate, void* stream);
/* cdf_layernorm_c_bwd_planes (round 6): cdf_layernorm_c_bwd whose dx is ALSO written as bf16 hi / lo planes (pitch ld_planes, the cdf_split_bf16 of the
 * stored value) for the GEMMs that consume this gradient. */
int cdf_layernorm_c_bwd_planes(const float* dy, int lddy, const float* x, int ldx, const float* g, const float* mean, const float* rstd,
                               float* dx, int lddx, const float* add, int ldadd, float* dg, float* db, float* part, long long M, int C,
                               int accumulate_dx, int accumulate_param, void* dx_hi, void* dx_lo, int ld_planes, void* stream);
int cdf_layernorm_c_bwd(const float* dy, int lddy, const float* x, int ldx, const float* g, const float* mean,
                        const float* rstd, float* dx, int lddx, const float* add, int ldadd, float* dg, float* db, float* part,
                        long long M, int C, int accumulate_dx, int accumulate_param, void* stream);
/* bf16 activation storage.  fwd: x_bf16 != 0 -> x is a bf16 tensor (outputs as before: fp32 y and / or the bf16 plane y_hi).
 * bwd: io_bf16 bit mask 1 dy | 2 x | 4 dx | 8 add are bf16; instantiated: 0, 7 (dy, x, dx: ConvNeXt block) and 14 (x, dx, add: the
 * attention block, whose own gradient dxn is fp32). */
int cdf_layernorm_c_fwd_io(const void* x, int ldx, float* y, int ldy, const float* g, const float* b, float* mean,
                           float* rstd, long long M, int C, float eps, void* y_hi, void* y_lo, int ld_ys, int x_bf16, void* stream);
int cdf_layernorm_c_bwd_io(const void* dy, int lddy, const void* x, int ldx, const float* g, const float* mean,
                           const float* rstd, void* dx, int lddx, const void* add, int ldadd, float* dg, float* db, float* part,
                           long long M, int C, int accumulate_dx, int accumulate_param, int io_bf16, void* stream);
/* GroupNorm(groups) [+ SiLU] (Model2.py:27-33): statistics per (sample, group) over C/groups channels
 * and all HW pixels; mean/rstd [B][groups].  fwd ws >= B*nchunk*2*C floats;
 * bwd ws >= B*nchunk*2*C + B*2*C + B*groups*2 floats, nchunk = cdf_groupnorm_nchunk(HW). */
int cdf_groupnorm_nchunk(int HW);
int cdf_groupnorm_fwd(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta, float* mean,
                      float* rstd, float* ws, int B, int HW, int C, int groups, float eps, int silu, void* stream);
/* the same with the tail of ResnetBlock.forward fused in (Model2.py:118-126): p_drop > 0 applies cdf_dropout's mask (seed, element
 * index row * C + c) to the activated output; y_hi / y_lo (nullable, pitch ld_ys): the result again as bf16 hi / lo planes, bit-equal
 * to cdf_split_bf16 of y (y_lo null: hi only); y itself may then be null (planes only). */
/* backward of the same: p_drop > 0 says dy is the gradient of the DROPPED output; the mask is applied while dy is read */
int cdf_groupnorm_bwd_ex(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* beta,
                         const float* mean, const float* rstd, float* dx, int lddx, float* dgamma, float* dbeta, float* ws,
                         int B, int HW, int C, int groups, int silu, int accumulate_dx, int accumulate_param, float p_drop,
                         long long seed, void* stream);
int cdf_groupnorm_fwd_ex(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta, float* mean,
                         float* rstd, float* ws, int B, int HW, int C, int groups, float eps, int silu, float p_drop,
                         long long seed, void* y_hi, void* y_lo, int ld_ys, void* stream);
int cdf_groupnorm_bwd(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* beta,
                      const float* mean, const float* rstd, float* dx, int lddx, float* dgamma, float* dbeta, float* ws,
                      int B, int HW, int C, int groups, int silu, int accumulate_dx, int accumulate_param,
                      void* stream);

/* ---- depthwise 7x7 (ConvNeXt ds_conv + time bias, deblurring_diffusion_pytorch.py:145,157-162) ---
 * w packed [49][ldw] (cdf_pack_weight, R=1); flip=1 mirrors the taps (data gradient).
 * wgrad: dw in the parameter layout [C][1][7][7], dbias [C], dsb [B][ld_dsb] (time-bias gradient,
 * overwritten); ws >= B * cdf_dwconv7_wgrad_nchunk(H) * 50 * C floats. */
int cdf_dwconv7(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* sbias, int ld_sbias,
                float* y, int ldy, int B, int H, int W, int C, int flip, int accumulate,
                const float* res, int ldr, void* stream);
/* cdf_dwconv7_planes (round 6): cdf_dwconv7 on fp32 tensors whose result is ALSO written as bf16 hi / lo planes (pitch ld_ys in bf16 elements, the
 * cdf_split_bf16 of the stored value): the data-gradient pass hands its result to the next block's GEMMs in that form. */
int cdf_dwconv7_planes(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* sbias, int ld_sbias, float* y, int ldy,
                       int B, int H, int W, int C, int flip, int accumulate, const float* res, int ldr, void* y_hi, void* y_lo, int ld_ys,
                       void* stream);
int cdf_dwconv7_wgrad_nchunk(int H);
int cdf_dwconv7_wgrad(const float* x, int ldx, const float* dy, int lddy, float* dw, float* dbias, float* dsb,
                      int ld_dsb, float* ws, int B, int H, int W, int C, int accumulate, void* stream);
/* bf16 activation storage: io_bf16 != 0 -> x, y, res (forward / data gradient) resp. x, dy (weight gradient) are bf16 tensors; weights,
 * biases, partial sums and the gradients written stay fp32.  cdf_dwconv7_io with io_bf16 == 2: x and res bf16, y fp32 (the data gradient
 * the bf16 stream hands to the fp32 tensors of the image-side block). */
int cdf_dwconv7_io(const void* x, int ldx, const float* w, int ldw, const float* bias, const float* sbias, int ld_sbias,
                   void* y, int ldy, int B, int H, int W, int C, int flip, int accumulate, const void* res, int ldr, int io_bf16,
                   void* stream);
int cdf_dwconv7_wgrad_io(const void* x, int ldx, const void* dy, int lddy, float* dw, float* dbias, float* dsb,
                         int ld_dsb, float* ws, int B, int H, int W, int C, int accumulate, int io_bf16, void* stream);

/* ---- attention -------------------------------------------------------------------------------------
 * LinearAttention core (deblurring_diffusion_pytorch.py:176-187) on qkv [B,n,ld] = (q|k|v), each
 * heads*32 channels, n = H*W tokens.  The softmax over n and the context are column reductions:
 *   cdf_linattn_context : kmax/ksum [B,HD], ctx[b,h,d,e] = sum_n softmax_n(k)[d,n] v[e,n], ctxs = scale*ctx
 *   out[n, h*32+e] = sum_d q[n, h*32+d] ctxs[h,d,e]   -> a K=32 GEMM per (b, head): cdf_conv_gemm
 * backward:
 *   cdf_linattn_dcontext: dctx = scale * sum_n q[n,d] dout[n,e] ; rvec[d] = sum_e dctx*ctx
 *   dq = dout . ctxs^T, dP = v . dctx^T, dv = P . dctx  (cdf_conv_gemm) with P from cdf_linattn_softk,
 *   dk = P * (dP - rvec)                                  (cdf_linattn_dk)
 * ws >= cdf_linattn_ws_floats(B,n,heads) floats. */
int cdf_linattn_nsplit(int n);
size_t cdf_linattn_ws_floats(int B, int n, int heads);
/* koff: channel offset of k inside a row (v follows at koff + heads*32): heads*32 for the reference's (q|k|v) tensor, 0 for a (k|v)
 * tensor (the q-free form of colddiff/ops.py linattn_fold: q never exists when dim <= heads*32). */
/* onepass: 1 (what the package passes) = k and v are read ONCE (per-tile maxima, partials rescaled in the finalize kernel: online softmax);
 * 0 = column-max pass + context pass.  Same results up to fp32 rounding. */
int cdf_linattn_context(const float* qkv, int ld, int koff, float* ctx, float* ctxs, float* kmax, float* ksum, float* ws, int B,
                        int n, int heads, float scale, int onepass, void* stream);
/* cdf_linattn_kvctx (round 2): the k | v projection kv = xn . Wkv^T ([B,n,256], 4 heads; Wkv as bf16 hi [/ lo] planes [256][ldk], K contiguous:
 * cdf_pack_weight_bf16; w_lo == NULL: single bf16 operands) AND the context partials of the same pixels in one pass -- k and v are written once
 * and not read back by the forward pass.  n % 128 == 0, dim % 32 == 0.  ws: (2*128 + 4*1024) * B * cdf_linattn_kvctx_parts(B, n, slots) floats;
 * cdf_linattn_finalize(ws, nparts = cdf_linattn_kvctx_parts(B, n, slots), ...) then yields what cdf_linattn_context yields (ctx, ctxs, kmax, ksum). */
/* slots (both): target block count per launch, <= 0 = the default 512 (about two blocks per CU queued); the SAME value must be given to
 * cdf_linattn_kvctx_parts (workspace size, nparts of cdf_linattn_finalize) and to cdf_linattn_kvctx. */
int cdf_linattn_kvctx_parts(int B, int n, int slots);
int cdf_linattn_kvctx(const float* xn, int ldx, const void* w_hi, const void* w_lo, int ldk, float* kv, int ldkv, float* ws, int B, int n,
                      int dim, int heads, int slots, void* stream);
int cdf_linattn_finalize(const float* ws, int nparts, float* ctx, float* ctxs, float* kmax, float* ksum, int B, int heads, float scale,
                         void* stream);
int cdf_linattn_dcontext(const float* qkv, int ld, const float* dout, int lddo, const float* ctx, float* dctx,
                         float* rvec, float* ws, int B, int n, int heads, float scale, void* stream);
int cdf_linattn_softk(const float* qkv, int ld, const float* kmax, const float* ksum, float* pn, int ldp, int B, int n,
                      int heads, void* stream);
int cdf_linattn_dk(const float* pn, int ldp, const float* dp, int lddp, const float* rvec, float* dk, int lddk, int B,
                   int n, int heads, void* stream);
/* cdf_linattn_bwd_kv: the k / v part of the attention backward in one pass (replaces cdf_linattn_softk + two K = 32 products +
 * cdf_linattn_dk): dk[n,d] = P[n,d] (sum_e v[n,e] dctx[d,e] - rvec[d]),  dv[n,e] = sum_d P[n,d] dctx[d,e],  P = softmax_n(k) recomputed
 * from kmax / ksum; k | v are read at channel offset koff of qkv's rows, dk | dv written at channel offset dkoff of dqkv's rows
 * (pitch lddq).  heads <= 4. */
int cdf_linattn_bwd_kv(const float* qkv, int ld, int koff, const float* dctx, const float* rvec, const float* kmax, const float* ksum,
                       float* dqkv, int lddq, int dkoff, int B, int n, int heads, void* stream);
/* ... with dk | dv written as bf16 hi / lo planes [B n][ldpl] (dkv_lo nullable) instead of fp32: the operand form of the k | v projection's
 * data- and weight-gradient GEMMs (cdf_conv_gemm_bf16x, cdf_conv_wgrad_bf16x) -- same bytes, no split downstream. */
int cdf_linattn_bwd_kv_planes(const float* qkv, int ld, int koff, const float* dctx, const float* rvec, const float* kmax, const float* ksum,
                              void* dkv_hi, void* dkv_lo, int ldpl, int dkoff, int B, int n, int heads, void* stream);
/* cdf_linattn_dctx_finish: dctx[i] = scale * raw[i]; rvec[row] = sum_e dctx[row][e] * ctx[row][e] over rows of 32 (rows = B * heads * 32):
 * the tail of the fused attention backward, where raw = d(scale * ctx) comes out of a batched GEMM (see colddiff/ops.py linattn_bwd). */
int cdf_linattn_dctx_finish(const float* raw, const float* ctx, float* dctx, float* rvec, long long rows, float scale, void* stream);
/* AttnBlock (Model2.py:164-188) row softmax of the score matrix: p = softmax(scale*s) per row;
 * ds = scale * p * (dp - sum(dp*p)) */
int cdf_softmax_rows_fwd(const float* s, float* p, long long rows, int n, int ld, float scale, void* stream);
int cdf_softmax_rows_bwd(const float* p, const float* dp, float* ds, long long rows, int n, int ld, float scale,
                         void* stream);

/* ---- small fused ops ---------------------------------------------------------------------------------
 * sinusoidal time embedding (DEBLUR:91-103, MODEL2:6-24): out[b] = (sin(t f_j) | cos(t f_j)),
 * freq[dim/2] = the init-time frequency table exp(-j ln(1e4)/(dim/2-1)) */
int cdf_sinusoidal(const int64_t* t, const float* freq, float* out, int ldo, int B, int dim, void* stream);
/* Direct convolution for layers with <= 4 input channels (image-side convs: DEBLUR:145-165 with dim = channels):
 * x is NHWC with pitch exactly 4 (channel padding zero), stride 1, k = 1 or 3, "same" padding; w = weights packed
 * [k*k][4][ldw] fp32 (rows >= Cin zero).  Cout = 4 * (power of two <= 64).
 *   fwd  : y (nullable) / pre (nullable pre-activation) / y_hi,y_lo (nullable bf16 planes) = act(conv(x) + bias)
 *   dgrad: dx[M][4] (+)= sum_taps dy * w  (channel 3 of dx receives the zero-weight sum, i.e. 0)
 *   wgrad: part[nchunk][k*k*Cin][Cout] and bsum[nchunk][Cout] partial sums, nchunk = cdf_conv_cin4_nchunk(B*H*W);
 *          reduce with cdf_unpack_reduce (T = k*k, R = Cin).  cdf_pack_cin4 builds w from the PyTorch [Cout][Cin][k][k] weight. */
int cdf_conv_cin4_fwd(const float* x, const float* w, int ldw, const float* bias, float* y, int ldy, float* pre, int ldp, void* y_hi,
                      void* y_lo, int ld_ys, int B, int H, int W, int Cout, int k, int act, void* stream);
int cdf_conv_cin4_dgrad(const float* dy, int ldd, const float* w, int ldw, float* dx, int B, int H, int W, int Cout, int k, int accumulate,
                        void* stream);
/* Second stage of the two-stage 3x3 data gradient (first stage: the 1x1 GEMM z[q][c*9 + ky*3 + kx] = sum_co dy[q][co] W[co][c][ky][kx],
 * whose [Cout][9 Cin] weight matrix is the parameter in its PyTorch layout):  dx[p][c] (+)= sum_{ky,kx} z[p - (ky-1, kx-1)][c*9 + ky*3 + kx],
 * zero outside the image; dx is [B,H,W,4] (channels >= Cin zero), z rows have pitch ldz >= 9 Cin (ldz % 4 == 0, <= 64). */
int cdf_conv_cin4_tapsum3(const float* z, int ldz, float* dx, int B, int H, int W, int Cin, int accumulate, void* stream);
int cdf_conv_cin4_nchunk(long long M);
int cdf_conv_cin4_wgrad(const float* x, const float* dy, int ldd, float* part, float* bsum, int B, int H, int W, int Cin, int Cout,
                        int k, void* stream);
int cdf_pack_cin4(const float* w, float* dst, int ldw, int Cout, int Cin, int k, void* stream);
/* Skinny linear layers (M = batch rows; the time-embedding MLPs DEBLUR:96-103,142-144,160, MODEL2:44-48,238-245):
 *   out[m][j] = bias[j] + sum_i in[m][i] * Wm[i*ldw + j]        (columns J..ldo-1 of out are zeroed)
 * forward: Wm = the weight packed [K][N] (cdf_pack_weight), i = k, j = n; data gradient: Wm = the PyTorch [N][K]
 * weight itself, i = n, j = k.  cdf_linear_small_wgrad ACCUMULATES dW[n][k] += sum_m dy[m][n] x[m][k], db[n] += sum_m dy[m][n]. */
int cdf_linear_small(const float* in, int ldi, const float* Wm, int ldw, const float* bias, float* out, int ldo, int M, int I, int J,
                     void* stream);
int cdf_linear_small_wgrad(const float* dy, int ldd, const float* x, int ldx, float* dW, float* db, int M, int N, int K, void* stream);
/* act 1 = exact GELU, 2 = SiLU, 3 = ReLU (forward only) on [rows, C] with pitches */
int cdf_act_fwd(const float* x, int ldx, float* y, int ldy, long long rows, int C, int act, void* stream);
int cdf_act_bwd(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, long long rows, int C, int act,
                int accumulate, void* stream);
/* dst = alpha*dst + beta*src (alpha == 0 ignores the old dst) */
int cdf_axpby(float* dst, int ldd, const float* src, int lds, long long rows, int C, float alpha, float beta,
              void* stream);
/* nearest x2 upsample (Model2.py:47) and its adjoint */
int cdf_upsample2(const float* x, int ldx, float* y, int ldy, int B, int H, int W, int C, void* stream);
int cdf_upsample2_bwd(const float* dy, int lddy, float* dx, int lddx, int B, int H, int W, int C, int accumulate,
                      void* stream);
/* dropout with a counter-based mask (same seed => same mask, used again in backward) */
int cdf_dropout(const float* x, int ldx, float* y, int ldy, long long rows, int C, float p, long long seed,
                void* stream);
/* torch.optim.Adam defaults over a flat arena (DEBLUR:1117,1200); `step` is the 1-based step count */
int cdf_adam_step(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2,
                  double eps, int step, void* stream);
/* EMA.update_average (DEBLUR:78-81): ma = ma*beta + (1-beta)*p */
int cdf_ema_update(float* ma, const float* p, long long n, double beta, void* stream);
int cdf_scale(float* x, long long n, float s, void* stream);
int cdf_zero(void* p, long long bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COLDDIFF_H */

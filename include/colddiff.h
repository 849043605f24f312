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

/* L1 / L2 training loss (deblurring_diffusion_pytorch.py:966-971): out[0] = mean|x-y| or
 * mean (x-y)^2; backward writes d loss / d y scaled by gout[0]. partial: >= 1024 floats. */
int cdf_loss_fwd(const float* x, const float* y, float* out, float* partial, long long n, int l2, void* stream);
int cdf_loss_bwd(const float* x, const float* y, const float* gout, float* gy, long long n, int l2, void* stream);

/* NCHW image <-> NHWC feature map (pitch ld); `add` (NCHW, nullable) is the Unet residual=True */
int cdf_nchw_to_nhwc(const float* x, float* y, int B, int C, int HW, int ldy, void* stream);
int cdf_nhwc_to_nchw(const float* x, float* y, const float* add, int B, int C, int HW, int ldx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COLDDIFF_H */

/*
 * mulactseg_hip.h -- C ABI of libmulactseg_hip.so: the MI355X (gfx950) implementation of the
 * MulActSeg hot path (per-superpixel BvSB acquisition scorer + stage-1 partial-label losses).
 *
 * Boundary contract:
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says "host";
 *   - the caller allocates every output / workspace and keeps inputs alive until `stream` has run
 *     the call; no allocation, no ownership transfer, no host synchronisation inside;
 *   - every launch goes to `stream` (a hipStream_t passed as void*; NULL = the null stream);
 *   - return value: 0 = OK, > 0 = hipError_t of the failed launch, < 0 = argument error
 *     (mas_error_string() explains); never throws;
 *   - thread-safe when each host thread uses its own stream and buffers; no global state.
 *
 * The reference (sehyun03/MulActSeg) is pure Python over PyTorch + torch_scatter; each entry point
 * below names the reference code (file:line, relative to the reference root) it replaces.  The
 * Python-side binding a maintainer of the reference would add is a ctypes stub -- see INTEGRATION.md.
 *
 * Numeric conventions (mulactseg_amd/csrc/detmath.h is the normative arithmetic):
 *   - logits z are f32 NCHW contiguous, exactly what `model(images)` returns;
 *   - temperature enters as invT = float32(1 / float32(T));
 *   - region / class / loss accumulators are unsigned 64-bit FIXED-POINT sums, so results do not
 *     depend on thread, wave, workgroup or GPU count;
 *   - on exact ties the LOWEST class index / LOWEST pixel index wins.
 */
#ifndef MULACTSEG_HIP_H
#define MULACTSEG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* superpixel-id element types accepted by every `spx` argument */
#define MAS_ID_I64 0   /* torch.long, what the reference data layer yields (ext_transforms.py:406) */
#define MAS_ID_I32 1
#define MAS_ID_U16 2   /* compact resident pool maps (S <= 65535) */

#define MAS_MAX_CLASSES 32

/* fixed-point fractional bits of the accumulators (see detmath.h) */
#define MAS_SCORE_FRAC_BITS 40
#define MAS_PROB_FRAC_BITS 31
#define MAS_LOSS_FRAC_BITS 32

int mas_abi_version(void);
const char* mas_error_string(int code);

/* ---------------------------------------------------------------------------------------------
 * K2  class-prior pass.  Replaces, per batch,
 *       preds_prob = softmax(preds / ce_temp, dim=1); cum += mean(preds_prob, dim=(0,2,3))
 *     active_selection/my_bvsb_predclsbal_pwr_banignore.py:41-42 (VOC twin ..._pwr.py:41-42).
 * Adds, for every image b and class c, sum_p floor(softmax(z_p * invT)_c * 2^31) into
 * prob_sum[b*C + c]  (caller zeroes prob_sum; the host turns the integer sums into the reference's
 * mean-of-batch-means and the class weight (coeff*cum+1)^-2).
 * --------------------------------------------------------------------------------------------- */
int mas_class_prob_sum(const float* z, int B, int C, int H, int W, float invT,
                       uint64_t* prob_sum /* [B,C] += */, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K1+K3  per-superpixel accumulation of the (class-weighted) BvSB margin and of the arg-max-class
 * histogram.  Replaces
 *       bvsb, top1 = softmax_bvsb(preds)                         active_selection/my_bvsb.py:19-27
 *       w = cls_weight[top1]; scatter(bvsb*w, spx, 'mean')       ..._pwr_banignore.py:57-65
 *       scatter(one_hot(top1), spx, 'sum')                       ..._pwr_banignore.py:67-69
 * cls_w == NULL gives the unweighted my_bvsb.py:66-73 variant.  Pixels whose id is outside [0,S)
 * are skipped.  Adds into score_sum[b*S+s] (fixed point, 40 fractional bits) and
 * hist[(b*S+s)*C + c] (pixel counts); the caller zeroes both.
 * --------------------------------------------------------------------------------------------- */
int mas_bvsb_region_accum(const float* z, const void* spx, int spx_dtype, const float* cls_w /* [C] or NULL */,
                          int B, int C, int H, int W, int S, float invT,
                          uint64_t* score_sum /* [B,S] += */, uint32_t* hist /* [B,S,C] += */, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K3 tail + K4 ban.  Replaces the division of scatter(...,'mean') and
 *       dominant = argmax(region_ntop1); score[dominant == C-1] = 0      ..._pwr_banignore.py:79-84
 * score[r] = mean (0 for an empty region); dominant[r] = first arg-max of hist[r,:];
 * ban_class >= 0 zeroes the score of regions whose dominant class equals it.
 * Optional outputs may be NULL: dominant, count, hist_i64 (region_ntop1 as the reference's int64).
 * --------------------------------------------------------------------------------------------- */
int mas_region_finalize(const uint64_t* score_sum, const uint32_t* hist, int64_t n_regions, int C, int ban_class,
                        float* score /* [n_regions] */, int32_t* dominant, uint32_t* count, int64_t* hist_i64,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MULACTSEG_HIP_H */

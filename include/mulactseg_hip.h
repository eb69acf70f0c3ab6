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
 *   - thread-safe when each host thread uses its own stream and buffers; no global mode state (per-call option structs / flag
 *     arguments carry every A/B switch).
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
#define MAS_MAP_U8 3   /* label maps of mas_train_augment only */

#define MAS_MAX_CLASSES 32

/* fixed-point fractional bits of the accumulators (see detmath.h) */
#define MAS_SCORE_FRAC_BITS 40
#define MAS_PROB_FRAC_BITS 23
#define MAS_LOSS_FRAC_BITS 32

/* The version of THIS header.  Any change of an exported signature or of the meaning of an argument bumps it; a binding compares
 * mas_abi_version() of the library it loaded with the MAS_ABI_VERSION it was written against and refuses a mismatch
 * (mulactseg_amd/_lib.py:load does).  History: 1 = rounds 1-4 (the round-4 additions -- mas_sk_opts, the split-bf16 entry points --
 * should have bumped it and did not); 5 = round 5 (mas_single_pass_accum_lowres_opt replaces the process-wide
 * mas_single_pass_lowres_generic switch; mas_test_occupy moved to the test-support library; the BatchNorm-fused forms of
 * mas_conv_bx_fwd); 6 = role 2 of mas_conv_bx_pack / _packed_bytes / _pack_job and ksize 3 at stride 2 in mas_conv_bx_supported /
 * mas_conv_bx_fwd (a library of version 5 answers "unsupported" to both). */
#define MAS_ABI_VERSION 7
int mas_abi_version(void);
const char* mas_error_string(int code);

/* ---------------------------------------------------------------------------------------------
 * K2  class-prior pass.  Replaces, per batch,
 *       preds_prob = softmax(preds / ce_temp, dim=1); cum += mean(preds_prob, dim=(0,2,3))
 *     active_selection/my_bvsb_predclsbal_pwr_banignore.py:41-42 (VOC twin ..._pwr.py:41-42).
 * Adds, for every image b and class c, sum_p round(softmax(z_p * invT)_c * 2^23) into
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

/* =============================================================================================
 * Stage-1 partial-label losses (K5 merged-positive CE, K6 group / MIL max-pool CE)
 * ============================================================================================= */

/* `flags` of the loss entry points */
#define MAS_LOSS_CE 1               /* compute the merged-positive CE sums */
#define MAS_LOSS_GROUP 2            /* compute the group (max-pool) loss */
#define MAS_LOSS_GROUP_ONLY_MULTI 4 /* group loss only over superpixels with > 1 target bit
                                       (GroupMultiLabelCE_onlymulti, ..._mclossablation2.py:36,51-53) */
#define MAS_LOSS_TCE 16             /* stage-2 temperature cross entropy (utils/loss.py:10-21): the `spx` map holds CLASS LABELS (S = C),
                                     * `mask` = label != ignore_index; per valid pixel l = -log softmax(z / T)[label] (no epsilon) into
                                     * the ce sum; mas_loss_values / mas_loss_scales divide by n, not 1 + n.  Combine with MAS_LOSS_CE. */
#define MAS_LOSS_DECOMP 8           /* separate one-hot (ce) / multi-hot (mc) sums and normalisers
                                       (OnehotCEMultihotChoice, ..._lossdecomp.py:58-72); otherwise one
                                       merged sum (MultiChoiceCE_, active_joint_multi_predignore.py:59-61) */

/* layout of the caller-zeroed accumulator `acc` (8 x uint64) */
#define MAS_ACC_SUM_CE 0     /* fixed point, 32 fractional bits */
#define MAS_ACC_SUM_MC 1
#define MAS_ACC_N_CE 2
#define MAS_ACC_N_MC 3
#define MAS_ACC_N_EMPTY 4    /* selected pixels whose superpixel has no target bit (the reference asserts
                                there are none, ..._lossdecomp.py:67) */
#define MAS_ACC_SUM_GROUP 5
#define MAS_ACC_N_GROUP 6
#define MAS_ACC_WORDS 8

/* Multi-hot target rows u8 [n_rows, cols_stored] -> one bit mask per row over the first `cols_used`
 * columns (cols_used = cols_stored for the "predignore" losses, cols_stored-1 for the base classes of
 * utils/loss.py:104,124,571 which drop the last column). */
int mas_target_bits(const uint8_t* targets, int64_t n_rows, int cols_stored, int cols_used,
                    uint32_t* bits /* [n_rows] */, void* stream);

/* Forward scan.  Replaces the bodies of
 *   OnehotCEMultihotChoice.forward   trainer/active_joint_multi_predignore_lossdecomp.py:21-72
 *   GroupMultiLabelCE_onlymulti.fwd  trainer/active_joint_multi_predignore_mclossablation2.py:22-75
 *   (and MultiChoiceCE_/GroupMultiLabelCE_, utils/loss.py base classes via `flags`).
 * z [N,C,H,W] f32; spx [N,H,W]; mask [N,H,W] u8/bool (selected pixels); bits [N,S] from mas_target_bits.
 * Adds fixed-point sums / counts into acc[0..4]; fills gmax [N,S,C] (caller-zeroed) with packed words
 * (float bits of max_p softmax_c) << 32 | (0xffffffff - arg pixel index), 0 = no entry. */
int mas_partial_loss_fwd(const float* z, const void* spx, int spx_dtype, const uint8_t* mask, const uint32_t* bits,
                         int N, int C, int H, int W, int S, float invT, int flags,
                         uint64_t* gmax /* [N,S,C] */, uint64_t* acc /* [8] */, void* stream);

/* sum over table entries of -log(max + 1e-8) and their count -> acc[5], acc[6]
 * (..._mclossablation2.py:67-73). */
int mas_group_finalize(const uint64_t* gmax, int64_t n_entries, uint64_t* acc, void* stream);

/* losses[0..2] = ce, mc, group (DECOMP) or merged-positive, 0, group: sum / (1 + n), f32 on device. */
int mas_loss_values(const uint64_t* acc, int flags, float* losses /* [3] */, void* stream);

/* scale[k] = grad_out[k] / (1 + n_k): the factor the backward scan multiplies in (device-side, so the
 * training step needs no host synchronisation between forward and backward). */
int mas_loss_scales(const uint64_t* acc, const float* grad_out /* [3] */, int flags, float* scale /* [3] */, void* stream);

/* The trainer's objective and its chain rule in the same two launches (fewer, smaller kernels around the scans):
 *   losses4 = (ce, mc, group, (w[0]*ce + w[1]*mc) + w[2]*group)      -- `coeff*ce + coeff_mc*mc + coeff_gm*group`,
 *                                                                        trainer/active_joint_multi_predignore_lossdecomp.py:104
 *   scale[k] = (grad_total * w[k]) / (1 + n_k)                        -- what mas_partial_loss_bwd[_lowres] multiplies in.
 * weights, grad_total: device pointers (3 floats / 1 float). */
int mas_loss_values_weighted(const uint64_t* acc, int flags, const float* weights /* [3] */, float* losses4 /* [4] */, void* stream);
int mas_loss_scales_weighted(const uint64_t* acc, const float* grad_total /* [1] */, const float* weights /* [3] */, int flags,
                             float* scale /* [3] */, void* stream);

/* Backward scan: writes dz [N,C,H,W] completely (zeros outside the mask).  Autograd equivalent of the
 * reference losses; the group loss sends gradient only to the arg-max pixel of each (superpixel, class)
 * (torch_scatter scatter_max backward). */
int mas_partial_loss_bwd(const float* z, const void* spx, int spx_dtype, const uint8_t* mask, const uint32_t* bits,
                         const uint64_t* gmax, const float* scale /* [3] */,
                         int N, int C, int H, int W, int S, float invT, int flags, float* dz, void* stream);

/* Quarter-resolution forms of the two scans.  The model emits cosine logits zq [N,C,h,w] and upsamples them x4 bilinearly to
 * the crop (models/segmentation/utils.py:25, F.interpolate(..., mode='bilinear', align_corners=False)) before the losses read
 * them.  These entry points take zq and evaluate that interpolation per selected pixel (operation order of
 * mas_upsample_bilinear_fwd), so the forward sums / gmax equal those of mas_partial_loss_fwd on the materialised tensor bit
 * for bit, and neither the [N,C,H,W] logits nor their gradient ever exist in memory.
 * Backward: dzq_fix [N,C,h,w] int64, caller-zeroed; every selected pixel adds round_to_nearest_even(d * ly * lx * 2^44) for
 * each of the four elements its class-c logit was interpolated from (integer atomics: order-independent, run-to-run
 * identical); mas_fix_to_float(dzq_fix, n, 44, dzq) then yields the f32 gradient of zq. */
int mas_partial_loss_fwd_lowres(const float* zq, int h, int w, const void* spx, int spx_dtype, const uint8_t* mask,
                                const uint32_t* bits, int N, int C, int H, int W, int S, float invT, int flags,
                                uint64_t* gmax /* [N,S,C] */, uint64_t* acc /* [8] */, void* stream);
int mas_partial_loss_bwd_lowres(const float* zq, int h, int w, const void* spx, int spx_dtype, const uint8_t* mask,
                                const uint32_t* bits, const uint64_t* gmax, const float* scale /* [3] */, int N, int C, int H, int W,
                                int S, float invT, int flags, int64_t* dzq_fix /* [N,C,h,w] */, void* stream);
#define MAS_GRAD_FRAC_BITS 44
/* out[i] = (float)(fix[i] * 2^-frac_bits) */
int mas_fix_to_float(const int64_t* fix, int64_t n, int frac_bits, float* out, void* stream);

/* Fused forms: ONE call launches everything a direction needs (what the loss modules of mulactseg_amd/utils/loss.py run).
 * Forward = a prep launch (zero the accumulators and the (superpixel, class) table; `targets` [N,S,cols_stored] u8 multi-hot rows ->
 * bit masks over the first cols_used columns -- or pass ready `bits` [N,S] and targets = NULL), the forward scan, and the group
 * finalize whose last workgroup writes `losses` ([3], or [4] with `weights` [3]: the fourth is (w0*ce + w1*mc) + w2*group).
 * losses = NULL: no values -- a data-parallel caller all-reduces the first 8 u64 words of `work` (the accumulators) and then calls
 * mas_loss_values[_weighted] on them.  h, w > 0: z is the model's quarter-resolution tensor [N,C,h,w] (as mas_partial_loss_fwd_lowres);
 * h = w = 0: z [N,C,H,W].  `work`: mas_partial_loss_work_bytes(N, S, C, flags) bytes, 8-byte aligned, NOT initialised by the
 * caller; layout acc u64[8] | gmax u64[N,S,C] (group flags only) | bits u32[N,S]; it is the state the backward call reads.
 * Backward = (h > 0: a memset of `dzq_fix` [N,C,h,w] i64 scratch) + the backward scan, which forms its scale factors itself from the
 * accumulators in `work` and `grad` ([3] upstream gradients of (ce, mc, group); with `weights`: [1], dL/dtotal) + (h > 0: the
 * conversion into dz = dzq [N,C,h,w] f32).  h = 0: dz [N,C,H,W], dzq_fix unused.  bits = NULL: the masks the forward call left in
 * `work`.  Same kernels and bits as the step-by-step entry points above. */
size_t mas_partial_loss_work_bytes(int N, int S, int C, int flags);
int mas_partial_loss_fwd_fused(const float* z, int h, int w, const void* spx, int spx_dtype, const uint8_t* mask, const uint8_t* targets,
                               int cols_stored, int cols_used, const uint32_t* bits, int N, int C, int H, int W, int S, float invT, int flags,
                               const float* weights, void* work, size_t work_bytes, float* losses, void* stream);
int mas_partial_loss_bwd_fused(const float* z, int h, int w, const void* spx, int spx_dtype, const uint8_t* mask, const uint32_t* bits,
                               const void* work, const float* grad, const float* weights, int N, int C, int H, int W, int S, float invT,
                               int flags, float* dz, int64_t* dzq_fix, void* stream);

/* =============================================================================================
 * K4  ordering of the region scores and the budgeted selection walk
 * ============================================================================================= */

/* One 64-bit key per region, [order-preserving f32 score bits : 32 | path rank : 32-b | id : b],
 * b = bits of S-1; DESCENDING key order == the reference's `sorted(scores, reverse=True)` over
 * (score, "img,lbl,spx", id) tuples (active_selection/base.py:37).  img_rank[i] = rank of image i's
 * joined path string in ascending order.  valid (u8 [n_img*S], may be NULL) marks the ids still in
 * pool_set.suppix (active_selection/my_bvsb.py:41-46); other regions get key 0. */
int mas_region_keys(const float* score, const uint8_t* valid, const int32_t* img_rank, int64_t n_img, int S,
                    uint64_t* keys /* [n_img*S] */, void* stream);

/* bytes of caller-owned scratch needed by mas_sort_keys_desc / mas_budget_walk for n keys (host call) */
size_t mas_select_workspace_bytes(int64_t n);

/* keys_out = keys_in sorted descending (device radix sort; keys_in and keys_out must not overlap). */
int mas_sort_keys_desc(const uint64_t* keys_in, int64_t n, uint64_t* keys_out, void* workspace, size_t ws_bytes,
                       void* stream);

/* The walk of RegionActiveDataset.expand_training_set (dataloader/region_active_dataset.py:31-73) over
 * descending keys: cost_i = region_cost[img*S+id] (= multi_hot_cls[img,id].sum() under fair counting +
 * or-labeling; NULL -> 1);
 * the region that makes the running cost exceed `budget` is the last one taken.  Writes the number of
 * taken regions to *n_selected (device) and decodes the first max_out keys (entries past n_selected
 * are -1).  img_of_rank inverts img_rank. */
int mas_budget_walk(const uint64_t* sorted_keys, int64_t n, const uint8_t* region_cost, const int32_t* img_of_rank,
                    int S, int64_t budget, int64_t max_out, int64_t* n_selected,
                    int32_t* sel_img, int32_t* sel_id, float* sel_score, void* workspace, size_t ws_bytes, void* stream);

/* score[r] <- (dominant[r] == ban_class ? 0 : score[r]) * (cls_w ? cls_w[dominant[r]] : 1), in place.
 * The post-normalisation ban and the region-level class balancing of
 * active_selection/my_bvsb_banignore.py:58-61 and my_bvsb_clsbal_v2_banignore.py:60-74 (ban_class < 0: no ban). */
int mas_region_reweight(float* score, const int32_t* dominant, int64_t n, int ban_class, const float* cls_w /* [C] or NULL */,
                        void* stream);

/* counts[c] += number of regions whose dominant class is c (my_bvsb_clsbal_v2.py:65-66: est_label_dist). */
int mas_dominant_hist(const int32_t* dominant, int64_t n, int C, uint64_t* counts /* [C] */, void* stream);

/* In-place min-max normalisation of the plain BvSB selector (active_selection/my_bvsb.py:79-81):
 *   u <- (u - min(u[u != 0])) / max(u - min(u[u != 0]))     over all n region scores.
 * scratch2: 2 x uint32 of caller-owned device scratch. */
int mas_minmax_normalize(float* scores, int64_t n, uint32_t* scratch2, void* stream);

/* =============================================================================================
 * mIoU counters
 * ============================================================================================= */

/* counts layout (uint64, caller-zeroed, accumulated across calls):
 *   [0,C) seen, [C,2C) correct, [2C,3C) positive  over pixels with target != ignore_label
 *   [3C] [3C+1] [3C+2]  seen / correct / positive of the "undefined" class (IoUIgnore).
 * outputs      int64 [n]: predicted labels for the C classes (MeanIoU._after_step, utils/miou.py:23-38)
 * outputs_all  int64 [n] or NULL: predictions including the extra class, value C = "undefined"
 *              (IoUIgnore._after_step, utils/miou_evalignore.py:20-32). */
int mas_iou_counts(const int64_t* outputs, const int64_t* outputs_all, const int64_t* targets, int64_t n,
                   int num_classes, int64_t ignore_label, uint64_t* counts /* [3C+3] */, void* stream);

/* Fused form for the evaluation loop (trainer/active_joint_multi_predignore.py:196-203): one read of
 * logits [B,channels,H,W] f32 (channels = num_classes or num_classes+1) yields argmax over the first
 * num_classes channels, argmax over all channels and every counter above; first maximum wins. */
int mas_logits_iou_counts(const float* z, const int64_t* targets, int B, int channels, int H, int W, int num_classes,
                          int64_t ignore_label, uint64_t* counts /* [3C+3] */, void* stream);

/* =============================================================================================
 * Single-pass acquisition scan (one read of the logits, one model forward per pool image)
 * ============================================================================================= */

/* Fuses mas_class_prob_sum and mas_bvsb_region_accum: the class weight depends only on the pixel's arg-max
 * class, so it factors out of the region sum.  Adds into
 *   prob_sum [B,C]      as mas_class_prob_sum,
 *   class_sum[B,S,C]    fixed-point (40 fractional bits) sum of the UNWEIGHTED margins of the pixels of region s
 *                       whose arg-max class is c,
 *   hist     [B,S,C]    pixel counts (as mas_bvsb_region_accum).
 * The caller zeroes all three.  Replaces both loops of my_bvsb_predclsbal_pwr_banignore.py:35-72. */
int mas_single_pass_accum(const float* z, const void* spx, int spx_dtype, int B, int C, int H, int W, int S, float invT,
                          uint64_t* prob_sum, uint64_t* class_sum, uint32_t* hist, void* stream);

/* K8 form of the scan: zq [B,C,h,w] are the model's QUARTER-resolution cosine logits; their x4 bilinear upsampling to H x W
 * (models/segmentation/utils.py:25, F.interpolate(..., 'bilinear', align_corners=False)) is evaluated in registers, in the
 * operation order of mas_upsample_bilinear_fwd, so prob_sum / class_sum / hist equal those of mas_single_pass_accum on the
 * materialised tensor bit for bit -- without that tensor (671 MB per Cityscapes batch) or the pass that writes it.
 * C in {19, 20, 21}; H / h and W / w >= ~3.8 (MAS_ERR_RANGE otherwise). */
int mas_single_pass_accum_lowres(const float* zq, int h, int w, const void* spx, int spx_dtype, int B, int C, int H, int W, int S,
                                 float invT, uint64_t* prob_sum, uint64_t* class_sum, uint32_t* hist, void* stream);
/* the same call with per-call option bits (tests / A-B measurements; results are bit-identical with every combination):
 * MAS_LOWRES_GENERIC = keep the generic tap reads at the exact x4 ratio too.  flags = 0 is mas_single_pass_accum_lowres. */
#define MAS_LOWRES_GENERIC 1u
int mas_single_pass_accum_lowres_opt(const float* zq, int h, int w, const void* spx, int spx_dtype, int B, int C, int H, int W, int S,
                                     float invT, uint64_t* prob_sum, uint64_t* class_sum, uint32_t* hist, unsigned flags, void* stream);

/* score[r] = floor(((sum_c class_sum[r,c] * w31[c]) >> 31) / n_r) * 2^-40, w31[c] = floor(cls_weight[c] * 2^31)
 * (exact integer arithmetic); dominant class, ban and optional outputs as mas_region_finalize.
 * With w31[c] = 2^31 for all c the scores equal those of mas_bvsb_region_accum(cls_w = NULL) + mas_region_finalize
 * bit for bit. */
int mas_region_finalize_weighted(const uint64_t* class_sum, const uint32_t* hist, int64_t n_regions, int C,
                                 const uint32_t* w31 /* [C] */, int ban_class, float* score, int32_t* dominant,
                                 uint32_t* count, int64_t* hist_i64, void* stream);

/* Class weights of the PixBal selectors from the per-picture class-probability sums, on the device:
 *   cum[c]   = (sum over reference batches b of mean_{pictures of b, pixels} p_c) / n_batches       (f64)
 *   cls_w[c] = (float)(1 / (coeff * cum[c] + 1)^2),   w31[c] = floor(cls_w[c] * 2^31)  (w31 may be NULL)
 * prob_sum [n_img, C] as mas_class_prob_sum / mas_single_pass_accum wrote it (23 fractional bits); batch b holds
 * pictures [b*batch_size, min((b+1)*batch_size, n_img)); a batch without pictures contributes nothing but counts in
 * n_batches.  Batch means are added in batch order.  Replaces my_bvsb_predclsbal_pwr_banignore.py:42,45,47
 * (cumulated_pred_prob / len(loader); cls_weight = 1 / (coeff * cum + 1)^2) without a host round trip. */
int mas_class_weight(const uint64_t* prob_sum, int n_img, int C, int64_t hw, int batch_size, int n_batches, double coeff,
                     double* cum, float* cls_w, uint32_t* w31, void* stream);

/* =============================================================================================
 * K9  stage-2 cosine pseudo labels with one-ring propagation
 * (trainer/eval_save_cosplbl_prop.py:121-314, ..._includeonehot.py); one image per call.
 * feat [Ch,fh,fw] is the L2-normalised feature map of feat_forward BEFORE its bilinear upsampling (fh x fw may
 * equal H x W); every kernel interpolates to pixel (y,x) of the H x W grid on the fly (align_corners=False).
 * Prototypes are listed in (superpixel, class) order: proto_start [S+1] prefix offsets, proto_cls [n], proto_pix [n]
 * (arg-max pixels taken from the gmax table of mas_partial_loss_fwd run with invT = 1 and the group flags).
 * ============================================================================================= */

/* P [n_proto, Ch] = features at the prototype pixels (:197-199). */
int mas_stage2_gather_protos(const float* feat, int Ch, int fh, int fw, int H, int W, const int32_t* proto_pix, int n_proto,
                             float* P, void* stream);

/* For every selected pixel whose superpixel owns prototypes: nn_proto = index of the most similar prototype of that
 * superpixel (first maximum), nn_sim = that similarity; -1 / 0 elsewhere (:203-232). */
int mas_stage2_assign(const float* feat, int Ch, int fh, int fw, int H, int W, const int64_t* spx, const uint8_t* mask, int S,
                      const int32_t* proto_start, const float* P, int32_t* nn_proto /* [H*W] */, float* nn_sim /* [H*W] */,
                      void* stream);

/* adj [S, ceil(S/32)] (caller-zeroed bit matrix): bit g of row t set iff superpixel g owns prototypes and some pixel
 * of g lies in the 3x3 neighbourhood of a pixel of t -- the binary_dilation + unique of :259-266, for all superpixels. */
int mas_stage2_adjacency(const int64_t* spx, int H, int W, int S, const int32_t* proto_start, uint32_t* adj, void* stream);

/* out [H*W] int64: 255, overwritten in ascending id order by every adjacent valid superpixel whose prototypes accept
 * the pixel (some similarity above that prototype's threshold thr[j]; label = class of the most similar prototype),
 * finally by the pixel's own nearest prototype (:272-311). */
int mas_stage2_propagate(const float* feat, int Ch, int fh, int fw, int H, int W, const int64_t* spx, int S, const uint32_t* adj,
                         const int32_t* proto_start, const int32_t* proto_cls, const float* P, const float* thr,
                         const int32_t* nn_proto, int64_t* out, void* stream);

/* =============================================================================================
 * K7  ASPP: the three dilated depthwise 3x3 convolutions from one read of the feature map
 * (models/segmentation/deeplabv3.py:168-201,216-245 after convert_to_separable_conv :249-261; dilations 6/12/18 at
 * output stride 16).  x, y_d, dy_d, dx are [N,C,H,W] f32 contiguous; w_d, dw_d are [C,1,3,3]; zero padding = dilation,
 * stride 1, no bias.  The pointwise 1x1 convolutions that follow stay GEMMs (MFMA).
 * ============================================================================================= */
int mas_aspp_dw3_fwd(const float* x, const float* w0, const float* w1, const float* w2, int N, int C, int H, int W,
                     int d0, int d1, int d2, float* y0, float* y1, float* y2, void* stream);
int mas_aspp_dw3_bwd_x(const float* dy0, const float* dy1, const float* dy2, const float* w0, const float* w1, const float* w2,
                       int N, int C, int H, int W, int d0, int d1, int d2, float* dx, void* stream);
/* deterministic (fixed reduction order): one workgroup per channel */
int mas_aspp_dw3_bwd_w(const float* x, const float* dy0, const float* dy1, const float* dy2, int N, int C, int H, int W,
                       int d0, int d1, int d2, float* dw0, float* dw1, float* dw2, void* stream);

/* Single depthwise 3x3, stride 1, padding = dilation, no bias: y[n,c] = w[c] (*) x[n,c].  Replaces the depthwise half
 * of AtrousSeparableConvolution (models/segmentation/deeplabv3.py:168-192) in the decoder (classifier.classifier.0/3:
 * 304 and 256 channels), forward / data gradient / weight gradient.  (16 + 2d) * (W + 2d) * 4 bytes must fit 64 KB.
 * bwd_w: `partial` is caller-owned scratch of N*C*9 floats; the reduction order is fixed (deterministic). */
int mas_depthwise3x3_fwd(const float* x, const float* w, int N, int C, int H, int W, int dilation, float* y, void* stream);
int mas_depthwise3x3_bwd_x(const float* dy, const float* w, int N, int C, int H, int W, int dilation, float* dx, void* stream);
int mas_depthwise3x3_bwd_w(const float* x, const float* dy, int N, int C, int H, int W, int dilation, float* partial, float* dw,
                           void* stream);

/* Training-time geometry of ONE sample on the device: dataloader/transform.py:105-113 = ExtRandomScale (Pillow BILINEAR
 * for the picture, NEAREST for up to two label / superpixel maps; ext_transforms.py:172-192), pad-if-needed + random crop
 * (:443-520), horizontal flip (:323-341), to-tensor + normalise (:384-437).  `mean`, `std` (3 floats) and `fill` (3
 * bytes) are HOST pointers, everything else device memory.
 * img u8 [H,W,3]; scaled size th x tw; hbounds[tw,2] / hk[tw,hks] and vbounds[th,2] / vk[th,vks]: Pillow's
 * (first tap, tap count) and 22-bit fixed-point weights of the horizontal / vertical pass; xidx[tw], yidx[th]: source
 * index of the nearest-neighbour resize; gap_*: padding on each side; (crop_i, crop_j): crop origin in the padded
 * image; out_img f32 [3,out_h,out_w]; maps: MAS_ID_* / MAS_MAP_U8 in, int64 (or uint8 when out*_u8) out,
 * pad value outside the scaled image.  map pointers may be NULL. */
int mas_train_augment(const uint8_t* img, int H, int W, int th, int tw, const int32_t* hbounds, const int32_t* hk, int hks,
                      const int32_t* vbounds, const int32_t* vk, int vks, const int32_t* xidx, const int32_t* yidx, int gap_y,
                      int gap_x, int crop_i, int crop_j, int flip, int out_h, int out_w, const float* mean, const float* std,
                      const uint8_t* fill, const void* map0, int map0_dtype, int64_t pad0, void* out_map0, int out0_u8,
                      const void* map1, int map1_dtype, int64_t pad1, void* out_map1, int out1_u8, float* out_img, void* stream);

/* K8 (part): F.interpolate(mode='bilinear', align_corners=False) of x [NC,Hi,Wi] -> y [NC,Ho,Wo]
 * (models/segmentation/utils.py:25, deeplabv3.py:116) and its backward as a deterministic gather (no atomics).
 * bwd requires Wo <= 6 * Wi. */
int mas_upsample_bilinear_fwd(const float* x, int64_t NC, int Hi, int Wi, int Ho, int Wo, float* y, void* stream);
int mas_upsample_bilinear_bwd(const float* gy, int64_t NC, int Hi, int Wi, int Ho, int Wo, float* gx, void* stream);

/* BatchNorm2d fused with the following ReLU and residual add (models/segmentation/backbone/resnet.py:143-160 Bottleneck,
 * the conv -> bn -> relu triples of the stem / ASPP / decoder, deeplabv3.py:93-110,216-245).  x, y, residual: [N,C,HW]
 * f32 NCHW; gamma / beta may be NULL (affine=False).  Deterministic reductions (fixed order, double).
 * train_fwd: batch statistics (biased variance for the normalisation, unbiased for running_var, momentum update as
 * torch.nn.BatchNorm2d; running_* / num_batches_tracked may be NULL), y = relu?((x-mean)*invstd*gamma+beta+residual);
 * saves mean / invstd [C] for the backward.  workspace: mas_bn_workspace_bytes(N, C, HW) bytes of device memory.
 * relu_mask (optional, mas_bn_mask_bytes(N, C, HW) bytes): one byte per 16-byte-aligned group of four outputs of a plane
 * ((HW + 3) / 4 + 1 groups per plane, whatever the plane's alignment), bit k = output k of the group is positive;
 * train_bwd: dx, dresidual (= masked dy; may be NULL), dgamma, dbeta (may be NULL); the ReLU mask comes from `relu_mask`
 * when given, else from the forward output `y`.
 * eval_fwd: the same map with the running statistics.
 * counters (train_fwd / train_bwd; may be NULL): C zeroed 32-bit words in device memory that STAY zeroed between calls (the caller
 * keeps one such array per stream).  With it the workgroup that writes a channel's last partial sums turns them into the channel's
 * statistics (forward: mean / invstd / running statistics; backward: dgamma / dbeta / the two means) -- two launches instead of
 * three, same sums in the same order, same bits. */
int64_t mas_bn_workspace_bytes(int N, int C, int HW);
int64_t mas_bn_mask_bytes(int N, int C, int HW);
int mas_bn_act_train_fwd(const float* x, const float* gamma, const float* beta, const float* residual, int N, int C, int HW, float eps,
                         float momentum, int relu, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                         float* save_mean, float* save_invstd, void* workspace, float* y, uint8_t* relu_mask, uint32_t* counters, void* stream);
/* train_fwd with the partial sums formed by the producer of x (mas_conv_sk_stats): partials [C][per_channel] pairs of doubles */
int mas_bn_act_train_fwd_stats(const float* x, const double* partials, int per_channel, const float* gamma, const float* beta,
                               const float* residual, int N, int C, int HW, float eps, float momentum, int relu, float* running_mean,
                               float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_invstd, float* y,
                               uint8_t* relu_mask, void* stream);
int mas_bn_act_eval_fwd(const float* x, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        const float* residual, int N, int C, int HW, float eps, int relu, float* y, void* stream);
int mas_bn_act_train_bwd(const float* dy, const float* x, const float* y, const uint8_t* relu_mask, const float* gamma,
                         const float* save_mean, const float* save_invstd, int N, int C, int HW, int relu, void* workspace, float* dx,
                         float* dresidual, float* dgamma, float* dbeta, uint32_t* counters, void* stream);

/* K8: cosine classifier of DeepLabHeadV3PlusWN (models/segmentation/deeplabv3.py:121-124):
 * logits[n,k,p] = <feat[n,:,p], proxy_hat[k,:]> / max(|feat[n,:,p]|, eps) for unit-norm proxies proxy_hat [K,Ch]
 * (K in {19, 20, 21}); inv_norm [N,HW] is saved for the backward.  bwd: dfeat only (the proxy gradient is a GEMM). */
int mas_cosine_head_fwd(const float* feat, const float* proxy_hat, int N, int Ch, int K, int HW, float eps, float* logits,
                        float* inv_norm, void* stream);
int mas_cosine_head_bwd(const float* feat, const float* proxy_hat, const float* logits, const float* inv_norm, const float* dlogits,
                        int N, int Ch, int K, int HW, float* dfeat, void* stream);

/* The 1x1 convolution of the ASPP image-pooling branch on its 1 x 1 map (models/segmentation/deeplabv3.py:194-207:
 * AdaptiveAvgPool2d(1) -> Conv2d(2048, 256, 1) -> BatchNorm -> ReLU) and its backward: y[n,m] = sum_k x[n,k] w[m,k] for a handful of
 * rows (x [N,K], w [M,K] = the convolution weight, y [N,M]); dx [N,K] and / or dw [M,K] (NULL = not wanted) from dy [N,M].  Fixed
 * summation order: run-to-run identical (vendor GEMMs pick atomic split-K solutions for this shape). */
int mas_dense_small_fwd(const float* x, const float* w, int N, int K, int M, float* y, void* stream);
int mas_dense_small_bwd(const float* dy, const float* x, const float* w, int N, int K, int M, float* dx, float* dw, void* stream);

/* MaxPool2d(kernel 3, stride 2, padding 1) of x [NC,H,W] (models/segmentation/backbone/resnet.py:171,206):
 * y [NC,Ho,Wo] with Ho = (H - 1) / 2 + 1, and a one-byte arg-max offset (0..8 inside the window, first maximum) per
 * output; bwd: dx [NC,H,W] gathered from dy through `arg` in a fixed order (no atomics). */
int mas_maxpool3s2_fwd(const float* x, int64_t NC, int H, int W, float* y, uint8_t* arg, void* stream);
int mas_maxpool3s2_bwd(const float* dy, const uint8_t* arg, int64_t NC, int H, int W, float* dx, void* stream);

/* 1x1 convolution y[n,m,p] = sum_k w[m,k] x[n,k,p] for the small-K layers (models/segmentation/backbone/resnet.py:129-160:
 * conv1 / conv3 / downsample of layer1, the stem's neighbours), x [N,K,HW], y [N,M,HW], HW % 4 == 0, M % 32 == 0;
 * `w_t` is the TRANSPOSED weight [K,M].  With scale / shift (both or neither) the inference BatchNorm
 * (y * scale[m] + shift[m]), the residual add and the ReLU that follow the convolution are applied in the epilogue. */
int mas_conv1x1_fwd(const float* x, const float* w_t, int N, int K, int M, int HW, const float* scale, const float* shift,
                    const float* residual, int relu, float* y, void* stream);

/* Dense convolution on the f32 matrix cores (v_mfma_f32_32x32x2_f32; exact f32 products, k-ordered accumulation),
 * NCHW: y[n,m,oy,ox] = sum_{c,r,s} w[m,c,r,s] * x[n,c, oy*stride + (r-1)*dil, ox*stride + (s-1)*dil]   (ksize 3, padding = dil)
 *       y[n,m,oy,ox] = sum_c w[m,c] * x[n,c, oy*stride, ox*stride]                                     (ksize 1)
 * with Ho = (H-1)/stride + 1.  `wt` is the re-arranged weight [Cin/CK][KC/8][2][Cout][4], KC = ksize*ksize*CK,
 * CK = mas_conv_chunk(ksize, Cin) (0 = unsupported channel count): element (chunk, q, h, m, j) is
 * w[m, chunk*CK + 2*cp + h, tap] with k-step kk = 4*q + j, tap = kk / (CK/2), cp = kk % (CK/2); the M extent of `wt` is
 * Cout rounded up to a multiple of 64 (zero rows), y / residual / scale / shift have Cout channels.  With scale / shift (both or neither)
 * the epilogue applies y*scale[m] + shift[m] (inference BatchNorm), then `residual` (NULL or [N,Cout,Ho,Wo]) is added
 * and, if `relu`, max(.,0) taken -- the conv-bn-relu / conv-bn-add-relu triples of
 * models/segmentation/backbone/resnet.py:143-160 and the 1x1 projections of deeplabv3.py:85-137,216-245 in one kernel. */
int mas_conv_chunk(int ksize, int Cin);
int mas_conv_fwd(const float* x, const float* wt, int N, int Cin, int H, int W, int Cout, int ksize, int stride, int dil,
                 const float* scale, const float* shift, const float* residual, int relu, float* y, void* stream);

/* The same convolutions (mas_conv_fwd: formulas, epilogue, reference lines) on the bf16 matrix cores with f32 operands and f32
 * results (csrc/conv_bx.hip): every f32 operand is split exactly into three bf16 terms (8 + 8 + 8 significand bits) and a
 * product is accumulated in f32 from its six partial products of order <= 2 (v_mfma_f32_32x32x16_bf16); the dropped terms are
 * below 2^-23 of the product, i.e. one f32 rounding -- same error bound as the f32 MFMA form, exact on integer data, 2.67x its
 * matrix peak.  Supported (mas_conv_bx_supported != 0): ksize 1 at stride 1 (any plane) or stride 2 (H even, W % 8 == 0, x 16-byte
 * aligned); ksize 3 with stride 1, dil 1 | 2 (padding = dil), W >= 32; ksize 3 with stride 2, dil 1 (padding 1: conv2 of
 * layer2.0 / layer3.0, resnet.py:140-150), Cin % 32 == 0, H even, W % 8 == 0, x 16-byte aligned -- computed as nine shifted 1x1
 * stride-2 products into one accumulator set, `wp` = the image of role 2; any Cin otherwise (a last partial chunk is
 * zero-padded), any Cout.
 * `wp` is the weight [Cout][Cin][ksize][ksize] split and laid out ONCE (per checkpoint load) by mas_conv_bx_pack into
 * mas_conv_bx_packed_bytes(ksize, Cin, Cout) bytes of caller-owned, 16-byte aligned device memory: the sequence of LDS images
 * [M tile][chunk][term h|m|l][k group][BM rows][8 bf16] (1x1: chunk = 32 channels, k = channel; 3x3: chunk = 8 channels,
 * k group = tap, + one zero tap; BM = 128 if ksize == 1 and Cout % 128 == 0, else 64). */
int mas_conv_bx_supported(int ksize, int stride, int dil, int Cin, int Cout, int H, int W);
long long mas_conv_bx_packed_bytes(int ksize, int Cin, int Cout, int role);
int mas_conv_bx_pack(const float* w, const float* row_scale, int Cout, int Cin, int ksize, int role, void* wp, void* stream);
/* role 0: the image of the forward product.  role 2 (ksize 3, Cin % 32 == 0): the forward image of the STRIDE-2 3x3 convolution --
 * 9 x (Cin / 32) chunks of the 1x1 form in tap-major order [M tile][tap * (Cin / 32) + channel chunk][term][k group][BM][8]
 * (BM as for ksize 1; `row_scale` allowed).  role 1: the image with which mas_conv_bx_fwd computes the INPUT GRADIENT of the
 * stride-1 convolution (the backward of the nn.Conv2d calls above inside trainer/active_joint_multi_predignore_lossdecomp.py:83-116):
 * channel axes swapped, taps mirrored -- call mas_conv_bx_fwd(dY, wp1, N, Cout, H, W, Cin, ksize, 1, dil, NULL, NULL, residual, 0, dX):
 * `residual` adds the gradient of x's other consumer.  In training the weights move every optimizer step: all images of a model
 * are rewritten by ONE launch from a device-resident job table (mas_conv_bx_pack_job fills a host record of
 * mas_conv_bx_pack_job_bytes() bytes and returns the job's block count, 0 if it rejects the arguments; records in ascending
 * first_block order; mas_conv_bx_pack_multi runs `njobs` records covering `nblocks` blocks). */
/* `row_scale` (role 0; NULL or [Cout]): the image holds w[m,:,:,:] * row_scale[m] -- an inference BatchNorm's scale folded into the
 * weight, which is what lets two convolutions share one accumulator:
 * mas_conv_bx_fwd_dual:  y = relu?( conv1x1(x1, w1') + conv1x1(x2, w2') + shift[m] ),  x1 [N,Cin1,H,W], x2 [N,Cin2,H,W] on the same
 * plane, both 1x1 stride 1 -- `bn3(conv3(out)) + downsample(x)` of the first Bottleneck of a stage whose downsample has stride 1
 * (models/segmentation/backbone/resnet.py:143-160: layer1.0, and layer4.0 at output stride 16) in ONE kernel: the identity branch
 * is never written to memory and read back.  w1' / w2' = the images packed with the two BatchNorm scales, shift = the sum of the
 * two BatchNorm shifts. */
int mas_conv_bx_fwd_dual(const float* x1, const void* wp1, int Cin1, const float* x2, const void* wp2, int Cin2, int N, int H, int W,
                         int Cout, const float* shift, int relu, float* y, void* stream);
size_t mas_conv_bx_pack_job_bytes(void);
unsigned mas_conv_bx_pack_job(void* job_host, const float* w, int Cout, int Cin, int ksize, int role, void* wp, unsigned first_block);
int mas_conv_bx_pack_multi(const void* jobs_dev, int njobs, unsigned nblocks, void* stream);
int mas_conv_bx_fwd(const float* x, const void* wp, int N, int Cin, int H, int W, int Cout, int ksize, int stride, int dil,
                    const float* scale, const float* shift, const float* residual, int relu, float* y, void* stream);
/* Presplit activations ("bx3"): x [N,C,H,W] f32 as [N][ceil(C/8)][3 terms][H*W][8 bf16] -- one 16-byte unit = one term (h | m | l of
 * csrc/bx_split.h) of 8 consecutive channels of one pixel, channels beyond C zero: exactly a unit of the LDS operand image of
 * mas_conv_bx_fwd, so a consumer stages it with 16-byte copies and no VALU work (the split is then done once per element by the
 * producer instead of once per M tile by every consumer; 6 instead of 4 bytes per element in HBM).  mas_bx3_split: the conversion as
 * a pass of its own; mas_conv_bx_fwd_pre: mas_conv_bx_fwd (stride 1) on such an input -- same products, same bits.
 * (models/segmentation/backbone/resnet.py:143-160: conv2 / conv3 of a Bottleneck read what conv1 / conv2 wrote.) */
long long mas_bx3_bytes(int N, int C, int H, int W);
int mas_bx3_split(const float* x, int N, int C, int H, int W, void* x3, void* stream);
int mas_conv_bx_fwd_pre(const void* x3, const void* wp, int N, int Cin, int H, int W, int Cout, int ksize, int dil, const float* scale,
                        const float* shift, const float* residual, int relu, float* y, void* stream);
/* The bare stride-1 products of a TRAINING step (forward with a role-0 image, input gradient with a role-1 image + the gradient of the
 * input's other consumer as `residual`; models/segmentation/backbone/resnet.py:143-160 under
 * trainer/active_joint_multi_predignore_lossdecomp.py:83-116) with a work-splitting plan for launches with fewer workgroups than
 * the chip has slots (the 48 x 48 planes of layer3 / layer4 / ASPP at the training crop): the K chunks of every tile are dealt to
 * `ksplit` workgroups, part 0 stores into y (+ residual), the others into `workspace`, and a second launch adds the parts in index
 * order (run-to-run identical; no flags, no waiting: nothing that needs co-residency, unlike the stream-K hand-off of mas_conv_sk).
 * 3x3: `tile_w` 32 (8 x 32 output pixels per tile), 16 (16 x 16: no padded quarter on 48 x 48 planes) or 1 ("flat": 256 consecutive
 * pixels of the plane in row-major order over a patch of the full rows they touch -- narrow planes whose width is no multiple of 16:
 * the 49 x 49 planes of the 769 crop are 10 such tiles against 14 / 16).
 * mas_conv_bx_train_plan: out3 = {ksplit, tile_w, workgroups} the library would choose; ksplit / tile_w <= 0 in mas_conv_bx_train
 * take the plan's.  workspace: mas_conv_bx_train_workspace_bytes(N, Cout, H, W, ksplit) bytes, 16-byte aligned (none for ksplit 1). */
int mas_conv_bx_train_plan(int N, int Cin, int H, int W, int Cout, int ksize, int dil, int* out3);
size_t mas_conv_bx_train_workspace_bytes(int N, int Cout, int H, int W, int ksplit);
int mas_conv_bx_train(const float* x, const void* wp, int N, int Cin, int H, int W, int Cout, int ksize, int dil, const float* residual,
                      float* y, int ksplit, int tile_w, void* workspace, size_t workspace_bytes, double* stats, void* stream);
/* `stats` (forward products without residual; may be NULL): [Cout][slots][2] doubles, slots = mas_conv_bx_train_stat_slots(...) > 0 --
 * per output channel the sums (sum y, sum y^2) of the stored outputs over disjoint pixel sets: with ksplit 1 formed in the kernel's
 * epilogue from the accumulators (one slot per pixel tile and wave column; cross-lane halving with v_permlane16_swap / DPP adds),
 * with ksplit > 1 by the reduction pass (one slot per picture and 4096-element chunk of a plane; H * W % 4 == 0, else 0 slots): the
 * BatchNorm partial sums of `bn(conv(x))` (resnet.py:143-160) without a reduction pass of their own over y; feed them to
 * mas_bn_act_train_fwd_stats.  Every slot of every channel is written. */
int mas_conv_bx_train_stat_slots(int N, int H, int W, int Cout, int ksize, int dil, int ksplit, int tile_w);

/* Weight gradient of a 1x1 stride-1 convolution on the bf16 matrix cores with f32 operands and results (csrc/conv_wgrad_bx.hip; the
 * operand split of mas_conv_bx_fwd applied to BOTH operands): dW[m,c] = sum_{n,p} dY[n,m,p] * X[n,c,p], x [N,Cin,H,W], dy [N,Cout,H,W],
 * dw [Cout,Cin] -- the backward of the 1x1 nn.Conv2d layers (mas_conv_wgrad: reference lines).  Split K over the pixels with a
 * fixed-order reduction through `workspace` (mas_conv_wgrad_bx_workspace_bytes(Cin, Cout), caller-owned): run-to-run identical.
 * Supported (mas_conv_wgrad_bx_supported != 0): any plane (a picture's last, partial 32-pixel chunk is masked), tensors of one
 * picture below 2 GiB. */
int mas_conv_wgrad_bx_supported(int N, int Cin, int H, int W, int Cout);
size_t mas_conv_wgrad_bx_workspace_bytes(int Cin, int Cout);
int mas_conv_wgrad_bx(const float* x, const float* dy, int N, int Cin, int H, int W, int Cout, float* dw, void* workspace,
                      size_t workspace_bytes, void* stream);
/* The same for the 3x3 stride-1 convolutions (padding = dilation = 1 | 2; models/segmentation/backbone/resnet.py:129-171: the deep
 * stem, conv2 of every Bottleneck): dW[m,c,ty,tx] = sum_{n,y,x} dY[n,m,y,x] * X[n,c,y+(ty-1)d,x+(tx-1)d], dw [Cout,Cin,3,3].  With
 * K = pixels a tap is a one-pixel shift of the K axis: the X patch is staged channel-contiguous (as the forward kernel stages it, a
 * tap = a whole-unit offset) and read with gfx950's transposing LDS read (ds_read_b64_tr_b16), which hands the MFMA its K-contiguous
 * fragment.  Workspace: mas_conv_wgrad_bx3_workspace_bytes(Cin, Cout); split-K partial tiles added in index order. */
int mas_conv_wgrad_bx3_supported(int N, int Cin, int H, int W, int Cout, int dil);
size_t mas_conv_wgrad_bx3_workspace_bytes(int Cin, int Cout);
int mas_conv_wgrad_bx3(const float* x, const float* dy, int N, int Cin, int H, int W, int Cout, int dil, float* dw, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Forward and input gradient of a dense convolution in training, as a persistent stream-K implicit GEMM on the f32 matrix
 * cores (csrc/conv_sk.hip).  `wp` is the weight as mas_conv_sk_pack writes it from PyTorch's [Cout][Cin][ksize][ksize] tensor
 * (mas_conv_sk_packed_elems floats, 16-byte aligned; one image per role: dgrad 0 / 1; one small launch per optimizer step): the
 * sequence of LDS images [M tile][K chunk][KC / 8][2][BM][4] the kernel copies linearly.  The nn.Conv2d forward / backward-input of models/segmentation/backbone/resnet.py:129-171 and
 * models/segmentation/deeplabv3.py:85-137,168-245 inside trainer/active_joint_multi_predignore_lossdecomp.py:83-116.
 *   dgrad = 0:  y[n,m,oy,ox] = sum_{c,r,s} w[m,c,r,s] x[n,c, oy*stride + r*dil - pad, ox*stride + s*dil - pad]   x [N,Cin,H,W], y [N,Cout,Ho,Wo]
 *   dgrad = 1:  y[n,c,iy,ix] = sum_{m,r,s} w[m,c,r,s] x[n,m, iy - r*dil + pad, ix - s*dil + pad]  (stride 1)       x [N,Cout,H,W], y [N,Cin,H,W]
 * pad = dil (ksize 3) / 0 (ksize 1); ksize 1 | 3; stride 1 | 2 (stride 2: forward only, dil 1); dil 1 | 2 | 4 (ksize 3).  Epilogue as
 * mas_conv_fwd: y*scale[m] + shift[m] (both or neither), + residual (same shape as y, may be NULL), ReLU if `relu`.
 * One workgroup per CU; the (tile, K-chunk) iterations of the layer are dealt to the workgroups in equal runs, tiles that
 * straddle two workgroups are combined through `workspace` (mas_conv_sk_workspace_bytes(), zero-filled ONCE by the caller, then
 * owned by launches of ONE stream) in a fixed order: run-to-run identical results.  `epoch` must be non-zero and differ from the
 * epoch of the previous launch on the same workspace.
 * Co-residency contract: a workgroup that CONTRIBUTES to a split tile publishes its part before it does anything else and never
 * waits; a workgroup that FINISHES a split tile waits, at the end of its work, for contributors with a higher index only.  So the
 * launch completes whenever its workgroups are eventually scheduled, in any order and beside any other kernel (RCCL's all-reduce
 * under DistributedDataParallel, a second stream); nothing requires all workgroups to be resident at once.  The wait is bounded
 * all the same (mas_sk_opts.spin_limit polls per contributor, default 2^22 = seconds): a finisher that gives up sets the
 * workspace's error word (sticky; mas_conv_sk_error copies it to the host, trainers read it with the loss) and writes NaN into its
 * tile -- a launch that gave up never hands out plausible numbers.  MAS_SK_NOSPLIT deals whole tiles only (no hand-off, hence no
 * wait at all; the tail of the layer then runs on fewer CUs): the plan to re-run a launch on after a give-up.
 * All options are per call (mas_sk_opts, NULL = defaults); the library keeps no mode state. */
#define MAS_SK_DMA 1u       /* stage the K chunks with an LDS-DMA ring (global_load_lds into 2..4 LDS buffers, up to three chunks in
                             * flight, no staging registers) instead of global -> registers -> LDS: measured 3-5 % slower on the
                             * training shapes, kept for A/B runs */
#define MAS_SK_NOSPLIT 2u   /* whole tiles only: no tile is split between workgroups */
typedef struct mas_sk_opts {
    unsigned flags;         /* MAS_SK_* */
    unsigned spin_limit;    /* polls a finisher waits per contributor before it gives up; 0 = default */
    void* stamps;           /* tools: device buffer [512][4] uint64 of per-workgroup wall-clock stamps (100 MHz; start, pipeline
                             * primed, last tile done, end), indexed by the logical workgroup; NULL = off */
} mas_sk_opts;
size_t mas_conv_sk_workspace_bytes(void);
size_t mas_conv_sk_packed_elems(int Cin, int Cout, int ksize, int stride, int dgrad);
int mas_conv_sk_pack(const float* w, int Cin, int Cout, int ksize, int stride, int dgrad, float* out, void* stream);
/* every weight of a network in one launch: the caller fills `njobs` records of mas_conv_sk_pack_job_bytes() bytes each in host
 * memory with mas_conv_sk_pack_job (which returns the job's block count; first_block = the sum of the counts before it), copies
 * the table to the device once -- the tensors are updated in place by the optimizer, so the table stays valid -- and calls
 * mas_conv_sk_pack_multi(table, njobs, total blocks) after every optimizer step. */
size_t mas_conv_sk_pack_job_bytes(void);
unsigned mas_conv_sk_pack_job(void* job_host, const float* w, int Cin, int Cout, int ksize, int stride, int dgrad, float* out,
                              unsigned first_block);
int mas_conv_sk_pack_multi(const void* jobs_dev, int njobs, unsigned nblocks, void* stream);
int mas_conv_sk(const float* x, const float* wp, int N, int Cin, int H, int W, int Cout, int ksize, int stride, int dil, int dgrad,
                const float* scale, const float* shift, const float* residual, int relu, float* y, void* workspace,
                size_t workspace_bytes, unsigned epoch, const mas_sk_opts* opts, void* stream);
int mas_conv_sk_error(const void* workspace, unsigned* out_host);
/* mas_conv_sk in the forward role without epilogue that also forms the BatchNorm partial sums of its output in the epilogue of
 * every tile (the relu(bn(conv(x))) triples of backbone/resnet.py:143-160 in training mode: removes the reduction pass over y):
 * stats [Cout][mas_conv_sk_stats_slots(...)] pairs of doubles (sum y, sum y^2) over disjoint pixel sets, every entry written;
 * fixed summation order (run-to-run identical).  Consumer: mas_bn_act_train_fwd_stats. */
int mas_conv_sk_stats_slots(int N, int Cin, int H, int W, int Cout, int ksize, int stride, int dil, unsigned flags /* MAS_SK_DMA changes the tiling */);
int mas_conv_sk_stats(const float* x, const float* wp, int N, int Cin, int H, int W, int Cout, int ksize, int stride, int dil, float* y,
                      double* stats, void* workspace, size_t workspace_bytes, unsigned epoch, const mas_sk_opts* opts, void* stream);
/* Input gradient of a 3x3, stride-2, padding-1 convolution (torch.autograd of nn.Conv2d(k=3, stride=2, padding=1), the conv2 of
 * layer2.0 / layer3.0: backbone/resnet.py:129-141), one parity class per launch: sub = 2 py + px writes dx[n, c, 2 i + py, 2 j + px]
 * as a stride-1 product over dy [N,Cout,(H-1)/2+1,(W-1)/2+1] with (1 + py) x (1 + px) taps; wp = mas_conv_sk_pack(..., ksize 3,
 * stride 2, dgrad = 2 + sub).  The four classes write every pixel of dx [N,Cin,H,W] exactly once.  (Class 0 is a one-tap product written to the
 * pixels (2i, 2j): with the input-gradient image of a 1x1 weight (mas_conv_sk_pack(..., ksize 1, dgrad = 1)) and a zero-filled dx it
 * is the input gradient of the 1x1 stride-2 `downsample` convolutions, backbone/resnet.py:215-223.)  Epilogue, workspace and epoch
 * as mas_conv_sk (a fresh epoch per launch). */
int mas_conv_sk_dgrad_s2(const float* dy, const float* wp, int N, int Cin, int H, int W, int Cout, int sub, const float* scale,
                         const float* shift, const float* residual, int relu, float* dx, void* workspace, size_t workspace_bytes,
                         unsigned epoch, const mas_sk_opts* opts, void* stream);

/* Weight gradient of a dense convolution on the f32 matrix cores (csrc/conv_wgrad.hip), NCHW operands as autograd holds them:
 *   dw[m,c,r,s] = sum_{n,oy,ox} dy[n,m,oy,ox] * x[n,c, oy*stride + r*dil - pad, ox*stride + s*dil - pad],  pad = dil (ksize 3) / 0 (ksize 1)
 * -- the backward of the nn.Conv2d modules of models/segmentation/backbone/resnet.py:129-171 and
 * models/segmentation/deeplabv3.py:85-137,168-245 with respect to their weights (what loss.backward() of
 * trainer/active_joint_multi_predignore_lossdecomp.py:83-116 runs through MIOpen in the reference).
 * x [N,Cin,H,W], dy [N,Cout,Ho,Wo] with Ho = (H-1)/stride + 1, dw [Cout,Cin,ksize,ksize] (overwritten).  ksize 1 | 3, stride 1 | 2,
 * dil 1 (ksize 1, stride 2), 1 | 2 | 4 (ksize 3, stride 1).  Any channel counts and plane sizes (16-byte loads when the
 * planes allow them).  Split-K partial sums go through `workspace` (mas_conv_wgrad_workspace_bytes) and are added in a fixed
 * order: run-to-run identical results. */
size_t mas_conv_wgrad_workspace_bytes(int N, int Cin, int H, int W, int Cout, int ksize, int stride, int dil);
/* the launch plan of mas_conv_wgrad for a geometry (host only, for tools): out6 = {BM, BC, pixels per K chunk, K chunks, split S,
 * workgroups} */
int mas_conv_wgrad_plan(int N, int Cin, int H, int W, int Cout, int ksize, int stride, int dil, int* out6);
int mas_conv_wgrad(const float* x, const float* dy, int N, int Cin, int H, int W, int Cout, int ksize, int stride, int dil,
                   float* dw, void* workspace, size_t workspace_bytes, void* stream);

/* The first convolution of the deep stem (models/segmentation/backbone/resnet.py:163-171, conv1[0..2]): x [N,3,H,W] ->
 * y [N,Cout,(H-1)/2+1,(W-1)/2+1], 3x3, stride 2, padding 1, w [Cout,3,3,3] as PyTorch stores it, optional inference BatchNorm
 * (scale / shift, both or neither) and ReLU in the epilogue.  Cout % 16 == 0, W % 8 == 0, 16-byte aligned tensors. */
int mas_stem_conv_fwd(const float* x, const float* w, int N, int H, int W, int Cout, const float* scale, const float* shift, int relu,
                      float* y, void* stream);

/* =============================================================================================
 * AdamW over all parameter tensors of the model in ONE launch -- the optimizer of trainer/base.py:64-66
 * (`optim.AdamW(params=[backbone @ lr, classifier @ cls_lr_scale * lr], weight_decay=wd)`, torch 1.11's single-tensor update:
 * p *= 1 - lr wd;  m = m b1 + g (1 - b1);  v = v b2 + (1 - b2) g g;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps), f32, every
 * operation rounded separately).  One job record per parameter tensor (mas_adamw_job fills a host record of mas_adamw_job_bytes()
 * bytes and returns the job's block count, 0 if it rejects the arguments; `group` < mas_adamw_max_groups() names the tensor's
 * parameter group; records in ascending first_block order; the table is then copied to the device).  `lr_host[ngroups]`: the
 * groups' learning rates of THIS step (host memory; the poly schedule of utils/scheduler.py:5-14 moves them every step).  `step_dev`: device float, the number of steps taken so far -- the launch uses t = *step_dev + 1 and then
 * increments it.  `skip_dev` (NULL or a device float): non-zero leaves parameters, moments and the step count untouched (the
 * stream-K give-up word of mas_conv_sk, trainer/base.py:guard_optimizer_step).  p, g, m, v: f32, n elements each.
 * ============================================================================================= */
size_t mas_adamw_job_bytes(void);
int mas_adamw_max_groups(void);
unsigned mas_adamw_job(void* job_host, float* p, const float* g, float* m, float* v, long long n, int group, unsigned first_block);
int mas_adamw_multi(const void* jobs_dev, int njobs, unsigned nblocks, const float* lr_host, int ngroups, double beta1, double beta2, double eps,
                    double weight_decay, float* step_dev, const float* skip_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MULACTSEG_HIP_H */

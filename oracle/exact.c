/*
 * exact.c -- plain-C CPU restatement of the MulActSeg hot path in the NORMATIVE arithmetic of
 * mulactseg_amd/csrc/detmath.h.
 *
 * ORACLE / TEST INFRASTRUCTURE ONLY: built into oracle/libexact.so by oracle/Makefile and loaded by
 * oracle/exact.py; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 * Nothing under mulactseg_amd/ links or loads it.
 *
 * Role in the parity chain (DESIGN.md "Oracle"):
 *   reference (executed) -> tests/golden/ (npz) -> oracle/port.py   bit-exact (same f32 op order)
 *   oracle/port.py  <->  this file      integers identical, floats within 1e-5 (different exp/log)
 *   this file       <->  HIP kernels    bit-exact, every output (same arithmetic, fixed-point sums)
 *
 * Everything here is sequential scalar C: one pixel at a time, one class at a time.  Each function
 * names the reference lines it restates (paths relative to the reference root).
 * Compile with -ffp-contract=off (see detmath.h).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../mulactseg_amd/csrc/detmath.h"

#define MAXC 64

/* softmax(z * invT) of one pixel; operation order as in csrc/common.h:mas_softmax_regs.
 * On return e[c] are the un-normalised exponentials; the function value is rinv = 1/sum (p_c = e_c * rinv). */
static float softmax_row(const float* z, size_t stride, int C, float invT, float* e) {
    float m = z[0];
    int c;
    for (c = 1; c < C; ++c)
        if (z[(size_t)c * stride] > m) m = z[(size_t)c * stride];
    const float negM = -(m * invT);
    float sum = 0.0f;
    for (c = 0; c < C; ++c) {
        e[c] = mas_expf_np(mas_fmaf(z[(size_t)c * stride], invT, negM));
        sum = (c == 0) ? e[c] : (sum + e[c]);
    }
    return 1.0f / sum;
}

/* K2: per-image fixed-point sums of softmax(z/T) -- the integer form of
 * `cum += mean(softmax(preds / ce_temp, dim=1), dim=(0,2,3))`
 * (active_selection/my_bvsb_predclsbal_pwr_banignore.py:41-42). */
void exact_class_prob_sum(const float* z, int B, int C, int H, int W, float invT, uint64_t* prob_sum) {
    const size_t HW = (size_t)H * W;
    float p[MAXC];
    int b, c;
    size_t i;
    for (b = 0; b < B; ++b)
        for (i = 0; i < HW; ++i) {
            const float R = softmax_row(z + (size_t)b * C * HW + i, HW, C, invT, p) * 8388608.0f;
            for (c = 0; c < C; ++c) prob_sum[(size_t)b * C + c] += mas_probq(p[c], R);
        }
}

/* Host step between the passes: mean of per-batch means, then (coeff*cum+1)^-2
 * (my_bvsb_predclsbal_pwr_banignore.py:42,45,47), evaluated in f64 from the integer sums and
 * rounded once to f32.  batch_of[i] = reference batch index of image i; n_batches = len(loader). */
void exact_class_weight(const uint64_t* prob_sum, int n_img, int C, int64_t HW, const int32_t* batch_of, int n_batches,
                        double coeff, double* cum /* [C] */, float* cls_w /* [C] */) {
    int c, i, b;
    for (c = 0; c < C; ++c) {
        double acc = 0.0;
        for (b = 0; b < n_batches; ++b) {
            uint64_t s = 0;
            int64_t n = 0;
            for (i = 0; i < n_img; ++i)
                if (batch_of[i] == b) { s += prob_sum[(size_t)i * C + c]; n += 1; }
            if (n) acc += ((double)s / 8388608.0) / ((double)n * (double)HW);
        }
        cum[c] = acc / (double)n_batches;
        const double t = coeff * cum[c] + 1.0;
        cls_w[c] = (float)(1.0 / (t * t));
    }
}

/* K1+K3: per-superpixel fixed-point sum of (weighted) BvSB and arg-max-class histogram
 * (active_selection/my_bvsb.py:19-27; my_bvsb_predclsbal_pwr_banignore.py:57-69). */
void exact_bvsb_region_accum(const float* z, const int64_t* spx, const float* cls_w, int B, int C, int H, int W, int S,
                             float invT, uint64_t* score_sum, uint32_t* hist) {
    const size_t HW = (size_t)H * W;
    int b, c;
    size_t i;
    for (b = 0; b < B; ++b)
        for (i = 0; i < HW; ++i) {
            const int64_t id = spx[(size_t)b * HW + i];
            if (id < 0 || id >= S) continue;
            const float* zp = z + (size_t)b * C * HW + i;
            float b1 = -INFINITY, b2 = -INFINITY;
            int a1 = 0;
            for (c = 0; c < C; ++c) {
                const float v = zp[(size_t)c * HW];
                if (v > b1) { b2 = b1; b1 = v; a1 = c; }      /* strict >: lowest index wins ties */
                else if (v > b2) b2 = v;
            }
            float v = mas_bvsb(b1, b2, invT);
            if (cls_w) v = v * cls_w[a1];
            score_sum[(size_t)b * S + id] += mas_fix(v, MAS_SCORE_FRAC);
            hist[((size_t)b * S + id) * C + a1] += 1u;
        }
}

/* K3 tail + K4 ban: mean (0 for an empty region), dominant = first arg-max of the histogram,
 * ban-ignore (my_bvsb_predclsbal_pwr_banignore.py:79-84). */
void exact_region_finalize(const uint64_t* score_sum, const uint32_t* hist, int64_t n_regions, int C, int ban_class,
                           float* score, int32_t* dominant, uint32_t* count) {
    int64_t r;
    int c;
    for (r = 0; r < n_regions; ++r) {
        uint64_t n = 0;
        uint32_t best = 0;
        int arg = 0;
        for (c = 0; c < C; ++c) {
            const uint32_t v = hist[r * C + c];
            n += v;
            if (v > best) { best = v; arg = c; }
        }
        float s = 0.0f;
        if (n) s = mas_fixed_mean(score_sum[r], n, MAS_SCORE_FRAC);
        if (ban_class >= 0 && arg == ban_class) s = 0.0f;
        score[r] = s;
        if (dominant) dominant[r] = arg;
        if (count) count[r] = (uint32_t)n;
    }
}

/* elementwise probes of the arithmetic spec, for tests/test_detmath.py */
void exact_expf_array(const float* x, float* y, int64_t n) { int64_t i; for (i = 0; i < n; ++i) y[i] = mas_expf(x[i]); }
void exact_expf_np_array(const float* x, float* y, int64_t n) { int64_t i; for (i = 0; i < n; ++i) y[i] = mas_expf_np(x[i]); }
void exact_logf_array(const float* x, float* y, int64_t n) { int64_t i; for (i = 0; i < n; ++i) y[i] = mas_logf(x[i]); }
void exact_fix_array(const float* x, int frac, uint64_t* y, int64_t n) { int64_t i; for (i = 0; i < n; ++i) y[i] = mas_fix(x[i], frac); }
void exact_softmax_rows(const float* z, int64_t n, int C, float invT, float* p) {
    int64_t i;
    for (i = 0; i < n; ++i) {
        const float rinv = softmax_row(z + i * C, 1, C, invT, p + i * C);
        int c;
        for (c = 0; c < C; ++c) p[i * C + c] = p[i * C + c] * rinv;
    }
}

/* =============================================================================================
 * Stage-1 partial-label losses
 * ============================================================================================= */
#define LOSS_CE 1
#define LOSS_GROUP 2
#define LOSS_GROUP_ONLY_MULTI 4
#define LOSS_DECOMP 8
#define LOSS_TCE 16 /* `spx` holds class labels: target = the label, l = -log p_label (no epsilon), mean over the valid pixels -- utils/loss.py:10-21 */
enum { ACC_SUM_CE = 0, ACC_SUM_MC = 1, ACC_N_CE = 2, ACC_N_MC = 3, ACC_N_EMPTY = 4, ACC_SUM_GROUP = 5, ACC_N_GROUP = 6 };

static int popcount32(uint32_t v) { int n = 0; while (v) { n += (int)(v & 1u); v >>= 1; } return n; }

/* multi-hot rows -> bit masks over the first cols_used columns (utils/loss.py:104,124,571 drop the
 * last column in the base classes; the predignore subclasses use every column). */
void exact_target_bits(const uint8_t* tgt, int64_t n_rows, int cols_stored, int cols_used, uint32_t* bits) {
    int64_t r;
    int c;
    for (r = 0; r < n_rows; ++r) {
        uint32_t b = 0;
        for (c = 0; c < cols_used; ++c) b |= (tgt[r * cols_stored + c] ? 1u : 0u) << c;
        bits[r] = b;
    }
}

/* K5 + K6 forward over the selected pixels.
 *   K5: pos = sum_{c in Y} softmax_c ; l = -log(pos + 1e-8); one-hot rows -> ce, multi-hot -> mc
 *       (trainer/active_joint_multi_predignore_lossdecomp.py:53-70)
 *   K6: M[s,c] = max over the selected pixels of superpixel s (onlymulti: only if #Y_s > 1) of softmax_c,
 *       first pixel wins ties (trainer/active_joint_multi_predignore_mclossablation2.py:51-60) */
void exact_partial_loss_fwd(const float* z, const int64_t* spx, const uint8_t* mask, const uint32_t* bits, int N, int C, int H,
                            int W, int S, float invT, int flags, uint64_t* gmax, uint64_t* acc) {
    const size_t HW = (size_t)H * W;
    float p[MAXC];
    int n, c;
    size_t i;
    for (n = 0; n < N; ++n)
        for (i = 0; i < HW; ++i) {
            if (!mask[(size_t)n * HW + i]) continue;
            const int64_t id = spx[(size_t)n * HW + i];
            if (id < 0 || id >= S) continue;
            const uint32_t Y = (flags & LOSS_TCE) ? (1u << id) : bits[(size_t)n * S + id];
            const int nb = popcount32(Y);
            if (nb == 0) { acc[ACC_N_EMPTY] += 1; continue; }
            {
                const float rinv = softmax_row(z + (size_t)n * C * HW + i, HW, C, invT, p);
                for (c = 0; c < C; ++c) p[c] = p[c] * rinv;
            }
            if (flags & LOSS_CE) {
                float pos = 0.0f;
                for (c = 0; c < C; ++c)
                    if ((Y >> c) & 1u) pos = pos + p[c];
                const float l = -mas_logf((flags & LOSS_TCE) ? (pos < 1.17549435e-38f ? 1.17549435e-38f : pos) : pos + 1e-8f);
                const uint64_t q = mas_fix(l, MAS_LOSS_FRAC);
                if (nb == 1) { acc[ACC_SUM_CE] += q; acc[ACC_N_CE] += 1; }
                else { acc[ACC_SUM_MC] += q; acc[ACC_N_MC] += 1; }
            }
            if ((flags & LOSS_GROUP) && (!(flags & LOSS_GROUP_ONLY_MULTI) || nb > 1)) {
                for (c = 0; c < C; ++c)
                    if ((Y >> c) & 1u) {
                        const uint64_t w = ((uint64_t)mas_f2u(p[c]) << 32) | (uint64_t)(0xffffffffu - (uint32_t)i);
                        uint64_t* g = &gmax[((size_t)n * S + id) * C + c];
                        if (w > *g) *g = w;
                    }
            }
        }
}

/* sum of -log(M + 1e-8) over the non-zero table entries and their count (..._mclossablation2.py:67-73) */
void exact_group_finalize(const uint64_t* gmax, int64_t n_entries, uint64_t* acc) {
    int64_t i;
    for (i = 0; i < n_entries; ++i) {
        const uint32_t pb = (uint32_t)(gmax[i] >> 32);
        if (pb) {
            const float l = -mas_logf(mas_u2f(pb) + 1e-8f);
            acc[ACC_SUM_GROUP] += mas_fix(l, MAS_LOSS_FRAC);
            acc[ACC_N_GROUP] += 1;
        }
    }
}

static float loss_value_n(uint64_t sum, uint64_t n, int one) {
    union { double d; uint64_t u; } s;
    s.u = (uint64_t)(1023 - MAS_LOSS_FRAC) << 52;
    return (float)(((double)sum * s.d) / (double)(n + one));
}

static float loss_value(uint64_t sum, uint64_t n) { return loss_value_n(sum, n, 1); }

/* loss / num_valid with num_valid starting at 1 (utils/loss.py:106,556; lossdecomp.py:32-35,72) */
void exact_loss_values(const uint64_t* acc, int flags, float* out) {
    if (flags & LOSS_TCE) {             /* nn.CrossEntropyLoss(reduction='mean'): sum / n over the valid pixels */
        out[0] = loss_value_n(acc[ACC_SUM_CE], acc[ACC_N_CE], 0);
        out[1] = 0.0f;
        out[2] = 0.0f;
        return;
    }
    if (flags & LOSS_DECOMP) {
        out[0] = loss_value(acc[ACC_SUM_CE], acc[ACC_N_CE]);
        out[1] = loss_value(acc[ACC_SUM_MC], acc[ACC_N_MC]);
    } else {
        out[0] = loss_value(acc[ACC_SUM_CE] + acc[ACC_SUM_MC], acc[ACC_N_CE] + acc[ACC_N_MC]);
        out[1] = 0.0f;
    }
    out[2] = loss_value(acc[ACC_SUM_GROUP], acc[ACC_N_GROUP]);
}

void exact_loss_scales(const uint64_t* acc, const float* grad_out, int flags, float* scale) {
    if (flags & LOSS_TCE) {
        scale[0] = grad_out[0] / (float)acc[ACC_N_CE];
        scale[1] = 0.0f;
        scale[2] = 0.0f;
        return;
    }
    if (flags & LOSS_DECOMP) {
        scale[0] = grad_out[0] / (float)(acc[ACC_N_CE] + 1);
        scale[1] = grad_out[1] / (float)(acc[ACC_N_MC] + 1);
    } else {
        const float s = grad_out[0] / (float)(acc[ACC_N_CE] + acc[ACC_N_MC] + 1);
        scale[0] = s;
        scale[1] = s;
    }
    scale[2] = grad_out[2] / (float)(acc[ACC_N_GROUP] + 1);
}

/* d(sum_k scale_k * loss_k) / dz: analytic gradient of the losses above.
 *   CE part   : dz_j = scale*invT/(pos+eps) * p_j * (pos - Y_j)
 *   group part: for every class c whose arg-max pixel is this pixel:
 *               dz_j += t_c * p_c * (delta_cj - p_j),  t_c = -(scale_g*invT) / (p_c + eps)  */
void exact_partial_loss_bwd(const float* z, const int64_t* spx, const uint8_t* mask, const uint32_t* bits, const uint64_t* gmax,
                            const float* scale, int N, int C, int H, int W, int S, float invT, int flags, float* dz) {
    const size_t HW = (size_t)H * W;
    const float a_ce = scale[0] * invT, a_mc = scale[1] * invT, g6 = scale[2] * invT;
    float p[MAXC], t[MAXC];
    int n, c;
    size_t i;
    memset(dz, 0, sizeof(float) * (size_t)N * C * HW);
    for (n = 0; n < N; ++n)
        for (i = 0; i < HW; ++i) {
            if (!mask[(size_t)n * HW + i]) continue;
            const int64_t id = spx[(size_t)n * HW + i];
            if (id < 0 || id >= S) continue;
            const uint32_t Y = (flags & LOSS_TCE) ? (1u << id) : bits[(size_t)n * S + id];
            const int nb = popcount32(Y);
            if (nb == 0) continue;
            {
                const float rinv = softmax_row(z + (size_t)n * C * HW + i, HW, C, invT, p);
                for (c = 0; c < C; ++c) p[c] = p[c] * rinv;
            }
            float coef = 0.0f, pos = 0.0f;
            if (flags & LOSS_CE) {
                for (c = 0; c < C; ++c)
                    if ((Y >> c) & 1u) pos = pos + p[c];
                coef = ((nb == 1) ? a_ce : a_mc) * (1.0f / ((flags & LOSS_TCE) ? (pos < 1.17549435e-38f ? 1.17549435e-38f : pos) : pos + 1e-8f));
            }
            uint32_t A = 0;
            float u = 0.0f;
            if ((flags & LOSS_GROUP) && (!(flags & LOSS_GROUP_ONLY_MULTI) || nb > 1)) {
                const uint32_t key = 0xffffffffu - (uint32_t)i;
                for (c = 0; c < C; ++c) {
                    t[c] = 0.0f;
                    if ((Y >> c) & 1u) {
                        const uint64_t w = gmax[((size_t)n * S + id) * C + c];
                        if ((uint32_t)w == key && (uint32_t)(w >> 32) != 0u) {
                            A |= 1u << c;
                            t[c] = -(g6 / (p[c] + 1e-8f));
                            u = u + t[c] * p[c];
                        }
                    }
                }
            }
            for (c = 0; c < C; ++c) {
                const float yj = ((Y >> c) & 1u) ? 1.0f : 0.0f;
                float d = coef * (p[c] * (pos - yj));
                if (A) {
                    if ((A >> c) & 1u) d = d + t[c] * p[c];
                    d = d - p[c] * u;
                }
                dz[((size_t)n * C + c) * HW + i] = d;
            }
        }
}

/* F.interpolate(x, size=(Ho, Wo), mode='bilinear', align_corners=False) of models/segmentation/utils.py:25, in the
 * operation order of csrc/upsample.hip (ATen's area_pixel_compute_source_index; every product and sum rounded once). */
static void bilinear_tap(float scale, int o, int n_in, int* i0, int* i1, float* l0, float* l1) {
    float s = scale * ((float)o + 0.5f) - 0.5f;
    s = s < 0.0f ? 0.0f : s;
    *i0 = (int)s;
    *i1 = *i0 + (*i0 < n_in - 1 ? 1 : 0);
    *l1 = s - (float)*i0;
    *l0 = 1.0f - *l1;
}

void exact_upsample_bilinear(const float* x, int64_t NC, int Hi, int Wi, int Ho, int Wo, float* y) {
    const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
    int64_t nc;
    int oy, ox;
    for (nc = 0; nc < NC; ++nc)
        for (oy = 0; oy < Ho; ++oy) {
            int y0, y1;
            float ly0, ly1;
            bilinear_tap(sh, oy, Hi, &y0, &y1, &ly0, &ly1);
            const float* r0 = x + ((size_t)nc * Hi + y0) * Wi;
            const float* r1 = x + ((size_t)nc * Hi + y1) * Wi;
            for (ox = 0; ox < Wo; ++ox) {
                int x0, x1;
                float lx0, lx1;
                bilinear_tap(sw, ox, Wi, &x0, &x1, &lx0, &lx1);
                y[((size_t)nc * Ho + oy) * Wo + ox] = ly0 * (lx0 * r0[x0] + lx1 * r0[x1]) + ly1 * (lx0 * r1[x0] + lx1 * r1[x1]);
            }
        }
}

/* Backward of (bilinear x4 upsampling -> partial-label losses) with respect to the QUARTER-resolution logits zq [N,C,h,w]
 * (mas_partial_loss_bwd_lowres): every full-resolution gradient d (exact_partial_loss_bwd on the interpolated logits) adds
 * round_to_nearest_even(d * (ly * lx) * 2^44) into the four elements its logit was interpolated from -- integer sums, so the
 * result does not depend on the order; dzq = (float)(sum * 2^-44). */
void exact_partial_loss_bwd_lowres(const float* zq, int h, int w, const int64_t* spx, const uint8_t* mask, const uint32_t* bits,
                                   const uint64_t* gmax, const float* scale, int N, int C, int H, int W, int S, float invT, int flags,
                                   int64_t* dzq_fix, float* dzq) {
    const size_t HW = (size_t)H * W, hw = (size_t)h * w;
    float* z = (float*)malloc(sizeof(float) * (size_t)N * C * HW);
    float* dz = (float*)malloc(sizeof(float) * (size_t)N * C * HW);
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    int n, c, oy, ox;
    size_t i;
    exact_upsample_bilinear(zq, (int64_t)N * C, h, w, H, W, z);
    exact_partial_loss_bwd(z, spx, mask, bits, gmax, scale, N, C, H, W, S, invT, flags, dz);
    memset(dzq_fix, 0, sizeof(int64_t) * (size_t)N * C * hw);
    for (n = 0; n < N; ++n)
        for (c = 0; c < C; ++c) {
            int64_t* q = dzq_fix + ((size_t)n * C + c) * hw;
            const float* g = dz + ((size_t)n * C + c) * HW;
            for (oy = 0; oy < H; ++oy) {
                int y0, y1;
                float ly0, ly1;
                bilinear_tap(sh, oy, h, &y0, &y1, &ly0, &ly1);
                for (ox = 0; ox < W; ++ox) {
                    const float d = g[(size_t)oy * W + ox];
                    int x0, x1;
                    float lx0, lx1;
                    if (d == 0.0f) continue;                 /* rounds to 0 quanta in any case */
                    bilinear_tap(sw, ox, w, &x0, &x1, &lx0, &lx1);
                    q[(size_t)y0 * w + x0] += (int64_t)llrint((double)(d * (ly0 * lx0)) * 17592186044416.0);
                    q[(size_t)y0 * w + x1] += (int64_t)llrint((double)(d * (ly0 * lx1)) * 17592186044416.0);
                    q[(size_t)y1 * w + x0] += (int64_t)llrint((double)(d * (ly1 * lx0)) * 17592186044416.0);
                    q[(size_t)y1 * w + x1] += (int64_t)llrint((double)(d * (ly1 * lx1)) * 17592186044416.0);
                }
            }
        }
    for (i = 0; i < (size_t)N * C * hw; ++i) dzq[i] = (float)((double)dzq_fix[i] * (1.0 / 17592186044416.0));
    free(z);
    free(dz);
}

/* =============================================================================================
 * Single-pass acquisition scan (csrc/single_pass.hip)
 * ============================================================================================= */

/* One scan producing the class-probability quanta of exact_class_prob_sum, and per (region, arg-max class) the
 * fixed-point sum of the UNWEIGHTED margins plus the pixel counts (the weight of
 * my_bvsb_predclsbal_pwr_banignore.py:59-61 depends only on the arg-max class and factors out of the region sum). */
void exact_single_pass_accum(const float* z, const int64_t* spx, int B, int C, int H, int W, int S, float invT,
                             uint64_t* prob_sum, uint64_t* class_sum, uint32_t* hist) {
    const size_t HW = (size_t)H * W;
    float e[MAXC];
    int b, c;
    size_t i;
    for (b = 0; b < B; ++b)
        for (i = 0; i < HW; ++i) {
            const float* zp = z + (size_t)b * C * HW + i;
            const float R = softmax_row(zp, HW, C, invT, e) * 8388608.0f;
            for (c = 0; c < C; ++c) prob_sum[(size_t)b * C + c] += mas_probq(e[c], R);
            const int64_t id = spx[(size_t)b * HW + i];
            if (id < 0 || id >= S) continue;
            float b1 = -INFINITY, b2 = -INFINITY;
            int a1 = 0;
            for (c = 0; c < C; ++c) {
                const float v = zp[(size_t)c * HW];
                if (v > b1) { b2 = b1; b1 = v; a1 = c; }
                else if (v > b2) b2 = v;
            }
            class_sum[((size_t)b * S + id) * C + a1] += mas_fix(mas_bvsb(b1, b2, invT), MAS_SCORE_FRAC);
            hist[((size_t)b * S + id) * C + a1] += 1u;
        }
}

/* score = floor(((sum_c class_sum[c] * w31[c]) >> 31) / n) * 2^-40, w31 = floor(w * 2^31); dominant class; ban. */
void exact_region_finalize_weighted(const uint64_t* class_sum, const uint32_t* hist, int64_t n_regions, int C,
                                    const uint32_t* w31, int ban_class, float* score, int32_t* dominant, uint32_t* count) {
    int64_t r;
    int c;
    for (r = 0; r < n_regions; ++r) {
        uint64_t n = 0, hi = 0, lo = 0;
        uint32_t best = 0;
        int arg = 0;
        for (c = 0; c < C; ++c) {
            const uint32_t v = hist[r * C + c];
            n += v;
            if (v > best) { best = v; arg = c; }
            if (v) mas_mac_u64_u32(class_sum[r * C + c], w31[c], &hi, &lo);
        }
        float s = 0.0f;
        if (n) s = mas_fixed_mean(mas_shr31_u128(hi, lo), n, MAS_SCORE_FRAC);
        if (ban_class >= 0 && arg == ban_class) s = 0.0f;
        score[r] = s;
        if (dominant) dominant[r] = arg;
        if (count) count[r] = (uint32_t)n;
    }
}

/* =============================================================================================
 * Stage-2 cosine pseudo labels with one-ring propagation (csrc/stage2.hip)
 * trainer/eval_save_cosplbl_prop.py:121-314 and ..._includeonehot.py
 * ============================================================================================= */

/* feature k of output pixel (y, x): the feature map is [Ch, fh, fw]; when (fh, fw) != (H, W) it is upsampled
 * bilinearly with align_corners=False exactly as F.interpolate does in feat_forward (models/segmentation/utils.py:28-34):
 * src = scale*(dst+0.5)-0.5 clamped at 0, lambda = src - floor(src), value = l0h*(l0w*v00 + l1w*v01) + l1h*(l0w*v10 + l1w*v11). */
static float stage2_feat(const float* f, int k, int fh, int fw, int H, int W, int y, int x) {
    const float* p = f + (size_t)k * fh * fw;
    if (fh == H && fw == W) return p[(size_t)y * W + x];
    const float sh = (float)fh / (float)H, sw = (float)fw / (float)W;
    float sy = sh * ((float)y + 0.5f) - 0.5f; if (sy < 0.0f) sy = 0.0f;
    float sx = sw * ((float)x + 0.5f) - 0.5f; if (sx < 0.0f) sx = 0.0f;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < fh - 1 ? 1 : 0), x1 = x0 + (x0 < fw - 1 ? 1 : 0);
    const float l1h = sy - (float)y0, l0h = 1.0f - l1h, l1w = sx - (float)x0, l0w = 1.0f - l1w;
    return l0h * (l0w * p[(size_t)y0 * fw + x0] + l1w * p[(size_t)y0 * fw + x1]) +
           l1h * (l0w * p[(size_t)y1 * fw + x0] + l1w * p[(size_t)y1 * fw + x1]);
}

static int cmp_float(const void* a, const void* b) {
    const float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* One image.  gmax [S,C] comes from exact_partial_loss_fwd(flags = GROUP [| ONLY_MULTI], invT = 1): packed
 * (softmax prob bits << 32 | ~arg pixel).  out [H*W] int32, 255 = no label.  Returns the number of prototypes. */
int exact_stage2_plbl(const float* feat, int Ch, int fh, int fw, int H, int W, const int64_t* spx, const uint8_t* mask,
                      const uint64_t* gmax, int S, int C, int32_t* out) {
    const size_t HW = (size_t)H * W;
    int s, c, j, k;
    size_t i;
    /* prototypes ordered by (superpixel, class) */
    int n_proto = 0;
    for (s = 0; s < S; ++s) for (c = 0; c < C; ++c) if (gmax[(size_t)s * C + c]) n_proto++;
    int* p_start = (int*)calloc(S + 1, sizeof(int));
    int* p_cls = (int*)malloc(sizeof(int) * (n_proto + 1));
    float* P = (float*)malloc(sizeof(float) * (size_t)(n_proto + 1) * Ch);
    j = 0;
    for (s = 0; s < S; ++s) {
        p_start[s] = j;
        for (c = 0; c < C; ++c) {
            const uint64_t w = gmax[(size_t)s * C + c];
            if (!w) continue;
            const uint32_t pix = 0xffffffffu - (uint32_t)w;
            for (k = 0; k < Ch; ++k) P[(size_t)j * Ch + k] = stage2_feat(feat, k, fh, fw, H, W, (int)(pix / W), (int)(pix % W));
            p_cls[j++] = c;
        }
    }
    p_start[S] = j;
    /* nearest prototype of the own superpixel for every valid pixel */
    int32_t* nn = (int32_t*)malloc(sizeof(int32_t) * HW);
    float* nn_sim = (float*)malloc(sizeof(float) * HW);
    float* fx = (float*)malloc(sizeof(float) * Ch);
    for (i = 0; i < HW; ++i) {
        nn[i] = -1;
        out[i] = 255;
        if (!mask[i]) continue;
        const int64_t id = spx[i];
        if (id < 0 || id >= S || p_start[id + 1] == p_start[id]) continue;
        for (k = 0; k < Ch; ++k) fx[k] = stage2_feat(feat, k, fh, fw, H, W, (int)(i / W), (int)(i % W));
        float best = 0.0f;
        for (j = p_start[id]; j < p_start[id + 1]; ++j) {
            float acc = 0.0f;
            for (k = 0; k < Ch; ++k) acc = mas_fmaf(P[(size_t)j * Ch + k], fx[k], acc);
            if (nn[i] < 0 || acc > best) { best = acc; nn[i] = j; }       /* first maximum wins */
        }
        nn_sim[i] = best;
    }
    /* per prototype: lower median of the similarities of its pixels, 1.0 when it has none */
    float* thr = (float*)malloc(sizeof(float) * (n_proto + 1));
    float* buf = (float*)malloc(sizeof(float) * HW);
    for (j = 0; j < n_proto; ++j) {
        size_t n = 0;
        for (i = 0; i < HW; ++i) if (nn[i] == j) buf[n++] = nn_sim[i];
        if (n) { qsort(buf, n, sizeof(float), cmp_float); thr[j] = buf[(n - 1) / 2]; }
        else thr[j] = 1.0f;
    }
    /* ring[t] = valid superpixels within Chebyshev distance 1 of a pixel of t (3x3 dilation, symmetric) */
    const int words = (S + 31) / 32;
    uint32_t* adj = (uint32_t*)calloc((size_t)S * words, sizeof(uint32_t));
    for (i = 0; i < HW; ++i) {
        const int64_t t = spx[i];
        if (t < 0 || t >= S) continue;
        const int y = (int)(i / W), x = (int)(i % W);
        int dy, dx;
        for (dy = -1; dy <= 1; ++dy) for (dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const int64_t g = spx[(size_t)yy * W + xx];
            if (g < 0 || g >= S || p_start[g + 1] == p_start[g]) continue;
            adj[(size_t)t * words + (g >> 5)] |= 1u << (g & 31);
        }
    }
    /* propagation in ascending superpixel id: the last valid neighbour that accepts the pixel wins */
    for (i = 0; i < HW; ++i) {
        const int64_t t = spx[i];
        int label = 255, have = 0;
        if (t >= 0 && t < S) {
            for (s = 0; s < S; ++s) {
                if (!((adj[(size_t)t * words + (s >> 5)] >> (s & 31)) & 1u)) continue;
                if (!have) { for (k = 0; k < Ch; ++k) fx[k] = stage2_feat(feat, k, fh, fw, H, W, (int)(i / W), (int)(i % W)); have = 1; }
                float best = 0.0f;
                int arg = -1, ok = 0;
                for (j = p_start[s]; j < p_start[s + 1]; ++j) {
                    float acc = 0.0f;
                    for (k = 0; k < Ch; ++k) acc = mas_fmaf(P[(size_t)j * Ch + k], fx[k], acc);
                    if (arg < 0 || acc > best) { best = acc; arg = j; }
                    if (thr[j] < acc) ok = 1;
                }
                if (ok) label = p_cls[arg];
            }
        }
        if (nn[i] >= 0) label = p_cls[nn[i]];
        out[i] = label;
    }
    free(p_start); free(p_cls); free(P); free(nn); free(nn_sim); free(fx); free(thr); free(buf); free(adj);
    return n_proto;
}

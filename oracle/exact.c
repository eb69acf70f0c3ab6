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

/* softmax(z * invT) of one pixel; operation order as in csrc/common.h:mas_softmax_regs */
static void softmax_row(const float* z, size_t stride, int C, float invT, float* p) {
    float m = z[0] * invT;
    int c;
    for (c = 0; c < C; ++c) {
        p[c] = z[(size_t)c * stride] * invT;
        if (p[c] > m) m = p[c];
    }
    float sum = 0.0f;
    for (c = 0; c < C; ++c) {
        p[c] = mas_expf(p[c] - m);
        sum = (c == 0) ? p[c] : (sum + p[c]);
    }
    const float rinv = 1.0f / sum;
    for (c = 0; c < C; ++c) p[c] = p[c] * rinv;
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
            softmax_row(z + (size_t)b * C * HW + i, HW, C, invT, p);
            for (c = 0; c < C; ++c) prob_sum[(size_t)b * C + c] += mas_fix(p[c], MAS_PROB_FRAC);
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
            if (n) acc += ((double)s / 2147483648.0) / ((double)n * (double)HW);
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
void exact_logf_array(const float* x, float* y, int64_t n) { int64_t i; for (i = 0; i < n; ++i) y[i] = mas_logf(x[i]); }
void exact_fix_array(const float* x, int frac, uint64_t* y, int64_t n) { int64_t i; for (i = 0; i < n; ++i) y[i] = mas_fix(x[i], frac); }
void exact_softmax_rows(const float* z, int64_t n, int C, float invT, float* p) {
    int64_t i;
    for (i = 0; i < n; ++i) softmax_row(z + i * C, 1, C, invT, p + i * C);
}

// metrics.hip -- mIoU counters (a-12 of SURVEY.md section 8) for gfx950.
//
// Reference: utils/miou.py:23-38 (MeanIoU._after_step: per class seen / correct / positive over the
// non-ignored pixels), utils/miou_evalignore.py:20-32 (IoUIgnore: the "undefined" class), and the caller
// trainer/active_joint_multi_predignore.py:175-215 which feeds argmax(z[:, :-1]) and argmax(z).
//
//   k_iou_counts     label maps in (what MeanIoU._after_step receives), counters out.
//   k_logits_iou     fused: one read of the logits gives both arg-max maps and all counters; nothing
//                    is materialised (the reference builds two int64 maps of the image size per batch
//                    and issues 19 x 3 host-synchronising reductions).
// Exact integer work: per-workgroup LDS histograms, one 64-bit global atomic per non-zero counter.
#include "common.h"

namespace {
constexpr int kThreads = 256;
constexpr int kMaxCnt = 3 * (MAS_MAX_CLASSES + 1);

__device__ __forceinline__ void flush(unsigned* s_cnt, int n, mas_u64* out) {
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kThreads)
        if (s_cnt[i]) atomicAdd(&out[i], (mas_u64)s_cnt[i]);
}

// counts layout: seen[C], correct[C], positive[C], then (ignore_seen, ignore_correct, ignore_positive)
__device__ __forceinline__ void tally(unsigned* s_cnt, int C, long long t, long long o_cls, long long o_all, long long ignore_label,
                                      bool with_ignore_iou) {
    if (t != ignore_label) {
        if (t >= 0 && t < C) {
            atomicAdd(&s_cnt[t], 1u);
            if (o_cls == t) atomicAdd(&s_cnt[C + t], 1u);
        }
        if (o_cls >= 0 && o_cls < C) atomicAdd(&s_cnt[2 * C + o_cls], 1u);
    }
    if (with_ignore_iou) {
        const bool tig = (t == ignore_label), oig = (o_all == C);
        if (tig) atomicAdd(&s_cnt[3 * C], 1u);
        if (tig && oig) atomicAdd(&s_cnt[3 * C + 1], 1u);
        if (oig) atomicAdd(&s_cnt[3 * C + 2], 1u);
    }
}

__global__ __launch_bounds__(kThreads) void k_iou_counts(const long long* __restrict__ outputs, const long long* __restrict__ outputs_all,
                                                          const long long* __restrict__ targets, long long n, int C,
                                                          long long ignore_label, mas_u64* __restrict__ counts) {
    __shared__ unsigned s_cnt[kMaxCnt];
    const int ncnt = 3 * C + 3;
    for (int i = threadIdx.x; i < ncnt; i += kThreads) s_cnt[i] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads)
        tally(s_cnt, C, targets[i], outputs ? outputs[i] : -1, outputs_all ? outputs_all[i] : -1, ignore_label,
              outputs_all != nullptr);
    flush(s_cnt, ncnt, counts);
}

// logits [B, CH, HW]; classes = CH - 1 when the model predicts the extra "undefined" channel
__global__ __launch_bounds__(kThreads) void k_logits_iou(const float* __restrict__ z, const long long* __restrict__ targets, int B,
                                                          int CH, int HW, int C, long long ignore_label, int blocks_per_image,
                                                          mas_u64* __restrict__ counts) {
    __shared__ unsigned s_cnt[kMaxCnt];
    const int ncnt = 3 * C + 3;
    for (int i = threadIdx.x; i < ncnt; i += kThreads) s_cnt[i] = 0;
    __syncthreads();
    const int b = blockIdx.x / blocks_per_image;
    const int j = blockIdx.x - b * blocks_per_image;
    const float* zb = z + (size_t)b * CH * HW;
    for (int p = j * kThreads + threadIdx.x; p < HW; p += blocks_per_image * kThreads) {
        float best = zb[p];
        int arg = 0;
        for (int c = 1; c < C; ++c) {
            const float v = zb[(size_t)c * HW + p];
            if (v > best) { best = v; arg = c; }        // first maximum wins (torch.max)
        }
        int arg_all = arg;
        if (CH > C && zb[(size_t)C * HW + p] > best) arg_all = C;
        tally(s_cnt, C, targets[(size_t)b * HW + p], arg, arg_all, ignore_label, CH > C);
    }
    flush(s_cnt, ncnt, counts);
}
}  // namespace

extern "C" int mas_iou_counts(const int64_t* outputs, const int64_t* outputs_all, const int64_t* targets, int64_t n,
                              int num_classes, int64_t ignore_label, uint64_t* counts, void* stream) {
    if (!targets || !counts || (!outputs && !outputs_all)) return MAS_ERR_NULL;
    if (n <= 0) return MAS_ERR_SHAPE;
    if (num_classes < 1 || num_classes > MAS_MAX_CLASSES) return MAS_ERR_CLASSES;
    long long nblk = (n + kThreads - 1) / kThreads;
    if (nblk > 2048) nblk = 2048;
    hipLaunchKernelGGL(k_iou_counts, dim3((unsigned)nblk), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const long long*>(outputs), reinterpret_cast<const long long*>(outputs_all),
                       reinterpret_cast<const long long*>(targets), (long long)n, num_classes, (long long)ignore_label,
                       reinterpret_cast<mas_u64*>(counts));
    return mas_launch_status();
}

extern "C" int mas_logits_iou_counts(const float* z, const int64_t* targets, int B, int channels, int H, int W, int num_classes,
                                     int64_t ignore_label, uint64_t* counts, void* stream) {
    if (!z || !targets || !counts) return MAS_ERR_NULL;
    if (B <= 0 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL / 2) return MAS_ERR_SHAPE;
    if (num_classes < 1 || num_classes > MAS_MAX_CLASSES || (channels != num_classes && channels != num_classes + 1))
        return MAS_ERR_CLASSES;
    const int HW = H * W;
    int bpi = 2048 / B;
    const int max_bpi = (HW + kThreads - 1) / kThreads;
    if (bpi > max_bpi) bpi = max_bpi;
    if (bpi < 1) bpi = 1;
    hipLaunchKernelGGL(k_logits_iou, dim3((unsigned)(B * bpi)), dim3(kThreads), 0, static_cast<hipStream_t>(stream), z,
                       reinterpret_cast<const long long*>(targets), B, channels, HW, num_classes, (long long)ignore_label, bpi,
                       reinterpret_cast<mas_u64*>(counts));
    return mas_launch_status();
}

// K9 — Adam(+L2 weight decay) over the flat arena-layout gradient K7 produced.
// Replaces torch.optim.Adam.step() as the reference drives it through DPOptimizer (optimizer built at
// Model_Code/timeautoencoder.py:658 with weight_decay=1e-6, stepped at :751): ~8 multi-tensor launches and three
// more P-sized passes become ONE launch that reads grad/m/v/param once and writes m/v/param once.
// HBM-bound and tiny (P <= ~12 M floats); one thread owns 4 consecutive arena elements.
#include "dpk_common.h"

namespace {

struct AdamWin {      // one parameter's window inside an arena row
    float* param;     // dense row-major [rows, cols]
    int64_t offset;   // first arena element
    int64_t rows, cols, ld;  // ld >= cols: arena row stride (padded rows carry zeros)
};
static_assert(sizeof(AdamWin) == 40, "table layout is part of the C ABI: 5 x int64 per window");

__global__ void __launch_bounds__(256) adam_kernel(const float* __restrict__ grad, float* __restrict__ exp_avg,
                                                   float* __restrict__ exp_avg_sq, const AdamWin* __restrict__ win,
                                                   const float* __restrict__ hyper, int n_win,
                                                   const float* __restrict__ step_dev, int64_t P) {
    const int64_t j0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (j0 >= P) return;
    // windows are 64-float aligned and ordered: the 4 elements of this thread share one window
    int lo = 0, hi = n_win - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (win[mid].offset <= j0) lo = mid; else hi = mid - 1;
    }
    const AdamWin w = win[lo];
    const int64_t local = j0 - w.offset;
    if (local < 0 || local >= w.rows * w.ld) return;  // alignment gap between windows
    const float* hp = hyper + lo * 8;
    const float lr = hp[0], b1 = hp[1], b2 = hp[2], eps = hp[3], wd = hp[4];
    const float omb1 = hp[5], omb2 = hp[6];  // 1-beta rounded from double, as torch passes them
    const double step = (double)step_dev[0] + 1.0;
    const float bc1 = (float)(1.0 - pow((double)b1, step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, step));
    const float step_size = lr / bc1;
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(grad + j0);
    f32x4 m4 = *reinterpret_cast<const f32x4*>(exp_avg + j0);
    f32x4 v4 = *reinterpret_cast<const f32x4*>(exp_avg_sq + j0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t rr = (local + e) / w.ld;  // a narrow dense window may cross rows inside these 4 elements
        const int64_t cc = (local + e) - rr * w.ld;
        if (rr >= w.rows || cc >= w.cols) continue;  // row padding / tail of the window
        float* pp = w.param + rr * w.cols + cc;
        const float p = *pp;
        float g = g4[e];
        if (wd != 0.f) g = fmaf(wd, p, g);
        const float m = m4[e] + (g - m4[e]) * omb1;
        const float v = fmaf(v4[e], b2, omb2 * g * g);
        m4[e] = m;
        v4[e] = v;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        *pp = p - step_size * (m / denom);
    }
    *reinterpret_cast<f32x4*>(exp_avg + j0) = m4;
    *reinterpret_cast<f32x4*>(exp_avg_sq + j0) = v4;
}

}  // namespace

extern "C" int dpk_adam_step(const float* grad, float* exp_avg, float* exp_avg_sq, const int64_t* win_table,
                             const float* hyper, int n_windows, const float* step_dev, int64_t P, void* stream) {
    if (!grad || !exp_avg || !exp_avg_sq || !win_table || !hyper || !step_dev) return DPK_EINVAL;
    if (n_windows <= 0 || P <= 0) return DPK_EINVAL;
    if (P % 4) return DPK_EALIGN;
    if (!dpk_aligned16(grad) || !dpk_aligned16(exp_avg) || !dpk_aligned16(exp_avg_sq)) return DPK_EALIGN;
    const int64_t blocks = (P / 4 + 255) / 256;
    if (blocks > 0x7fffffff) return DPK_ERANGE;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad, exp_avg,
                       exp_avg_sq, reinterpret_cast<const AdamWin*>(win_table), hyper, n_windows, step_dev, P);
    return dpk_launch_status();
}

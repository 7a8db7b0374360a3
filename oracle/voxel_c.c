/* Oracle (plain C): events -> voxel grid.  TEST INFRASTRUCTURE ONLY.
 *
 * Restates utils/event_utils.py:27-59 (events_to_voxel_torch) and :4-24
 * (events_to_image_torch, sequential index_put_(accumulate=True)) of the reference.
 * fp32, one IEEE rounding per operation: build with -ffp-contract=off, no fast-math.
 * Per (bin, pixel) cell the adds happen in event order, exactly as the reference's
 * per-bin passes do, so looping events-outer/bins-inner gives identical bits.
 * Used by tests (checker) and by bench.py's cpu_baseline leg ("port").
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static float lin_tnorm(int64_t i, int64_t n, int B) {
    /* torch.linspace(0, B-1, n) scalar formula, see oracle/voxel.py:linspace_f32 */
    if (n == 1) return 0.0f;
    float start = 0.0f, end = (float)(B - 1);
    float step = (end - start) / (float)(n - 1);
    if (i < n / 2) { float m = step * (float)i; return start + m; }
    float m = step * (float)(n - i - 1);
    return end - m;
}

/* one window; out[B*H*W] must be zeroed by the caller */
static void voxel_one(const float* x, const float* y, const float* t, const float* p,
                      int64_t n, int B, int H, int W, float* out) {
    if (n <= 0) return;
    const float t0 = t[0];
    const float dt = t[n - 1] - t0;
    const int lin = ((double)dt < 1e-9);
    const float bm1 = (float)(B - 1);
    const int64_t HW = (int64_t)H * W;
    for (int64_t i = 0; i < n; ++i) {
        float tn;
        if (lin) tn = lin_tnorm(i, n, B);
        else { float a = t[i] - t0; float q = a / dt; tn = q * bm1; }
        const int64_t pix = (int64_t)y[i] * W + (int64_t)x[i];   /* .long() truncation */
        for (int b = 0; b < B; ++b) {
            float d = tn - (float)b;
            float w = 1.0f - fabsf(d);
            if (!(w > 0.0f)) w = 0.0f;            /* torch.max(zeros, .) */
            float v = p[i] * w;
            out[(int64_t)b * HW + pix] += v;
        }
    }
}

/* C-ABI mirror of evr_voxelize on host memory (same argument meaning). */
int oracle_voxelize(const float* x, const float* y, const float* t, const float* p,
                    const int64_t* win_offsets, int n_windows, int B, int H, int W, float* out) {
    const int64_t vol = (int64_t)B * H * W;
    for (int w = 0; w < n_windows; ++w) {
        float* o = out + (int64_t)w * vol;
        memset(o, 0, sizeof(float) * (size_t)vol);
        int64_t a = win_offsets[w], b = win_offsets[w + 1];
        voxel_one(x + a, y + a, t + a, p + a, b - a, B, H, W, o);
    }
    return 0;
}

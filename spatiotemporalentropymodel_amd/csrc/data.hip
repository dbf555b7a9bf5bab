// Device side of the training-data pipeline: the host only decodes PNGs and draws the augmentation
// scalars; cropping, temporal flip, uint8 -> float conversion and the synthesis of the quality maps run here, one launch
// per batch.
//   stem_crop_u8_to_f32   stem/dataset_vidseq.py:12-21,79-85 and stem_roi/stem_roi_dataset.py:89-101
//                         (shared crop window for the 7 frames, reversed frame order, ToTensor = uint8 / 255)
//   stem_qmap_render      stem_roi/stem_roi_dataset.py:106-148 (uniform / gradation / sum-of-Gaussians maps)
// HBM-bound byte work: 3 B read + 12 B written per output pixel.
#include "stem_common.h"

namespace {

__global__ void crop_u8_kernel(const unsigned char *src, float *dst, const int *params, int B, int T, int H, int W, int c)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)T * B * c * c;
    if (i >= total) return;
    const int x = (int)(i % c);
    size_t r = i / c;
    const int y = (int)(r % c);
    r /= c;
    const int b = (int)(r % B), t = (int)(r / B);
    const int top = params[3 * b], left = params[3 * b + 1], flip = params[3 * b + 2];
    const int ts = flip ? T - 1 - t : t;
    const unsigned char *p = src + ((((size_t)b * T + ts) * H + top + y) * W + left + x) * 3;
    float *o = dst + (((size_t)t * B + b) * 3) * c * c + (size_t)y * c + x;
    o[0] = (float)p[0] / 255.0f;                       // ToTensor: float(byte) / 255 (IEEE division, exact match)
    o[(size_t)c * c] = (float)p[1] / 255.0f;
    o[(size_t)2 * c * c] = (float)p[2] / 255.0f;
}

// per-sample parameter block (doubles): [0] mode (0 uniform, 1 gradation, 2 gaussians), [1] value | v1 | count,
// [2] v2 | final factor, [3] transpose flag, [4 + 4k .. ] mu_row, mu_col, var_row, var_col of Gaussian k (k < 20)
constexpr int QP = 4 + 4 * 20;

__device__ __forceinline__ double qmap_raw(const double *q, int mode, int y, int x, int c)
{
    if (mode == 0) return q[1];
    if (mode == 1) {
        // np.tile(np.linspace(v1, v2, c), (c, 1)) [.T]: y_i = i * step + start (two roundings, no fma), last = stop exactly
        const int i = q[3] != 0.0 ? y : x;
        if (i == c - 1 && c > 1) return q[2];
        const double step = __ddiv_rn(__dsub_rn(q[2], q[1]), (double)(c - 1));
        return __dadd_rn(__dmul_rn((double)i, step), q[1]);
    }
    // exp(MultivariateNormal(loc, diag(var)).log_prob(grid)) in fp32 as torch evaluates it, accumulated in fp64
    double s = 0.0;
    const int n = (int)q[1];
    for (int k = 0; k < n; ++k) {
        const float mr = (float)q[4 + 4 * k], mc = (float)q[5 + 4 * k];
        const float sr = sqrtf((float)q[6 + 4 * k]), sc = sqrtf((float)q[7 + 4 * k]);
        const float dr = __fdiv_rn(__fsub_rn((float)y, mr), sr), dc = __fdiv_rn(__fsub_rn((float)x, mc), sc);
        const float m = __fadd_rn(__fmul_rn(dr, dr), __fmul_rn(dc, dc));
        const float hld = __fadd_rn(logf(sr), logf(sc));
        const float lp = __fsub_rn(__fmul_rn(-0.5f, __fadd_rn(3.6757541328186907f, m)), hld);     // 2 log(2 pi)
        s += (double)expf(lp);
    }
    return s;
}

__global__ __launch_bounds__(1024) void qmap_kernel(const double *params, float *out, int c, float inv_range)
{
    __shared__ double red[1024];
    const double *q = params + (size_t)blockIdx.x * QP;
    const int mode = (int)q[0];
    const int n = c * c;
    double scale = 1.0;
    if (mode == 2) {
        double mx = 0.0;
        for (int i = threadIdx.x; i < n; i += 1024) {
            const double v = qmap_raw(q, mode, i / c, i % c, c);
            mx = v > mx ? v : mx;
        }
        red[threadIdx.x] = mx;
        __syncthreads();
        for (int k = 512; k > 0; k >>= 1) {
            if ((int)threadIdx.x < k) red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + k] ? red[threadIdx.x] : red[threadIdx.x + k];
            __syncthreads();
        }
        scale = __dmul_rn(__ddiv_rn(100.0, red[0]), q[2]);        // qmap *= 100 / qmap.max() * (0.5 r + 0.5)
    }
    float *o = out + (size_t)blockIdx.x * n;
    for (int i = threadIdx.x; i < n; i += 1024) {
        double v = qmap_raw(q, mode, i / c, i % c, c);
        if (mode == 2) v = __dmul_rn(v, scale);
        o[i] = __fmul_rn((float)v, inv_range);                     // torch.FloatTensor(qmap) *= 1 / level_range[1]
    }
}

}   // namespace

STEM_EXPORT int stem_crop_u8_to_f32(const unsigned char *src, float *dst, const int *params, int B, int T, int H, int W, int crop,
                                    void *stream)
{
    STEM_CHECK_ARG(src && dst && params && B > 0 && T > 0 && crop > 0 && crop <= H && crop <= W, "stem_crop_u8_to_f32: bad arguments");
    const size_t total = (size_t)T * B * crop * crop;
    hipLaunchKernelGGL(crop_u8_kernel, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, params, B, T, H, W, crop);
    STEM_LAUNCH_CHECK("crop_u8");
    return 0;
}

STEM_EXPORT int stem_qmap_params_per_sample(void) { return QP; }

STEM_EXPORT int stem_qmap_render(const double *params, float *out, int B, int crop, float inv_range, void *stream)
{
    STEM_CHECK_ARG(params && out && B > 0 && crop > 0, "stem_qmap_render: bad arguments");
    hipLaunchKernelGGL(qmap_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, params, out, crop, inv_range);
    STEM_LAUNCH_CHECK("qmap_render");
    return 0;
}

// Fused optimiser kernels over flat fp32 buffers: global-norm reduction and clip + Adam.
// Replaces torch.nn.utils.clip_grad_norm_ + two torch.optim.Adam.step() calls of
// stem/trainSTEM.py:213-218 (utils.py:127-134): 41 tensors x ~6 elementwise passes become one
// HBM-bound pass (16 B read + 12 B written per parameter).
#include "stem_common.h"

namespace {

// Two deterministic stages (no float atomics: the clip coefficient derived from this sum scales every gradient, so the
// sum itself must not depend on workgroup scheduling): per-workgroup partials into the caller's scratch, then one
// workgroup adds them in a fixed order onto acc[0].
__global__ __launch_bounds__(256) void sumsq_kernel(const f32x4 *g4, const float *g, size_t n4, size_t n, double *part)
{
    __shared__ double red[256];
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = g4[i];
        s += (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) s += (double)g[i] * g[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const double *part, int nparts, double *acc, int overwrite)
{
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) acc[0] = overwrite ? red[0] : acc[0] + red[0];
}

// torch.optim.Adam single-tensor update (no amsgrad, no weight decay):
//   m.lerp_(g, 1-b1); v = v*b2 + (1-b2) g*g; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// dev_scal != nullptr: step_size / inv_sqrt_bc2 come from device memory (written by adam_prepare_kernel), so that a
// captured hipGraph replays with the CURRENT step count and learning rate instead of the ones baked in at capture.
// zero_g: the gradient buffer is cleared in the same pass (the explicit training schedule accumulates into it next step;
// saves the separate 72 MB memset launch on the critical path)
__global__ __launch_bounds__(256) void adam_kernel(float *p, float *g, float *m, float *v, size_t n, const double *sumsq,
                                                   float max_norm, float gscale, float step_size, float b1, float b2,
                                                   float inv_sqrt_bc2, float eps, const float *dev_scal, int zero_g)
{
    if (dev_scal) {
        step_size = dev_scal[0];
        inv_sqrt_bc2 = dev_scal[1];
    }
    float coef = gscale;
    if (max_norm > 0.f && sumsq) {
        const float total = (float)sqrt(sumsq[0]) * gscale;
        coef *= fminf(max_norm / (total + 1e-6f), 1.0f);
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i] * coef;
        if (zero_g) g[i] = 0.f;
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    }
}

// The same update over contiguous chunks of STEM_ADAM_CHUNK parameters per workgroup, which also leaves max |p_new| of every chunk
// in bmax[4 * chunk .. 4 * chunk + 3] (one value per wavefront): the fp16 weight packing that follows an optimiser step (conv_f16x3.hip) takes its power-of-two scale from the
// maxima of the chunks a tensor touches instead of a reduction pass of its own (a chunk shared with a neighbouring tensor can
// only raise the bound).  Same arithmetic per element as adam_kernel.
__global__ __launch_bounds__(256) void adam_bmax_kernel(float *p, float *g, float *m, float *v, size_t n, const double *sumsq,
                                                        float max_norm, float gscale, float step_size, float b1, float b2,
                                                        float inv_sqrt_bc2, float eps, int zero_g, float *bmax)
{
    float coef = gscale;
    if (max_norm > 0.f && sumsq) {
        const float total = (float)sqrt(sumsq[0]) * gscale;
        coef *= fminf(max_norm / (total + 1e-6f), 1.0f);
    }
    const size_t base = (size_t)blockIdx.x * STEM_ADAM_CHUNK;
    float mx = 0.f;
    // four elements per thread and pass, all sixteen loads issued before the first use (the pass is HBM-bound: 28 B per parameter)
#pragma unroll 1
    for (int k0 = 0; k0 < STEM_ADAM_CHUNK / 256; k0 += 4) {
        float gi[4], mi[4], vi[4], pi[4];
        size_t idx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            idx[u] = base + (size_t)(k0 + u) * 256 + threadIdx.x;
            const bool ok = idx[u] < n;
            const size_t j = ok ? idx[u] : 0;
            gi[u] = ok ? g[j] : 0.f;
            mi[u] = ok ? m[j] : 0.f;
            vi[u] = ok ? v[j] : 0.f;
            pi[u] = ok ? p[j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (idx[u] >= n) continue;
            const float gs = gi[u] * coef;
            if (zero_g) g[idx[u]] = 0.f;
            const float mn = mi[u] + (gs - mi[u]) * (1.f - b1);
            const float vn = vi[u] * b2 + (1.f - b2) * gs * gs;
            m[idx[u]] = mn;
            v[idx[u]] = vn;
            const float pn = pi[u] - step_size * (mn / (sqrtf(vn) * inv_sqrt_bc2 + eps));
            p[idx[u]] = pn;
            mx = fmaxf(mx, fabsf(pn));
        }
    }
    mx = wave_max(mx);                                         // one slot per wavefront: no barrier in an HBM-bound pass
    if ((threadIdx.x & 63) == 0) bmax[blockIdx.x * 4 + (threadIdx.x >> 6)] = mx;
}

// One thread: ++step (device-resident), then the two scalars of this step's update exactly as stem_adam_step derives them
// on the host: step_size = lr / (1 - b1^t), inv_sqrt_bc2 = 1 / sqrt(1 - b2^t), in double.
__global__ void adam_prepare_kernel(long long *step, const float *lr, float b1, float b2, float *scal)
{
    const long long t = step[0] + 1;
    step[0] = t;
    const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
    scal[0] = (float)((double)lr[0] / bc1);
    scal[1] = (float)(1.0 / sqrt(bc2));
}

__global__ void counter_add_kernel(long long *ctr, long long inc) { ctr[0] += inc; }

// torch.nn.utils.clip_grad_norm_ on its own: g *= min(1, max_norm / (sqrt(sumsq) + 1e-6)), coefficient computed on the device
__global__ __launch_bounds__(256) void clip_scale_kernel(float *g, size_t n, const double *sumsq, float max_norm)
{
    const float coef = fminf(max_norm / ((float)sqrt(sumsq[0]) + 1e-6f), 1.0f);
    if (coef >= 1.0f) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) g[i] *= coef;
}

// y += a * x (gradient accumulation across the frames of a GOP in the data-parallel variable-rate loop)
__global__ __launch_bounds__(256) void axpy_kernel(float *y, const float *x, float a, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] += a * x[i];
}

}   // namespace

namespace {
int sumsq_impl(const float *g, size_t n, double *acc, int overwrite, void *stream);
}
STEM_EXPORT int stem_sumsq(const float *g, size_t n, double *acc, void *stream) { return sumsq_impl(g, n, acc, 0, stream); }
STEM_EXPORT int stem_sumsq_set(const float *g, size_t n, double *acc, void *stream) { return sumsq_impl(g, n, acc, 1, stream); }
namespace {
int sumsq_impl(const float *g, size_t n, double *acc, int overwrite, void *stream)
{
    STEM_CHECK_ARG(g && acc, "stem_sumsq: null pointer");
    if (n == 0) {
        if (overwrite && hipMemsetAsync(acc, 0, sizeof(double), (hipStream_t)stream) != hipSuccess) return -2;
        return 0;
    }
    const bool al = (((uintptr_t)g) & 15) == 0;
    const size_t n4 = al ? n / 4 : 0;
    size_t nb = cdivz(n4 ? n4 : 1, 256);
    if (nb > STEM_SUMSQ_SCRATCH) nb = STEM_SUMSQ_SCRATCH;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const f32x4 *>(g), g, n4, n, acc + 1);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, acc + 1, (int)nb, acc, overwrite);
    STEM_LAUNCH_CHECK("sumsq");
    return 0;
}
}   // namespace

STEM_EXPORT int stem_adam_step(float *p, const float *g, float *m, float *v, size_t n, const double *sumsq, float max_norm,
                               float gscale, float lr, float beta1, float beta2, float eps, int step, void *stream)
{
    STEM_CHECK_ARG(p && g && m && v && step >= 1, "stem_adam_step: bad arguments");
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    size_t nb = cdivz(n, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, const_cast<float *>(g), m, v, n, sumsq, max_norm,
                       gscale, (float)(lr / bc1), beta1, beta2, (float)(1.0 / sqrt(bc2)), eps, (const float *)nullptr, 0);
    STEM_LAUNCH_CHECK("adam");
    return 0;
}

STEM_EXPORT int stem_adam_step_zero(float *p, float *g, float *m, float *v, size_t n, const double *sumsq, float max_norm,
                                    float gscale, float lr, float beta1, float beta2, float eps, int step, void *stream)
{
    STEM_CHECK_ARG(p && g && m && v && step >= 1, "stem_adam_step_zero: bad arguments");
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    size_t nb = cdivz(n, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, sumsq, max_norm,
                       gscale, (float)(lr / bc1), beta1, beta2, (float)(1.0 / sqrt(bc2)), eps, (const float *)nullptr, 1);
    STEM_LAUNCH_CHECK("adam_zero");
    return 0;
}

STEM_EXPORT size_t stem_adam_chunk(void) { return STEM_ADAM_CHUNK; }

STEM_EXPORT int stem_adam_step_bmax(float *p, float *g, float *m, float *v, size_t n, const double *sumsq, float max_norm,
                                    float gscale, float lr, float beta1, float beta2, float eps, int step, int zero_grad, float *bmax,
                                    void *stream)
{
    STEM_CHECK_ARG(p && g && m && v && bmax && step >= 1, "stem_adam_step_bmax: bad arguments");
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_bmax_kernel, dim3((unsigned)cdivz(n, STEM_ADAM_CHUNK)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, sumsq,
                       max_norm, gscale, (float)(lr / bc1), beta1, beta2, (float)(1.0 / sqrt(bc2)), eps, zero_grad, bmax);
    STEM_LAUNCH_CHECK("adam_bmax");
    return 0;
}

STEM_EXPORT int stem_adam_step_dev(float *p, const float *g, float *m, float *v, size_t n, const double *sumsq, float max_norm,
                                   float gscale, const float *lr_dev, float beta1, float beta2, float eps, long long *step_dev,
                                   float *scal_dev, void *stream)
{
    STEM_CHECK_ARG(p && g && m && v && lr_dev && step_dev && scal_dev, "stem_adam_step_dev: null pointer");
    if (n == 0) return 0;
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, lr_dev, beta1, beta2, scal_dev);
    size_t nb = cdivz(n, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, const_cast<float *>(g), m, v, n, sumsq, max_norm,
                       gscale, 0.f, beta1, beta2, 0.f, eps, (const float *)scal_dev, 0);
    STEM_LAUNCH_CHECK("adam_dev");
    return 0;
}

STEM_EXPORT int stem_counter_add(long long *ctr, long long inc, void *stream)
{
    STEM_CHECK_ARG(ctr, "stem_counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, ctr, inc);
    STEM_LAUNCH_CHECK("counter_add");
    return 0;
}

STEM_EXPORT int stem_clip_scale(float *g, size_t n, const double *sumsq, float max_norm, void *stream)
{
    STEM_CHECK_ARG(g && sumsq && max_norm > 0.f, "stem_clip_scale: bad arguments");
    if (n == 0) return 0;
    size_t nb = cdivz(n, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(clip_scale_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, g, n, sumsq, max_norm);
    STEM_LAUNCH_CHECK("clip_scale");
    return 0;
}

STEM_EXPORT int stem_axpy(float *y, const float *x, float a, size_t n, void *stream)
{
    STEM_CHECK_ARG(y && x, "stem_axpy: null pointer");
    if (n == 0) return 0;
    size_t nb = cdivz(n, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, y, x, a, n);
    STEM_LAUNCH_CHECK("axpy");
    return 0;
}

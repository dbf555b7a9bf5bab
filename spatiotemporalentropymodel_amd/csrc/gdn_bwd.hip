// GDN / IGDN backward: composite of the igemm / wgrad kernels plus four small elementwise kernels.
//   n   = beta' + gamma' x^2                (igemm, GDN operand transforms, EPI_NORM)
//   g   = dL/dn = dy x (-1/2 n^-3/2 | +1/2 n^-1/2),   u = dy (n^-1/2 | n^1/2)        (prep, HBM-bound)
//   s   = g gamma'                          (igemm 1x1 with the reparametrised, transposed gamma)
//   dx  = u + 2 x s                         (finish, HBM-bound)
//   dgamma' = sum_pix g (x) x^2, dbeta' = sum_pix g      (wgrad with squared gather operand + its bias-gradient path)
//   dgamma, dbeta through out = max(p, bound)^2 - pedestal with the LowerBound pass-through rule
#include "stem_common.h"

namespace {

constexpr float kPed = 1.4551915228366852e-11f;     // 2^-36

__global__ void gdn_gammaT_kernel(const float *gamma, float *gT, int C)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * C) return;
    const int k = i / C, c = i - k * C;                 // gT[k][c] = gamma'[c][k]
    const float v = fmaxf(gamma[c * C + k], 3.814697265625e-06f);
    gT[i] = v * v - kPed;
}

// n (in) -> g (in place), u (out); all dense [npix][C]
__global__ void gdn_bwd_prep_kernel(float *n_g, float *u, const float *x, int ldx, const float *dy, int lddy, size_t npix, int C,
                                    int inverse)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * C) return;
    const size_t p = i / C;
    const int c = (int)(i - p * C);
    const float n = n_g[i], xv = x[p * ldx + c], dv = dy[p * lddy + c];
    const float rs = __builtin_amdgcn_rsqf(n);
    if (inverse) {
        n_g[i] = 0.5f * dv * xv * rs;
        u[i] = dv * n * rs;                              // dy * sqrt(n)
    } else {
        n_g[i] = -0.5f * dv * xv * rs * rs * rs;
        u[i] = dv * rs;
    }
}

__global__ void gdn_bwd_finish_kernel(const float *u, const float *s, const float *x, int ldx, float *dx, int lddx, size_t npix, int C)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * C) return;
    const size_t p = i / C;
    const int c = (int)(i - p * C);
    dx[p * lddx + c] = u[i] + 2.f * x[p * ldx + c] * s[i];
}

// d(raw) = 2 lb d(prime), passed iff raw >= bound or the gradient is negative (bound_ops.py:28-31)
__global__ void gdn_reparam_bwd_kernel(const float *raw, const float *dprime, float *draw, int n, float bound)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float r = raw[i], lb = fmaxf(r, bound), g = 2.f * lb * dprime[i];
    draw[i] = (r >= bound || g < 0.f) ? g : 0.f;
}

struct Layout {
    size_t n_g, u, s, gT, dgp, dbp, slabs, total;
    int splits;
};
Layout layout(int B, int H, int W, int C)
{
    Layout L;
    const size_t M = (size_t)B * H * W, act = (M * C + 3) / 4 * 4;
    L.splits = stem_wgrad_splits(B, H, W, C, C, 1, 1);
    size_t o = 0;
    L.n_g = o; o += act;
    L.u = o; o += act;
    L.s = o; o += act;
    L.gT = o; o += (size_t)C * C;
    L.dgp = o; o += (size_t)C * C;
    L.dbp = o; o += (size_t)(C + 3) / 4 * 4;
    L.slabs = o; o += stem_wgrad_workspace_elems(L.splits, C, C, 1, 1, (int)M);
    L.total = o;
    return L;
}
inline unsigned nb(size_t n) { return (unsigned)cdivz(n, 256); }

}   // namespace

STEM_EXPORT size_t stem_gdn_bwd_workspace_bytes(int B, int H, int W, int C)
{
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    return layout(B, H, W, C).total * sizeof(float);
}

STEM_EXPORT int stem_gdn_bwd(const float *x, int ldx, const float *dy, int lddy, const float *beta, const float *gamma,
                             float *dx, int lddx, float *dbeta, float *dgamma, int B, int H, int W, int C, int inverse,
                             float beta_min, void *ws, size_t ws_bytes, void *stream)
{
    STEM_CHECK_ARG(x && dy && beta && gamma && dx && dbeta && dgamma && ws, "stem_gdn_bwd: null pointer");
    STEM_CHECK_ARG(C % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0, "stem_gdn_bwd: channel count / pitches must be multiples of 4");
    const Layout L = layout(B, H, W, C);
    STEM_CHECK_ARG(ws_bytes >= L.total * sizeof(float), "stem_gdn_bwd: workspace too small (%zu < %zu)", ws_bytes, L.total * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    float *w = (float *)ws;
    const size_t M = (size_t)B * H * W;
    // n = beta' + gamma' x^2  (mode 2 of stem_gdn_fwd writes the denominator)
    if (int rc = stem_gdn_fwd(x, ldx, beta, gamma, w + L.n_g, C, B, H, W, C, 2, beta_min, stream)) return rc;
    hipLaunchKernelGGL(gdn_bwd_prep_kernel, dim3(nb(M * C)), dim3(256), 0, st, w + L.n_g, w + L.u, x, ldx, dy, lddy, M, C, inverse ? 1 : 0);
    STEM_LAUNCH_CHECK("gdn_bwd_prep");
    hipLaunchKernelGGL(gdn_gammaT_kernel, dim3(nb((size_t)C * C)), dim3(256), 0, st, gamma, w + L.gT, C);
    STEM_LAUNCH_CHECK("gdn_gammaT");
    // s[pix][k] = sum_i g[pix][i] gamma'[i][k]: a 1x1 convolution whose packed weight [k][i] is gamma'^T
    if (int rc = stem_conv2d_fwd(w + L.n_g, C, w + L.gT, nullptr, w + L.s, C, B, H, W, C, C, 1, 1, 1, 0, STEM_ACT_NONE, 0.f, nullptr, 0, stream))
        return rc;
    hipLaunchKernelGGL(gdn_bwd_finish_kernel, dim3(nb(M * C)), dim3(256), 0, st, w + L.u, w + L.s, x, ldx, dx, lddx, M, C);
    STEM_LAUNCH_CHECK("gdn_bwd_finish");
    // dgamma'[i][j] = sum_pix g_i x_j^2 ; dbeta'[i] = sum_pix g_i   (P = g, G = x squared while staged)
    if (int rc = stem_conv2d_wgrad(x, ldx, w + L.n_g, C, w + L.slabs, w + L.dbp, B, H, W, C, C, 1, 1, 1, 0, L.splits, STEM_WGRAD_SQUARE_G, stream))
        return rc;
    if (int rc = stem_unpack_wgrad(w + L.slabs, w + L.dgp, C, C, 1, 1, L.splits, 0, stream)) return rc;
    hipLaunchKernelGGL(gdn_reparam_bwd_kernel, dim3(nb((size_t)C * C)), dim3(256), 0, st, gamma, w + L.dgp, dgamma, C * C, 3.814697265625e-06f);
    STEM_LAUNCH_CHECK("gdn_reparam_bwd(gamma)");
    hipLaunchKernelGGL(gdn_reparam_bwd_kernel, dim3(nb(C)), dim3(256), 0, st, beta, w + L.dbp, dbeta, C,
                       (float)sqrt((double)beta_min + 1.4551915228366852e-11));
    STEM_LAUNCH_CHECK("gdn_reparam_bwd(beta)");
    return 0;
}

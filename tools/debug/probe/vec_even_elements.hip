// Probe for a hipcc (ROCm 7.2, gfx950) code-generation problem met in csrc/ar_persistent.hip: __builtin_bit_cast(float, v[i]) applied
// DIRECTLY to an element of an ext_vector_type value reads the wrong element (v[2] gives v[0]): a float4 built from elements 0 and 2 of
// two 16-byte loads came out as {lo[0], lo[0], hi[0], hi[0]}.  Neither the polling loop, nor the tag compares, nor the kind of load
// matter (variants 0-3 below all gather 1 1 3 3 5 5 7 7); copying the element to a scalar first is correct (variant 4: 1 2 3 4 ...).
//   hipcc --offload-arch=gfx950 -O3 vec_even_elements.hip -o vec_even_elements && ./vec_even_elements
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ void gather(const long long *words, int n, int tag, float *out)
{
    const int l4 = threadIdx.x * 4;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<long long *>(words), 0, n * 8, 0x00020000);
    // V = 0: as met in ar_persistent.hip (polling loop, tags checked); 1: no loop, no tag check; 2: elements through scalar floats first;
    // 3: plain global loads instead of buffer loads; 4: each element copied to a scalar `unsigned` before the bit_cast
    f32x4 x[3];
    long spins = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (t < 2 && t * 256 + l4 < n) {
                u32x4 lo, hi;
                if (V == 3) {
                    lo = *reinterpret_cast<const u32x4 *>(words + t * 256 + l4);
                    hi = *reinterpret_cast<const u32x4 *>(words + t * 256 + l4 + 2);
                } else {
                    lo = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (t * 256 + l4) * 8, 0, 16));
                    hi = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (t * 256 + l4) * 8 + 16, 0, 16));
                }
                if (V == 4) {            // the element copied to a scalar first: bit_cast of an rvalue
                    const unsigned a = lo[0], b = lo[2], c = hi[0], d = hi[2];
                    x[t] = f32x4{__builtin_bit_cast(float, a), __builtin_bit_cast(float, b), __builtin_bit_cast(float, c), __builtin_bit_cast(float, d)};
                } else if (V == 2) {
                    const float a = __builtin_bit_cast(float, lo[0]), b = __builtin_bit_cast(float, lo[2]), c = __builtin_bit_cast(float, hi[0]),
                                d = __builtin_bit_cast(float, hi[2]);
                    x[t][0] = a; x[t][1] = b; x[t][2] = c; x[t][3] = d;
                } else {
                    x[t] = f32x4{__builtin_bit_cast(float, lo[0]), __builtin_bit_cast(float, lo[2]), __builtin_bit_cast(float, hi[0]), __builtin_bit_cast(float, hi[2])};
                }
                if (V != 1) ok = ok && (int)lo[1] == tag && (int)lo[3] == tag && (int)hi[1] == tag && (int)hi[3] == tag;
            }
        if (V == 1 || __all(ok)) break;
        if (++spins > 1000) break;
    }
    for (int t = 0; t < 2; ++t)
        if (t * 256 + l4 < n)
            for (int c = 0; c < 4; ++c) out[t * 256 + l4 + c] = x[t][c];
}

int main()
{
    const int n = 384, tag = 7;
    std::vector<long long> h(n);
    for (int i = 0; i < n; ++i) {
        const float v = 1.0f + i;
        unsigned bits;
        std::memcpy(&bits, &v, 4);
        h[i] = ((long long)tag << 32) | bits;
    }
    long long *d;
    float *o;
    hipMalloc(&d, n * 8);
    hipMalloc(&o, 512 * 4);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    int worst = 0;
    for (int v = 0; v < 5; ++v) {
        hipMemset(o, 0, 512 * 4);
        if (v == 0) hipLaunchKernelGGL(gather<0>, dim3(1), dim3(64), 0, 0, d, n, tag, o);
        if (v == 1) hipLaunchKernelGGL(gather<1>, dim3(1), dim3(64), 0, 0, d, n, tag, o);
        if (v == 2) hipLaunchKernelGGL(gather<2>, dim3(1), dim3(64), 0, 0, d, n, tag, o);
        if (v == 3) hipLaunchKernelGGL(gather<3>, dim3(1), dim3(64), 0, 0, d, n, tag, o);
        if (v == 4) hipLaunchKernelGGL(gather<4>, dim3(1), dim3(64), 0, 0, d, n, tag, o);
        std::vector<float> got(512);
        hipMemcpy(got.data(), o, 512 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < n; ++i) bad += got[i] != 1.0f + i;
        printf("variant %d: first eight gathered: %g %g %g %g %g %g %g %g -> %s (%d of %d wrong)\n", v, got[0], got[1], got[2], got[3], got[4], got[5],
               got[6], got[7], bad ? "WRONG" : "ok", bad, n);
        worst += bad;
    }
    return worst != 0;
}

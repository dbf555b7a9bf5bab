// Probe for a hipcc (ROCm 7.2, gfx950) code-generation problem met in csrc/ar_persistent.hip: a float4 built from elements 0 and 2
// of two 16-byte buffer loads inside a polling loop came out as {lo[0], lo[0], hi[0], hi[0]}.  This file isolates the construct:
//   hipcc --offload-arch=gfx950 -O3 vec_even_elements.hip -o vec_even_elements && ./vec_even_elements
// prints the four floats every lane gathered; "ok" if they are {v0, v1, v2, v3}.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void gather(const long long *words, int n, int tag, float *out)
{
    const int l4 = threadIdx.x * 4;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<long long *>(words), 0, n * 8, 0x00020000);
    f32x4 x[3];
    long spins = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (t < 2 && t * 256 + l4 < n) {
                const u32x4 lo = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (t * 256 + l4) * 8, 0, 16));
                const u32x4 hi = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (t * 256 + l4) * 8 + 16, 0, 16));
                x[t] = f32x4{__builtin_bit_cast(float, lo[0]), __builtin_bit_cast(float, lo[2]), __builtin_bit_cast(float, hi[0]), __builtin_bit_cast(float, hi[2])};
                ok = ok && (int)lo[1] == tag && (int)lo[3] == tag && (int)hi[1] == tag && (int)hi[3] == tag;
            }
        if (__all(ok)) break;
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
    hipMemset(o, 0, 512 * 4);
    hipLaunchKernelGGL(gather, dim3(1), dim3(64), 0, 0, d, n, tag, o);
    std::vector<float> got(512);
    hipMemcpy(got.data(), o, 512 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += got[i] != 1.0f + i;
    printf("first eight gathered: %g %g %g %g %g %g %g %g -> %s (%d of %d wrong)\n", got[0], got[1], got[2], got[3], got[4], got[5], got[6], got[7],
           bad ? "WRONG" : "ok", bad, n);
    return bad != 0;
}

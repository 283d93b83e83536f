// micro-benchmark: f32 MFMA 16x16x4 + the VALU patterns of a screening sweep (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int V> __global__ __launch_bounds__(256, 4) void k(const float *in, float *out, int iters, int *cand)
{
    const int lane = threadIdx.x & 63;
    float a0 = in[lane], a1 = in[64 + lane], a2 = in[128 + lane], a3 = in[192 + lane];
    float b = in[256 + lane];
    const f4 zero = { 0, 0, 0, 0 };
    f4 acc0 = zero, acc1 = zero, acc2 = zero, acc3 = zero;
    float best[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) best[q] = V == 3 ? -1e30f : 1e30f;
    int hits = 0;
    for (int it = 0; it < iters; ++it) {
        if (V == 0) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b, acc3, 0, 0, 0);
        } else {
            f4 c[4];
            c[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, zero, 0, 0, 0);
            c[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, zero, 0, 0, 0);
            c[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b, zero, 0, 0, 0);
            c[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b, zero, 0, 0, 0);
            if (V == 2) {
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) best[rt * 4 + r] = fminf(best[rt * 4 + r], c[rt][r]);
            } else if (V == 3) {          // threshold test, rare hit path
                bool any = false;
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) any |= (c[rt][r] <= best[rt * 4 + r]);
                if (__builtin_expect(any, 0)) { hits++; atomicAdd(cand, 1); }
            }
            b += 1.0f;
        }
    }
    float s = acc0[0] + acc1[1] + acc2[2] + acc3[3] + hits;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += best[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V> int run(const char *name, int blocks, int threads, int iters, float *in, float *out, int *cand)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, in, out, iters, cand);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, in, out, iters, cand);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double waves = (double)blocks * threads / 64;
    double mfma = waves * iters * 4.0;
    double tf = mfma * 2048.0 / (ms * 1e-3) / 1e12;
    double cyc = (ms * 1e-3 * 2.4e9) / (mfma / 1024.0);
    printf("%-28s blocks=%5d thr=%4d  %8.3f ms  %7.2f TFLOP/s  ~%6.1f cyc/MFMA/SIMD@2.4GHz\n", name, blocks, threads, ms, tf, cyc);
    return 0;
}

int main()
{
    float *in, *out; int *cand;
    CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 8 * 1024 * 1024)); CK(hipMalloc(&cand, 4)); CK(hipMemset(cand, 0, 4));
    float h[512]; for (int i = 0; i < 512; ++i) h[i] = sinf(i * 0.37f) * 1000.0f;
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    const int it = 40000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        int blocks = 256 * wps;
        printf("--- %d wave(s) per SIMD\n", wps);
        run<0>("F0 f32 mfma only", blocks, 256, it, in, out, cand);
        run<2>("F2 f32 mfma + 4 min", blocks, 256, it, in, out, cand);
        run<3>("F3 f32 mfma + 4 cmp + branch", blocks, 256, it, in, out, cand);
    }
    return 0;
}

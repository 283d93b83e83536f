// micro-benchmarks for the fp64 MFMA nearest-neighbour inner loop (dev tool, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// V0: MFMA only, 4 independent accumulator chains
template <int V> __global__ __launch_bounds__(256) void k(const double *in, double *out, int iters)
{
    const int lane = threadIdx.x & 63;
    double a0 = in[lane], a1 = in[64 + lane], a2 = in[128 + lane], a3 = in[192 + lane];
    double b = in[256 + lane];
    const d4 zero = { 0, 0, 0, 0 };
    d4 acc0 = zero, acc1 = zero, acc2 = zero, acc3 = zero;
    double best[16]; int bt[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { best[q] = V == 8 ? 0.0 : 1e300; bt[q] = 0; }
    for (int it = 0; it < iters; ++it) {
        if (V == 0) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b, acc3, 0, 0, 0);
        } else {
            d4 c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, zero, 0, 0, 0);
            d4 c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, zero, 0, 0, 0);
            d4 c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b, zero, 0, 0, 0);
            d4 c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b, zero, 0, 0, 0);
            d4 c[4] = { c0, c1, c2, c3 };
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = rt * 4 + r;
                    if (V == 1) {           // cmp + select value + select index
                        bool lt = c[rt][r] < best[q];
                        best[q] = lt ? c[rt][r] : best[q];
                        bt[q] = lt ? it : bt[q];
                    } else if (V == 2) {    // min only
                        best[q] = fmin(best[q], c[rt][r]);
                    } else if (V == 8) {    // handled below (prefilter)
                    } else if (V == 4) {    // positive metric: u64 compare + 3 selects
                        unsigned long long nv = __double_as_longlong(c[rt][r]), ob = __double_as_longlong(best[q]);
                        bool lt = nv < ob;
                        best[q] = lt ? c[rt][r] : best[q];
                        bt[q] = lt ? it : bt[q];
                    } else if (V == 5) {    // positive metric: hi/lo 32-bit compares
                        unsigned long long nv = __double_as_longlong(c[rt][r]), ob = __double_as_longlong(best[q]);
                        unsigned nh = nv >> 32, nl = (unsigned)nv, oh = ob >> 32, ol = (unsigned)ob;
                        bool lt = (nh < oh) | ((nh == oh) & (nl < ol));
                        best[q] = lt ? c[rt][r] : best[q];
                        bt[q] = lt ? it : bt[q];
                    } else if (V == 6) {    // value-only u64 compare (index deferred to group end)
                        unsigned long long nv = __double_as_longlong(c[rt][r]), ob = __double_as_longlong(best[q]);
                        best[q] = nv < ob ? c[rt][r] : best[q];
                    } else if (V == 7) {    // value-only hi/lo compare
                        unsigned long long nv = __double_as_longlong(c[rt][r]), ob = __double_as_longlong(best[q]);
                        unsigned nh = nv >> 32, nl = (unsigned)nv, oh = ob >> 32, ol = (unsigned)ob;
                        bool lt = (nh < oh) | ((nh == oh) & (nl < ol));
                        best[q] = lt ? c[rt][r] : best[q];
                    } else if (V == 3) {    // min + index via compare-equal
                        double nb = fmin(best[q], c[rt][r]);
                        bt[q] = (nb != best[q]) ? it : bt[q];
                        best[q] = nb;
                    }
                }
            if (V == 8) {           // hi-word prefilter, wave-uniform skip of the exact update
                bool pass = false;
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        pass |= ((unsigned)(__double_as_longlong(c[rt][r]) >> 32) <= (unsigned)bt[rt * 4 + r]);
                if (__builtin_amdgcn_ballot_w64(pass) != 0) {
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int q = rt * 4 + r;
                            bool lt = c[rt][r] < best[q];
                            best[q] = lt ? c[rt][r] : best[q];
                            bt[q] = lt ? (int)(__double_as_longlong(c[rt][r]) >> 32) : bt[q];
                        }
                }
            }
            b += 1.0;      // keep the MFMA inputs changing
        }
    }
    double s = acc0[0] + acc1[1] + acc2[2] + acc3[3];
#pragma unroll
    for (int q = 0; q < 16; ++q) s += best[q] + bt[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V> int run(const char *name, int blocks, int threads, int iters, double *in, double *out)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double waves = (double)blocks * threads / 64;
    double mfma = waves * iters * 4.0;
    double tf = mfma * 2048.0 / (ms * 1e-3) / 1e12;
    // cycles per MFMA per SIMD assuming 2.4 GHz and all 1024 SIMDs busy
    double cyc = (ms * 1e-3 * 2.4e9) / (mfma / 1024.0);
    printf("%-28s blocks=%5d thr=%4d  %8.3f ms  %7.2f TFLOP/s  ~%6.1f cyc/MFMA/SIMD@2.4GHz\n", name, blocks, threads, ms, tf, cyc);
    return 0;
}

int main()
{
    double *in, *out;
    CK(hipMalloc(&in, 4096 * 8)); CK(hipMalloc(&out, 8 * 1024 * 1024));
    double h[512]; for (int i = 0; i < 512; ++i) h[i] = sin(i * 0.37) * 1000.0;
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    const int it = 20000;
    for (int wps = 2; wps <= 4; wps *= 2) {            // waves per SIMD: blocks of 256 threads = 1 wave/SIMD each
        int blocks = 256 * wps;
        printf("--- %d wave(s) per SIMD\n", wps);
        run<0>("V0 mfma only (acc chains)", blocks, 256, it, in, out);
        run<2>("V2 mfma + min", blocks, 256, it, in, out);
        run<3>("V3 mfma + min + idx(ne)", blocks, 256, it, in, out);
        run<1>("V1 mfma + cmp + 3 cndmask", blocks, 256, it, in, out);
        run<6>("V6 u64 cmp + 2 cndmask", blocks, 256, it, in, out);
        run<8>("V8 hi-word prefilter (never)", blocks, 256, it, in, out);
    }
    return 0;
}

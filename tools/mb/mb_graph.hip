// microbenchmark: 31 dependent short kernels on one stream, launched one by one vs replayed as a captured hipGraph.
// Host time per chain (time spent inside the launch calls) and wall time until the chain has finished.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void step_kernel(double *x, int k)
{
    // ~15 us of dependent work in one wave of every block
    double v = x[blockIdx.x * 64 + (threadIdx.x & 63)];
    for (int i = 0; i < 2500; ++i) v = v * 1.0000001 + (double)k * 1e-9;
    x[blockIdx.x * 64 + (threadIdx.x & 63)] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    double *x;
    (void)hipMalloc(&x, 512 * 64 * sizeof(double));
    (void)hipMemset(x, 0, 512 * 64 * sizeof(double));
    hipStream_t st;
    (void)hipStreamCreate(&st);
    const int N = 31, R = 200;
    for (int w = 0; w < 3; ++w) for (int k = 0; k < N; ++k) hipLaunchKernelGGL(step_kernel, dim3(313), dim3(256), 0, st, x, k);
    (void)hipStreamSynchronize(st);
    double host = 0, t0 = now();
    for (int r = 0; r < R; ++r) {
        double a = now();
        for (int k = 0; k < N; ++k) hipLaunchKernelGGL(step_kernel, dim3(313), dim3(256), 0, st, x, k);
        host += now() - a;
        (void)hipStreamSynchronize(st);
    }
    printf("plain launches : host %.1f us per chain of %d, wall %.1f us per chain\n", host / R * 1e6, N, (now() - t0) / R * 1e6);
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int k = 0; k < N; ++k) hipLaunchKernelGGL(step_kernel, dim3(313), dim3(256), 0, st, x, k);
    (void)hipStreamEndCapture(st, &g);
    double ti = now();
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    printf("instantiate: %s, %.1f us\n", hipGetErrorString(e), (now() - ti) * 1e6);
    for (int w = 0; w < 3; ++w) (void)hipGraphLaunch(ge, st);
    (void)hipStreamSynchronize(st);
    host = 0; t0 = now();
    for (int r = 0; r < R; ++r) {
        double a = now();
        (void)hipGraphLaunch(ge, st);
        host += now() - a;
        (void)hipStreamSynchronize(st);
    }
    printf("graph replay   : host %.1f us per chain of %d, wall %.1f us per chain\n", host / R * 1e6, N, (now() - t0) / R * 1e6);
    // three chains side by side (three streams), as the registration batch runs them
    hipStream_t s3[3]; hipGraphExec_t g3[3];
    for (int i = 0; i < 3; ++i) { (void)hipStreamCreate(&s3[i]); (void)hipGraphInstantiate(&g3[i], g, nullptr, nullptr, 0); }
    for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(g3[i], s3[i]);
    (void)hipDeviceSynchronize();
    t0 = now(); host = 0;
    for (int r = 0; r < R; ++r) {
        double a = now();
        for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(g3[i], s3[i]);
        host += now() - a;
        (void)hipDeviceSynchronize();
    }
    printf("3 graphs       : host %.1f us per 3 chains, wall %.1f us\n", host / R * 1e6, (now() - t0) / R * 1e6);
    t0 = now(); host = 0;
    for (int r = 0; r < R; ++r) {
        double a = now();
        for (int k = 0; k < N; ++k) for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(step_kernel, dim3(313), dim3(256), 0, s3[i], x + i * 64 * 512 / 4, k);
        host += now() - a;
        (void)hipDeviceSynchronize();
    }
    printf("3 plain chains : host %.1f us per 3 chains, wall %.1f us\n", host / R * 1e6, (now() - t0) / R * 1e6);
    return 0;
}

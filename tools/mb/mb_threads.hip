// microbenchmark: launch throughput of one host thread vs two (each with its own stream) -- does the HIP runtime let two
// threads enqueue short kernels side by side?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
__global__ void tiny_kernel(double *x, int k) { if (threadIdx.x == 0) x[blockIdx.x] += k; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void worker(double *x, int launches, double *seconds)
{
    hipStream_t st;
    (void)hipStreamCreate(&st);
    for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(tiny_kernel, dim3(64), dim3(64), 0, st, x, k);
    (void)hipStreamSynchronize(st);
    const double t0 = now();
    for (int k = 0; k < launches; ++k) hipLaunchKernelGGL(tiny_kernel, dim3(64), dim3(64), 0, st, x, k);
    (void)hipStreamSynchronize(st);
    *seconds = now() - t0;
}
int main()
{
    double *x;
    (void)hipMalloc(&x, 4096 * sizeof(double));
    (void)hipMemset(x, 0, 4096 * sizeof(double));
    const int L = 20000;
    double t1 = 0, ta = 0, tb = 0;
    worker(x, L, &t1);
    printf("one thread : %.2f us per launch (%d launches)\n", t1 / L * 1e6, L);
    std::thread A(worker, x, L, &ta), B(worker, x + 2048, L, &tb);
    A.join(); B.join();
    printf("two threads: %.2f / %.2f us per launch each -> %.2f us per launch overall\n", ta / L * 1e6, tb / L * 1e6, (ta > tb ? ta : tb) / (2 * L) * 1e6);
    return 0;
}

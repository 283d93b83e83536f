// How many kernel launches per second can T host threads (one stream each) push through the HIP runtime, and what does the
// same chain cost as one hipGraph launch?  Usage: launch_rate [kernels_per_chain=130] [chains=200]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void tiny_kernel(int *p, int v)
{
    if (p && threadIdx.x == 0 && blockIdx.x == 0 && v < 0) *p = v;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double run(int threads, int chain, int chains, bool graph, int blocks)
{
    std::vector<std::thread> th;
    auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; ++t)
        th.emplace_back([=]() {
            hipStream_t st;
            CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            int *d = nullptr;
            CK(hipMalloc(&d, 4));
            hipGraphExec_t exec = nullptr;
            if (graph) {
                hipGraph_t g;
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                for (int k = 0; k < chain; ++k) tiny_kernel<<<blocks, 256, 0, st>>>(d, k);
                CK(hipStreamEndCapture(st, &g));
                CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
            }
            for (int c = 0; c < chains; ++c) {
                if (graph) CK(hipGraphLaunch(exec, st));
                else for (int k = 0; k < chain; ++k) tiny_kernel<<<blocks, 256, 0, st>>>(d, k);
                if ((c & 3) == 3) CK(hipStreamSynchronize(st));          // a frame loop syncs now and then
            }
            CK(hipStreamSynchronize(st));
        });
    for (auto &x : th) x.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return (double)threads * chain * chains / s;
}

int main(int argc, char **argv)
{
    const int chain = argc > 1 ? atoi(argv[1]) : 130, chains = argc > 2 ? atoi(argv[2]) : 200;
    CK(hipSetDevice(0));
    (void)run(1, chain, 20, false, 1);
    for (int blocks : { 1, 500 })
        for (int graph = 0; graph < 2; ++graph)
            for (int threads : { 1, 2, 4, 8 }) {
                const double r = run(threads, chain, chains, graph != 0, blocks);
                printf("%s blocks %3d threads %d: %8.0f kernels/s total  (%.2f us per kernel per thread, chain of %d = %.3f ms)\n", graph ? "graph " : "stream", blocks, threads, r,
                       1e6 * threads / r, chain, 1e3 * chain * threads / r);
            }
    return 0;
}

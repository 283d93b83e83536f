// microbenchmark: hipcub radix sort pairs, u32 vs u64 keys at the sizes the voxel filter sorts
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <cstdint>
template <class K> static void run(int n, int end_bit, const char *name)
{
    std::vector<K> h(n);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (K)(s & ((end_bit >= 64 ? ~0ull : ((1ull << end_bit) - 1)))); }
    K *ki, *ko; int *vi, *vo;
    hipMalloc(&ki, n * sizeof(K)); hipMalloc(&ko, n * sizeof(K)); hipMalloc(&vi, n * 4); hipMalloc(&vo, n * 4);
    hipMemcpy(ki, h.data(), n * sizeof(K), hipMemcpyHostToDevice);
    size_t bytes = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, ki, ko, vi, vo, n, 0, end_bit);
    void *tmp; hipMalloc(&tmp, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipcub::DeviceRadixSort::SortPairs(tmp, bytes, ki, ko, vi, vo, n, 0, end_bit);
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) hipcub::DeviceRadixSort::SortPairs(tmp, bytes, ki, ko, vi, vo, n, 0, end_bit);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s n=%7d bits=%2d  %.1f us per sort\n", name, n, end_bit, ms / 20 * 1e3);
    hipFree(ki); hipFree(ko); hipFree(vi); hipFree(vo); hipFree(tmp);
}
// rocPRIM called directly with the merge-sort size limit set to 0: Onesweep (one scatter kernel per 8-bit digit) also below 1M keys
using onesweep_cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;
template <class K> static void run_onesweep(int n, int end_bit, const char *name)
{
    std::vector<K> h(n);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (K)(s & ((end_bit >= 64 ? ~0ull : ((1ull << end_bit) - 1)))); }
    K *ki, *ko; int *vi, *vo;
    hipMalloc(&ki, n * sizeof(K)); hipMalloc(&ko, n * sizeof(K)); hipMalloc(&vi, n * 4); hipMalloc(&vo, n * 4);
    hipMemcpy(ki, h.data(), n * sizeof(K), hipMemcpyHostToDevice);
    size_t bytes = 0;
    rocprim::radix_sort_pairs<onesweep_cfg>(nullptr, bytes, ki, ko, vi, vo, (size_t)n, 0u, (unsigned)end_bit, (hipStream_t)0);
    void *tmp; hipMalloc(&tmp, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) rocprim::radix_sort_pairs<onesweep_cfg>(tmp, bytes, ki, ko, vi, vo, (size_t)n, 0u, (unsigned)end_bit, (hipStream_t)0);
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) rocprim::radix_sort_pairs<onesweep_cfg>(tmp, bytes, ki, ko, vi, vo, (size_t)n, 0u, (unsigned)end_bit, (hipStream_t)0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<K> out(n); hipMemcpy(out.data(), ko, n * sizeof(K), hipMemcpyDeviceToHost);
    bool ok = true; for (int i = 1; i < n; ++i) if (out[i - 1] > out[i]) { ok = false; break; }
    printf("%-28s n=%7d bits=%2d  %.1f us per sort  sorted=%d tmp=%zu\n", name, n, end_bit, ms / 20 * 1e3, (int)ok, bytes);
    hipFree(ki); hipFree(ko); hipFree(vi); hipFree(vo); hipFree(tmp);
}
int main()
{
    for (int n : { 31000, 90000, 340000 }) {
        run<uint64_t>(n, 64, "u64 keys hipcub");
        run_onesweep<uint64_t>(n, 64, "u64 keys onesweep");
        run_onesweep<uint64_t>(n, 40, "u64 keys onesweep");
        run_onesweep<uint64_t>(n, 24, "u64 keys onesweep");
        run_onesweep<uint32_t>(n, 32, "u32 keys onesweep");
        run_onesweep<uint32_t>(n, 24, "u32 keys onesweep");
        run_onesweep<uint32_t>(n, 16, "u32 keys onesweep");
    }
    for (int n : { 31000, 64000, 283000, 1000000 }) {
        run<uint64_t>(n, 63, "u64 keys");
        run<uint64_t>(n, 40, "u64 keys");
        run<uint32_t>(n, 32, "u32 keys");
        run<uint32_t>(n, 24, "u32 keys");
        run<uint32_t>(n, 16, "u32 keys");
    }
    return 0;
}

// Radix sort of the reprojection pass's (pixel, item) pairs: how the digit width of rocPRIM's onesweep changes the time.
// 8 294 400 pairs, keys uniform below 1922 * 1082 (21 bits).   hipcc -O3 --offload-arch=gfx950 -w sort_bits.hip -o sort_bits
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>

template <unsigned BITS, unsigned BS, unsigned IPT, rocprim::block_radix_rank_algorithm ALG = rocprim::block_radix_rank_algorithm::default_algorithm>
static void run(const uint32_t* k_in, uint32_t* k_out, const uint32_t* v_in, uint32_t* v_out, size_t n, const char* name)
{
    using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                           rocprim::radix_sort_onesweep_config<rocprim::kernel_config<BS, IPT>, rocprim::kernel_config<BS, IPT>, BITS, ALG>>;
    size_t bytes = 0;
    if (rocprim::radix_sort_pairs<cfg>(nullptr, bytes, k_in, k_out, v_in, v_out, n, 0, 21) != hipSuccess) { printf("%s: size query failed\n", name); return; }
    void* tmp = nullptr;
    hipMalloc(&tmp, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) rocprim::radix_sort_pairs<cfg>(tmp, bytes, k_in, k_out, v_in, v_out, n, 0, 21);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; i++) rocprim::radix_sort_pairs<cfg>(tmp, bytes, k_in, k_out, v_in, v_out, n, 0, 21);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint32_t> h(1000);
    hipMemcpy(h.data(), k_out, 4000, hipMemcpyDeviceToHost);
    bool ok = true;
    for (int i = 1; i < 1000; i++) ok &= h[i - 1] <= h[i];
    printf("%-28s %8.1f us per sort  (temp %zu KB, head sorted: %d)\n", name, ms / reps * 1e3, bytes >> 10, (int)ok);
    hipFree(tmp);
}

int main()
{
    const size_t n = 8294400;
    std::vector<uint32_t> k(n), v(n);
    std::mt19937 g(7);
    for (size_t i = 0; i < n; i++) { k[i] = g() % (1922u * 1082u); v[i] = (uint32_t)i; }
    uint32_t *k_in, *k_out, *v_in, *v_out;
    hipMalloc(&k_in, 4 * n); hipMalloc(&k_out, 4 * n); hipMalloc(&v_in, 4 * n); hipMalloc(&v_out, 4 * n);
    hipMemcpy(k_in, k.data(), 4 * n, hipMemcpyHostToDevice);
    hipMemcpy(v_in, v.data(), 4 * n, hipMemcpyHostToDevice);
    {
        size_t bytes = 0;
        rocprim::radix_sort_pairs(nullptr, bytes, k_in, k_out, v_in, v_out, n, 0, 21);
        void* tmp; hipMalloc(&tmp, bytes);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; i++) rocprim::radix_sort_pairs(tmp, bytes, k_in, k_out, v_in, v_out, n, 0, 21);
        hipEventRecord(e0);
        for (int i = 0; i < 20; i++) rocprim::radix_sort_pairs(tmp, bytes, k_in, k_out, v_in, v_out, n, 0, 21);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.1f us per sort\n", "default config", ms / 20 * 1e3);
    }
    constexpr auto M = rocprim::block_radix_rank_algorithm::match;
    run<7, 256, 12>(k_in, k_out, v_in, v_out, n, "7 bits, 256 x 12");
    run<8, 256, 12>(k_in, k_out, v_in, v_out, n, "8 bits, 256 x 12");
    run<8, 256, 16>(k_in, k_out, v_in, v_out, n, "8 bits, 256 x 16");
    run<7, 256, 12, M>(k_in, k_out, v_in, v_out, n, "7 bits, 256 x 12, match");
    run<8, 256, 12, M>(k_in, k_out, v_in, v_out, n, "8 bits, 256 x 12, match");
    run<8, 512, 12, M>(k_in, k_out, v_in, v_out, n, "8 bits, 512 x 12, match");
#ifdef TRY11
    run<11, 256, 12, M>(k_in, k_out, v_in, v_out, n, "11 bits, 256 x 12, match");
    run<11, 512, 12, M>(k_in, k_out, v_in, v_out, n, "11 bits, 512 x 12, match");
    run<11, 1024, 8, M>(k_in, k_out, v_in, v_out, n, "11 bits, 1024 x 8, match");
#endif
    return 0;
}

// radix_bits.hip — how long does rocPRIM's onesweep radix sort take on the render cache's (pixel, item) pairs
// (8 294 400 pairs, 21 key bits) with 7 / 8 / 11 bits per pass?  (profiles/r03_rc_reproject.md)
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/radix_bits tools/ubench/radix_bits.hip && /tmp/radix_bits
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class Config>
static void run(const char* name, const uint32_t* k, uint32_t* ko, const uint32_t* v, uint32_t* vo, size_t n, unsigned bits,
                const std::vector<uint32_t>& want)
{
    size_t bytes = 0;
    CHECK((rocprim::radix_sort_pairs<Config>(nullptr, bytes, k, ko, v, vo, n, 0, bits, 0)));
    void* tmp; CHECK(hipMalloc(&tmp, bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; w++) CHECK((rocprim::radix_sort_pairs<Config>(tmp, bytes, k, ko, v, vo, n, 0, bits, 0)));
    CHECK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int r = 0; r < reps; r++) CHECK((rocprim::radix_sort_pairs<Config>(tmp, bytes, k, ko, v, vo, n, 0, bits, 0)));
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint32_t> got(n);
    CHECK(hipMemcpy(got.data(), vo, n * 4, hipMemcpyDeviceToHost));
    printf("%-28s %8.1f us per sort   %s\n", name, ms / reps * 1e3, got == want ? "stable order ok" : "WRONG ORDER");
    CHECK(hipFree(tmp));
}

int main()
{
    const size_t n = 1920u * 1080u * 4u;
    const uint32_t cells = 1922u * 1082u;
    std::vector<uint32_t> hk(n), hv(n);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        hk[i] = (s >> 20) % 5 == 0 ? 0xffffffffu : (uint32_t)((s >> 24) % cells);
        hv[i] = (uint32_t)i;
    }
    // the stable order, by a counting sort on the host (the no-pixel key sorts last: its low 21 bits are all ones)
    std::vector<uint32_t> want(n);
    {
        std::vector<uint32_t> cnt((1u << 21) + 1, 0);
        for (size_t i = 0; i < n; i++) cnt[(hk[i] & 0x1fffffu) + 1]++;
        for (size_t b = 0; b < (1u << 21); b++) cnt[b + 1] += cnt[b];
        for (size_t i = 0; i < n; i++) want[cnt[hk[i] & 0x1fffffu]++] = (uint32_t)i;
    }
    uint32_t *k, *ko, *v, *vo;
    CHECK(hipMalloc(&k, n * 4)); CHECK(hipMalloc(&ko, n * 4)); CHECK(hipMalloc(&v, n * 4)); CHECK(hipMalloc(&vo, n * 4));
    CHECK(hipMemcpy(k, hk.data(), n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(v, hv.data(), n * 4, hipMemcpyHostToDevice));
    using namespace rocprim;
    run<default_config>("default", k, ko, v, vo, n, 21, want);
#define ONESWEEP(BITS, BLOCK, IPT) radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<256, 12>, kernel_config<BLOCK, IPT>, BITS, block_radix_rank_algorithm::match>>
    run<ONESWEEP(7, 512, 15)>("7 bits, 512 x 15", k, ko, v, vo, n, 21, want);
    run<ONESWEEP(8, 512, 15)>("8 bits, 512 x 15", k, ko, v, vo, n, 21, want);
#ifdef WIDE
    run<ONESWEEP(11, 512, 15)>("11 bits, 512 x 15", k, ko, v, vo, n, 21, want);
    run<ONESWEEP(11, 1024, 12)>("11 bits, 1024 x 12", k, ko, v, vo, n, 21, want);
#endif
    return 0;
}

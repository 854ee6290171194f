// Dependent random loads under the gather's kind of load: how long is one memory round trip on this machine when N
// independent walks each chase through their own region of a large allocation?
//   ./chase TOTAL_GB CHAINS ITERS [LINES_PER_TRIP] [SECOND_ARRAY] [STORES_PER_TRIP] [STRIDE_BYTES] [HOT_BYTES]
// Every lane is one chain (64 per wavefront) with its own region; a trip loads LINES_PER_TRIP consecutive 64-byte lines
// at a pseudo-random 256-byte-aligned offset of the region (and, with SECOND_ARRAY = 1, one more line at the same
// relative offset in a second allocation: a record whose parts live in two arrays, like NodeStats / NodeKids), then
// writes STORES_PER_TRIP (<= 2) words into the record just read; the next offset depends on the loaded data.
// STRIDE_BYTES: distance between the chains' regions (0: TOTAL / CHAINS); HOT_BYTES: the walk stays in the first
// HOT_BYTES of its region (0: all of it) -- do same-index records of regularly spaced arenas meet on a memory channel?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(64) k_chase(uint4* a, const uint4* b, size_t region_bytes, size_t hot_bytes, uint32_t iters,
                                               int lines, int second, int stores, uint32_t* sink) {
    const size_t chain = (size_t)blockIdx.x * 64 + threadIdx.x;
    char* base = (char*)a + chain * region_bytes;
    const char* base2 = (const char*)b + chain * (region_bytes / 2);
    const uint32_t slots = (uint32_t)(hot_bytes / 256);  // (hot_bytes <= region_bytes: every access stays inside the region)
    uint32_t x = (uint32_t)chain * 2654435761u + 12345u, acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        const uint32_t s = (uint32_t)(((uint64_t)(x >> 4) * slots) >> 28) % slots;
        char* p = base + (size_t)s * 256;
        uint32_t v = 0;
        for (int l = 0; l < lines; ++l) v += ((const uint4*)(p + 64 * l))->x;
        if (second) v += ((const uint4*)(base2 + (size_t)s * 128))->x;
        for (int w = 0; w < stores; ++w) *(uint32_t*)(p + 128 * w + 16) = 0u;  // (zero: the data the walk reads stays zero)
        acc += v;
        x += v;  // (the data is zero: the walk does not change, the dependency stays)
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

int main(int argc, char** argv) {
    const double total_gb = argc > 1 ? atof(argv[1]) : 8.0;
    const size_t chains = argc > 2 ? (size_t)atoll(argv[2]) / 64 * 64 : 65536;
    const uint32_t iters = argc > 3 ? (uint32_t)atoi(argv[3]) : 1000;
    int lines = argc > 4 ? atoi(argv[4]) : 1;
    const int second = argc > 5 ? atoi(argv[5]) : 0;
    int stores = argc > 6 ? atoi(argv[6]) : 0;
    const size_t stride = argc > 7 ? (size_t)atoll(argv[7]) : 0;
    size_t hot = argc > 8 ? (size_t)atoll(argv[8]) : 0;
    if (lines < 1) lines = 1;
    if (lines > 4) lines = 4;
    if (stores > 2) stores = 2;
    size_t region = (stride ? stride : (size_t)(total_gb * 1e9 / (double)chains)) / 256 * 256;
    if (region < 1024) region = 1024;
    if (hot == 0 || hot > region) hot = region;
    hot = hot / 256 * 256;
    uint4 *a = nullptr, *b = nullptr;
    uint32_t* sink = nullptr;
    CHECK(hipMalloc(&a, region * chains));
    CHECK(hipMemset(a, 0, region * chains));
    CHECK(hipMalloc(&b, region / 2 * chains + 256));
    CHECK(hipMemset(b, 0, region / 2 * chains + 256));
    CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_chase, dim3((unsigned)(chains / 64)), dim3(64), 0, 0, a, b, region, hot, 64u, lines, second, stores, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_chase, dim3((unsigned)(chains / 64)), dim3(64), 0, 0, a, b, region, hot, iters, lines, second, stores, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double waves_per_simd = (double)chains / 64.0 / 1024.0;
    const double gens = waves_per_simd > 8.0 ? waves_per_simd / 8.0 : 1.0;  // (8 wavefronts per SIMD resident at most)
    printf("chains %zu (%.2f wavefronts/SIMD) stride %zu B hot %zu B lines %d second %d stores %d iters %u: %.3f ms, %.3f us per trip per "
           "chain, %.2f G trips/s, %.1f GB/s of lines read\n",
           chains, waves_per_simd, region, hot, lines, second, stores, iters, ms, ms * 1e3 / iters / gens,
           (double)chains * iters / ms / 1e6, (double)chains * iters * (lines + second) * 64 / ms / 1e6);
    return 0;
}

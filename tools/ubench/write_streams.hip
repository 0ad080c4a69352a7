// Microbenchmark: what do the list writes of the table-gradient scatter's first kernel (k_grid_bwd_bin) cost by PATTERN?
//
// Every workgroup (512 threads) plays one K1 tile of a fine level: 2048 twelve-byte items. The item VALUES and the workgroup
// count are the same in every mode; only the addresses differ:
//   bucketed   K1's pattern: 256 bucket lists per level `cap` items apart, the tile appends a run of 8 items (96 bytes, not
//              aligned to anything) to each; run position = tile index (no atomics: the reservation is not what is measured)
//   bucketedN  the same with N x as many items per run and 1/N of the buckets visited by a tile (what a kernel that keeps
//              per-bucket remainders in LDS across N tiles would write): runs of 96 N bytes
//   aligned32  runs of 32 items = 384 bytes = three whole 128-byte lines, 384-byte aligned, 64 buckets per tile
//   dense      the tile's 24 KB written contiguously (each workgroup its own region): the write rate itself
//   granules   384-byte aligned granules at pseudo-random positions of ONE dense pool shared by everybody (a bump-allocated
//              chunk pool): whole lines, no stream structure
// and by MAPPING of levels to XCDs: `paired` = the product's (XCD k works on levels 15-k and k: two level regions per XCD),
// `flat` = every XCD works on all 16 levels at once.
//   hipcc --offload-arch=gfx950 -O3 -o write_streams.bin write_streams.hip && ./write_streams.bin [tiles_per_level=6400]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

constexpr uint32_t kThreads = 512, kItems = 2048, kLevels = 16, kBuckets = 256;

struct Item { uint32_t a, b, c; };

enum Mode { BUCKETED = 0, ALIGNED32 = 1, DENSE = 2, GRANULES = 3 };

// run: items per (tile, bucket) visit; a tile visits kItems / run buckets, rotating through the 256 with the tile index
// NT: 0 = plain stores, 1 = __builtin_nontemporal_store (global_store ... nt), 2 = sc1 (system-scope write-through hint), 3 = sc0 sc1 nt
template <int MODE, int NT = 0>
__global__ __launch_bounds__(kThreads) void k_write(Item* __restrict__ out, uint32_t tiles, uint64_t cap, uint32_t run, int flat,
                                                     uint64_t pool_granules) {
    // (level, tile) of this workgroup
    const uint32_t xcd = blockIdx.x & 7u, local = blockIdx.x >> 3;
    uint32_t level, tile;
    if (flat) {
        level = local % kLevels; tile = (local / kLevels) * 8u + xcd;
    } else {
        level = local < tiles ? 15u - xcd : xcd; tile = local < tiles ? local : local - tiles;
    }
    if (tile >= tiles) return;
    const uint64_t level_base = (uint64_t)level * kBuckets * cap;
#pragma unroll
    for (uint32_t i = 0; i < kItems / kThreads; i++) {
        const uint32_t k = i * kThreads + threadIdx.x;
        Item it = {k ^ blockIdx.x, level, tile};
        uint64_t dst;
        if (MODE == DENSE) {
            dst = ((uint64_t)level * tiles + tile) * kItems + k;
        } else if (MODE == GRANULES) {
            const uint64_t g = ((uint64_t)level * tiles + tile) * (kItems / 32) + (k >> 5);
            const uint64_t pos = (g * 2654435761ull) % pool_granules;   // odd multiplier, pool_granules a power of two: a permutation
            dst = pos * 32 + (k & 31u);
        } else {
            const uint32_t visits = kItems / run;                 // buckets this tile writes to
            const uint32_t v = k / run, j = k - v * run;
            const uint32_t bucket = (v + tile * visits) % kBuckets;
            const uint64_t round = ((uint64_t)tile * visits) / kBuckets;   // how many runs this bucket has received before
            uint64_t slot = round * run + j;
            if (MODE == BUCKETED) slot += 3;                      // lists start wherever: not line aligned
            dst = level_base + (uint64_t)bucket * cap + slot;
        }
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
        const u32x3 v3 = {it.a, it.b, it.c};
        if (NT == 0) out[dst] = it;
        else if (NT == 1) asm volatile("global_store_dwordx3 %0, %1, off nt" :: "v"(out + dst), "v"(v3) : "memory");
        else if (NT == 2) asm volatile("global_store_dwordx3 %0, %1, off sc1" :: "v"(out + dst), "v"(v3) : "memory");
        else asm volatile("global_store_dwordx3 %0, %1, off sc0 sc1 nt" :: "v"(out + dst), "v"(v3) : "memory");
    }
}

template <int MODE, int NT = 0>
float run_mode(Item* buf, uint32_t tiles, uint64_t cap, uint32_t run, int flat, uint64_t pool_granules) {
    const uint32_t grid = 2 * tiles * 8;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_write<MODE, NT>), dim3(grid), dim3(kThreads), 0, 0, buf, tiles, cap, run, flat, pool_granules);
    hipEventRecord(a);
    const int n = 5;
    for (int i = 0; i < n; i++) hipLaunchKernelGGL((k_write<MODE, NT>), dim3(grid), dim3(kThreads), 0, 0, buf, tiles, cap, run, flat, pool_granules);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / n;
}

int main(int argc, char** argv) {
    const uint32_t tiles = argc > 1 ? (uint32_t)atoi(argv[1]) : 6400u;      // per level (3.27 M points / 512)
    const uint64_t per_level = (uint64_t)tiles * kItems;
    uint64_t cap = (per_level / kBuckets) * 5 / 4 + 256;                     // the product's list capacity: uniform share + 25 %
    cap = (cap + 31) / 32 * 32;                                              // (whole granules, so that ALIGNED32 is 384-byte aligned)
    uint64_t pool = 1;
    while (pool < per_level * kLevels / 32) pool <<= 1;
    const uint64_t items = (uint64_t)kLevels * kBuckets * cap > pool * 32 ? (uint64_t)kLevels * kBuckets * cap : pool * 32;
    Item* buf = nullptr;
    if (hipMalloc(&buf, items * sizeof(Item)) != hipSuccess) { printf("hipMalloc of %.2f GB failed\n", items * 12 / 1e9); return 1; }
    const double gb = (double)per_level * kLevels * sizeof(Item) / 1e9;
    printf("tiles per level %u, 16 levels, %.3f GB of items per launch, lists %.1f KB apart, buffer %.2f GB\n", tiles, gb, cap * 12 / 1e3,
           items * 12 / 1e9);
    for (int flat = 0; flat < 2; flat++) {
        const char* map = flat ? "flat  " : "paired";
        struct { const char* name; int mode; uint32_t run; } rows[] = {
            {"bucketed   run   8 ( 96 B)", BUCKETED, 8},  {"bucketed   run  16 (192 B)", BUCKETED, 16}, {"bucketed   run  32 (384 B)", BUCKETED, 32},
            {"bucketed   run 128 (1.5 KB)", BUCKETED, 128}, {"aligned32  run  32 (384 B)", ALIGNED32, 32}, {"aligned32  run  64 (768 B)", ALIGNED32, 64},
            {"dense      24 KB per tile", DENSE, 0},      {"granules   384 B anywhere", GRANULES, 0},
        };
        if (argc > 2) {   // cache-policy variants of K1's pattern (any second argument)
            const char* pol[] = {"plain", "nt", "sc1", "sc0 sc1 nt"};
            for (uint32_t run : {8u, 16u, 32u}) {
                float ms[4] = {run_mode<BUCKETED, 0>(buf, tiles, cap, run, flat, pool), run_mode<BUCKETED, 1>(buf, tiles, cap, run, flat, pool),
                               run_mode<BUCKETED, 2>(buf, tiles, cap, run, flat, pool), run_mode<BUCKETED, 3>(buf, tiles, cap, run, flat, pool)};
                for (int q = 0; q < 4; q++) printf("%s  bucketed run %3u (%4u B)  stores %-11s %8.1f us  %6.2f TB/s\n", map, run, run * 12, pol[q], ms[q] * 1e3, gb / ms[q]);
            }
            continue;
        }
        for (auto& r : rows) {
            float ms = 0;
            switch (r.mode) {
                case BUCKETED: ms = run_mode<BUCKETED>(buf, tiles, cap, r.run, flat, pool); break;
                case ALIGNED32: ms = run_mode<ALIGNED32>(buf, tiles, cap, r.run, flat, pool); break;
                case DENSE: ms = run_mode<DENSE>(buf, tiles, cap, r.run, flat, pool); break;
                default: ms = run_mode<GRANULES>(buf, tiles, cap, r.run, flat, pool); break;
            }
            printf("%s  %-28s %8.1f us  %6.2f TB/s\n", map, r.name, ms * 1e3, gb / ms);
        }
    }
    hipFree(buf);
    return 0;
}

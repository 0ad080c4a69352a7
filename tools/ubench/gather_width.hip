// Microbenchmark: cost of divergent gathers from an L2-resident 2 MiB table on MI355X, by access width
// (4/8/16 B per lane) and by how many neighbouring lanes share a 128-byte line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int W, int ITER>
__global__ void k_gather(const uint32_t* __restrict__ table, uint32_t mask_words, int share_log2, uint32_t* out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll 8
    for (int i = 0; i < ITER; i++) {
        // lanes in groups of 2^share_log2 fall into the same 128-byte line (different words of it)
        const uint32_t grp = tid >> share_log2;
        const uint32_t line = mix(grp * 977u + i * 7919u) & (mask_words >> 5);
        uint32_t word = (line << 5) + ((mix(tid + i) & 31u) & ~(uint32_t)(W - 1));
        if (W == 1) acc += table[word];
        else if (W == 2) { const uint2 v = *reinterpret_cast<const uint2*>(table + word); acc += v.x ^ v.y; }
        else { const uint4 v = *reinterpret_cast<const uint4*>(table + word); acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0xdeadbeef) out[tid] = acc;
}

template <int W>
void run(const uint32_t* table, uint32_t words, int share_log2, uint32_t* out) {
    constexpr int ITER = 64;
    const int blocks = 256 * 16, threads = 256;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k_gather<W, ITER>), dim3(blocks), dim3(threads), 0, 0, table, words - 1, share_log2, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k_gather<W, ITER>), dim3(blocks), dim3(threads), 0, 0, table, words - 1, share_log2, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double instr = (double)blocks * threads / 64 * ITER;   // wave-level gather instructions
    const double cyc = ms * 1e-3 * 2.4e9 * 256 / instr;           // CU-cycles per wave-instruction (at 2.4 GHz)
    printf("width %2d B  lanes/line %2d : %7.3f ms  %6.1f CU-cycles per wave gather  %7.1f G lane-gathers/s\n", W * 4,
           1 << share_log2, ms, cyc, instr * 64 / ms / 1e6);
}

int main() {
    const uint32_t words = 1u << 19;  // 2 MiB
    uint32_t *table, *out;
    (void)hipMalloc(&table, words * 4); (void)hipMemset(table, 1, words * 4);
    (void)hipMalloc(&out, 256 * 16 * 256 * 4);
    for (int s = 0; s <= 4; s += 1) { run<1>(table, words, s, out); run<2>(table, words, s, out); run<4>(table, words, s, out); }
    return 0;
}

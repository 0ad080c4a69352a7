// Microbenchmark: float atomic-add throughput on random addresses of a 4 MiB table, by memory scope,
// and with every workgroup restricted to "its XCD's" slice of the table (hardware XCC_ID).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
__device__ __forceinline__ uint32_t rng(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int SCOPE, bool XCD_SLICE>
__global__ void k_atomics(float* table, uint32_t rows_mask, uint32_t per_thread, uint32_t* xcd_hist) {
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    uint32_t base = 0, mask = rows_mask;
    if (XCD_SLICE) {
        const uint32_t x = xcc_id();
        mask = rows_mask >> 3;
        base = x * (mask + 1);
        if (threadIdx.x == 0) atomicAdd(&xcd_hist[x], 1u);
    }
    for (uint32_t i = 0; i < per_thread; i++) {
        const uint32_t r = base + (rng(s) & mask);
        if (SCOPE == 0) __hip_atomic_fetch_add(table + r, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (SCOPE == 1) __hip_atomic_fetch_add(table + r, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (SCOPE == 2) __hip_atomic_fetch_add(table + r, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        else table[r] += 1.0f;  // plain racy RMW (upper bound of the memory system)
    }
}

template <int SCOPE, bool XS>
double run(float* table, uint32_t rows, uint32_t* hist, const char* name) {
    const uint32_t blocks = 2048, threads = 256, per = 256;
    hipMemset(table, 0, rows * 4);
    hipMemset(hist, 0, 64);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_atomics<SCOPE, XS>), dim3(blocks), dim3(threads), 0, 0, table, rows - 1, 8u, hist);
    hipDeviceSynchronize();
    hipMemset(table, 0, rows * 4);
    hipEventRecord(a);
    hipLaunchKernelGGL((k_atomics<SCOPE, XS>), dim3(blocks), dim3(threads), 0, 0, table, rows - 1, per, hist);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    std::vector<float> h(rows);
    hipMemcpy(h.data(), table, rows * 4, hipMemcpyDeviceToHost);
    double sum = 0;
    for (float v : h) sum += v;
    const double n = (double)blocks * threads * per;
    printf("%-28s %8.3f ms  %7.2f G ops/s  sum/expected = %.6f\n", name, ms, n / ms / 1e6, sum / n);
    return ms;
}

int main() {
    const uint32_t rows = 1u << 20;  // 4 MiB of floats
    float* table; uint32_t* hist;
    hipMalloc(&table, rows * 4);
    hipMalloc(&hist, 64);
    run<0, false>(table, rows, hist, "agent scope, whole table");
    run<1, false>(table, rows, hist, "workgroup scope, whole");
    run<2, false>(table, rows, hist, "wavefront scope, whole");
    run<3, false>(table, rows, hist, "plain RMW (racy), whole");
    run<0, true>(table, rows, hist, "agent scope, XCD slice");
    run<1, true>(table, rows, hist, "workgroup scope, XCD slice");
    run<2, true>(table, rows, hist, "wavefront scope, XCD slice");
    run<3, true>(table, rows, hist, "plain RMW (racy), XCD slice");
    uint32_t hh[16];
    hipMemcpy(hh, hist, 64, hipMemcpyDeviceToHost);
    printf("blocks per XCC_ID:");
    for (int i = 0; i < 8; i++) printf(" %u", hh[i]);
    printf("\n");
    return 0;
}

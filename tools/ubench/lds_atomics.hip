// Microbenchmark: LDS accumulate throughput on random addresses of a 2048 x 2 accumulator, by operation:
//   ds_add_f32 (float atomic)      what the first version of k_grid_bwd_reduce used
//   ds_add_u32 / ds_add_u64        integer atomics (k_grid_bwd_bin's histogram, k_grid_bwd_reduce_fixed)
//   plain read-modify-write        no atomicity (upper bound; racy)
//   ticket                         per-wave private accumulator + one-byte ticket (k_grid_bwd_reduce_ticket)
// PMC on the real kernel showed ~390 LDS-busy cycles per ds_add_f32 wave instruction against 4-5 for the integer
// atomics (profiles/r01_pmc_gridbwd_before.txt); this isolates the effect.
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomics.bin lds_atomics.hip && ./lds_atomics.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr uint32_t kRows = 2048, kThreads = 256, kIters = 4096;

__device__ __forceinline__ uint32_t rng(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE>
__global__ __launch_bounds__(kThreads) void k_lds(float* out) {
    __shared__ unsigned long long acc64[kRows * 2];                      // 32 KiB, reinterpreted per mode
    __shared__ unsigned char tag[(kThreads / 64) * kRows];
    float* accf = reinterpret_cast<float*>(acc64);
    uint32_t* accu = reinterpret_cast<uint32_t*>(acc64);
    for (uint32_t i = threadIdx.x; i < kRows * 2; i += kThreads) acc64[i] = 0ull;
    __syncthreads();
    uint32_t s = (blockIdx.x * kThreads + threadIdx.x) * 2654435761u + 99u;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t it = 0; it < kIters; it++) {
        const uint32_t r = (rng(s) & (kRows - 1)) * 2;
        if (MODE == 0) { atomicAdd(&accf[r], 1.0f); atomicAdd(&accf[r + 1], 2.0f); }
        if (MODE == 1) { atomicAdd(&accu[r], 1u); atomicAdd(&accu[r + 1], 2u); }
        if (MODE == 2) { atomicAdd(&acc64[r], 1ull); atomicAdd(&acc64[r + 1], 2ull); }
        if (MODE == 3) { accf[r] += 1.0f; accf[r + 1] += 2.0f; }
        if (MODE == 4) {   // per-wave private float2 accumulator (8 KiB per wave here: 1024 rows) + ticket byte
            float2* acc = reinterpret_cast<float2*>(acc64) + wave * (kRows / 4);
            volatile unsigned char* tg = tag + wave * kRows;
            const uint32_t rr = (r >> 1) & (kRows / 4 - 1);
            bool pending = true;
            while (__ballot(pending)) {
                if (pending) tg[rr] = (unsigned char)lane;
                if (pending && tg[rr] == (unsigned char)lane) {
                    float2 a = acc[rr];
                    a.x += 1.0f; a.y += 2.0f;
                    acc[rr] = a;
                    pending = false;
                }
            }
        }
    }
    __syncthreads();
    float sum = 0.f;
    for (uint32_t i = threadIdx.x; i < kRows * 2; i += kThreads) sum += (MODE == 1 ? (float)accu[i] : MODE == 2 ? (float)acc64[i] : accf[i]);
    if (sum == 12345.678f) out[0] = sum;   // keep the work alive
}

template <int MODE>
void run(const char* name, float* out) {
    const uint32_t blocks = 1024;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_lds<MODE>, dim3(blocks), dim3(kThreads), 0, 0, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k_lds<MODE>, dim3(blocks), dim3(kThreads), 0, 0, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double items = (double)blocks * kThreads * kIters;            // one item = two channel updates
    printf("%-44s %8.3f ms  %8.1f G items/s  (%5.1f CU-cycles per wave-item at 2.4 GHz, 256 CUs)\n", name, ms, items / ms / 1e6,
           ms * 1e-3 * 2.4e9 * 256 / (items / 64));
}

int main() {
    float* out;
    hipMalloc(&out, 64);
    run<0>("ds_add_f32 x2 (float atomics)", out);
    run<1>("ds_add_u32 x2 (integer atomics)", out);
    run<2>("ds_add_u64 x2 (64-bit integer atomics)", out);
    run<3>("plain RMW x2 (racy)", out);
    run<4>("per-wave accumulator + ticket byte", out);
    hipFree(out);
    return 0;
}

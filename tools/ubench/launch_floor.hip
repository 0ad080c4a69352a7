// Microbenchmark (round 2): what a one-wave-per-ray kernel over 4096 rays costs before it moves a byte — the floor under
// the compositor / fused render kernels at BASELINE's 4096 rays (their 12-30 MB are 2-4 us at the HBM roofline).
//   (a) empty kernel, 1088 workgroups x 256 threads (the launch shape of k_render_train_fwd for 4096 rays + 64 padding groups)
//   (b) the same, every wave doing `chain` DEPENDENT 64-lane loads of 28 B per lane from a 32 MB array (one load's address
//       depends on the previous result: the carried transmittance of a ray's 64-sample chunks), chain = 1, 2, 5, 8
// Launch-to-launch time inside a stream of back-to-back launches (hipEvent around 200 launches).
//   hipcc --offload-arch=gfx950 -O3 -o launch_floor.bin launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k_empty(const int* rays, float* out) {
    const uint32_t n = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (rays[n & 4095] == -12345) out[n] = 1.f;
}

__global__ __launch_bounds__(256) void k_chain(const float* __restrict__ data, uint32_t words, int chain, float* out) {
    const uint32_t n = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t off = (n * 7919u * 448u) % (words - 64 * 7 * 16);
    float carry = 0.f;
    for (int c = 0; c < chain; c++) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 7; k++) acc += data[off + k * 64 + lane];          // 7 coalesced 256-byte rows = 28 B per lane
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);           // the wave-wide scan / reduction of a chunk
        carry += acc;
        off = (off + 448u + ((uint32_t)(carry != 12345.f) - 1u)) % (words - 64 * 7 * 16);   // next chunk's address depends on the result
    }
    if (carry == 12345.f) out[n] = carry;
}

int main() {
    const uint32_t words = 8u << 20;   // 32 MB
    float *data, *out; int* rays;
    (void)hipMalloc(&data, words * 4); (void)hipMemset(data, 0, words * 4);
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&rays, 4096 * 4); (void)hipMemset(rays, 0, 4096 * 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int reps = 200, blocks = 1088;
    auto time = [&](auto launch, const char* name) {
        for (int i = 0; i < 20; i++) launch();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        for (int i = 0; i < reps; i++) launch();
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("%-44s %7.2f us per launch\n", name, ms * 1e3 / reps);
    };
    time([&] { hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, 0, rays, out); }, "empty, 1088 x 256");
    for (int chain : {1, 2, 3, 5, 8}) {
        char name[64]; snprintf(name, sizeof name, "%d dependent 64-sample chunk load(s) per wave", chain);
        time([&] { hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(256), 0, 0, data, words, chain, out); }, name);
    }
    return 0;
}

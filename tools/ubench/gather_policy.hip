// Microbenchmark (round 2): divergent 16-byte gathers from a 2 MiB table on MI355X by cache policy of the load
// (plain, sc0, sc1, sc0 sc1, nt, sc0 sc1 nt) and by whether an instruction re-touches the lines of the previous one
// (the "second load" of k_grid_forward: row r1 of an x-pair when it is not in r0's 16-byte block).
// Question: does any policy move fewer than 128 bytes per touched line from L2, and what does an L1 / in-flight hit cost?
//   hipcc --offload-arch=gfx950 -O3 -o gather_policy.bin gather_policy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// Four gathers per asm statement, with the wait inside: the compiler must never see a register whose load is still in
// flight (an asm load returns at once; a register it "defined" may be copied or reused before the data arrives).
#define POL_STR(P) ((P) == 0 ? "" : (P) == 1 ? " sc0" : (P) == 2 ? " sc1" : (P) == 3 ? " sc0 sc1" : (P) == 4 ? " nt" : " sc0 sc1 nt")

template <int POLICY>
__device__ __forceinline__ void ld16x4(const uint32_t* p0, const uint32_t* p1, const uint32_t* p2, const uint32_t* p3, uint4 v[4]) {
#define LD16X4(POL)                                                                                                     \
    asm volatile("global_load_dwordx4 %0, %4, off" POL "\n\tglobal_load_dwordx4 %1, %5, off" POL                          \
                 "\n\tglobal_load_dwordx4 %2, %6, off" POL "\n\tglobal_load_dwordx4 %3, %7, off" POL "\n\ts_waitcnt vmcnt(0)" \
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory")
    if (POLICY == 0) LD16X4("");
    if (POLICY == 1) LD16X4(" sc0");
    if (POLICY == 2) LD16X4(" sc1");
    if (POLICY == 3) LD16X4(" sc0 sc1");
    if (POLICY == 4) LD16X4(" nt");
    if (POLICY == 5) LD16X4(" sc0 sc1 nt");
#undef LD16X4
}
// the same four 16-byte gathers, each followed by a 4-byte load of another word of ITS line (exec-masked by the caller)
template <int POLICY>
__device__ __forceinline__ void ld16x4_retouch(const uint32_t* p0, const uint32_t* p1, const uint32_t* p2, const uint32_t* p3,
                                               uint4 v[4], uint32_t u[4], bool second) {
    const uint32_t *q0 = p0 + ((p0 - (const uint32_t*)0) & 16 ? -16 : 16), *q1 = p1 + ((p1 - (const uint32_t*)0) & 16 ? -16 : 16),
                   *q2 = p2 + ((p2 - (const uint32_t*)0) & 16 ? -16 : 16), *q3 = p3 + ((p3 - (const uint32_t*)0) & 16 ? -16 : 16);
    u[0] = u[1] = u[2] = u[3] = 0;
    // all eight loads in flight together; the 4-byte ones only for lanes with `second`
    ld16x4<POLICY>(p0, p1, p2, p3, v);
    if (second) {
#define LD4X4(POL)                                                                                                      \
        asm volatile("global_load_dword %0, %4, off" POL "\n\tglobal_load_dword %1, %5, off" POL                          \
                     "\n\tglobal_load_dword %2, %6, off" POL "\n\tglobal_load_dword %3, %7, off" POL "\n\ts_waitcnt vmcnt(0)" \
                     : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]) : "v"(q0), "v"(q1), "v"(q2), "v"(q3) : "memory")
        if (POLICY == 0) LD4X4("");
        if (POLICY == 1) LD4X4(" sc0");
        if (POLICY == 2) LD4X4(" sc1");
        if (POLICY == 3) LD4X4(" sc0 sc1");
        if (POLICY == 4) LD4X4(" nt");
        if (POLICY == 5) LD4X4(" sc0 sc1 nt");
#undef LD4X4
    }
}
template <int POLICY>
__device__ __forceinline__ void ld4x4(const uint32_t* p0, const uint32_t* p1, const uint32_t* p2, const uint32_t* p3, uint32_t u[4]) {
#define LD4X4(POL)                                                                                                      \
    asm volatile("global_load_dword %0, %4, off" POL "\n\tglobal_load_dword %1, %5, off" POL                              \
                 "\n\tglobal_load_dword %2, %6, off" POL "\n\tglobal_load_dword %3, %7, off" POL "\n\ts_waitcnt vmcnt(0)"     \
                 : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory")
    if (POLICY == 0) LD4X4("");
    if (POLICY == 1) LD4X4(" sc0");
    if (POLICY == 2) LD4X4(" sc1");
    if (POLICY == 3) LD4X4(" sc0 sc1");
    if (POLICY == 4) LD4X4(" nt");
    if (POLICY == 5) LD4X4(" sc0 sc1 nt");
#undef LD4X4
}

// MODE 0: every instruction touches 64 fresh random lines (16 B per lane)
// MODE 1: as 0, then a 4-byte load of ANOTHER word of the same line by a quarter of the lanes (k_grid_forward's second load;
//         issued after the 16-byte data has arrived, i.e. an L1 hit if the line survived)
// MODE 2: as 1 but by all lanes
// MODE 3: 4-byte loads only (one per lane, fresh lines)
template <int POLICY, int MODE, int ITER>
__global__ void k_gather(const uint32_t* __restrict__ table, uint32_t mask_words, uint32_t* out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (int i = 0; i < ITER; i += 4) {
        const uint32_t* p[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t line = mix(tid * 977u + (i + k) * 7919u) & (mask_words >> 5);
            p[k] = table + (line << 5) + (mix(tid + i + k) & 28u);
        }
        uint4 v[4] = {};
        uint32_t u[4] = {0, 0, 0, 0};
        if (MODE == 0) ld16x4<POLICY>(p[0], p[1], p[2], p[3], v);
        if (MODE == 1) ld16x4_retouch<POLICY>(p[0], p[1], p[2], p[3], v, u, (tid & 3u) == 3u);
        if (MODE == 2) ld16x4_retouch<POLICY>(p[0], p[1], p[2], p[3], v, u, true);
        if (MODE == 3) ld4x4<POLICY>(p[0], p[1], p[2], p[3], u);
#pragma unroll
        for (int k = 0; k < 4; k++) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w ^ u[k];
    }
    if (acc == 0xdeadbeef) out[tid] = acc;
}

template <int POLICY, int MODE>
void run(const uint32_t* table, uint32_t words, uint32_t* out, int blocks) {
    constexpr int ITER = 64;
    const int threads = 256;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k_gather<POLICY, MODE, ITER>), dim3(blocks), dim3(threads), 0, 0, table, words - 1, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k_gather<POLICY, MODE, ITER>), dim3(blocks), dim3(threads), 0, 0, table, words - 1, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double instr = (double)blocks * threads / 64 * ITER;   // wave-level 64-line gathers
    const double cyc = ms * 1e-3 * 2.4e9 * 256 / instr;
    static const char* pol[] = {"plain", "sc0", "sc1", "sc0 sc1", "nt", "sc0 sc1 nt"};
    static const char* mode[] = {"16B fresh lines", "16B + 4B re-touch by 1/4 lanes", "16B + 4B re-touch by all lanes", "4B fresh lines"};
    printf("%-11s %-32s blocks %5d: %7.3f ms  %6.1f CU-cycles per 64-line gather  %6.2f TB/s of 128-B lines\n", pol[POLICY], mode[MODE],
           blocks, ms, cyc, instr * 64 * 128 / ms / 1e9);
}

template <int POLICY>
void run_policy(const uint32_t* table, uint32_t words, uint32_t* out) {
    run<POLICY, 0>(table, words, out, 4096);
    run<POLICY, 1>(table, words, out, 4096);
    run<POLICY, 2>(table, words, out, 4096);
    run<POLICY, 3>(table, words, out, 4096);
}

int main() {
    const uint32_t words = 1u << 19;  // 2 MiB: one hashed level of the fp16 table
    uint32_t *table, *out;
    (void)hipMalloc(&table, words * 4); (void)hipMemset(table, 1, words * 4);
    (void)hipMalloc(&out, 4096 * 256 * 4);
    run_policy<0>(table, words, out);
    run_policy<1>(table, words, out);
    run_policy<2>(table, words, out);
    run_policy<3>(table, words, out);
    run_policy<4>(table, words, out);
    run_policy<5>(table, words, out);
    // one XCD only (blocks 0, 8, 16, ... land on XCD 0 when the grid is 8x larger and the others exit at once is not
    // expressible here; instead: 512 blocks = 2 per CU, the latency-bound end)
    run<0, 0>(table, words, out, 512);
    run<0, 0>(table, words, out, 1024);
    run<0, 0>(table, words, out, 2048);
    return 0;
}

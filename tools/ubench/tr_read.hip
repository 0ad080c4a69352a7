// Probe: what ds_read_b64_tr_b16 (gfx950's transposing LDS read) returns. LDS holds halves whose bit pattern is their own index;
// every lane passes its own byte address; the four 16-bit results per lane are printed. Three address patterns:
//   P0  lane l -> byte 8 l                 (consecutive 8-byte pieces)
//   P1  all lanes -> byte 0                (uniform address)
//   P2  lane l -> row (l & 15) >> 2 of pitch 64 B, piece l & 3, + 1024 B per 16-lane group   (the layout of csrc/field.hip's plan)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k(uint32_t* out) {
    __shared__ uint16_t lds[8192];
    for (uint32_t i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t l = threadIdx.x;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds;   // LDS byte offset
    uint32_t addr[3];
    addr[0] = 8 * l;
    addr[1] = 0;
    addr[2] = (l >> 4) * 1024 + (((l & 15) >> 2) * 64) + (l & 3) * 8;
    for (int p = 0; p < 3; p++) {
        uint32_t a = base + addr[p];
        uint2 v;
        asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        out[(p * 64 + l) * 2 + 0] = v.x;
        out[(p * 64 + l) * 2 + 1] = v.y;
    }
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 3 * 64 * 2 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    std::vector<uint32_t> h(3 * 64 * 2);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    for (int p = 0; p < 3; p++) {
        printf("pattern %d (element indices = byte offset / 2)\n", p);
        for (int l = 0; l < 64; l++) {
            const uint32_t x = h[(p * 64 + l) * 2], y = h[(p * 64 + l) * 2 + 1];
            printf("  lane %2d: %5u %5u %5u %5u\n", l, x & 0xffff, x >> 16, y & 0xffff, y >> 16);
        }
    }
    return 0;
}

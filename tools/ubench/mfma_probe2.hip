// Which B element does A element (hi_a, j_a) of row 0 multiply in v_mfma_f32_32x32x16_f16?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out) {
    const unsigned l = threadIdx.x;
    for (int ha = 0; ha < 2; ha++) for (int ja = 0; ja < 8; ja++) {
        h8 a, b;
        for (int j = 0; j < 8; j++) { a[j] = (l == (unsigned)(32 * ha) && j == ja) ? (_Float16)1 : (_Float16)0; b[j] = (_Float16)(float)(l * 8 + j); }
        f32x16 d; for (int i = 0; i < 16; i++) d[i] = 0;
        d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
        if (l < 32) out[(ha * 8 + ja) * 32 + l] = d[0];   // row 0 (reg 0 of lanes 0..31), col l
    }
}
int main() {
    float* o; (void)hipMalloc(&o, 16 * 32 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
    float h[16 * 32]; (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    for (int ha = 0; ha < 2; ha++) for (int ja = 0; ja < 8; ja++) {
        const float v0 = h[(ha * 8 + ja) * 32 + 0], v5 = h[(ha * 8 + ja) * 32 + 5];
        printf("A(hi=%d,j=%d): D[0][0]=%g -> B lane %d elem %d ; D[0][5]=%g -> B lane %d elem %d\n", ha, ja, v0, (int)v0 / 8, (int)v0 % 8, v5, (int)v5 / 8, (int)v5 % 8);
    }
    return 0;
}

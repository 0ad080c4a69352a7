// Probe: v_permlane32_swap semantics and the lane/register layout of v_mfma_f32_32x32x16_f16 on gfx950,
// in the form the fused field kernels rely on (lane = sample, weights as the A operand).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

// in-place form: a = [a.lo | b.lo], b = [a.hi | b.hi]; the wait states cover the VALU-write -> permlane-read hazard
__device__ __forceinline__ void swap32(unsigned& a, unsigned& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

__global__ void k_swap(unsigned* out) {
    const unsigned l = threadIdx.x;
    unsigned a = 1000 + l, b = 2000 + l;
    const u2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[l] = r.x; out[64 + l] = r.y;
}

// out[m][n] = sum_k W[m][k] * X[n][k], m < 32, n < 64 (two n-blocks), k < 16; lane = sample n holds X[n][0..15]
__global__ void k_mfma(const _Float16* W, const _Float16* X, float* out, float* raw) {
    const unsigned l = threadIdx.x, hi = l >> 5;
    // A fragment: lane (m = l & 31, hi): W[m][8 hi + j]
    h8 a;
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = W[(l & 31) * 16 + 8 * hi + j];
    // this lane's sample: 16 features packed as 8 words
    unsigned w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const _Float16 lo = X[l * 16 + 2 * i], hi_ = X[l * 16 + 2 * i + 1];
        w[i] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi_) << 16);
    }
    uint4 b0, b1;
    { const u2 r = __builtin_amdgcn_permlane32_swap(w[0], w[4], false, false); b0.x = r.x; b1.x = r.y; }
    { const u2 r = __builtin_amdgcn_permlane32_swap(w[1], w[5], false, false); b0.y = r.x; b1.y = r.y; }
    { const u2 r = __builtin_amdgcn_permlane32_swap(w[2], w[6], false, false); b0.z = r.x; b1.z = r.y; }
    { const u2 r = __builtin_amdgcn_permlane32_swap(w[3], w[7], false, false); b0.w = r.x; b1.w = r.y; }
    f32x16 d0, d1;
#pragma unroll
    for (int i = 0; i < 16; i++) { d0[i] = 0; d1[i] = 0; }
    d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(h8, b0), d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(h8, b1), d1, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; r++) { raw[r * 64 + l] = d0[r]; raw[1024 + r * 64 + l] = d1[r]; }
    // back to lane = sample: after the swap e0[r] = row rho(r,0), e1[r] = row rho(r,1) of this lane's sample
#pragma unroll
    for (int r = 0; r < 16; r++) {
        unsigned x0 = __float_as_uint(d0[r]), x1 = __float_as_uint(d1[r]);
        swap32(x0, x1);
        const u2 s = {x0, x1};
        const int row0 = (r & 3) + 8 * (r >> 2);
        const unsigned y0 = s.x, y1 = s.y;   // NOTE: __builtin_bit_cast on a vector ELEMENT reads element 0 (clang); copy to a scalar first
        out[(row0) * 64 + l] = __uint_as_float(y0);
        out[(row0 + 4) * 64 + l] = __uint_as_float(y1);
    }
}

int main() {
    unsigned* o; (void)hipMalloc(&o, 128 * 4);
    hipLaunchKernelGGL(k_swap, dim3(1), dim3(64), 0, 0, o);
    unsigned h[128]; (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    printf("swap: x[0]=%u x[31]=%u x[32]=%u x[63]=%u | y[0]=%u y[31]=%u y[32]=%u y[63]=%u\n", h[0], h[31], h[32], h[63], h[64], h[95], h[96], h[127]);
    _Float16 W[32 * 16], X[64 * 16]; float ref[32 * 64];
    for (int i = 0; i < 32 * 16; i++) W[i] = (_Float16)(((i * 37) % 23 - 11) / 8.0f);
    for (int i = 0; i < 64 * 16; i++) X[i] = (_Float16)(((i * 53) % 19 - 9) / 4.0f);
    for (int m = 0; m < 32; m++) for (int n = 0; n < 64; n++) { float s = 0; for (int k = 0; k < 16; k++) s += (float)W[m * 16 + k] * (float)X[n * 16 + k]; ref[m * 64 + n] = s; }
    _Float16 *dW, *dX; float* dO;
    (void)hipMalloc(&dW, sizeof(W)); (void)hipMalloc(&dX, sizeof(X)); (void)hipMalloc(&dO, sizeof(ref));
    (void)hipMemcpy(dW, W, sizeof(W), hipMemcpyHostToDevice); (void)hipMemcpy(dX, X, sizeof(X), hipMemcpyHostToDevice);
    float* dR; (void)hipMalloc(&dR, 2048 * 4);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dW, dX, dO, dR);
    float got[32 * 64]; (void)hipMemcpy(got, dO, sizeof(got), hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 32 * 64; i++) err = fmax(err, fabs(got[i] - ref[i]));
    float raw[2048]; (void)hipMemcpy(raw, dR, sizeof(raw), hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0;
    for (int r = 0; r < 16; r++) for (int l = 0; l < 64; l++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        e0 = fmax(e0, fabs(raw[r * 64 + l] - ref[row * 64 + col]));
        e1 = fmax(e1, fabs(raw[1024 + r * 64 + l] - ref[row * 64 + 32 + col]));
    }
    printf("raw D layout check: block0 err %g, block1 err %g\n", e0, e1);
    for (int n : {3, 40}) {
        printf("col %d: out row -> ref row:", n);
        for (int m = 0; m < 32; m++) { int f = -1; for (int q = 0; q < 32; q++) if (fabs(got[m * 64 + n] - ref[q * 64 + n]) < 1e-4) { f = q; break; } printf(" %d>%d", m, f); }
        printf("\n");
    }
    printf("mfma lane=sample round trip: max |err| = %g (ref[5][40]=%g got=%g)\n", err, ref[5 * 64 + 40], got[5 * 64 + 40]);
    return 0;
}

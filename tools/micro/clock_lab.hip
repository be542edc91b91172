// Ratio of the shader cycle counter to the 100 MHz wall clock while all CUs run a VALU/MFMA loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void spin(unsigned long long* out, int iters, int mfma) {
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    f32x16 acc = {};
    bf16x8 a = {}, b = {};
    float x = threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
        if (mfma) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        else x = x * 1.0001f + 0.5f;
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
    if (x == 12345.f || acc[0] == 1.f) out[2] = 1;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    for (int mfma = 0; mfma < 2; ++mfma) {
        spin<<<1024, 256>>>(d, 200000, mfma);
        hipDeviceSynchronize();
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%s: cycles %llu wall(100MHz) %llu -> %.0f MHz\n", mfma ? "mfma" : "valu", h[0], h[1], h[0] / (h[1] / 100.0));
    }
    return 0;
}

// Issue rate of plain wave64 vector instructions on one SIMD at several occupancies.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = seed + threadIdx.x * 0.001f + i;
    float s = seed * 0.5f;
    unsigned long long acc = 0;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP8(asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %2, %2, %1\n v_max_f32 %3, %3, %1\n v_max_f32 %4, %4, %1" : "+v"(r[0]), "+v"(s), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));) }
        if (OP == 1) { REP8(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %2, %2, %1, %1\n v_fma_f32 %3, %3, %1, %1\n v_fma_f32 %4, %4, %1, %1" : "+v"(r[0]), "+v"(s), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));) }
        if (OP == 2) { REP8(asm volatile("v_sub_f32 %0, %0, %1\n v_sub_f32 %2, %2, %1\n v_sub_f32 %3, %3, %1\n v_sub_f32 %4, %4, %1" : "+v"(r[0]), "+v"(s), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));) }
        if (OP == 3) { REP8(asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %2, %2, %1\n v_mul_f32 %3, %3, %1\n v_mul_f32 %4, %4, %1" : "+v"(r[0]), "+v"(s), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));) }
        if (OP == 4) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %2, %1\n v_cmp_lt_f32 vcc, %3, %1\n v_cmp_lt_f32 vcc, %4, %1" : "+v"(r[0]), "+v"(s), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : : "vcc");) }
        if (OP == 5) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4" : "+v"(r[0]), "+v"(s), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));) }
        if (OP == 6) { REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %2, %2, %1\n v_cvt_pk_bf16_f32 %3, %3, %1\n v_cvt_pk_bf16_f32 %4, %4, %1" : "+v"(r[0]), "+v"(s), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));) }
        if (OP == 7) { REP8(asm volatile("v_max_f32 %0, %0, %1\n v_min_f32 %2, %2, %1\n v_sub_f32 %3, %3, %1\n v_mul_f32 %4, %4, %1" : "+v"(r[0]), "+v"(s), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = r[0] + r[1] + r[2] + r[3] + s + (float)acc;
}

template <int OP>
void run(const char* name, float* d, int wg_per_cu) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 4000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<cus * wg_per_cu, 256>>>(d, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<cus * wg_per_cu, 256>>>(d, iters, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // each wave issues iters*32 instructions; a SIMD hosts wg_per_cu waves
    const double inst_per_simd = (double)iters * 32 * wg_per_cu;
    printf("%-22s waves/SIMD %d : %.3f ms  -> %.2f ns per instruction per SIMD (x clock GHz = cycles)\n", name, wg_per_cu, ms,
           ms * 1e6 / inst_per_simd);
}

int main() {
    float* d; hipMalloc(&d, 256 * 4096 * 64);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_max_f32", d, w); run<1>("v_fma_f32", d, w); run<2>("v_sub_f32", d, w); run<3>("v_mul_f32", d, w);
        run<4>("v_cmp_lt_f32", d, w); run<5>("v_exp_f32", d, w); run<6>("v_cvt_pk_bf16_f32", d, w); run<7>("mix max/min/sub/mul", d, w);
    }
    return 0;
}

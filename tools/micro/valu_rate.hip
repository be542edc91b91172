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
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[5];
    for (int i = 0; i < 5; ++i) p[i] = f2{seed + i, seed - i};
    float q[3] = {seed, seed + 1, seed + 2};
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
        if (OP == 8) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %2, %2, %1\n v_pk_mul_f32 %3, %3, %1\n v_pk_mul_f32 %4, %4, %1" : "+v"(p[0]), "+v"(p[4]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]));) }
        if (OP == 9) { REP8(asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %2, %2, %1\n v_pk_add_f32 %3, %3, %1\n v_pk_add_f32 %4, %4, %1" : "+v"(p[0]), "+v"(p[4]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]));) }
        if (OP == 10) { REP8(asm volatile("v_cmp_nlt_f32_e64 s[20:21], %0, %1\n v_cmp_nlt_f32_e64 s[22:23], %2, %1\n v_cmp_nlt_f32_e64 s[24:25], %3, %1\n v_cmp_nlt_f32_e64 s[26:27], %4, %1" : "+v"(r[0]), "+v"(s), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (OP == 11) { REP8(asm volatile("v_max_f32 %0, s20, %0\n v_min_f32 %2, s21, %2\n v_max_f32 %3, s22, %3\n v_min_f32 %4, s23, %4" : "+v"(r[0]), "+v"(s), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));) }
        // the mask kernel's column step for two columns (17 instructions, counted as 16 here + 1)
        if (OP == 12) { REP8(asm volatile(
            "v_max_f32 %0, s20, %5\n v_max_f32 %1, s21, %6\n v_min_f32 %2, s22, %7\n v_min_f32 %3, s23, %8\n"
            "v_max_f32 %4, s24, %5\n v_max_f32 %9, s25, %6\n v_min_f32 %10, s26, %7\n v_min_f32 %11, s27, %8\n"
            "v_sub_f32 %0, %2, %0\n v_sub_f32 %4, %10, %4\n v_max_f32 %0, 0, %0\n v_max_f32 %4, 0, %4\n"
            "v_sub_f32 %1, %3, %1\n v_sub_f32 %9, %11, %9\n v_mul_f32 %0, %0, %1\n v_mul_f32 %4, %4, %9"
            : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(s), "+v"(q[0]), "+v"(q[1]), "+v"(q[2]));) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = r[0] + r[1] + r[2] + r[3] + r[4] + r[5] + r[6] + r[7] + s + (float)acc + p[0].x + p[1].y + p[2].x + p[3].y + p[4].x + q[0] + q[1] + q[2];
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
        run<8>("v_pk_mul_f32", d, w); run<9>("v_pk_add_f32", d, w); run<10>("v_cmp_nlt_f32 -> sgpr", d, w); run<11>("v_max/min sgpr src", d, w);
        run<12>("mask step (16 per 32)", d, w);
    }
    return 0;
}

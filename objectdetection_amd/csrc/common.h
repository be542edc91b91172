// Shared device/host helpers for the gfx950 kernels of libodamd.so.
#ifndef ODAMD_COMMON_H
#define ODAMD_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define ODAMD_ABI_VERSION 3

#include <mutex>
#include <type_traits>

// Compile-time loop: f(std::integral_constant<int, I>) for I = FROM .. TO - 1.  `#pragma unroll` is a request that the
// unroller drops once the unrolled body passes its size threshold; indices that must be constants (register arrays that
// would otherwise land in private memory) are iterated with this instead.
template <int FROM, int TO, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (FROM < TO) {
        f(std::integral_constant<int, FROM>{});
        static_for<FROM + 1, TO>(f);
    }
}
// Raises a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) on the current device only when the
// limit set so far is smaller: the attribute call is a driver round trip, and the launch paths would make it per launch.
inline hipError_t odamd_ensure_lds(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    struct Slot { const void* k; int dev; size_t bytes; };
    static std::mutex mu;
    static Slot slots[1024];
    static int n = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> guard(mu);
    for (int i = 0; i < n; ++i)
        if (slots[i].k == kernel && slots[i].dev == dev) {
            if (slots[i].bytes >= bytes) return hipSuccess;
            e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e == hipSuccess) slots[i].bytes = bytes;
            return e;
        }
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && n < 1024) slots[n++] = {kernel, dev, bytes};
    return e;
}

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // 8 bf16 = one 16-byte MFMA fragment
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// float -> bf16 bits, round to nearest even, NaN stays NaN: the plain cast is v_cvt_pk_bf16_f32.
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    const __bf16 h = (__bf16)f;
    return __builtin_bit_cast(uint16_t, h);
}
// two floats -> packed bf16 pair (low half = a), one v_cvt_pk_bf16_f32
typedef __attribute__((ext_vector_type(2))) __bf16 odamd_bf16x2_t;
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float a, float b) {
    const odamd_bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// 1 / (1 + exp(-z)) from the hardware exp2 and reciprocal (about 1 ulp each, two quarter-rate
// instructions and two full-rate ones); for kernels whose results are rounded to bf16.  The library
// forms (__expf, __frcp_rn, operator/) expand to range fix-ups and a full IEEE division sequence.
__device__ __forceinline__ float sigmoid_fast(float z) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f));
}

#endif

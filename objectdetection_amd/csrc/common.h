// Shared device/host helpers for the gfx950 kernels of libodamd.so.
#ifndef ODAMD_COMMON_H
#define ODAMD_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define ODAMD_ABI_VERSION 1

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // 8 bf16 = one 16-byte MFMA fragment
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// float -> bf16 bits, round to nearest even (NaN kept a NaN by forcing a mantissa bit).
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

#endif

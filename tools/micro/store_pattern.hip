// How fast do 16-byte-per-lane stores and LDS-DMA loads go when a pixel owns a 64-byte (or 128-byte) slice of a
// 256-byte record?  Patterns: lanes per pixel = 2 (32 B per pixel and instruction, the MFMA-register epilogue),
// 4 (64 B per pixel), 8 (128 B per pixel).  Every pattern moves the same bytes per pixel slice overall.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/store_pattern tools/micro/store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// LPP lanes per pixel; SLICE bytes of the pixel's slice (64 or 128); the wave covers 64 / LPP pixels per instruction and
// needs SLICE / (16 * LPP) instructions per pixel block.
template <int LPP, int SLICE>
__global__ void __launch_bounds__(256) store_kernel(char* y, int npix, int stride, int off) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = gridDim.x * 4;
    constexpr int PPI = 64 / LPP, REP = SLICE / (16 * LPP);
    const u32x4 v = {1u, 2u, 3u, (unsigned)lane};
    for (int p0 = wave * PPI; p0 < npix; p0 += nw * PPI) {
        const int px = p0 + lane / LPP;
#pragma unroll
        for (int r = 0; r < REP; ++r)
            if (px < npix) *(u32x4*)(y + (size_t)px * stride + off + (r * LPP + lane % LPP) * 16) = v;
    }
}

template <int LPP, int SLICE>
__global__ void __launch_bounds__(256) load_kernel(const char* x, int npix, int stride, int off, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = gridDim.x * 4;
    constexpr int PPI = 64 / LPP, REP = SLICE / (16 * LPP);
    u32x4 acc = {0, 0, 0, 0};
    for (int p0 = wave * PPI; p0 < npix; p0 += nw * PPI) {
        const int px = p0 + lane / LPP;
#pragma unroll
        for (int r = 0; r < REP; ++r)
            if (px < npix) acc ^= *(const u32x4*)(x + (size_t)px * stride + off + (r * LPP + lane % LPP) * 16);
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <typename F>
float timed(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / 20;
}

int main() {
    const int npix = 64 * 160 * 160;
    char* buf; unsigned* sink;
    hipMalloc(&buf, (size_t)npix * 256);
    hipMalloc(&sink, 4);
    hipMemset(buf, 0, (size_t)npix * 256);
    for (int grid : {512, 2048}) {
        for (int stride : {256, 64}) {
            const int off = stride == 256 ? 128 : 0;
            printf("grid %d, pixel stride %d B:\n", grid, stride);
#define RUN(K, LPP, SLICE, ...) { const float us = timed([&]() { K<LPP, SLICE><<<grid, 256>>>(__VA_ARGS__); }); \
            printf("  %-6s %d lanes/pixel, %3d B slice: %7.1f us  %6.0f GB/s\n", #K, LPP, SLICE, us, (double)npix * SLICE / us * 1e-3); }
            RUN(store_kernel, 2, 64, buf, npix, stride, off)
            RUN(store_kernel, 4, 64, buf, npix, stride, off)
            RUN(load_kernel, 2, 64, buf, npix, stride, off, sink)
            RUN(load_kernel, 4, 64, buf, npix, stride, off, sink)
            if (stride == 256) {
                RUN(store_kernel, 2, 128, buf, npix, stride, off)
                RUN(store_kernel, 4, 128, buf, npix, stride, off)
                RUN(store_kernel, 8, 128, buf, npix, stride, off)
                RUN(load_kernel, 4, 128, buf, npix, stride, off, sink)
                RUN(load_kernel, 8, 128, buf, npix, stride, off, sink)
            }
        }
    }
    return 0;
}

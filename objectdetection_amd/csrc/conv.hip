// Implicit-GEMM convolution for gfx950 (MI355X): 1x1 / 3x3, stride 1 / 2, NHWC activations,
// bf16 or f32 inputs with f32 accumulation on the matrix cores, fused per-channel affine
// (folded BatchNorm or bias) + SiLU epilogue, output written at a channel offset of a wider
// NHWC buffer (so a concat costs nothing) or as float32 NCHW (the head layout).
//
// GEMM view: D[cout][pixel] = sum_k W[cout][k] * X[k][pixel], k = (ky, kx, cin).
//   A operand = weights  (rows = 32 output channels), B operand = pixels (cols = 32 pixels),
//   one v_mfma_f32_32x32x16_bf16 per 16 input channels (bf16) or four v_mfma_f32_32x32x2_f32
//   per 8 input channels (f32; exact f32 fma chain).
// A workgroup (4 waves) owns a TH x TW patch of output pixels (<= 32*PG pixels) of one image
// and 32*CG output channels.  Per input-channel chunk it stages the input halo patch
// ((TH-1)*stride+k) x ((TW-1)*stride+k) and the weight slab in LDS once; the k*k taps then
// read shifted windows of the same patch, so every activation is fetched from HBM/L2 once per
// workgroup.  LDS rows are padded by 16 bytes so the 16-byte fragment reads are conflict free.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "odamd.h"

namespace {

template <int DT> struct Elem;
template <> struct Elem<ODAMD_DTYPE_BF16> { static constexpr int SIZE = 2; };
template <> struct Elem<ODAMD_DTYPE_F32> { static constexpr int SIZE = 4; };

// One k-step: 32 bytes of input channels per pixel (16 bf16 / 8 f32); lane half h holds the
// 16 bytes at offset 16*h of both operands.
template <int DT>
__device__ __forceinline__ void mma_step(f32x16_t& acc, const uint4& a, const uint4& b) {
    if constexpr (DT == ODAMD_DTYPE_BF16) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                      acc, 0, 0, 0);
    } else {
        const f32x4_t fa = __builtin_bit_cast(f32x4_t, a), fb = __builtin_bit_cast(f32x4_t, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[j], acc, 0, 0, 0);
    }
}

__device__ __forceinline__ uint4 vmax16(const uint4& p, const uint4& q, int dt) {
    uint4 r;
    if (dt == ODAMD_DTYPE_F32) {
        const float* a = (const float*)&p; const float* b = (const float*)&q; float* o = (float*)&r;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = fmaxf(a[i], b[i]);
    } else {
        const uint16_t* a = (const uint16_t*)&p; const uint16_t* b = (const uint16_t*)&q; uint16_t* o = (uint16_t*)&r;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = bf16_to_f32(a[i]) >= bf16_to_f32(b[i]) ? a[i] : b[i];
    }
    return r;
}

template <int DT>
__device__ __forceinline__ float silu(float v) {
    if constexpr (DT == ODAMD_DTYPE_F32) return v / (1.0f + expf(-v));
    else return v * __frcp_rn(1.0f + __expf(-v));
}

struct ConvP {
    const char* x; const char* w; const float* scale; const float* shift; char* y;
    int B, H, W, Ho, Wo, Cin, Cout, x_ld, y_ld;
    int Hr, Wr;  // raw input size (differs from H, W only when pooling on load)
    int TH, TW, tiles_x, tiles_y, ncb, IH, IW;
    int act, pool, nchw;
    int nblocks;
};

// XCD-aware block order: blocks b and b+8 share an XCD (and its L2), so give each XCD a
// contiguous run of logical tiles; neighbouring tiles then share halo rows and, when
// cout > 32*CG, the channel blocks of one pixel patch share the staged input through L2.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, xcd = bid & 7, slot = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// CG: 32-channel groups per workgroup, PG: 32-pixel groups per workgroup,
// WP x WC = 4 waves over (pixel groups, channel groups), CKB: bytes of input channels per chunk.
template <int DT, int KS, int ST, int CG, int PG, int WP, int WC, int CKB>
__global__ void __launch_bounds__(256) conv_igemm_kernel(ConvP p) {
    constexpr int ES = Elem<DT>::SIZE;
    constexpr int TAPS = KS * KS;
    constexpr int PS = CKB + 16;            // LDS bytes per staged pixel
    constexpr int WRS = TAPS * CKB + 16;    // LDS bytes per staged weight row
    constexpr int PGW = PG / WP, CGW = CG / WC;
    constexpr int PARTS = CKB / 16;
    constexpr int PAD = (KS - 1) / 2;
    static_assert(WP * WC == 4, "four waves");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* lds_in = lds;
    char* lds_w = lds + ((p.IH * p.IW * PS + 255) & ~255);

    const int lid = xcd_remap(blockIdx.x, p.nblocks);
    const int cb = lid % p.ncb;
    const int tile = (lid / p.ncb) % (p.tiles_x * p.tiles_y);
    const int b = lid / (p.ncb * p.tiles_x * p.tiles_y);
    const int oy0 = (tile / p.tiles_x) * p.TH, ox0 = (tile % p.tiles_x) * p.TW;
    const int co0 = cb * 32 * CG;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wp = wave / WC, wc = wave % WC;

    // per-lane LDS offsets of this wave's pixel groups / channel groups
    int b_off[PGW], a_off[CGW];
    int q_y[PGW], q_x[PGW];
#pragma unroll
    for (int i = 0; i < PGW; ++i) {
        const int q = (wp * PGW + i) * 32 + r;
        int ty = q / p.TW, tx = q - ty * p.TW;
        q_y[i] = ty; q_x[i] = tx;
        if (ty >= p.TH) { ty = 0; tx = 0; }
        b_off[i] = ((ty * ST) * p.IW + tx * ST) * PS + 16 * h;
    }
#pragma unroll
    for (int i = 0; i < CGW; ++i) a_off[i] = ((wc * CGW + i) * 32 + r) * WRS + 16 * h;

    f32x16_t acc[PGW][CGW];
#pragma unroll
    for (int i = 0; i < PGW; ++i)
#pragma unroll
        for (int j = 0; j < CGW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int n_in = p.IH * p.IW * PARTS;
    const int n_w = 32 * CG * TAPS * PARTS;
    const int iy0 = oy0 * ST - PAD, ix0 = ox0 * ST - PAD;
    const size_t img = (size_t)b * p.H * p.W;      // pixels before this image (pooled grid if pool)
    const uint4 zero = make_uint4(0, 0, 0, 0);

    for (int c0 = 0; c0 < p.Cin * ES; c0 += CKB) {  // c0 in bytes
        // ---- stage the input halo patch -------------------------------------------------------
        for (int i = tid; i < n_in; i += 256) {
            const int pix = i / PARTS, part = i - pix * PARTS;
            const int iy = pix / p.IW, ix = pix - iy * p.IW;
            const int gy = iy0 + iy, gx = ix0 + ix;
            uint4 v = zero;
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
                if (!p.pool) {
                    v = *(const uint4*)(p.x + ((img + (size_t)gy * p.W + gx) * p.x_ld) * ES + c0 + part * 16);
                } else {  // 2x2/2 max pool of the raw (2H x 2W) tensor on the fly
                    const size_t rw = (size_t)p.Wr;
                    const char* s = p.x + ((((size_t)b * p.Hr + 2 * gy) * rw + 2 * gx) * p.x_ld) * ES + c0 + part * 16;
                    const size_t px = (size_t)p.x_ld * ES;
                    v = vmax16(vmax16(*(const uint4*)s, *(const uint4*)(s + px), DT),
                               vmax16(*(const uint4*)(s + rw * px), *(const uint4*)(s + rw * px + px), DT), DT);
                }
            }
            *(uint4*)(lds_in + pix * PS + part * 16) = v;
        }
        // ---- stage the weight slab: rows = output channels, [tap][chunk] -------------------------
        for (int i = tid; i < n_w; i += 256) {
            const int row = i / (TAPS * PARTS), rem = i - row * (TAPS * PARTS);
            const int tap = rem / PARTS, part = rem - tap * PARTS;
            const int co = co0 + row;
            uint4 v = zero;
            if (co < p.Cout) v = *(const uint4*)(p.w + ((size_t)(co * TAPS + tap) * p.Cin) * ES + c0 + part * 16);
            *(uint4*)(lds_w + row * WRS + tap * CKB + part * 16) = v;
        }
        __syncthreads();
        // ---- k*k taps over the staged patch ------------------------------------------------------
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int tap_off = ((tap / KS) * p.IW + (tap % KS)) * PS;
#pragma unroll
            for (int s = 0; s < CKB / 32; ++s) {
                uint4 a[CGW], bb[PGW];
#pragma unroll
                for (int j = 0; j < CGW; ++j) a[j] = *(const uint4*)(lds_w + a_off[j] + tap * CKB + s * 32);
#pragma unroll
                for (int i = 0; i < PGW; ++i) bb[i] = *(const uint4*)(lds_in + b_off[i] + tap_off + s * 32);
#pragma unroll
                for (int i = 0; i < PGW; ++i)
#pragma unroll
                    for (int j = 0; j < CGW; ++j) mma_step<DT>(acc[i][j], a[j], bb[i]);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds pixel (group i, r) x channels {8g + 4h + e} of each 32-channel group ----
#pragma unroll
    for (int i = 0; i < PGW; ++i) {
        const int oy = oy0 + q_y[i], ox = ox0 + q_x[i];
        if (q_y[i] >= p.TH || oy >= p.Ho || ox >= p.Wo) continue;
        const size_t opix = ((size_t)b * p.Ho + oy) * p.Wo + ox;
#pragma unroll
        for (int j = 0; j < CGW; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = co0 + (wc * CGW + j) * 32 + 8 * g + 4 * h;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = co + e;
                    float t = acc[i][j][4 * g + e];
                    if (c < p.Cout) {
                        if (p.scale) t *= p.scale[c];
                        if (p.shift) t += p.shift[c];
                    }
                    v[e] = p.act ? silu<DT>(t) : t;
                }
                if (p.nchw) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < p.Cout)
                            ((float*)p.y)[(((size_t)b * p.Cout + co + e) * p.Ho + oy) * p.Wo + ox] = v[e];
                } else if (co + 3 < p.Cout) {
                    char* dst = p.y + (opix * p.y_ld + co) * ES;
                    if constexpr (DT == ODAMD_DTYPE_F32) {
                        *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        uint2 pk;
                        pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
                        pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
                        *(uint2*)dst = pk;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < p.Cout) {
                            if constexpr (DT == ODAMD_DTYPE_F32) ((float*)p.y)[opix * p.y_ld + co + e] = v[e];
                            else ((uint16_t*)p.y)[opix * p.y_ld + co + e] = f32_to_bf16(v[e]);
                        }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// first layer: float32 NCHW image, 3 input channels, 3x3 stride 1 -> NHWC activations.
// Weights are wave-uniform and come through scalar loads; one thread = one output pixel.
// ------------------------------------------------------------------------------------------
template <int DT>
__global__ void __launch_bounds__(256)
conv_first_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ scale,
                  const float* __restrict__ shift, char* __restrict__ y, int H, int W, int Cout, int y_ld) {
    constexpr int T = 16;
    __shared__ float patch[3][T + 2][T + 3];
    const int b = blockIdx.z, oy0 = blockIdx.y * T, ox0 = blockIdx.x * T;
    for (int i = threadIdx.x; i < 3 * (T + 2) * (T + 2); i += 256) {
        const int c = i / ((T + 2) * (T + 2)), rem = i - c * (T + 2) * (T + 2);
        const int iy = rem / (T + 2), ix = rem - iy * (T + 2);
        const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
        patch[c][iy][ix] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? img[(((size_t)b * 3 + c) * H + gy) * W + gx] : 0.0f;
    }
    __syncthreads();
    const int ty = threadIdx.x / T, tx = threadIdx.x % T;
    const int oy = oy0 + ty, ox = ox0 + tx;
    float in[27];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int c = 0; c < 3; ++c) in[(ky * 3 + kx) * 3 + c] = patch[c][ty + ky][tx + kx];
    if (oy >= H || ox >= W) return;
    constexpr int ES = Elem<DT>::SIZE;
    char* dst = y + (((size_t)b * H + oy) * W + ox) * y_ld * ES;
    for (int c0 = 0; c0 < Cout; c0 += 8) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int co = min(c0 + e, Cout - 1);
            float a = 0.0f;
#pragma unroll
            for (int k = 0; k < 27; ++k) a = fmaf(in[k], w[co * 27 + k], a);
            a = a * scale[co] + shift[co];
            o[e] = silu<DT>(a);
        }
        if (c0 + 8 <= Cout) {
            if constexpr (DT == ODAMD_DTYPE_F32) {
                *(float4*)(dst + c0 * 4) = make_float4(o[0], o[1], o[2], o[3]);
                *(float4*)(dst + c0 * 4 + 16) = make_float4(o[4], o[5], o[6], o[7]);
            } else {
                uint4 pk;
                pk.x = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
                pk.y = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
                pk.z = (uint32_t)f32_to_bf16(o[4]) | ((uint32_t)f32_to_bf16(o[5]) << 16);
                pk.w = (uint32_t)f32_to_bf16(o[6]) | ((uint32_t)f32_to_bf16(o[7]) << 16);
                *(uint4*)(dst + c0 * 2) = pk;
            }
        } else {
            for (int e = 0; c0 + e < Cout; ++e) {
                if constexpr (DT == ODAMD_DTYPE_F32) ((float*)dst)[c0 + e] = o[e];
                else ((uint16_t*)dst)[c0 + e] = f32_to_bf16(o[e]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// pools / upsample / layout converters: one thread per (pixel, 16-byte channel vector)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 neg_inf16(int dt) {
    const uint32_t v = dt == ODAMD_DTYPE_F32 ? 0xFF800000u : 0xFF80FF80u;
    return make_uint4(v, v, v, v);
}

__global__ void __launch_bounds__(256)
maxpool2_kernel(const char* __restrict__ x, char* __restrict__ y, int B, int H, int W, int vecs, int x_ld, int y_ld,
                int es, int dt) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t n = (size_t)B * Ho * Wo * vecs;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int v = i % vecs;
        size_t pix = i / vecs;
        const int ox = pix % Wo; pix /= Wo;
        const int oy = pix % Ho;
        const int b = pix / Ho;
        const char* s = x + ((((size_t)b * H + 2 * oy) * W + 2 * ox) * x_ld) * es + v * 16;
        const size_t px = (size_t)x_ld * es;
        const uint4 m = vmax16(vmax16(*(const uint4*)s, *(const uint4*)(s + px), dt),
                               vmax16(*(const uint4*)(s + W * px), *(const uint4*)(s + W * px + px), dt), dt);
        *(uint4*)(y + ((((size_t)b * Ho + oy) * Wo + ox) * y_ld) * es + v * 16) = m;
    }
}

// Three stride-1 max pools (5, 9, 13; -inf padding) of the same tensor in one pass: the 5x5
// window is inside the 9x9 one, which is inside the 13x13 one, so one walk over 13x13 feeds all.
__global__ void __launch_bounds__(256)
spp_pool3_kernel(const char* __restrict__ x, char* __restrict__ y5, char* __restrict__ y9, char* __restrict__ y13,
                 int B, int H, int W, int vecs, int x_ld, int y_ld, int es, int dt) {
    const size_t n = (size_t)B * H * W * vecs;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int v = i % vecs;
        size_t pix = i / vecs;
        const int ox = pix % W; pix /= W;
        const int oy = pix % H;
        const int b = pix / H;
        uint4 m5 = neg_inf16(dt), m9 = m5, m13 = m5;
        for (int dy = -6; dy <= 6; ++dy) {
            const int gy = oy + dy;
            if (gy < 0 || gy >= H) continue;
            const int ady = dy < 0 ? -dy : dy;
            for (int dx = -6; dx <= 6; ++dx) {
                const int gx = ox + dx;
                if (gx < 0 || gx >= W) continue;
                const int adx = dx < 0 ? -dx : dx;
                const int rad = ady > adx ? ady : adx;
                const uint4 t = *(const uint4*)(x + ((((size_t)b * H + gy) * W + gx) * x_ld) * es + v * 16);
                m13 = vmax16(m13, t, dt);
                if (rad <= 4) m9 = vmax16(m9, t, dt);
                if (rad <= 2) m5 = vmax16(m5, t, dt);
            }
        }
        const size_t o = ((((size_t)b * H + oy) * W + ox) * y_ld) * es + v * 16;
        *(uint4*)(y5 + o) = m5;
        *(uint4*)(y9 + o) = m9;
        *(uint4*)(y13 + o) = m13;
    }
}

__global__ void __launch_bounds__(256)
upsample2_kernel(const char* __restrict__ x, char* __restrict__ y, int B, int H, int W, int vecs, int x_ld, int y_ld,
                 int es) {
    const int Ho = 2 * H, Wo = 2 * W;
    const size_t n = (size_t)B * Ho * Wo * vecs;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int v = i % vecs;
        size_t pix = i / vecs;
        const int ox = pix % Wo; pix /= Wo;
        const int oy = pix % Ho;
        const int b = pix / Ho;
        *(uint4*)(y + ((((size_t)b * Ho + oy) * Wo + ox) * y_ld) * es + v * 16) =
            *(const uint4*)(x + ((((size_t)b * H + (oy >> 1)) * W + (ox >> 1)) * x_ld) * es + v * 16);
    }
}

// NCHW float32 <-> NHWC activations through a 32x32 LDS tile (coalesced on both sides).
template <int DT, bool TO_NHWC>
__global__ void __launch_bounds__(256)
layout_kernel(const char* __restrict__ x, char* __restrict__ y, int C, int HW, int ld, int Ct) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    if (TO_NHWC) {
        for (int j = ty; j < 32; j += 8) {
            const int c = c0 + j, pp = p0 + tx;
            tile[j][tx] = (c < C && pp < HW) ? ((const float*)x)[((size_t)b * Ct + c) * HW + pp] : 0.0f;
        }
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const int pp = p0 + j, c = c0 + tx;
            if (c < C && pp < HW) {
                const float v = tile[tx][j];
                if (DT == ODAMD_DTYPE_F32) ((float*)y)[((size_t)b * HW + pp) * ld + c] = v;
                else ((uint16_t*)y)[((size_t)b * HW + pp) * ld + c] = f32_to_bf16(v);
            }
        }
    } else {
        for (int j = ty; j < 32; j += 8) {
            const int pp = p0 + j, c = c0 + tx;
            float v = 0.0f;
            if (c < C && pp < HW)
                v = DT == ODAMD_DTYPE_F32 ? ((const float*)x)[((size_t)b * HW + pp) * ld + c]
                                          : bf16_to_f32(((const uint16_t*)x)[((size_t)b * HW + pp) * ld + c]);
            tile[j][tx] = v;
        }
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const int c = c0 + j, pp = p0 + tx;
            if (c < C && pp < HW) ((float*)y)[((size_t)b * Ct + c) * HW + pp] = tile[tx][j];
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
constexpr int LDS_LIMIT = 64 * 1024;

struct TileChoice { int th, tw, ih, iw; bool ok; };

// Pick the output patch (TH x TW <= npix pixels) that stages the fewest input pixels overall
// and whose halo patch + weight slab fit in LDS.
TileChoice choose_tile(int Ho, int Wo, int npix, int ks, int st, int ps, size_t w_bytes) {
    const int cand[] = {Wo, 64, 40, 32, 20, 16, 8};
    TileChoice best = {0, 0, 0, 0, false};
    double best_cost = 1e30;
    for (int tw : cand) {
        if (tw > Wo || tw > npix || tw < 1) continue;
        int th = npix / tw;
        if (th > Ho) th = Ho;
        if (th < 1) continue;
        const int ih = (th - 1) * st + ks, iw = (tw - 1) * st + ks;
        if ((((size_t)ih * iw * ps + 255) & ~(size_t)255) + w_bytes > (size_t)LDS_LIMIT) continue;
        const double tiles = (double)((Wo + tw - 1) / tw) * ((Ho + th - 1) / th);
        const double cost = tiles * (ih * iw + 0.25 * npix);  // staged pixels + a share for idle lanes
        if (cost < best_cost) { best_cost = cost; best = {th, tw, ih, iw, true}; }
    }
    return best;
}

template <int DT, int KS, int ST, int CG, int PG, int WP, int WC, int CKB>
int launch_conv(ConvP p, hipStream_t st) {
    constexpr int PS = CKB + 16, WRS = KS * KS * CKB + 16;
    const size_t w_bytes = (size_t)32 * CG * WRS;
    TileChoice t = choose_tile(p.Ho, p.Wo, 32 * PG, KS, ST, PS, w_bytes);
    if (!t.ok) return ODAMD_E_SHAPE;
    p.TH = t.th; p.TW = t.tw; p.IH = t.ih; p.IW = t.iw;
    p.tiles_x = (p.Wo + t.tw - 1) / t.tw;
    p.tiles_y = (p.Ho + t.th - 1) / t.th;
    p.ncb = (p.Cout + 32 * CG - 1) / (32 * CG);
    p.nblocks = p.B * p.tiles_x * p.tiles_y * p.ncb;
    const size_t lds = (((size_t)t.ih * t.iw * PS + 255) & ~(size_t)255) + w_bytes;
    conv_igemm_kernel<DT, KS, ST, CG, PG, WP, WC, CKB><<<p.nblocks, 256, lds, st>>>(p);
    return (int)hipGetLastError();
}

template <int DT, int KS, int ST, int CKB>
int dispatch_cg(const ConvP& p, hipStream_t st) {
    constexpr int PG = ST == 2 ? 4 : 8;
    if (p.Cout <= 32) return launch_conv<DT, KS, ST, 1, PG, 4, 1, CKB>(p, st);
    if (p.Cout <= 64) return launch_conv<DT, KS, ST, 2, PG, 4, 1, CKB>(p, st);
    return launch_conv<DT, KS, ST, 4, PG, 2, 2, CKB>(p, st);
}

template <int DT>
int dispatch_conv(const ConvP& p, int ks, int stride, hipStream_t st) {
    const int cin_bytes = p.Cin * Elem<DT>::SIZE;
    if (ks == 1 && stride == 1) {
        if (cin_bytes % 64 == 0) return dispatch_cg<DT, 1, 1, 64>(p, st);
        return dispatch_cg<DT, 1, 1, 32>(p, st);
    }
    if (ks == 3 && stride == 1) return dispatch_cg<DT, 3, 1, 32>(p, st);
    if (ks == 3 && stride == 2) return dispatch_cg<DT, 3, 2, 32>(p, st);
    return ODAMD_E_ARG;
}

int grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

int check_vec(const void* x, const void* y, int c, int x_ld, int y_ld, int dtype, int* es) {
    if (!x || !y || c < 1) return ODAMD_E_ARG;
    if (dtype != ODAMD_DTYPE_F32 && dtype != ODAMD_DTYPE_BF16) return ODAMD_E_ARG;
    *es = dtype == ODAMD_DTYPE_F32 ? 4 : 2;
    const int per = 16 / *es;
    if (c % per || x_ld % per || y_ld % per || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return ODAMD_E_ALIGN;
    if (x_ld < c || y_ld < c) return ODAMD_E_SHAPE;
    return ODAMD_OK;
}

}  // namespace

extern "C" {

int odamd_conv_forward(const odamd_conv_desc* d, void* stream) {
    if (!d || !d->x || !d->w || !d->y) return ODAMD_E_ARG;
    if (d->batch < 1 || d->h < 1 || d->w_in < 1 || d->cin < 1 || d->cout < 1) return ODAMD_E_ARG;
    if (d->dtype != ODAMD_DTYPE_F32 && d->dtype != ODAMD_DTYPE_BF16) return ODAMD_E_ARG;
    if (!((d->ksize == 1 && d->stride == 1) || (d->ksize == 3 && (d->stride == 1 || d->stride == 2)))) return ODAMD_E_ARG;
    if (d->in_pool2 && d->ksize != 1) return ODAMD_E_ARG;
    const int es = d->dtype == ODAMD_DTYPE_F32 ? 4 : 2;
    // a k-step eats 32 bytes of input channels; all vector accesses are 16 bytes wide
    if ((d->cin * es) % 32 || (d->x_ld * es) % 16 || ((uintptr_t)d->x & 15) || ((uintptr_t)d->w & 15)) return ODAMD_E_ALIGN;
    if (d->x_ld < d->cin) return ODAMD_E_SHAPE;
    ConvP p = {};
    p.x = (const char*)d->x; p.w = (const char*)d->w; p.scale = d->scale; p.shift = d->shift; p.y = (char*)d->y;
    p.B = d->batch; p.Hr = d->h; p.Wr = d->w_in;
    p.H = d->in_pool2 ? d->h / 2 : d->h; p.W = d->in_pool2 ? d->w_in / 2 : d->w_in;
    p.Cin = d->cin; p.Cout = d->cout; p.x_ld = d->x_ld; p.y_ld = d->y_ld;
    const int pad = (d->ksize - 1) / 2;
    p.Ho = (p.H + 2 * pad - d->ksize) / d->stride + 1;
    p.Wo = (p.W + 2 * pad - d->ksize) / d->stride + 1;
    p.act = d->act; p.pool = d->in_pool2; p.nchw = d->out_nchw_f32;
    if (!p.nchw) {
        if (d->y_ld < d->cout) return ODAMD_E_SHAPE;
        if (((uintptr_t)d->y & (4 * es - 1)) || (d->y_ld * es) % (4 * es)) return ODAMD_E_ALIGN;
    }
    hipStream_t st = (hipStream_t)stream;
    return d->dtype == ODAMD_DTYPE_F32 ? dispatch_conv<ODAMD_DTYPE_F32>(p, d->ksize, d->stride, st)
                                       : dispatch_conv<ODAMD_DTYPE_BF16>(p, d->ksize, d->stride, st);
}

int odamd_conv_first_forward(const float* image, const float* w, const float* scale, const float* shift, void* y,
                             int batch, int h, int w_in, int cout, int y_ld, int dtype, void* stream) {
    if (!image || !w || !scale || !shift || !y || batch < 1 || h < 1 || w_in < 1 || cout < 1) return ODAMD_E_ARG;
    if (dtype != ODAMD_DTYPE_F32 && dtype != ODAMD_DTYPE_BF16) return ODAMD_E_ARG;
    const int es = dtype == ODAMD_DTYPE_F32 ? 4 : 2;
    if (y_ld < cout) return ODAMD_E_SHAPE;
    if (((uintptr_t)y & 15) || (y_ld * es) % 16) return ODAMD_E_ALIGN;
    dim3 grid((w_in + 15) / 16, (h + 15) / 16, batch);
    if (dtype == ODAMD_DTYPE_F32)
        conv_first_kernel<ODAMD_DTYPE_F32><<<grid, 256, 0, (hipStream_t)stream>>>(image, w, scale, shift, (char*)y, h, w_in, cout, y_ld);
    else
        conv_first_kernel<ODAMD_DTYPE_BF16><<<grid, 256, 0, (hipStream_t)stream>>>(image, w, scale, shift, (char*)y, h, w_in, cout, y_ld);
    return (int)hipGetLastError();
}

int odamd_maxpool2_forward(const void* x, void* y, int batch, int h, int w, int c, int x_ld, int y_ld, int dtype,
                           void* stream) {
    int es;
    int rc = check_vec(x, y, c, x_ld, y_ld, dtype, &es);
    if (rc) return rc;
    if (batch < 1 || h < 2 || w < 2) return ODAMD_E_ARG;
    const int vecs = c * es / 16;
    maxpool2_kernel<<<grid_for((size_t)batch * (h / 2) * (w / 2) * vecs), 256, 0, (hipStream_t)stream>>>(
        (const char*)x, (char*)y, batch, h, w, vecs, x_ld, y_ld, es, dtype);
    return (int)hipGetLastError();
}

int odamd_spp_pool3_forward(const void* x, void* y5, void* y9, void* y13, int batch, int h, int w, int c, int x_ld,
                            int y_ld, int dtype, void* stream) {
    int es;
    int rc = check_vec(x, y5, c, x_ld, y_ld, dtype, &es);
    if (rc) return rc;
    if (!y9 || !y13 || batch < 1 || h < 1 || w < 1) return ODAMD_E_ARG;
    if (((uintptr_t)y9 & 15) || ((uintptr_t)y13 & 15)) return ODAMD_E_ALIGN;
    const int vecs = c * es / 16;
    spp_pool3_kernel<<<grid_for((size_t)batch * h * w * vecs), 256, 0, (hipStream_t)stream>>>(
        (const char*)x, (char*)y5, (char*)y9, (char*)y13, batch, h, w, vecs, x_ld, y_ld, es, dtype);
    return (int)hipGetLastError();
}

int odamd_upsample2_forward(const void* x, void* y, int batch, int h, int w, int c, int x_ld, int y_ld, int dtype,
                            void* stream) {
    int es;
    int rc = check_vec(x, y, c, x_ld, y_ld, dtype, &es);
    if (rc) return rc;
    if (batch < 1 || h < 1 || w < 1) return ODAMD_E_ARG;
    const int vecs = c * es / 16;
    upsample2_kernel<<<grid_for((size_t)batch * 4 * h * w * vecs), 256, 0, (hipStream_t)stream>>>(
        (const char*)x, (char*)y, batch, h, w, vecs, x_ld, y_ld, es);
    return (int)hipGetLastError();
}

int odamd_nchw_to_nhwc(const float* x, void* y, int batch, int c, int h, int w, int x_c, int y_ld, int dtype,
                       void* stream) {
    if (!x || !y || batch < 1 || c < 1 || h < 1 || w < 1 || y_ld < c || x_c < c) return ODAMD_E_ARG;
    dim3 grid((h * w + 31) / 32, (c + 31) / 32, batch);
    if (dtype == ODAMD_DTYPE_F32)
        layout_kernel<ODAMD_DTYPE_F32, true><<<grid, 256, 0, (hipStream_t)stream>>>((const char*)x, (char*)y, c, h * w, y_ld, x_c);
    else if (dtype == ODAMD_DTYPE_BF16)
        layout_kernel<ODAMD_DTYPE_BF16, true><<<grid, 256, 0, (hipStream_t)stream>>>((const char*)x, (char*)y, c, h * w, y_ld, x_c);
    else
        return ODAMD_E_ARG;
    return (int)hipGetLastError();
}

int odamd_nhwc_to_nchw(const void* x, float* y, int batch, int c, int h, int w, int x_ld, int y_c, int dtype,
                       void* stream) {
    if (!x || !y || batch < 1 || c < 1 || h < 1 || w < 1 || x_ld < c || y_c < c) return ODAMD_E_ARG;
    dim3 grid((h * w + 31) / 32, (c + 31) / 32, batch);
    if (dtype == ODAMD_DTYPE_F32)
        layout_kernel<ODAMD_DTYPE_F32, false><<<grid, 256, 0, (hipStream_t)stream>>>((const char*)x, (char*)y, c, h * w, x_ld, y_c);
    else if (dtype == ODAMD_DTYPE_BF16)
        layout_kernel<ODAMD_DTYPE_BF16, false><<<grid, 256, 0, (hipStream_t)stream>>>((const char*)x, (char*)y, c, h * w, x_ld, y_c);
    else
        return ODAMD_E_ARG;
    return (int)hipGetLastError();
}

}  // extern "C"

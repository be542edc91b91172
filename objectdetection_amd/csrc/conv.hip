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
#include <algorithm>
#include <type_traits>

#include <atomic>
#include "common.h"
#include "odamd.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // 16-byte register carrier

template <int DT> struct Elem;
template <> struct Elem<ODAMD_DTYPE_BF16> { static constexpr int SIZE = 2; };
template <> struct Elem<ODAMD_DTYPE_F32> { static constexpr int SIZE = 4; };

// One k-step: 32 bytes of input channels per pixel (16 bf16 / 8 f32); lane half h holds the
// 16 bytes at offset 16*h of both operands.
template <int DT>
__device__ __forceinline__ void mma_step(f32x16_t& acc, const u32x4& a, const u32x4& b) {
    if constexpr (DT == ODAMD_DTYPE_BF16) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                      acc, 0, 0, 0);
    } else {
        const f32x4_t fa = __builtin_bit_cast(f32x4_t, a), fb = __builtin_bit_cast(f32x4_t, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[j], acc, 0, 0, 0);
    }
}

__device__ __forceinline__ u32x4 vmax16(const u32x4& p, const u32x4& q, int dt) {
    u32x4 r;
    if (dt == ODAMD_DTYPE_F32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = __float_as_uint(fmaxf(__uint_as_float(p[i]), __uint_as_float(q[i])));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // two bf16 per word; compare as floats, keep the bits
            const uint32_t a = p[i], b = q[i];
            const uint32_t lo = __uint_as_float(a << 16) >= __uint_as_float(b << 16) ? (a & 0xFFFFu) : (b & 0xFFFFu);
            const uint32_t hi = __uint_as_float(a & 0xFFFF0000u) >= __uint_as_float(b & 0xFFFF0000u) ? (a & 0xFFFF0000u)
                                                                                                   : (b & 0xFFFF0000u);
            r[i] = lo | hi;
        }
    }
    return r;
}
__device__ __forceinline__ uint4 vmax16(const uint4& p, const uint4& q, int dt) {
    const u32x4 r = vmax16(u32x4{p.x, p.y, p.z, p.w}, u32x4{q.x, q.y, q.z, q.w}, dt);
    return make_uint4(r[0], r[1], r[2], r[3]);
}

template <int DT>
__device__ __forceinline__ float silu(float v) {
    if constexpr (DT == ODAMD_DTYPE_F32) return v / (1.0f + expf(-v));
    else return v * sigmoid_fast(v);
}

// NHWC epilogue shared by the convolution kernels.  `acc[i][j]`: the wave's results for pixel
// group i (32 pixels, lane r = pixel) and channel group j; `opix[i]`: output pixel index of (group i,
// lane r) relative to `ybase`, -1 when outside; `stage`: wave-private LDS slice of 32 * (CGW*128 + 16) bytes.
// Per-lane affine constants of the NHWC epilogue: lane `lane` of a wave that holds channels [cw0, cw0 + 32*CGW)
// owns CPL = 16 / element-size consecutive channels in the read-back.  Loaded unconditionally from clamped
// (always valid) addresses so that all loads are in flight together; callers issue this before their main
// loop, which hides the latency completely (conditional loads serialise into one memory round trip each).
template <int DT, int CGW>
struct Affine {
    static constexpr int CPL = 16 / Elem<DT>::SIZE;
    float scl[CPL], shf[CPL];
    __device__ __forceinline__ void load(const float* __restrict__ scale, const float* __restrict__ shift, int lane, int cw0,
                                         int Cout) {
        constexpr int LPR = CGW * 32 / CPL;
        const int c_lane = cw0 + (lane % LPR) * CPL;
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
            const int c = min(c_lane + e, Cout - 1);
            scl[e] = scale ? scale[c] : 1.0f;
            shf[e] = shift ? shift[c] : 0.0f;
        }
    }
};

template <int DT, int PGW, int CGW>
__device__ __forceinline__ void nhwc_epilogue(const f32x16_t (&acc)[PGW][CGW], const int (&opix)[PGW], char* stage, int lane,
                                              int cw0, size_t ybase, const Affine<DT, CGW>& aff, int Cout,
                                              char* __restrict__ y, int y_ld, int act, int accum) {
    constexpr int ES = Elem<DT>::SIZE;
    const int r = lane & 31, h = lane >> 5;
    // NHWC: the wave parks its raw f32 accumulators in a private LDS slice [32 pixels][32*CGW
    // channels]; reading it back, every lane owns CPL fixed consecutive channels (so it fetches
    // their scale/shift once), applies affine + SiLU and stores 16 bytes; LPR lanes cover one
    // pixel's 64..256-byte channel run.
    constexpr int CPL = 16 / ES;             // channels per lane in the read-back (16-byte store)
    constexpr int NCH = CGW * 32;            // channels per pixel held by this wave
    constexpr int RS = NCH * 4 + 16;         // padded LDS row (f32)
    constexpr int LPR = NCH / CPL;           // lanes per pixel row
    constexpr int RPP = 64 / LPR;            // pixel rows per pass
    const int part = lane % LPR;
    const int c_lane = cw0 + part * CPL;
    const float (&scl)[CPL] = aff.scl;
    const float (&shf)[CPL] = aff.shf;
#pragma unroll
    for (int i = 0; i < PGW; ++i) {
#pragma unroll
        for (int j = 0; j < CGW; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(stage + r * RS + (j * 32 + 8 * g + 4 * h) * 4) =
                    make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < 32 / RPP; ++ps) {
            const int row = ps * RPP + lane / LPR;
            const int op = __shfl(opix[i], row);
            float v[CPL];
#pragma unroll
            for (int e = 0; e < CPL; e += 4) {
                const float4 t = *(const float4*)(stage + row * RS + (part * CPL + e) * 4);
                v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w;
            }
#pragma unroll
            for (int e = 0; e < CPL; ++e) v[e] = v[e] * scl[e] + shf[e];
            if (act) {
#pragma unroll
                for (int e = 0; e < CPL; ++e) v[e] = silu<DT>(v[e]);
            }
            if (op >= 0 && c_lane < Cout) {
                char* dst = y + ((ybase + op) * y_ld + c_lane) * ES;
                if (accum) {
                    if constexpr (DT == ODAMD_DTYPE_F32) {
                        for (int e = 0; e < CPL; ++e)
                            if (c_lane + e < Cout) v[e] += ((const float*)dst)[e];
                    } else {
                        for (int e = 0; e < CPL; ++e)
                            if (c_lane + e < Cout) v[e] += bf16_to_f32(((const uint16_t*)dst)[e]);
                    }
                }
                u32x4 pk;
                if constexpr (DT == ODAMD_DTYPE_F32) {
                    pk = u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                } else {
                    pk.x = f32x2_to_bf16x2(v[0], v[1]);
                    pk.y = f32x2_to_bf16x2(v[2], v[3]);
                    pk.z = f32x2_to_bf16x2(v[4 % CPL], v[5 % CPL]);
                    pk.w = f32x2_to_bf16x2(v[6 % CPL], v[7 % CPL]);
                }
                if (c_lane + CPL <= Cout) {
                    *(u32x4*)dst = pk;
                } else {   // ragged channel tail
                    for (int e = 0; c_lane + e < Cout; ++e) {
                        if constexpr (DT == ODAMD_DTYPE_F32) ((float*)dst)[e] = v[e];
                        else ((uint16_t*)dst)[e] = f32_to_bf16(v[e]);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Branch-free form of nhwc_epilogue for the common case (channel block entirely inside Cout, overwrite, output
// tensor below 2 GiB): the per-pass loop has no control flow, so the LDS reads, the SiLU chain and the stores
// of consecutive passes overlap (with branches every pass is its own serial LDS -> VALU -> store round trip).
// Stores go through a buffer resource: lanes whose pixel lies outside the image get offset 0xFFFFFFFF,
// which the range check of the resource drops, so no lane mask is needed.
// `ytensor`: the address the pixel indices in `opix` (+ ybase) are relative to; `ybytes`: bytes addressable from it.
// ACCUM: the results are added to what the destination holds (one 16-byte read per store; a position outside the
// resource reads zero and its store is dropped); each element has exactly one writer, so no atomics.
template <int DT, int PGW, int CGW, bool ACT, bool ACCUM = false>
__device__ __forceinline__ void nhwc_epilogue_fast(const f32x16_t (&acc)[PGW][CGW], const int (&opix)[PGW], char* stage, int lane,
                                                   int cw0, int ybase, const Affine<DT, CGW>& aff, char* ytensor,
                                                   unsigned ybytes, int y_ld, int c_end = 0x7fffffff) {
    constexpr int ES = Elem<DT>::SIZE;
    const int r = lane & 31, h = lane >> 5;
    constexpr int CPL = 16 / ES;
    constexpr int NCH = CGW * 32;
    constexpr int RS = NCH * 4 + 16;
    constexpr int LPR = NCH / CPL;
    constexpr int RPP = 64 / LPR;
    const int part = lane % LPR;
    const int c_lane = cw0 + part * CPL;
    const bool c_ok = c_lane + CPL <= c_end;       // c_end: first channel that does not exist (whole 16-byte pieces before it)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ytensor, 0, ybytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < PGW; ++i) {
#pragma unroll
        for (int j = 0; j < CGW; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(stage + r * RS + (j * 32 + 8 * g + 4 * h) * 4) =
                    make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < 32 / RPP; ++ps) {
            const int row = ps * RPP + lane / LPR;
            const int op = __shfl(opix[i], row);
            float v[CPL];
#pragma unroll
            for (int e = 0; e < CPL; e += 4) {
                const float4 t = *(const float4*)(stage + row * RS + (part * CPL + e) * 4);
                v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w;
            }
#pragma unroll
            for (int e = 0; e < CPL; ++e) v[e] = v[e] * aff.scl[e] + aff.shf[e];
            if (ACT) {
#pragma unroll
                for (int e = 0; e < CPL; ++e) v[e] = silu<DT>(v[e]);
            }
            const int voff = (op >= 0 && c_ok) ? ((ybase + op) * y_ld + c_lane) * ES : -1;
            if constexpr (ACCUM) {
                const u32x4 old = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
                if constexpr (DT == ODAMD_DTYPE_F32) {
                    v[0] += __uint_as_float(old.x); v[1] += __uint_as_float(old.y);
                    v[2] += __uint_as_float(old.z); v[3] += __uint_as_float(old.w);
                } else {
                    const uint32_t w[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[(2 * e) % CPL] += __uint_as_float(w[e] << 16);
                        v[(2 * e + 1) % CPL] += __uint_as_float(w[e] & 0xFFFF0000u);
                    }
                }
            }
            u32x4 pk;
            if constexpr (DT == ODAMD_DTYPE_F32) {
                pk = u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
            } else {
                pk.x = f32x2_to_bf16x2(v[0], v[1]);
                pk.y = f32x2_to_bf16x2(v[2], v[3]);
                pk.z = f32x2_to_bf16x2(v[4 % CPL], v[5 % CPL]);
                pk.w = f32x2_to_bf16x2(v[6 % CPL], v[7 % CPL]);
            }
            __builtin_amdgcn_raw_buffer_store_b128(pk, rsrc, voff, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Host side: may the branch-free epilogue serve this output?  (whole channel blocks, 32-bit byte offsets)
static inline bool fast_epilogue_ok(size_t npix, int y_ld, int es, int Cout, int block_channels, int accum) {
    (void)accum;      // the fast epilogue has an adding form
    return Cout % block_channels == 0 && npix * (size_t)y_ld * es < ((size_t)1 << 31);
}

struct ConvP {
    const char* x; const char* w; const float* scale; const float* shift; char* y;
    int B, H, W, Ho, Wo, Cin, Cout, x_ld, y_ld;
    int Hr, Wr;  // raw input size (differs from H, W only when pooling on load)
    int TH, TW, tiles_x, tiles_y, ncb, IH, IW;
    int act, pool, nchw;
    int upz;    // input is zero-upsampled x2 on load (data gradient of a stride-2 convolution)
    int accum;  // NHWC epilogue adds to the destination instead of overwriting it
    int upskip; // upz with even tiles: pixel groups are parity classes and skip the taps that only meet zeros
    int nblocks;
    float inv_tw, inv_iw;  // 1 / TW, 1 / IW for the index splits below
    int fast_epi;          // the branch-free epilogue applies (fast_epilogue_ok)
    unsigned ybytes;       // bytes addressable from y (for the buffer resource of the fast epilogue)
};

// floor(v / d) for 0 <= v < 2^21 given inv = 1.0f / d (d < 2^12): (v + 0.5) * inv stays at least 0.5 / d
// away from an integer while its rounding error is below v * 2^-22, so truncation is exact; three
// vector instructions instead of the ~30 of an integer division.
__device__ __forceinline__ int div_small(int v, float inv) { return (int)(((float)v + 0.5f) * inv); }

// XCD-aware block order: blocks b and b+8 share an XCD (and its L2), so give each XCD a
// contiguous run of logical tiles; neighbouring tiles then share halo rows and, when
// cout > 32*CG, the channel blocks of one pixel patch share the staged input through L2.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, xcd = bid & 7, slot = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// Upper bound of staged input pixels per workgroup (the host tile chooser enforces it), so
// that the per-thread prefetch registers have a compile-time size.
__host__ __device__ constexpr int max_in_pix(int ks, int st, int pg) {
    return ks == 1 ? 32 * pg : (st == 1 ? (pg == 8 ? 400 : 232) : (pg == 4 ? 608 : 1200));
}

constexpr int UPSKIP_MAX_PIX = 160;     // real input pixels under an even output tile of at most 256 pixels: (TH/2 + 1) * (TW/2 + 1)

// CG: 32-channel groups per workgroup, PG: 32-pixel groups per workgroup,
// WP x WC = 4 waves over (pixel groups, channel groups), CKB: bytes of input channels per chunk.
//
// Pipeline per input-channel chunk: the global loads of chunk c+1 are issued into registers
// before the matrix work of chunk c starts and are written to LDS after it (one LDS buffer,
// two barriers per chunk); the address arithmetic of every staged 16-byte piece is done once
// before the loop.  POOL (2x2 max pool on load) stages synchronously instead.
// Epilogue: each wave transposes its 32-pixel x (32*CGW)-channel results through a private LDS
// slice so that global stores are 16 bytes per lane and whole 64..256-byte runs per pixel.
template <int DT, int KS, int ST, int CG, int PG, int WP, int WC, int CKB, bool POOL, bool UPSKIP = false>
__global__ void __launch_bounds__(256, UPSKIP ? (DT == ODAMD_DTYPE_BF16 ? 3 : 2) : 1) conv_igemm_kernel(ConvP p) {
    constexpr int ES = Elem<DT>::SIZE;
    constexpr int TAPS = KS * KS;
    constexpr int PS = CKB + 16;            // LDS bytes per staged pixel
    constexpr int WRS = TAPS * CKB + 16;    // LDS bytes per staged weight row
    constexpr int PGW = PG / WP, CGW = CG / WC;
    constexpr int PARTS = CKB / 16;
    constexpr int PAD = (KS - 1) / 2;
    constexpr int NP_IN = ((UPSKIP ? UPSKIP_MAX_PIX : max_in_pix(KS, ST, PG)) * PARTS + 255) / 256;
    constexpr int NP_W = (32 * CG * TAPS * PARTS + 255) / 256;
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

    // ---- per-lane LDS offsets of this wave's pixel groups / channel groups ----------------------
    int b_off[PGW], a_off[CGW];
    int opix[PGW];   // output pixel index of (group i, lane r) within the image, -1 when outside
#pragma unroll
    for (int i = 0; i < PGW; ++i) {
        const int q = (wp * PGW + i) * 32 + r;
        int ty = div_small(q, p.inv_tw), tx = q - ty * p.TW;
        if (UPSKIP) {
            // zero-upsampled input: group g holds pixels of one (row parity, column parity) class
            // g & 3, so that whole MFMA groups can skip the taps that land on inserted zeros
            const int g = wp * PGW + i, hw = p.TW >> 1;
            const int m = (g >> 2) * 32 + r;
            const int my = m / hw, mx = m - my * hw;
            ty = 2 * my + ((g >> 1) & 1);
            tx = 2 * mx + (g & 1);
        }
        const int oy = oy0 + ty, ox = ox0 + tx;
        opix[i] = (ty < p.TH && oy < p.Ho && ox < p.Wo) ? oy * p.Wo + ox : -1;
        if (ty >= p.TH) { ty = 0; tx = 0; }
        // UPSKIP stages the REAL input pixels only: output (ty, tx) of class (py, px) reads patch pixel
        // (ty / 2 + (py & (ky >> 1)), tx / 2 + (px & (kx >> 1))) for its taps
        if (UPSKIP) b_off[i] = ((ty >> 1) * p.IW + (tx >> 1)) * PS + 16 * h;
        else b_off[i] = ((ty * ST) * p.IW + tx * ST) * PS + 16 * h;
    }
#pragma unroll
    for (int i = 0; i < CGW; ++i) a_off[i] = ((wc * CGW + i) * 32 + r) * WRS + 16 * h;

    // ---- staging descriptors: global byte offset (or -1 = zero fill) and LDS offset per piece ------
    const int n_in = p.IH * p.IW * PARTS;
    const int iy0 = oy0 * ST - PAD, ix0 = ox0 * ST - PAD;
    int in_g[NP_IN], w_g[NP_W];
#pragma unroll
    for (int j = 0; j < NP_IN; ++j) {
        const int i = tid + j * 256;
        const int pix = i / PARTS, part = i - pix * PARTS;
        const int iy = div_small(pix, p.inv_iw), ix = pix - iy * p.IW;
        const int gy = iy0 + iy, gx = ix0 + ix;
        const bool inside = i < n_in && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        if (UPSKIP) {
            const int ry = (oy0 >> 1) + iy, rx = (ox0 >> 1) + ix;
            in_g[j] = (i < n_in && ry < p.Hr && rx < p.Wr) ? (ry * p.Wr + rx) * p.x_ld * ES + part * 16 : -1;
        } else if (POOL) in_g[j] = inside ? ((2 * gy) * p.Wr + 2 * gx) * p.x_ld * ES + part * 16 : -1;
        else if (p.upz) in_g[j] = (inside && !((gy | gx) & 1)) ? ((gy >> 1) * p.Wr + (gx >> 1)) * p.x_ld * ES + part * 16 : -1;
        else in_g[j] = inside ? (gy * p.W + gx) * p.x_ld * ES + part * 16 : -1;
    }
#pragma unroll
    for (int j = 0; j < NP_W; ++j) {
        const int i = tid + j * 256;
        const int row = i / (TAPS * PARTS), rem = i - row * (TAPS * PARTS);
        const int tap = rem / PARTS, part = rem - tap * PARTS;
        const int co = co0 + row;
        w_g[j] = (row < 32 * CG && co < p.Cout) ? ((co * TAPS + tap) * p.Cin) * ES + part * 16 : -1;
    }
    const char* xb = p.x + (size_t)b * ((POOL || p.upz) ? (size_t)p.Hr * p.Wr : (size_t)p.H * p.W) * p.x_ld * ES;

    f32x16_t acc[PGW][CGW];
#pragma unroll
    for (int i = 0; i < PGW; ++i)
#pragma unroll
        for (int j = 0; j < CGW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const u32x4 zero = u32x4{0u, 0u, 0u, 0u};
    u32x4 v_in[NP_IN], v_w[NP_W];
#define ODAMD_FETCH(C0)                                                                                   \
    {                                                                                                     \
        _Pragma("unroll") for (int j = 0; j < NP_IN; ++j) {                                               \
            if (POOL) {                                                                                   \
                u32x4 v = zero;                                                                           \
                if (in_g[j] >= 0) {                                                                       \
                    const char* s_ = xb + in_g[j] + (C0);                                                 \
                    const size_t px_ = (size_t)p.x_ld * ES, row_ = (size_t)p.Wr * px_;                    \
                    v = vmax16(vmax16(*(const u32x4*)s_, *(const u32x4*)(s_ + px_), DT),                  \
                               vmax16(*(const u32x4*)(s_ + row_), *(const u32x4*)(s_ + row_ + px_), DT), DT); \
                }                                                                                         \
                v_in[j] = v;                                                                              \
            } else {                                                                                      \
                v_in[j] = in_g[j] >= 0 ? *(const u32x4*)(xb + in_g[j] + (C0)) : zero;                     \
            }                                                                                             \
        }                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < NP_W; ++j)                                                  \
            v_w[j] = w_g[j] >= 0 ? *(const u32x4*)(p.w + w_g[j] + (C0)) : zero;                           \
    }
#define ODAMD_COMMIT()                                                                                    \
    {                                                                                                     \
        _Pragma("unroll") for (int j = 0; j < NP_IN; ++j) {                                               \
            const int i_ = tid + j * 256;                                                                 \
            if (i_ < n_in) *(u32x4*)(lds_in + (i_ / PARTS) * PS + (i_ % PARTS) * 16) = v_in[j];           \
        }                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < NP_W; ++j) {                                                \
            const int i_ = tid + j * 256;                                                                 \
            if (i_ < 32 * CG * TAPS * PARTS)                                                              \
                *(u32x4*)(lds_w + (i_ / (TAPS * PARTS)) * WRS + (i_ % (TAPS * PARTS)) * 16) = v_w[j];     \
        }                                                                                                 \
    }

    const int cbytes = p.Cin * ES;
    Affine<DT, CGW> aff;
    if (!p.nchw) aff.load(p.scale, p.shift, lane, co0 + wc * CGW * 32, p.Cout);
    ODAMD_FETCH(0)
    for (int c0 = 0; c0 < cbytes; c0 += CKB) {
        ODAMD_COMMIT()
        __syncthreads();
        if (c0 + CKB < cbytes) ODAMD_FETCH(c0 + CKB)  // in flight while the matrix cores work
        // float32: every chunk accumulates into a fresh tile that is then added to the running sum,
        // so the rounding-error chain is (taps * channels per chunk) + (number of chunks) long
        // instead of the whole k*k*cin reduction (matters once k*k*cin reaches the thousands).
        f32x16_t part[DT == ODAMD_DTYPE_F32 ? PGW : 1][DT == ODAMD_DTYPE_F32 ? CGW : 1];
        if constexpr (DT == ODAMD_DTYPE_F32) {
#pragma unroll
            for (int i = 0; i < PGW; ++i)
#pragma unroll
                for (int j = 0; j < CGW; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) part[i][j][e] = 0.0f;
        }
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int tap_off = ((tap / KS) * p.IW + (tap % KS)) * PS;
#pragma unroll
            for (int s = 0; s < CKB / 32; ++s) {
                u32x4 a[CGW], bb[PGW];
#pragma unroll
                for (int j = 0; j < CGW; ++j) a[j] = *(const u32x4*)(lds_w + a_off[j] + tap * CKB + s * 32);
                if (!UPSKIP) {
#pragma unroll
                    for (int i = 0; i < PGW; ++i) bb[i] = *(const u32x4*)(lds_in + b_off[i] + tap_off + s * 32);
                }
#pragma unroll
                for (int i = 0; i < PGW; ++i) {
                    if (UPSKIP) {
                        // tap (ky, kx) meets data only where (row + ky) and (column + kx) are odd
                        const int g = wp * PGW + i;
                        if (!((((g >> 1) + tap / KS) & 1) && ((g + tap % KS) & 1))) continue;
                        const int shift = ((((g >> 1) & 1) & ((tap / KS) >> 1)) * p.IW + ((g & 1) & ((tap % KS) >> 1))) * PS;
                        bb[i] = *(const u32x4*)(lds_in + b_off[i] + shift + s * 32);
                    }
#pragma unroll
                    for (int j = 0; j < CGW; ++j) {
                        if constexpr (DT == ODAMD_DTYPE_F32) mma_step<DT>(part[i][j], a[j], bb[i]);
                        else mma_step<DT>(acc[i][j], a[j], bb[i]);
                    }
                }
            }
        }
        if constexpr (DT == ODAMD_DTYPE_F32) {
#pragma unroll
            for (int i = 0; i < PGW; ++i)
#pragma unroll
                for (int j = 0; j < CGW; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] += part[i][j][e];
        }
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    // lane (r, h) holds, for pixel (group i, r), channels {8g + 4h + e : g, e in 0..3} of group j.
    const int cw0 = co0 + wc * CGW * 32;     // first channel of this wave
    if (p.nchw) {   // float32 NCHW: lanes are consecutive pixels, stores coalesce as they are
        // The affine constants of the block's channels go through LDS (free after the main loop): every lane
        // needs 16 * CGW of them, and loading them from global memory between the stores serialises the epilogue
        // into one memory round trip per channel.
        float* tab = (float*)lds;
        for (int t = tid; t < 32 * CG; t += 256) {
            const int c = min(co0 + t, p.Cout - 1);
            tab[t] = p.scale ? p.scale[c] : 1.0f;
            tab[32 * CG + t] = p.shift ? p.shift[c] : 0.0f;
        }
        __syncthreads();
        float* yb = (float*)p.y + (size_t)b * p.Cout * p.Ho * p.Wo;
        const int plane = p.Ho * p.Wo;
#pragma unroll
        for (int j = 0; j < CGW; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = (wc * CGW + j) * 32 + 8 * g + 4 * h;
                const float4 sc4 = *(const float4*)(tab + cl), sh4 = *(const float4*)(tab + 32 * CG + cl);
                const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = co0 + cl + e;
#pragma unroll
                    for (int i = 0; i < PGW; ++i) {
                        float t = acc[i][j][4 * g + e] * scv[e] + shv[e];
                        if (p.act) t = silu<DT>(t);
                        if (c < p.Cout && opix[i] >= 0) yb[(size_t)c * plane + opix[i]] = t;
                    }
                }
            }
        return;
    }
    // NHWC: transposed through a wave-private LDS slice (see nhwc_epilogue)
    char* stage = lds + wave * (32 * (CGW * 128 + 16));
    if (p.fast_epi) {
        if (cw0 >= p.Cout) return;
        if (p.accum && !p.act) nhwc_epilogue_fast<DT, PGW, CGW, false, true>(acc, opix, stage, lane, cw0, b * p.Ho * p.Wo, aff, p.y, p.ybytes, p.y_ld, p.Cout);
        else if (p.act) nhwc_epilogue_fast<DT, PGW, CGW, true>(acc, opix, stage, lane, cw0, b * p.Ho * p.Wo, aff, p.y, p.ybytes, p.y_ld, p.Cout);
        else nhwc_epilogue_fast<DT, PGW, CGW, false>(acc, opix, stage, lane, cw0, b * p.Ho * p.Wo, aff, p.y, p.ybytes, p.y_ld, p.Cout);
        return;
    }
    nhwc_epilogue<DT, PGW, CGW>(acc, opix, stage, lane, cw0, (size_t)b * p.Ho * p.Wo, aff, p.Cout, p.y, p.y_ld, p.act, p.accum);
}
#undef ODAMD_FETCH
#undef ODAMD_COMMIT

// ------------------------------------------------------------------------------------------
// Data gradient of a stride-2 3x3 convolution with few channels (bf16; dX of at most 32 channels, dY of 32 or 64:
// the first two stride-2 layers of the network, whose dX maps are the largest tensors of the backward pass).
// dX(2a + py, 2b + px) only meets the taps with ky = 1 - py (mod 2), kx = 1 - px (mod 2), which read dY(a + dr, b + dc)
// with dr, dc in {0, 1}: a wave takes a block of 4 x 8 dY pixels, loads the four shifted copies of the block straight
// from global memory as matrix operands (lane = pixel, 16 bytes of channels per k-step; no patch in LDS), and runs the
// nine (class, tap) products into four accumulator tiles, one per parity class = 128 output pixels.  The flipped,
// transposed weights ([dX channel][tap][dY channel], all taps) stay in LDS for the life of the workgroup; waves walk
// the blocks independently (no barrier in the loop).
// ------------------------------------------------------------------------------------------
struct DgS2P {
    const char* dy; const char* w; char* dx;
    int B, Hr, Wr, dy_ld, dx_ld, Cout, accum;
    int blocks_x, blocks_y, nblocks;
    unsigned ybytes;
};

template <int KSTEPS>
__global__ void __launch_bounds__(256) dgrad_s2_direct_kernel(DgS2P p) {
    constexpr int DT = ODAMD_DTYPE_BF16;
    constexpr int CDY = 16 * KSTEPS;                 // dY channels
    constexpr int WRS = 9 * CDY * 2 + 16;            // LDS bytes per weight row (one dX channel)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* lds_w = lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    char* stage = lds + ((32 * WRS + 255) & ~255) + wave * (32 * (128 + 16));
    constexpr int PPR = 9 * CDY / 8;                 // 16-byte pieces per weight row
    for (int i = tid; i < 32 * PPR; i += 256) {
        const int row = i / PPR, part = i - row * PPR;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (row < p.Cout) v = *(const u32x4*)(p.w + ((size_t)row * PPR + part) * 16);
        *(u32x4*)(lds_w + row * WRS + part * 16) = v;
    }
    __syncthreads();
    Affine<DT, 1> aff;
    aff.load(nullptr, nullptr, lane, 0, p.Cout);
    const int a_off = r * WRS + 16 * h;
    const int Ho = 2 * p.Hr, Wo = 2 * p.Wr;
    const int per_img = p.blocks_x * p.blocks_y;
    for (int blk = blockIdx.x * 4 + wave; blk < p.nblocks; blk += gridDim.x * 4) {
        const int b = blk / per_img, rem = blk - b * per_img;
        const int by = rem / p.blocks_x, bx = rem - by * p.blocks_x;
        const int my = by * 4 + (r >> 3), mx = bx * 8 + (r & 7);
        u32x4 bf[2][2][KSTEPS];
#pragma unroll
        for (int dr = 0; dr < 2; ++dr)
#pragma unroll
            for (int dc = 0; dc < 2; ++dc) {
                const bool ok = my + dr < p.Hr && mx + dc < p.Wr;
                const char* src = p.dy + (((size_t)(b * p.Hr + my + dr) * p.Wr + mx + dc) * p.dy_ld + 8 * h) * 2;
#pragma unroll
                for (int s = 0; s < KSTEPS; ++s) bf[dr][dc][s] = ok ? *(const u32x4*)(src + s * 32) : u32x4{0u, 0u, 0u, 0u};
            }
        f32x16_t acc[4][1];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[c][0][e] = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int py = c >> 1, px = c & 1;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                if (((py + ky) & 1) == 0 || ((px + kx) & 1) == 0) continue;      // compile time after unrolling
                const int dr = py & (ky >> 1), dc = px & (kx >> 1);
#pragma unroll
                for (int s = 0; s < KSTEPS; ++s)
                    mma_step<DT>(acc[c][0], *(const u32x4*)(lds_w + a_off + (tap * CDY + 16 * s) * 2), bf[dr][dc][s]);
            }
        }
        const bool inside = my < p.Hr && mx < p.Wr;
        int opix[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) opix[c] = inside ? ((b * Ho + 2 * my + (c >> 1)) * Wo + 2 * mx + (c & 1)) : -1;
        if (p.accum) nhwc_epilogue_fast<DT, 4, 1, false, true>(acc, opix, stage, lane, 0, 0, aff, p.dx, p.ybytes, p.dx_ld, p.Cout);
        else nhwc_epilogue_fast<DT, 4, 1, false, false>(acc, opix, stage, lane, 0, 0, aff, p.dx, p.ybytes, p.dx_ld, p.Cout);
    }
}

// ------------------------------------------------------------------------------------------
// Streaming 1x1 convolution (stride 1, NHWC in, NHWC out; the kernel handles up to 512 bytes of input
// channels per pixel, the dispatcher uses it up to 256, where two workgroups fit a CU).  A 1x1 convolution has no spatial structure: the pixels of the whole batch form one flat
// list.  A persistent workgroup keeps the weight slab of its 32*CG output channels in LDS for its
// whole life and walks 128-pixel tiles tile, tile + gridDim.x, ...; the global loads of the next tile
// are in registers while the matrix cores work on the current one, so HBM latency is hidden
// inside the workgroup instead of relying on occupancy, and the weights are read once per
// workgroup instead of once per tile.
// ------------------------------------------------------------------------------------------
struct StreamP {
    const char* x; const char* w; const float* scale; const float* shift; char* y;
    int npix, Cin, Cout, x_ld, y_ld, ntiles, act, accum, w_off;
    int fast_epi; unsigned ybytes;
};

template <int DT, int CG, int WP, int WC, int NPF>
__global__ void __launch_bounds__(256) conv1x1_stream_kernel(StreamP p) {
    constexpr int ES = Elem<DT>::SIZE;
    constexpr int PG = 4, PGW = PG / WP, CGW = CG / WC;
    static_assert(WP * WC == 4, "four waves");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int cbytes = p.Cin * ES;          // multiple of 32, <= 512
    const int parts = cbytes >> 4;
    const int RSB = cbytes + 16;            // LDS bytes per staged pixel / weight row
    char* lds_in = lds;                     // [128][RSB], overlaid by the epilogue slices
    char* lds_w = lds + p.w_off;            // [32*CG][RSB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wp = wave / WC, wc = wave % WC;
    const int co0 = blockIdx.y * 32 * CG;

    // weights: once
    for (int i = tid; i < 32 * CG * parts; i += 256) {
        const int row = i / parts, part = i - row * parts;
        const int co = co0 + row;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (co < p.Cout) v = *(const u32x4*)(p.w + (size_t)co * cbytes + part * 16);
        *(u32x4*)(lds_w + row * RSB + part * 16) = v;
    }
    int a_off[CGW], b_off[PGW];
#pragma unroll
    for (int j = 0; j < CGW; ++j) a_off[j] = ((wc * CGW + j) * 32 + r) * RSB + 16 * h;
#pragma unroll
    for (int i = 0; i < PGW; ++i) b_off[i] = ((wp * PGW + i) * 32 + r) * RSB + 16 * h;
    // staged piece j of this thread: pixel (tid + 256 j) / parts of the tile, 16-byte part % parts
    int g_off[NPF], l_off[NPF], g_pix[NPF];
    const int n_in = 128 * parts;
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
        const int i = tid + j * 256;
        const int pix = i / parts, part = i - pix * parts;
        g_pix[j] = i < n_in ? pix : (1 << 30);
        g_off[j] = pix * p.x_ld * ES + part * 16;
        l_off[j] = pix * RSB + part * 16;
    }
    const u32x4 zero = u32x4{0u, 0u, 0u, 0u};
    u32x4 v_in[NPF];
    const size_t tile_bytes = (size_t)128 * p.x_ld * ES;
    Affine<DT, CGW> aff;
    aff.load(p.scale, p.shift, lane, co0 + wc * CGW * 32, p.Cout);
#define ODAMD_SFETCH(T)                                                                          \
    {                                                                                            \
        const char* xb_ = p.x + (size_t)(T) * tile_bytes;                                        \
        const int left_ = p.npix - (T) * 128;                                                    \
        _Pragma("unroll") for (int j = 0; j < NPF; ++j)                                          \
            v_in[j] = g_pix[j] < left_ ? *(const u32x4*)(xb_ + g_off[j]) : zero;                 \
    }
    int t = blockIdx.x;
    if (t < p.ntiles) ODAMD_SFETCH(t)
    for (; t < p.ntiles; t += gridDim.x) {
        __syncthreads();                     // epilogue slices (overlaying lds_in) of the previous tile are done
#pragma unroll
        for (int j = 0; j < NPF; ++j)
            if (g_pix[j] < 128) *(u32x4*)(lds_in + l_off[j]) = v_in[j];
        __syncthreads();
        if (t + (int)gridDim.x < p.ntiles) ODAMD_SFETCH(t + gridDim.x)   // in flight during the matrix work
        f32x16_t acc[PGW][CGW];
#pragma unroll
        for (int i = 0; i < PGW; ++i)
#pragma unroll
            for (int j = 0; j < CGW; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        for (int c0 = 0; c0 < cbytes; c0 += 128) {
            // float32: chunk sums first, then the running sum (same rounding structure as conv_igemm_kernel)
            f32x16_t part[DT == ODAMD_DTYPE_F32 ? PGW : 1][DT == ODAMD_DTYPE_F32 ? CGW : 1];
            if constexpr (DT == ODAMD_DTYPE_F32) {
#pragma unroll
                for (int i = 0; i < PGW; ++i)
#pragma unroll
                    for (int j = 0; j < CGW; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) part[i][j][e] = 0.0f;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (c0 + s * 32 < cbytes) {
                    u32x4 a[CGW], bb[PGW];
#pragma unroll
                    for (int j = 0; j < CGW; ++j) a[j] = *(const u32x4*)(lds_w + a_off[j] + c0 + s * 32);
#pragma unroll
                    for (int i = 0; i < PGW; ++i) bb[i] = *(const u32x4*)(lds_in + b_off[i] + c0 + s * 32);
#pragma unroll
                    for (int i = 0; i < PGW; ++i)
#pragma unroll
                        for (int j = 0; j < CGW; ++j) {
                            if constexpr (DT == ODAMD_DTYPE_F32) mma_step<DT>(part[i][j], a[j], bb[i]);
                            else mma_step<DT>(acc[i][j], a[j], bb[i]);
                        }
                }
            }
            if constexpr (DT == ODAMD_DTYPE_F32) {
#pragma unroll
                for (int i = 0; i < PGW; ++i)
#pragma unroll
                    for (int j = 0; j < CGW; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][j][e] += part[i][j][e];
            }
        }
        __syncthreads();                     // every wave has finished reading lds_in
        int opix[PGW];
#pragma unroll
        for (int i = 0; i < PGW; ++i) {
            const int q = t * 128 + (wp * PGW + i) * 32 + r;
            opix[i] = q < p.npix ? q : -1;
        }
        char* stage = lds + wave * (32 * (CGW * 128 + 16));
        const int cw0 = co0 + wc * CGW * 32;
        if (p.fast_epi) {
            if (cw0 < p.Cout) {
                if (p.accum && !p.act) nhwc_epilogue_fast<DT, PGW, CGW, false, true>(acc, opix, stage, lane, cw0, 0, aff, p.y, p.ybytes, p.y_ld);
                else if (p.act) nhwc_epilogue_fast<DT, PGW, CGW, true>(acc, opix, stage, lane, cw0, 0, aff, p.y, p.ybytes, p.y_ld);
                else nhwc_epilogue_fast<DT, PGW, CGW, false>(acc, opix, stage, lane, cw0, 0, aff, p.y, p.ybytes, p.y_ld);
            }
        } else {
            nhwc_epilogue<DT, PGW, CGW>(acc, opix, stage, lane, cw0, 0, aff, p.Cout, p.y, p.y_ld, p.act, p.accum);
        }
    }
#undef ODAMD_SFETCH
}

// 64 bytes of zeros: the source line of LDS-DMA pieces that lie outside the image (zero padding)
__device__ __attribute__((aligned(64))) const unsigned char g_zero_line[64] = {};

#include "conv_dma.h"
#include "conv_s3.h"
#include "conv_wres.h"
#include "conv_mp.h"
#include "conv_mpk.h"
#include "conv_pair.h"

// ------------------------------------------------------------------------------------------
// first layer, float32 compute: float32 NCHW image, 3 input channels, 3x3 stride 1 -> NHWC
// activations as one fmaf chain per output in (ky, kx, c) order.
// Weights are wave-uniform and come through scalar loads; one thread = one output pixel.
// ------------------------------------------------------------------------------------------
template <int DT>
__global__ void __launch_bounds__(256)
conv_first_valu_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ scale,
                  const float* __restrict__ shift, char* __restrict__ y, int H, int W, int Cout, int y_ld, int act) {
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
            if (scale) a = a * scale[co] + shift[co];
            o[e] = act ? silu<DT>(a) : a;
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
// first layer: float32 NCHW image, 3 input channels, 3x3 stride 1 -> NHWC activations.
// GEMM view D[co][pixel] with K = 27 (ky, kx, c) padded to 32 (bf16: one v_mfma_f32_16x16x32_bf16)
// or 28 (f32: seven v_mfma_f32_16x16x4_f32, an exact fma chain in k order).  A workgroup stages the
// f32 halo patch of a 16 x 64 pixel tile in LDS; a lane gathers its k values of one pixel through a
// per-lane offset table (loop invariant), 16 consecutive pixels per MFMA; the weights sit in LDS as
// [co][32] f32.  D gives a lane 4 consecutive channels of one pixel: the 64 lanes of a store cover
// 16 pixels x 16 channels contiguously.
// ------------------------------------------------------------------------------------------
template <int DT>
__global__ void __launch_bounds__(256)
conv_first_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ scale,
                  const float* __restrict__ shift, char* __restrict__ y, int H, int W, int Cout, int y_ld, int act) {
    constexpr int TH = 16, TW = 64, PH = TH + 2, PW = TW + 3;        // odd row pitch
    constexpr int ES = Elem<DT>::SIZE;
    extern __shared__ __attribute__((aligned(16))) float fl[];
    float* patch = fl;                          // [3][PH][PW]
    float* wl = fl + 3 * PH * PW;               // [cout16][32]
    const int b = blockIdx.z, oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cpad = (Cout + 15) & ~15;
    for (int i = tid; i < 3 * PH * (TW + 2); i += 256) {
        const int c = i / (PH * (TW + 2)), rem = i - c * PH * (TW + 2);
        const int iy = rem / (TW + 2), ix = rem - iy * (TW + 2);
        const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
        patch[(c * PH + iy) * PW + ix] =
            (gy >= 0 && gy < H && gx >= 0 && gx < W) ? img[(((size_t)b * 3 + c) * H + gy) * W + gx] : 0.0f;
    }
    // weights [co][ky][kx][c]: k = (ky*3 + kx)*3 + c is their own order; zero beyond 27 / Cout
    for (int i = tid; i < cpad * 32; i += 256) {
        const int co = i >> 5, k = i & 31;
        wl[i] = (co < Cout && k < 27) ? w[co * 27 + k] : 0.0f;
    }
    __syncthreads();
    const int j = lane & 15, kg = lane >> 4;
    // patch offset of every k this lane feeds (padding k point at a valid word; their weight is zero)
    constexpr int NK = DT == ODAMD_DTYPE_F32 ? 7 : 8;
    int off[NK];
#pragma unroll
    for (int e = 0; e < NK; ++e) {
        int k = DT == ODAMD_DTYPE_F32 ? 4 * e + kg : 8 * kg + e;
        if (k > 26) k = 26;
        const int c = k % 3, t = k / 3;
        off[e] = (c * PH + t / 3) * PW + t % 3;
    }
    for (int cg = 0; cg < cpad; cg += 16) {
        float wa[NK];
#pragma unroll
        for (int e = 0; e < NK; ++e) wa[e] = wl[(cg + j) * 32 + (DT == ODAMD_DTYPE_F32 ? 4 * e + kg : 8 * kg + e)];
        float sc[4], sh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = min(cg + 4 * kg + e, Cout - 1);
            sc[e] = scale ? scale[co] : 1.0f;
            sh[e] = scale ? shift[co] : 0.0f;
        }
        bf16x8_t wb;
        if constexpr (DT == ODAMD_DTYPE_BF16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) wb[e] = (short)f32_to_bf16(wa[e]);
        }
        // wave w owns the 16-pixel column strip w of every tile row (TW / 16 == 4 waves): the lane's
        // gather addresses are fixed and the row advances by an immediate offset in the unrolled loop
        static_assert(TW / 16 == 4, "one strip per wave");
        const int tx = wave * 16 + j;
        const float* src[NK];
#pragma unroll
        for (int e = 0; e < NK; ++e) src[e] = patch + tx + off[e];
#pragma unroll 2
        for (int ty = 0; ty < TH; ++ty) {
            f32x4_t d = {0.f, 0.f, 0.f, 0.f};
            if constexpr (DT == ODAMD_DTYPE_F32) {
#pragma unroll
                for (int e = 0; e < NK; ++e) d = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[e], src[e][ty * PW], d, 0, 0, 0);
            } else {
                u32x4 xw;
#pragma unroll
                for (int e = 0; e < 4; ++e) xw[e] = f32x2_to_bf16x2(src[2 * e][ty * PW], src[2 * e + 1][ty * PW]);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, __builtin_bit_cast(bf16x8_t, xw), d, 0, 0, 0);
            }
            const int oy = oy0 + ty, ox = ox0 + tx;
            if (oy >= H || ox >= W) continue;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = scale ? d[e] * sc[e] + sh[e] : d[e];
            if (act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = silu<DT>(o[e]);
            }
            char* dst = y + ((((size_t)b * H + oy) * W + ox) * y_ld + cg + 4 * kg) * ES;
            if (cg + 4 * kg + 4 <= Cout) {
                if constexpr (DT == ODAMD_DTYPE_F32) {
                    *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    uint2 pk;
                    pk.x = f32x2_to_bf16x2(o[0], o[1]);
                    pk.y = f32x2_to_bf16x2(o[2], o[3]);
                    *(uint2*)dst = pk;
                }
            } else {
                for (int e = 0; cg + 4 * kg + e < Cout; ++e) {
                    if constexpr (DT == ODAMD_DTYPE_F32) ((float*)dst)[e] = o[e];
                    else ((uint16_t*)dst)[e] = f32_to_bf16(o[e]);
                }
            }
        }
    }
}

// First layer with 32 or 64 output channels (the wider presets), bf16: same staging and gather as conv_first_kernel, but
// all NQ = Cout / 16 matrix instructions of a pixel row share ONE gathered and converted B operand, and the rows of the
// weight operand are dealt so that lane group kg ends up with the 4 * NQ CONSECUTIVE channels [4 NQ kg, 4 NQ (kg + 1))
// of its pixel: row m of instruction q is channel 4 NQ (m / 4) + 4 q + m % 4.  The four lanes of a pixel then store one
// contiguous run of 64 / 128 bytes with 16-byte stores (conv_first_kernel writes 8-byte pieces of 32-byte runs, one
// pass over the tile per 16 channels: 3 -> 64 at 640 x 640, batch 64, ran at 1.4 TB/s).
template <int NQ>
__global__ void __launch_bounds__(256)
conv_first_wide_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ scale,
                       const float* __restrict__ shift, char* __restrict__ y, int H, int W, int y_ld, int act) {
    constexpr int TH = 16, TW = 64, PH = TH + 2, PW = TW + 3, Cout = 16 * NQ;
    extern __shared__ __attribute__((aligned(16))) float fl[];
    float* patch = fl;                          // [3][PH][PW]
    float* wl = fl + 3 * PH * PW;               // [Cout][32]
    const int b = blockIdx.z, oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 3 * PH * (TW + 2); i += 256) {
        const int c = i / (PH * (TW + 2)), rem = i - c * PH * (TW + 2);
        const int iy = rem / (TW + 2), ix = rem - iy * (TW + 2);
        const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
        patch[(c * PH + iy) * PW + ix] =
            (gy >= 0 && gy < H && gx >= 0 && gx < W) ? img[(((size_t)b * 3 + c) * H + gy) * W + gx] : 0.0f;
    }
    for (int i = tid; i < Cout * 32; i += 256) {
        const int co = i >> 5, k = i & 31;
        wl[i] = k < 27 ? w[co * 27 + k] : 0.0f;
    }
    __syncthreads();
    const int j = lane & 15, kg = lane >> 4;
    int off[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int k = 8 * kg + e;
        if (k > 26) k = 26;
        const int c = k % 3, t = k / 3;
        off[e] = (c * PH + t / 3) * PW + t % 3;
    }
    bf16x8_t wb[NQ];
    float sc[NQ][4], sh[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int row = 4 * NQ * (j >> 2) + 4 * q + (j & 3);
#pragma unroll
        for (int e = 0; e < 8; ++e) wb[q][e] = (short)f32_to_bf16(wl[row * 32 + 8 * kg + e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = 4 * NQ * kg + 4 * q + e;
            sc[q][e] = scale ? scale[co] : 1.0f;
            sh[q][e] = scale ? shift[co] : 0.0f;
        }
    }
    const int tx = wave * 16 + j;
    const float* src[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) src[e] = patch + tx + off[e];
#pragma unroll 2
    for (int ty = 0; ty < TH; ++ty) {
        u32x4 xw;
#pragma unroll
        for (int e = 0; e < 4; ++e) xw[e] = f32x2_to_bf16x2(src[2 * e][ty * PW], src[2 * e + 1][ty * PW]);
        f32x4_t d[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
            d[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[q], __builtin_bit_cast(bf16x8_t, xw), zero, 0, 0, 0);
        }
        const int oy = oy0 + ty, ox = ox0 + tx;
        if (oy >= H || ox >= W) continue;
        uint32_t pk[2 * NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = scale ? d[q][e] * sc[q][e] + sh[q][e] : d[q][e];
            if (act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = silu<ODAMD_DTYPE_BF16>(o[e]);
            }
            pk[2 * q] = f32x2_to_bf16x2(o[0], o[1]);
            pk[2 * q + 1] = f32x2_to_bf16x2(o[2], o[3]);
        }
        char* dst = y + ((((size_t)b * H + oy) * W + ox) * y_ld + 4 * NQ * kg) * 2;
#pragma unroll
        for (int v = 0; v < NQ / 2; ++v) *(u32x4*)(dst + 16 * v) = u32x4{pk[4 * v], pk[4 * v + 1], pk[4 * v + 2], pk[4 * v + 3]};
    }
}

// ------------------------------------------------------------------------------------------
// first + second layer in one launch (bf16): cbs0 (3 -> 16, 3x3 stride 1) feeding cbs1 (16 -> 32,
// 3x3 stride 2).  The 16-channel full-resolution activation between them is the largest tensor of
// the network (writing and re-reading it is about as much HBM traffic as the image and the second
// layer's output together); here it only exists as a bf16 patch in LDS.
// Persistent workgroups (4 waves, as many as the device keeps resident) loop over tiles of 4 x 32 output pixels of
// the second layer; the weights of both layers stay in registers / LDS for the whole launch, and the image patch of
// the next tile is fetched into registers while the current one is computed.  Per tile:
//   A. image halo patch (11 x 67 pixels) -> LDS as bf16 (c0, c1, c2, 0) records
//   B. first layer for the 9 x 65 pixels the second layer reads: three v_mfma_f32_32x32x16_bf16 per 32
//      pixels (see first_weights_frag), affine + SiLU, bf16 into the LDS patch; pixels
//      outside the image are the second layer's zero padding.  Even and odd columns are stored apart,
//      so that the stride-2 reads of phase C walk consecutive 48-byte records (bank-conflict free).
//   C. second layer: wave w owns output row w (32 pixels x 32 channels), nine v_mfma_f32_32x32x16_bf16
//      (one per tap) with the weights held in registers; NHWC epilogue through LDS as everywhere.
// ------------------------------------------------------------------------------------------
#ifdef ODAMD_LAB_STAMPS
#define F2_STAMP(i) do { if (lane == 0 && blockIdx.x < 1024 && ((i) < 2 || j == 13)) g_lab_stamps[(blockIdx.x * 8 + wave) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#define F2_STAMP_AT(i, jj) do { if (lane == 0 && blockIdx.x < 1024 && j == (jj)) g_lab_stamps[(blockIdx.x * 8 + wave) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define F2_STAMP_AT(i, jj) do { } while (0)
#define F2_STAMP(i) do { } while (0)
#endif
namespace first2 {
constexpr int TH = 4, TW = 32;   // TH = waves per workgroup
constexpr int SEG = 10;          // vertically adjacent tiles a workgroup takes in one go
constexpr int P1H = 2 * TH + 1, P1W = 2 * TW + 1;      // first-layer pixels needed: 9 x 65
constexpr int P1EVEN = (P1W + 1) / 2;                   // even columns come first in a patch row
constexpr int P0H = P1H + 2, P0W = P1W + 2, P0P = P0W + 2;   // image patch 11 x 67 (+2 zero columns)
constexpr int PS = 48;                                  // bytes per first-layer pixel record (32 + 16 pad)
constexpr int P1_BYTES = (P1H * P1W * PS + 255) & ~255;
constexpr int STAGE_BYTES = TH * 32 * (128 + 16);       // epilogue slices of the waves (reuse the patch)
constexpr int P0_BYTES = (P0H * P0P * 8 + 15) & ~15;    // bf16 (c0, c1, c2, 0) per image pixel
constexpr int W1_BYTES = 9 * 64 * 16;                   // second-layer weights as MFMA fragments
constexpr int LDS_BYTES = (P1_BYTES > STAGE_BYTES ? P1_BYTES : STAGE_BYTES) + P0_BYTES + W1_BYTES;
}

// First layer on the matrix cores from a bf16 (c0,c1,c2,0) image patch in LDS: per kernel row ky the
// 3 taps x 4 channels of a pixel are 24 contiguous bytes, so K = (ky; kx, c) is three k-steps of
// v_mfma_f32_32x32x16_bf16 whose B fragment is ONE 16-byte LDS read per lane (half h = 0: taps 0 and 1,
// h = 1: tap 2 and four values that meet zero weights).  Rows 16..31 of A are zero.
// Returns this lane's weight fragments; w0 is [16][ky][kx][3] float32.
__device__ __forceinline__ void first_weights_frag(const float* __restrict__ w0, int r, int h, u32x4 (&wa)[3]) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // unconditional loads from clamped (always valid) addresses, so that all of them are in
            // flight together; the padding positions are zeroed afterwards
            const int kx = 2 * h + i / 4, c = i % 4;
            const float t = w0[(r & 15) * 27 + (ky * 3 + (kx < 3 ? kx : 2)) * 3 + (c < 3 ? c : 2)];
            v[i] = (r < 16 && kx < 3 && c < 3) ? t : 0.0f;
        }
        wa[ky] = u32x4{f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3]), f32x2_to_bf16x2(v[4], v[5]),
                       f32x2_to_bf16x2(v[6], v[7])};
    }
}

// One tile of the persistent loop: `j` counts this workgroup's tiles; a workgroup takes items (image, column of
// tiles, run of SEG vertically adjacent tiles) and walks down the run, so the three image rows two vertically
// adjacent tiles share are read again while still in the L2 of the same XCD.  Items are dealt to the workgroups of
// one XCD as a contiguous range (workgroup w runs on XCD w % 8), which keeps horizontal neighbours on one L2 as well.
struct First2Tile { int b, oy0, ox0; bool valid; };
// First tile (k = 0) of the item this workgroup takes in round i.
__device__ __forceinline__ First2Tile first2_item(int i, int nitems, int tiles_x, int segs_y) {
    using namespace first2;
    const int w = blockIdx.x;
    const int per_xcd = (nitems + 7) >> 3, slots = gridDim.x >> 3;  // the host launches a multiple of 8 workgroups
    const int idx = (w >> 3) + i * slots;
    const int item = (w & 7) * per_xcd + idx;
    First2Tile t;
    t.valid = idx < per_xcd && item < nitems;
    const int it = t.valid ? item : 0;
    const int tx = it % tiles_x, rest = it / tiles_x, sy = rest % segs_y;
    t.b = rest / segs_y;
    t.oy0 = sy * SEG * TH;
    t.ox0 = tx * TW;
    return t;
}
// Tile k of the item whose first tile is `first` (tiles past the bottom of the image do not exist).
__device__ __forceinline__ First2Tile first2_step(const First2Tile& first, int k, int tiles_y) {
    First2Tile t = first;
    t.oy0 = first.oy0 + k * first2::TH;
    t.valid = first.valid && t.oy0 < tiles_y * first2::TH;
    if (!t.valid) t.b = t.oy0 = t.ox0 = 0;
    return t;
}

__global__ void __launch_bounds__(64 * first2::TH) __attribute__((amdgpu_waves_per_eu(3, 3)))
conv_first2_kernel(const float* __restrict__ img, const float* __restrict__ w0, const float* __restrict__ scale0,
                   const float* __restrict__ shift0, const char* __restrict__ w1, const float* __restrict__ scale1,
                   const float* __restrict__ shift1, char* __restrict__ y, int H, int W, int Ho, int Wo, int y_ld,
                   unsigned ybytes, int nitems, int tiles_x, int tiles_y, int segs_y, int rounds) {
    using namespace first2;
    constexpr int DT = ODAMD_DTYPE_BF16;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* p1 = lds;
    char* p0 = lds + (P1_BYTES > STAGE_BYTES ? P1_BYTES : STAGE_BYTES);
    char* wl = p0 + P0_BYTES;              // second-layer weight fragments
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    int j = 0;
    F2_STAMP(0);

    // ---- constants of the whole launch --------------------------------------------------------------
    // second-layer weights [32][3][3][16] bf16 -> LDS in fragment order [tap][lane (r, h)][8 input channels]
    for (int t = tid; t < 9 * 64; t += 64 * TH) {
        const int tap = t >> 6, l = t & 63;
        *(u32x4*)(wl + t * 16) = *(const u32x4*)(w1 + (((l & 31) * 9 + tap) * 16 + 8 * (l >> 5)) * 2);
    }
    u32x4 wa[3];
    first_weights_frag(w0, r, h, wa);
    Affine<DT, 1> aff1;
    aff1.load(scale1, shift1, lane, 0, 32);
    float sc[8], sh[8];   // channels 4h+e (e < 4) and 8+4h+(e-4)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = (e < 4 ? 0 : 8) + 4 * h + (e & 3);
        sc[e] = scale0[c];
        sh[e] = shift0[c];
    }
    // first-layer pixel groups of this wave: group g = wave + TH * slot; 2 groups per patch row (columns 0..31,
    // 32..63) and one last group for column 64 of all rows.  Source / destination offsets do not depend on the tile.
    constexpr int NGRP = 2 * P1H + 1, NSLOT = (NGRP + TH - 1) / TH;
    static_assert(P1W == 65 && P1H <= 32, "group layout of the first-layer patch");
    int src_off[NSLOT], dst_off[NSLOT];
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
        const int g = min(wave + TH * sl, NGRP - 1);
        const bool col = g == NGRP - 1;
        const int py = col ? min(r, P1H - 1) : g >> 1, px = col ? P1W - 1 : (g & 1) * 32 + r;
        src_off[sl] = (py * P0P + px + 2 * h) * 8;
        dst_off[sl] = (py * P1W + (px & 1) * P1EVEN + (px >> 1)) * PS + 8 * h;
    }

    // image patch of a tile: P0H x P0P pixels dealt to the threads in row-major order (NSL per thread), three
    // colour planes each.  Offsets relative to the patch origin do not depend on the tile; a tile whose patch
    // crosses the image border clamps every address (always valid) and zeroes those positions when it writes.
    constexpr int NP0 = P0H * P0P, NSL = (NP0 + 64 * TH - 1) / (64 * TH);
    int yx[NSL];           // patch row << 8 | patch column
#pragma unroll
    for (int k = 0; k < NSL; ++k) {
        const int idx = min(tid + 64 * TH * k, NP0 - 1);
        const int iy = idx / P0P, ix = idx - iy * P0P;
        yx[k] = iy << 8 | ix;
    }
    float v[NSL][3];
    const int plane = H * W;
    auto is_border = [&](const First2Tile& t) {
        const int gy0 = 2 * t.oy0 - 2, gx0 = 2 * t.ox0 - 2;
        return gy0 < 0 || gy0 + P0H > H || gx0 < 0 || gx0 + P0P > W;
    };
    auto fetch = [&](const First2Tile& t) {
        const int gy0 = 2 * t.oy0 - 2, gx0 = 2 * t.ox0 - 2;
        const float* ib = img + (size_t)t.b * 3 * plane;
        if (is_border(t)) {
#pragma unroll
            for (int k = 0; k < NSL; ++k) {
                const int gy = min(max(gy0 + (yx[k] >> 8), 0), H - 1), gx = min(max(gx0 + (yx[k] & 255), 0), W - 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) v[k][c] = ib[c * plane + gy * W + gx];
            }
        } else {
            const float* org = ib + gy0 * W + gx0;
#pragma unroll
            for (int k = 0; k < NSL; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) v[k][c] = org[c * plane + (yx[k] >> 8) * W + (yx[k] & 255)];
        }
    };
    // (a fetch is never conditional: a fixed number of loads in flight lets the compiler count them instead of
    // draining the queue; a tile that does not exist fetches the pixels of tile (0, 0, 0), which nobody uses)
    First2Tile item = first2_item(0, nitems, tiles_x, segs_y);
    First2Tile cur = first2_step(item, 0, tiles_y);
    fetch(cur);
    F2_STAMP(1);

    // first layer of two pixel groups at a time (independent accumulators keep the matrix pipe and the
    // transcendental unit busy): slots s0 and s1 = s0 + 1 (s1 may not exist)
    auto layer1 = [&](auto border_tag, int s0, bool has1, int gy0, int gx0) {
        constexpr bool BORDER = decltype(border_tag)::value;
        const int s1 = s0 + 1;
        f32x16_t d0, d1;
#pragma unroll
        for (int e = 0; e < 16; ++e) d0[e] = d1[e] = 0.0f;
        const char* q0 = p0 + src_off[s0];
        const char* q1 = p0 + src_off[has1 ? s1 : s0];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const uint2 lo0 = *(const uint2*)(q0 + ky * P0P * 8), hi0 = *(const uint2*)(q0 + ky * P0P * 8 + 8);
            const uint2 lo1 = *(const uint2*)(q1 + ky * P0P * 8), hi1 = *(const uint2*)(q1 + ky * P0P * 8 + 8);
            mma_step<DT>(d0, wa[ky], u32x4{lo0.x, lo0.y, hi0.x, hi0.y});
            mma_step<DT>(d1, wa[ky], u32x4{lo1.x, lo1.y, hi1.x, hi1.y});
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int sl = u ? s1 : s0;
            if (u && !has1) break;                            // wave-uniform
            const f32x16_t& d = u ? d1 : d0;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = silu<DT>(d[e] * sc[e] + sh[e]);
            uint2 pk0 = make_uint2(f32x2_to_bf16x2(o[0], o[1]), f32x2_to_bf16x2(o[2], o[3]));
            uint2 pk1 = make_uint2(f32x2_to_bf16x2(o[4], o[5]), f32x2_to_bf16x2(o[6], o[7]));
            const int g = wave + TH * sl;
            const bool col = g == NGRP - 1;
            bool live = !col || r < P1H;
            if constexpr (BORDER) {                           // the second layer's zero padding
                const int py = col ? min(r, P1H - 1) : g >> 1, px = col ? P1W - 1 : (g & 1) * 32 + r;
                const int gy = gy0 + 1 + py, gx = gx0 + 1 + px;
                if (!(gy >= 0 && gy < H && gx >= 0 && gx < W)) pk0 = pk1 = make_uint2(0u, 0u);
            }
            if (live) {
                *(uint2*)(p1 + dst_off[sl]) = pk0;
                *(uint2*)(p1 + dst_off[sl] + 16) = pk1;
            }
        }
    };

    const int njobs = rounds * SEG;
    for (int k = 0; j < njobs; ++j) {
        if (++k == SEG) {           // the divisions of the item decode run once per SEG tiles
            k = 0;
            item = first2_item(j / SEG + 1, nitems, tiles_x, segs_y);
            item.valid = item.valid && j + 1 < njobs;
        }
        const First2Tile nxt = first2_step(item, k, tiles_y);
        if (!cur.valid) {           // workgroup-uniform: ragged ends of the item range and of a column of tiles
            fetch(nxt);
            cur = nxt;
            continue;
        }
        const int b = cur.b, oy0 = cur.oy0, ox0 = cur.ox0;
        const int gy0 = 2 * oy0 - 2, gx0 = 2 * ox0 - 2;
        // ---- A: image patch, converted to bf16 once ------------------------------------------------
        const bool border = is_border(cur);
#pragma unroll
        for (int k = 0; k < NSL; ++k) {
            const int iy = yx[k] >> 8, ix = yx[k] & 255;
            bool in = ix < P0W;
            if (border) {
                const int gy = gy0 + iy, gx = gx0 + ix;
                in = in && gy >= 0 && gy < H && gx >= 0 && gx < W;
            }
            const uint2 pk = make_uint2(f32x2_to_bf16x2(v[k][0], v[k][1]), f32x2_to_bf16x2(v[k][2], 0.0f));
            if (k + 1 < NSL || tid + 64 * TH * k < NP0) *(uint2*)(p0 + (iy * P0P + ix) * 8) = in ? pk : make_uint2(0u, 0u);
        }
        F2_STAMP(2);
        __syncthreads();
        F2_STAMP(3);
        // the next tile's pixels travel while this one is computed
        fetch(nxt);
        // ---- B: first layer into the bf16 patch -----------------------------------------------------
#pragma unroll
        for (int s0 = 0; s0 < NSLOT; s0 += 2) {
            if (wave + TH * s0 >= NGRP) break;                                   // wave-uniform
            const bool has1 = s0 + 1 < NSLOT && wave + TH * (s0 + 1) < NGRP;
            if (border) layer1(std::true_type{}, s0, has1, gy0, gx0);
            else layer1(std::false_type{}, s0, has1, gy0, gx0);
        }
        F2_STAMP(4);
        __syncthreads();
        F2_STAMP(5);
        // ---- C: second layer ------------------------------------------------------------------------
        f32x16_t acc[1][1];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][0][e] = 0.0f;
        const int ty = wave;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const u32x4 aa = *(const u32x4*)(wl + (tap * 64 + lane) * 16);
            const u32x4 bb = *(const u32x4*)(p1 + ((2 * ty + ky) * P1W + (kx & 1) * P1EVEN + r + (kx >> 1)) * PS + 16 * h);
            mma_step<DT>(acc[0][0], aa, bb);
        }
        F2_STAMP(6);
        __syncthreads();   // every wave is done reading the patch: its memory becomes the epilogue slices
        F2_STAMP(7);
        const int oy = oy0 + ty, ox = ox0 + r;
        int opix[1] = {(oy < Ho && ox < Wo) ? oy * Wo + ox : -1};
        if (ybytes)
            nhwc_epilogue_fast<DT, 1, 1, true>(acc, opix, lds + wave * (32 * (128 + 16)), lane, 0, b * Ho * Wo, aff1, y, ybytes, y_ld);
        else
            nhwc_epilogue<DT, 1, 1>(acc, opix, lds + wave * (32 * (128 + 16)), lane, 0, (size_t)b * Ho * Wo, aff1, 32, y, y_ld, 1, 0);
        F2_STAMP(8);
        F2_STAMP_AT(9, 12);
        cur = nxt;
    }
    F2_STAMP_AT(10, njobs);
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

// Same three pools as a cascade for small maps: pool9 = pool5(pool5(x)), pool13 = pool5(pool9) (maxima
// are exact, so the results are identical), each 5x5 pool as a row pass and a column pass.  One
// workgroup per (image, VPW consecutive 16-byte channel vectors) keeps the whole H x W plane of its channels in
// LDS: 30 window reads per output instead of 169, and every global access is a run of 16 VPW contiguous bytes
// per pixel (a workgroup per single vector touches 16 bytes of every 128-byte line).
template <int VPW>
__global__ void __launch_bounds__(256)
spp_cascade_kernel(const char* __restrict__ x, char* __restrict__ y5, char* __restrict__ y9, char* __restrict__ y13,
                   int H, int W, int vecs, int x_ld, int y_ld, int es, int dt) {
    extern __shared__ __attribute__((aligned(16))) uint4 plane[];
    const int hw = H * W, v0 = blockIdx.x * VPW, b = blockIdx.y;
    const int nv = min(VPW, vecs - v0);                 // vectors of this workgroup (the last one may hold fewer)
    const int n = hw * VPW;                             // items: (pixel, vector) with the vector fastest
    uint4* A = plane;
    uint4* Bp = plane + n;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int p = i / VPW, v = i - p * VPW;
        if (v < nv) A[i] = *(const uint4*)(x + (((size_t)b * hw + p) * x_ld) * es + (v0 + v) * 16);
    }
    char* outs[3] = {y5, y9, y13};
#pragma unroll 1
    for (int stage = 0; stage < 3; ++stage) {
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) {        // row pass A -> Bp
            const int p = i / VPW, px = p % W;
            uint4 m = A[i];
            for (int d = -2; d <= 2; ++d)
                if (d != 0 && px + d >= 0 && px + d < W) m = vmax16(m, A[i + d * VPW], dt);
            Bp[i] = m;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) {        // column pass Bp -> A, and out
            const int p = i / VPW, v = i - p * VPW, py = p / W;
            uint4 m = Bp[i];
            for (int d = -2; d <= 2; ++d)
                if (d != 0 && py + d >= 0 && py + d < H) m = vmax16(m, Bp[i + d * W * VPW], dt);
            A[i] = m;
            if (v < nv) *(uint4*)(outs[stage] + (((size_t)b * hw + p) * y_ld) * es + (v0 + v) * 16) = m;
        }
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

// NCHW float32 planes -> NHWC pixels of eight channels, for tensors of at most eight channels (the image): one
// thread per pixel reads its value from every plane (coalesced along the plane) and writes the pixel's whole
// vector, channels beyond C as zeros.  The tile kernel above would use 3 of its 32 channel lanes here.
template <int DT>
__global__ void __launch_bounds__(256)
few_planes_to_nhwc8_kernel(const float* __restrict__ x, char* __restrict__ y, int C, size_t HW, int Ct, size_t npix) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        const size_t b = i / HW, pp = i % HW;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = c < C ? x[(b * Ct + c) * HW + pp] : 0.0f;
        if (DT == ODAMD_DTYPE_F32) {
            float4* o = (float4*)(y + i * 32);
            o[0] = make_float4(v[0], v[1], v[2], v[3]);
            o[1] = make_float4(v[4], v[5], v[6], v[7]);
        } else {
            uint4 o;
            o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
            o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
            o.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
            o.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
            *(uint4*)(y + i * 16) = o;
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
TileChoice choose_tile(int Ho, int Wo, int npix, int ks, int st, int ps, size_t w_bytes, int max_pix, bool even = false,
                       size_t lds_limit = LDS_LIMIT) {
    const int cand[] = {Wo, 64, 40, 32, 20, 16, 8};
    TileChoice best = {0, 0, 0, 0, false};
    double best_cost = 1e30;
    for (int tw : cand) {
        if (tw > Wo || tw > npix || tw < 1) continue;
        int th = npix / tw;
        if (th > Ho) th = Ho;
        if (even) { if ((tw & 1) || th < 2) continue; th &= ~1; }
        // the tallest patch of this width whose halo and weight slab fit
        for (; th >= (even ? 2 : 1); th -= (even ? 2 : 1)) {
            const int ih = (th - 1) * st + ks, iw = (tw - 1) * st + ks;
            if (ih * iw > max_pix) continue;
            if ((((size_t)ih * iw * ps + 255) & ~(size_t)255) + w_bytes > lds_limit) continue;
            const double tiles = (double)((Wo + tw - 1) / tw) * ((Ho + th - 1) / th);
            const double cost = tiles * (ih * iw + 0.25 * npix);  // staged pixels + a share for idle lanes
            if (cost < best_cost) { best_cost = cost; best = {th, tw, ih, iw, true}; }
            break;
        }
    }
    return best;
}

template <int DT, int KS, int ST, int CG, int PG, int WP, int WC, int CKB, bool POOL>
int launch_conv(ConvP p, hipStream_t st) {
    constexpr int ES = Elem<DT>::SIZE;
    constexpr int PS = CKB + 16, WRS = KS * KS * CKB + 16;
    const size_t w_bytes = (size_t)32 * CG * WRS;
    TileChoice t = {0, 0, 0, 0, false};
    p.upskip = 0;
    if (KS == 3 && ST == 1 && p.upz && PG % 4 == 0) {   // even tiles make the parity-class tap skipping possible
        t = choose_tile(p.Ho, p.Wo, 32 * PG, KS, ST, PS, w_bytes, max_in_pix(KS, ST, PG), true);
        if (t.ok && (t.th / 2 + 1) * (t.tw / 2 + 1) > UPSKIP_MAX_PIX) t.ok = false;
        p.upskip = t.ok;
    }
    if (!t.ok) t = choose_tile(p.Ho, p.Wo, 32 * PG, KS, ST, PS, w_bytes, max_in_pix(KS, ST, PG));
    if (!t.ok) return ODAMD_E_SHAPE;
    p.TH = t.th; p.TW = t.tw; p.IH = t.ih; p.IW = t.iw;
    if (p.upskip) { p.IH = t.th / 2 + 1; p.IW = t.tw / 2 + 1; }      // the real pixels under the tile, no inserted zeros
    p.inv_tw = 1.0f / (float)t.tw; p.inv_iw = 1.0f / (float)p.IW;
    p.tiles_x = (p.Wo + t.tw - 1) / t.tw;
    p.tiles_y = (p.Ho + t.th - 1) / t.th;
    p.ncb = (p.Cout + 32 * CG - 1) / (32 * CG);
    p.nblocks = p.B * p.tiles_x * p.tiles_y * p.ncb;
    {
        const size_t npix = (size_t)p.B * p.Ho * p.Wo;
        // this kernel masks the lanes past the last channel, so whole 16-byte pieces are enough
        p.fast_epi = !p.nchw && !(p.accum && p.act) && fast_epilogue_ok(npix, p.y_ld, ES, p.Cout, 16 / ES, p.accum);
        p.ybytes = p.fast_epi ? (unsigned)((npix - 1) * p.y_ld * ES + (size_t)p.Cout * ES) : 0u;
    }
    size_t lds = (((size_t)p.IH * p.IW * PS + 255) & ~(size_t)255) + w_bytes;
    const size_t epi = (size_t)4 * 32 * ((CG / WC) * 32 * 4 + 16);    // the waves' f32 transpose slices
    if (lds < epi) lds = epi;
    if constexpr (KS == 3 && ST == 1 && PG % 4 == 0 && !POOL) {
        if (p.upskip) {
            conv_igemm_kernel<DT, KS, ST, CG, PG, WP, WC, CKB, POOL, true><<<p.nblocks, 256, lds, st>>>(p);
            return (int)hipGetLastError();
        }
    }
    conv_igemm_kernel<DT, KS, ST, CG, PG, WP, WC, CKB, POOL><<<p.nblocks, 256, lds, st>>>(p);
    return (int)hipGetLastError();
}

template <int DT, int KS, int ST, int CKB, bool POOL>
int dispatch_cg(const ConvP& p, hipStream_t st) {
    if (p.Cout <= 32) return launch_conv<DT, KS, ST, 1, ST == 2 ? 4 : 8, 4, 1, CKB, POOL>(p, st);
    if (p.Cout <= 64) return launch_conv<DT, KS, ST, 2, ST == 2 ? 4 : 8, 4, 1, CKB, POOL>(p, st);
    return launch_conv<DT, KS, ST, 4, 4, 2, 2, CKB, POOL>(p, st);
}

template <int DT, int CG, int WP, int WC>
int launch_stream(const ConvP& c, hipStream_t st) {
    constexpr int ES = Elem<DT>::SIZE;
    StreamP p = {};
    p.x = c.x; p.w = c.w; p.scale = c.scale; p.shift = c.shift; p.y = c.y;
    p.npix = c.B * c.H * c.W; p.Cin = c.Cin; p.Cout = c.Cout; p.x_ld = c.x_ld; p.y_ld = c.y_ld;
    p.ntiles = (p.npix + 127) / 128; p.act = c.act; p.accum = c.accum;
    p.fast_epi = !(p.accum && p.act) && fast_epilogue_ok((size_t)p.npix, p.y_ld, ES, p.Cout, 32 * (CG / WC), p.accum);
    p.ybytes = p.fast_epi ? (unsigned)(((size_t)p.npix - 1) * p.y_ld * ES + (size_t)p.Cout * ES) : 0u;
    const int cbytes = c.Cin * ES, rsb = cbytes + 16;
    const size_t epi = (size_t)4 * 32 * ((CG / WC) * 128 + 16);
    size_t in_bytes = (size_t)128 * rsb;
    if (in_bytes < epi) in_bytes = epi;
    p.w_off = (int)((in_bytes + 255) & ~(size_t)255);
    const size_t lds = p.w_off + (size_t)32 * CG * rsb;
    const int ncb = (c.Cout + 32 * CG - 1) / (32 * CG);
    int gx = (lds <= 80 * 1024 ? 512 : 256) / ncb;
    gx = (gx + 7) & ~7;                       // channel blocks of one tile land on the same XCD
    if (gx > p.ntiles) gx = p.ntiles;
    const dim3 grid(gx, ncb);
    const int parts = cbytes / 16;
#define ODAMD_STREAM(NPF)                                                                                        \
    {                                                                                                            \
        auto k = conv1x1_stream_kernel<DT, CG, WP, WC, NPF>;                                                     \
        (void)odamd_ensure_lds((const void*)k, lds);                                                            \
        k<<<grid, 256, lds, st>>>(p);                                                                            \
    }
    if (parts <= 8) ODAMD_STREAM(4)
    else if (parts <= 16) ODAMD_STREAM(8)
    else return ODAMD_E_SHAPE;
#undef ODAMD_STREAM
    return (int)hipGetLastError();
}

// ---- LDS-DMA kernel: configuration, launch -------------------------------------------------------
struct DmaConfig { int ks, pg, cg, wp, wc, ckb, nbuf, st; };
#ifdef ODAMD_LAB
static int g_lab_no_s3 = 0;
static int g_lab_no_dma_s2 = 0;
struct S3Config;
DmaConfig g_lab_force = {};      // tools/micro/conv_lab.hip tries instantiations through this
#endif

// Which conv_dma_kernel instantiation serves a layer (bf16, stride 1, plain NHWC in / out), or pg == 0.
// Output tile and patch of a stride-2 3x3 layer for the LDS-DMA kernel: TH x TW <= 32 * pg output pixels, patch rows of
// [TW + 1 even][TW + 1 odd] column slots.  Picks the shape that stages the fewest slots overall; ok = false when
// none fits `max_slots` slots per 16-byte part plane.
struct S2Tile { int th, tw, ih, iw, iwh, slots; bool ok; };
S2Tile choose_tile_s2(int Ho, int Wo, int pg, int max_slots) {
    const int cand[] = {Wo, 64, 40, 32, 20, 16, 10, 8};
    S2Tile best = {0, 0, 0, 0, 0, 0, false};
    double best_cost = 1e30;
    for (int tw : cand) {
        if (tw > Wo || tw > 32 * pg || tw < 1) continue;
        for (int th = std::min(Ho, 32 * pg / tw); th >= 1; --th) {
            const int ih = 2 * (th - 1) + 3, iwh = tw + 1, slots = ih * 2 * iwh;
            if (slots > max_slots) continue;
            const double tiles = (double)((Wo + tw - 1) / tw) * ((Ho + th - 1) / th);
            const double cost = tiles * (slots + 0.5 * 32 * pg);      // staged slots + a share for idle lanes
            if (cost < best_cost) { best_cost = cost; best = {th, tw, ih, 2 * (tw - 1) + 3, iwh, slots, true}; }
            break;
        }
    }
    return best;
}

DmaConfig dma_config(const ConvP& p, int ks, int stride, int dtype) {
    DmaConfig none = {};
    if (dtype != ODAMD_DTYPE_BF16 || p.pool || p.upz || p.nchw || p.accum) return none;
    if (stride == 2) {
#ifdef ODAMD_LAB
        if (g_lab_force.pg) return g_lab_force.ks == ks && g_lab_force.st == 2 ? g_lab_force : none;
        if (g_lab_no_dma_s2) return none;
#endif
        // mp.cbs3 (model/layer.py:143): 3x3 stride 2 with 64 .. 256 channels on <= 160 x 160 inputs; with 128 output
        // channels and more also the stem layers of the wider presets on their 320 x 320 / 640 x 640 inputs
        // (measured at batch 64: 128 -> 256 at 320 x 320 1.87 -> 1.54 ms, 64 -> 128 at 640 x 640 2.15 -> 2.01 ms)
        if (ks != 3 || p.Cin < 64 || p.Cin % 16 || p.Cout < 64) return none;
        if (p.H * p.W > 160 * 160 && p.Cout < 128) return none;
        if ((long long)p.B * ((p.Ho * p.Wo + 63) / 64) * ((p.Cout + 63) / 64) >= (1 << 21)) return none;   // block index range of the kernel
        if (!choose_tile_s2(p.Ho, p.Wo, 4, dma_max_in_pix(3, 4, 2)).ok) return none;
        // measured at batch 64 (tools/micro/conv_lab.hip): 128 output channels per workgroup from 128 channels up
        // ... and 256 output pixels per workgroup on 8 waves where such a tile exists: the stride-2 patch is four input pixels
        // per output pixel, and the larger tile halves the weight bytes staged per pixel (128 -> 256 at 320 x 320 1.60 ->
        // 1.37 ms, 256 -> 256 at 160 x 160 0.71 -> 0.62 ms, at 40 x 40 53 -> 48 us, 128 -> 128 at 80 x 80 63 -> 58 us)
        if (p.Cout >= 128) {
            // (only where that still leaves a workgroup per compute unit: 128 -> 128 at 40 x 40, 128 workgroups, 16 -> 22 us)
            const long long big_tiles = (long long)p.B * ((p.Ho * p.Wo + 255) / 256) * ((p.Cout + 127) / 128);
            if (big_tiles >= 256 && choose_tile_s2(p.Ho, p.Wo, 8, dma_max_in_pix(3, 8, 2)).ok) return {3, 8, 4, 4, 2, 32, 2, 2};
            return {3, 4, 4, 2, 2, 32, 2, 2};
        }
        return {3, 4, 2, 4, 1, 32, 2, 2};
    }
    if (stride != 1) return none;
    const long long npix = (long long)p.B * p.H * p.W;
#ifdef ODAMD_LAB
    if (g_lab_force.pg) return g_lab_force.ks == ks && g_lab_force.st != 2 ? g_lab_force : none;
#endif
    // measured against conv_igemm_kernel / conv1x1_stream_kernel at batch 64 (tools/micro/conv_lab.hip): the DMA
    // kernel wins where K = taps * cin is deep and the map is small (weights dominate the staged bytes)
    if (ks == 3) {
        // larger maps only from 128 input channels (K = 1152) up: 128 -> 128 at 160 x 160 0.74 -> 0.52 ms, at
        // 320 x 320 2.78 -> 2.07 ms (the ELAN chains and the stem of the widest preset)
        if (p.Cin < 64 || p.Cout < 64 || (p.H * p.W > 80 * 80 && p.Cin < 128)) return none;
        if ((long long)p.B * ((p.H * p.W + 63) / 64) * ((p.Cout + 63) / 64) >= (1 << 21)) return none;
        return {3, 8, 2, 4, 1, 32, 2, 1};
    }
    if (ks == 1) {
        if (p.Cin % 32 || p.Cin < 256 || p.Cout < 128 || npix > 64ll * 40 * 40) return none;
        return {1, 4, 4, 2, 2, 64, 3, 1};
    }
    return none;
}

template <int KS, int PG, int CG, int WP, int WC, int CKB, int NBUF, int ST = 1>
int launch_dma(const ConvP& c, hipStream_t st) {
    constexpr int PARTS = CKB / 16, KSTEPS = CKB / 32, TAPS = KS * KS, NW = WP * WC;
    constexpr size_t W_BYTES = (size_t)CG * TAPS * KSTEPS * 1024;
    constexpr size_t LDS_MAX = 160 * 1024;
    DmaP p = {};
    p.x = c.x; p.w = c.w; p.scale = c.scale; p.shift = c.shift; p.y = c.y;
    p.Cin = c.Cin; p.Cout = c.Cout; p.x_ld = c.x_ld; p.y_ld = c.y_ld; p.act = c.act;
    p.Ho = c.Ho; p.Wo = c.Wo;
    if (KS == 1) {            // no spatial structure: the batch is one row of pixels
        p.B = 1; p.H = 1; p.W = c.B * c.H * c.W;
        p.Ho = 1; p.Wo = p.W;
        p.TH = 1; p.TW = 32 * PG; p.IH = 1; p.IW = p.TW;
    } else if (ST == 2) {
        p.B = c.B; p.H = c.H; p.W = c.W;
        const S2Tile t = choose_tile_s2(c.Ho, c.Wo, PG, dma_max_in_pix(KS, PG, 2));
        if (!t.ok) return ODAMD_E_SHAPE;
        p.TH = t.th; p.TW = t.tw; p.IH = t.ih; p.IW = t.iw; p.IWH = t.iwh;
    } else {
        p.B = c.B; p.H = c.H; p.W = c.W;
        // per buffer: PARTS * npad * 16 + W_BYTES; the chooser sees bytes per staged pixel and the rest
        const size_t per_buf_limit = LDS_MAX / NBUF;
        TileChoice t = choose_tile(c.H, c.W, 32 * PG, KS, 1, PARTS * 16, W_BYTES + PARTS * 16 * 64, dma_max_in_pix(KS, PG), false,
                                   per_buf_limit);
        if (!t.ok) return ODAMD_E_SHAPE;
        p.TH = t.th; p.TW = t.tw; p.IH = t.ih; p.IW = t.iw;
    }
    p.IWP = ST == 2 ? 2 * p.IWH : p.IW;
    p.npad = (p.IH * p.IWP + 63) & ~63;
    p.inv_tw = 1.0f / (float)p.TW; p.inv_iw = 1.0f / (float)p.IWP;
    p.tiles_x = (p.Wo + p.TW - 1) / p.TW;
    p.tiles_y = (p.Ho + p.TH - 1) / p.TH;
    p.ncb = (p.Cout + 32 * CG - 1) / (32 * CG);
    p.nchunk = p.Cin * 2 / CKB;
    p.nblocks = p.B * p.tiles_x * p.tiles_y * p.ncb;
    {
        const size_t npix = (size_t)p.B * p.Ho * p.Wo;
        p.fast_epi = fast_epilogue_ok(npix, p.y_ld, 2, p.Cout, 32 * (CG / WC), 0);
        p.ybytes = p.fast_epi ? (unsigned)((npix - 1) * p.y_ld * 2 + (size_t)p.Cout * 2) : 0u;
    }
    p.inv_ncb = 1.0f / (float)p.ncb; p.inv_tiles = 1.0f / (float)(p.tiles_x * p.tiles_y); p.inv_tiles_x = 1.0f / (float)p.tiles_x;
    // div_small is exact for dividends below 2^21 and divisors below 2^12
    if (p.nblocks >= (1 << 21) || p.tiles_x * p.tiles_y >= 4096 || p.ncb >= 4096) return ODAMD_E_SHAPE;
    size_t lds = (size_t)NBUF * ((size_t)PARTS * p.npad * 16 + W_BYTES);
    const size_t epi = (size_t)NW * 32 * ((CG / WC) * 128 + 16);
    if (lds < epi) lds = epi;
    if (lds > LDS_MAX) return ODAMD_E_SHAPE;
    if (p.npad > ((dma_max_in_pix(KS, PG, ST) + 63) & ~63)) return ODAMD_E_SHAPE;
    auto k = conv_dma_kernel<KS, PG, CG, WP, WC, CKB, NBUF, ST>;
    (void)odamd_ensure_lds((const void*)k, lds);
    k<<<p.nblocks, 64 * NW, lds, st>>>(p);
    return (int)hipGetLastError();
}

int dispatch_dma(const ConvP& p, const DmaConfig& g, hipStream_t st) {
#define ODAMD_DMA_CASE(KS, PG, CG, WP, WC, CKB, NBUF)                                                          \
    if (g.st != 2 && g.ks == KS && g.pg == PG && g.cg == CG && g.wp == WP && g.wc == WC && g.ckb == CKB && g.nbuf == NBUF)  \
        return launch_dma<KS, PG, CG, WP, WC, CKB, NBUF>(p, st);
#define ODAMD_DMA_S2_CASE(PG, CG, WP, WC, CKB, NBUF)                                                           \
    if (g.st == 2 && g.ks == 3 && g.pg == PG && g.cg == CG && g.wp == WP && g.wc == WC && g.ckb == CKB && g.nbuf == NBUF)  \
        return launch_dma<3, PG, CG, WP, WC, CKB, NBUF, 2>(p, st);
    ODAMD_DMA_CASE(3, 8, 2, 4, 1, 32, 2)
    ODAMD_DMA_CASE(1, 4, 4, 2, 2, 64, 3)
    ODAMD_DMA_S2_CASE(4, 2, 4, 1, 32, 2)
    ODAMD_DMA_S2_CASE(4, 4, 2, 2, 32, 2)
    ODAMD_DMA_S2_CASE(8, 4, 4, 2, 32, 2)
#ifdef LAB_DMA_S2_VARIANTS
    LAB_DMA_S2_VARIANTS
#endif
#ifdef LAB_DMA_VARIANTS
    LAB_DMA_VARIANTS
#endif
#undef ODAMD_DMA_CASE
#undef ODAMD_DMA_S2_CASE
    return ODAMD_E_ARG;
}

// ---- persistent 3x3 kernel for the wide, shallow layers (conv_s3.h) --------------------------------------
struct S3Config { int ksteps, cg, st, pg, npad_max, nbuf; };

#ifdef ODAMD_LAB
S3Config g_lab_s3 = {};
#endif
S3Config s3_config(const ConvP& p, int ks, int stride, int dtype) {
    S3Config none = {};
    if (dtype != ODAMD_DTYPE_BF16 || ks != 3 || p.pool || p.upz || p.nchw || p.accum || !p.scale || !p.shift) return none;
#ifdef ODAMD_LAB
    if (g_lab_no_s3) return none;
    if (g_lab_s3.pg && g_lab_s3.ksteps * 16 == p.Cin && g_lab_s3.cg * 32 == p.Cout && g_lab_s3.st == stride) return g_lab_s3;
#endif
    const long long npix = (long long)p.B * p.Ho * p.Wo;
    // measured at batch 64 (tools/micro/conv_lab.hip and the per-layer profile of the network): stride 1 is ahead of
    // conv_igemm_kernel on the 320 x 320 map (0.25 -> 0.21 ms) and behind it on the 160 x 160 slices of the ELAN
    // concat buffers (64 of every 256 bytes: both kernels drop to ~2.3 TB/s there, this one pays its ragged
    // tile columns on top); stride 2 is ahead from 160 x 160 outputs up
    if (npix < (stride == 1 ? 4 : 1) * 1024 * 1024) return none;
    if (npix * p.y_ld * 2 >= (1ll << 31)) return none;   // 32-bit offsets of the buffer stores
#ifdef ODAMD_LAB
    if (g_lab_s3.pg) return g_lab_s3;
#endif
    if (stride == 1 && p.Cin == 32 && p.Cout == 32) return {2, 1, 1, 8, 384, 2};
    if (stride == 2 && p.Cin == 32 && p.Cout == 64) return {2, 2, 2, 4, 640, 3};
    return none;
}

template <int KSTEPS, int CG, int ST, int PG, int NPAD_MAX, int NBUF, bool ACT>
int launch_s3(const ConvP& c, hipStream_t st) {
    S3P p = {};
    p.x = c.x; p.w = c.w; p.scale = c.scale; p.shift = c.shift; p.y = c.y;
    p.H = c.H; p.W = c.W; p.Ho = c.Ho; p.Wo = c.Wo; p.x_ld = c.x_ld; p.y_ld = c.y_ld; p.act = c.act;
    p.twl = p.Wo % 32 == 0 ? 5 : p.Wo % 16 == 0 ? 4 : p.Wo % 8 == 0 ? 3 : 5;
    p.TW = 1 << p.twl;
    p.TH = PG * 32 / p.TW;
    p.IH = ST * (p.TH - 1) + 3;
    p.IW = ST * (p.TW - 1) + 3;
    p.IWH = p.TW + 1;
    if (p.IH * (ST == 2 ? 2 * p.IWH : p.IW) > NPAD_MAX || p.IW > 255) return ODAMD_E_SHAPE;
    p.npad = NPAD_MAX;          // fixed piece count per tile (the kernel's main loop has no branches)
    p.tiles_x = (p.Wo + p.TW - 1) / p.TW;
    p.tiles_y = (p.Ho + p.TH - 1) / p.TH;
    p.segs_y = (p.tiles_y + S3_SEG - 1) / S3_SEG;
    const long long nitems = (long long)c.B * p.tiles_x * p.segs_y;
    if (nitems > (1ll << 28)) return ODAMD_E_SHAPE;
    p.nitems = (int)nitems;
    p.ybytes = (unsigned)(((size_t)c.B * p.Ho * p.Wo - 1) * p.y_ld * 2 + (size_t)c.Cout * 2);
    const size_t lds = (size_t)CG * 9 * KSTEPS * 1024 + NBUF * (size_t)(2 * KSTEPS) * p.npad * 16;
    auto k = conv_s3_kernel<KSTEPS, CG, ST, PG, NPAD_MAX, NBUF, ACT>;
    static int resident[64];                 // workgroups the device keeps resident at once, per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return (int)hipErrorInvalidDevice;
    if (!resident[dev]) {
        // the largest patch this instantiation can be launched with sizes the occupancy (and the attribute)
        const size_t lds_max = (size_t)CG * 9 * KSTEPS * 1024 + NBUF * (size_t)(2 * KSTEPS) * NPAD_MAX * 16;
        if (lds_max > 160 * 1024) return ODAMD_E_SHAPE;
        if (lds_max > 64 * 1024 &&
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max) != hipSuccess)
            return (int)hipGetLastError();
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 256, lds_max) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu < 1)
            return (int)hipErrorInvalidDevice;
        resident[dev] = per_cu * cus / 8 * 8;
        if (resident[dev] < 8) { resident[dev] = 0; return (int)hipErrorInvalidDevice; }
    }
    const int per_xcd = (p.nitems + 7) / 8;
    const int slots = std::min(resident[dev] / 8, per_xcd);
    p.rounds = (per_xcd + slots - 1) / slots;
    k<<<slots * 8, 256, lds, st>>>(p);
    return (int)hipGetLastError();
}

int dispatch_s3(const ConvP& p, const S3Config& g, hipStream_t st) {
#define ODAMD_S3_CASE(KSTEPS, CG, ST, PG, NPAD, NBUF)                                                          \
    if (g.ksteps == KSTEPS && g.cg == CG && g.st == ST && g.pg == PG && g.npad_max == NPAD && g.nbuf == NBUF)  \
        return p.act ? launch_s3<KSTEPS, CG, ST, PG, NPAD, NBUF, true>(p, st) : launch_s3<KSTEPS, CG, ST, PG, NPAD, NBUF, false>(p, st);
    ODAMD_S3_CASE(2, 1, 1, 8, 384, 2)
    ODAMD_S3_CASE(2, 2, 2, 4, 640, 3)
#ifdef LAB_S3_VARIANTS
    LAB_S3_VARIANTS
#endif
#undef ODAMD_S3_CASE
    return ODAMD_E_ARG;
}

// ---- persistent 1x1 kernel with register-resident weights (conv_wres.h) ------------------------------------
struct WresConfig { int kch, cgw, nbuf; };

#ifdef ODAMD_LAB
static int g_lab_no_wres = 0;
WresConfig g_lab_wres = {};
#endif
// Compute units of the current device rounded down to whole XCD shares of 8 (0 when there is no device); queried once
// per device, the cached value is the same whoever writes it first.
int device_cus8() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (!n) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) return 0;
        n = n / 8 * 8;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// one workgroup per compute unit, 8 walkers (one per XCD) per channel block at least: can the device hold them?
bool wres_fits(int cout, int cgw) { return device_cus8() / (cout / (128 * cgw)) / 8 * 8 >= 8; }

WresConfig wres_config(const ConvP& p, int ks, int stride, int dtype) {
    WresConfig none = {};
    if (dtype != ODAMD_DTYPE_BF16 || ks != 1 || stride != 1 || p.pool || p.upz || p.nchw || p.accum || !p.scale || !p.shift)
        return none;
#ifdef ODAMD_LAB
    if (g_lab_no_wres) return none;
#endif
    const long long npix = (long long)p.B * p.H * p.W;
    // whole 64-pixel tiles only (the kernel has no ragged tile); 32-bit byte offsets in the buffer stores
    if (npix < 1024 || npix % 64 || npix * p.y_ld * 2 >= (1ll << 31) || npix >= (1ll << 24)) return none;
    if (p.Cout % 128) return none;
    int nbuf = 3;
#ifdef ODAMD_LAB
    if (g_lab_wres.nbuf) nbuf = g_lab_wres.nbuf;
    if (g_lab_wres.cgw == 1 && (p.Cin == 256 || p.Cin == 512)) return {p.Cin / 256, 1, nbuf};
#endif
    // a device with too few compute units for the layer's channel blocks (a partition, or very many output channels)
    // gets no packed form here, so that the caller packs for a kernel that can be launched
    WresConfig g = none;
    if (p.Cin == 256) g = {1, p.Cout % 256 == 0 ? 2 : 1, nbuf};
    else if (p.Cin == 512) g = {2, p.Cout % 256 == 0 ? 2 : 1, nbuf};
    else if (p.Cin == 1024) g = {4, 1, nbuf};
    if (g.kch && !wres_fits(p.Cout, g.cgw)) return none;
    return g;
}

template <int KCH, int CGW, int NBUF, bool ACT>
int launch_wres(const ConvP& c, hipStream_t st) {
    WresP p = {};
    p.x = c.x; p.w = c.w; p.scale = c.scale; p.shift = c.shift; p.y = c.y;
    p.npix = c.B * c.H * c.W; p.x_ld = c.x_ld; p.y_ld = c.y_ld; p.Cout = c.Cout;
    p.ntiles = (p.npix + 63) / 64;
    p.ncb = c.Cout / (128 * CGW);
    p.ybytes = (unsigned)(((size_t)p.npix - 1) * p.y_ld * 2 + (size_t)c.Cout * 2);
    const size_t lds = (size_t)NBUF * WRES_UNIT + 2 * 128 * CGW * sizeof(float) + 4 * 32 * (CGW * 64 + 16);
    auto k = conv1x1_wres_kernel<KCH, CGW, NBUF, ACT>;
    if (lds > 160 * 1024) return ODAMD_E_SHAPE;
    if (odamd_ensure_lds((const void*)k, lds) != hipSuccess) return (int)hipGetLastError();
    const int cus = device_cus8();           // one workgroup per compute unit (the ring fills the LDS)
    if (!cus) return (int)hipErrorInvalidDevice;
    // walkers: a multiple of 8 (one share per XCD), at most one workgroup per compute unit, no more than tiles
    int nwalk = cus / p.ncb / 8 * 8;
    if (nwalk < 8) return ODAMD_E_SHAPE;     // wres_config does not offer the kernel then
    const int need = (p.ntiles + 7) / 8 * 8;
    if (nwalk > need) nwalk = need;
    p.nwalk = nwalk;
    p.rounds = ((p.ntiles + nwalk - 1) / nwalk + 1) & ~1;
    k<<<nwalk * p.ncb, 256, lds, st>>>(p);
    return (int)hipGetLastError();
}

int dispatch_wres(const ConvP& p, const WresConfig& g, hipStream_t st) {
#define ODAMD_WRES_CASE(KCH, CGW, NBUF)                                                            \
    if (g.kch == KCH && g.cgw == CGW && g.nbuf == NBUF)                                            \
        return p.act ? launch_wres<KCH, CGW, NBUF, true>(p, st) : launch_wres<KCH, CGW, NBUF, false>(p, st);
    ODAMD_WRES_CASE(1, 2, 3)
    ODAMD_WRES_CASE(1, 1, 3)
    ODAMD_WRES_CASE(2, 2, 3)
    ODAMD_WRES_CASE(2, 1, 3)
    ODAMD_WRES_CASE(4, 1, 3)
#ifdef LAB_WRES_VARIANTS
    LAB_WRES_VARIANTS
#endif
#undef ODAMD_WRES_CASE
    return ODAMD_E_ARG;
}

// ---- both 1x1 convolutions of an mp block in one launch (conv_mp.h) -------------------------------------------
// Does the fused form exist for this pair?  `a`: the pooled branch (in_pool2), `b`: the plain branch of the same input.
bool mp_pair_ok(const odamd_conv_desc* a, const odamd_conv_desc* b) {
    if (!a || !b) return false;
    if (a->dtype != ODAMD_DTYPE_BF16 || b->dtype != ODAMD_DTYPE_BF16) return false;
    if (a->ksize != 1 || b->ksize != 1 || a->stride != 1 || b->stride != 1) return false;
    if (!a->in_pool2 || b->in_pool2 || a->out_nchw_f32 || b->out_nchw_f32 || a->in_upzero2 || b->in_upzero2 || a->accumulate ||
        b->accumulate || a->w_packed || b->w_packed || a->act != 1 || b->act != 1)
        return false;
    if (a->x != b->x || a->batch != b->batch || a->h != b->h || a->w_in != b->w_in || a->cin != b->cin || a->x_ld != b->x_ld) return false;
    if (a->h % 8 || a->w_in % 16) return false;
    const int cb = a->cin * 2;
    if (cb != 64 && cb != 128 && cb != 256) return false;
    auto couts_ok = [](int c) { return c == 32 || c == 64 || c == 128; };
    if (!couts_ok(a->cout) || !couts_ok(b->cout)) return false;
    const long long npix = (long long)a->batch * a->h * a->w_in;
    if (npix * b->y_ld * 2 >= (1ll << 31) || (npix / 4) * a->y_ld * 2 >= (1ll << 31) || npix >= (1ll << 26)) return false;
    if (a->y_ld < a->cout || b->y_ld < b->cout || (a->y_ld * 2) % 16 || (b->y_ld * 2) % 16 || (a->x_ld * 2) % 16) return false;
    return true;
}

template <int CG2, int CG1, int NPF>
int launch_mp(const odamd_conv_desc* a, const odamd_conv_desc* b, hipStream_t st) {
    constexpr int WC = CG2 >= 2 ? 2 : 1, CGW = CG2 / WC;
    MpP p = {};
    p.x = (const char*)a->x;
    p.w2 = (const char*)b->w; p.scale2 = b->scale; p.shift2 = b->shift; p.y2 = (char*)b->y;
    p.w1 = (const char*)a->w; p.scale1 = a->scale; p.shift1 = a->shift; p.y1 = (char*)a->y;
    p.B = a->batch; p.H = a->h; p.W = a->w_in; p.Cin = a->cin; p.x_ld = a->x_ld;
    p.y2_ld = b->y_ld; p.y1_ld = a->y_ld; p.Cout2 = b->cout; p.Cout1 = a->cout;
    p.tiles_x = p.W / 16; p.tiles_y = p.H / 8; p.ntiles = p.B * p.tiles_x * p.tiles_y;
    const size_t npix = (size_t)p.B * p.H * p.W;
    p.y2bytes = (unsigned)((npix - 1) * p.y2_ld * 2 + (size_t)p.Cout2 * 2);
    p.y1bytes = (unsigned)((npix / 4 - 1) * p.y1_ld * 2 + (size_t)p.Cout1 * 2);
    const size_t rsb = (size_t)p.Cin * 2 + 16;
    size_t in_bytes = 128 * rsb;
    const size_t epi = (size_t)4 * 32 * (CGW * 128 + 16);
    if (in_bytes < epi) in_bytes = epi;
    p.pool_off = (int)((in_bytes + 255) & ~(size_t)255);
    p.w2_off = p.pool_off + (int)((32 * rsb + 255) & ~(size_t)255);
    p.w1_off = p.w2_off + (int)(32 * CG2 * rsb);
    const size_t lds = p.w1_off + 32 * CG1 * rsb;
    auto k = conv_mp_kernel<CG2, CG1, NPF>;
    if (odamd_ensure_lds((const void*)k, lds) != hipSuccess) return (int)hipGetLastError();
    int grid = lds <= 80 * 1024 ? 512 : 256;
    if (grid > p.ntiles) grid = p.ntiles;
    k<<<grid, 256, lds, st>>>(p);
    return (int)hipGetLastError();
}

int dispatch_mp(const odamd_conv_desc* a, const odamd_conv_desc* b, hipStream_t st) {
    const int cg2 = b->cout / 32, cg1 = a->cout / 32, npf = a->cin * 2 / 32;
#define ODAMD_MP_CASE(CG2, CG1, NPF) if (cg2 == CG2 && cg1 == CG1 && npf == NPF) return launch_mp<CG2, CG1, NPF>(a, b, st);
#define ODAMD_MP_CASES(CG2, CG1) ODAMD_MP_CASE(CG2, CG1, 2) ODAMD_MP_CASE(CG2, CG1, 4) ODAMD_MP_CASE(CG2, CG1, 8)
    ODAMD_MP_CASES(1, 1) ODAMD_MP_CASES(2, 2) ODAMD_MP_CASES(4, 4)
    ODAMD_MP_CASES(1, 2) ODAMD_MP_CASES(2, 1) ODAMD_MP_CASES(2, 4) ODAMD_MP_CASES(4, 2) ODAMD_MP_CASES(1, 4) ODAMD_MP_CASES(4, 1)
#undef ODAMD_MP_CASES
#undef ODAMD_MP_CASE
    return ODAMD_E_SHAPE;
}

// ---- the same pair for 256 / 512 input channels: input channels in chunks, weights streamed (conv_mpk.h) -----------
bool mpk_pair_ok(const odamd_conv_desc* a, const odamd_conv_desc* b) {
    if (!a || !b) return false;
    if (a->dtype != ODAMD_DTYPE_BF16 || b->dtype != ODAMD_DTYPE_BF16) return false;
    if (a->ksize != 1 || b->ksize != 1 || a->stride != 1 || b->stride != 1) return false;
    if (!a->in_pool2 || b->in_pool2 || a->out_nchw_f32 || b->out_nchw_f32 || a->in_upzero2 || b->in_upzero2 || a->accumulate ||
        b->accumulate || a->w_packed || b->w_packed || a->act != 1 || b->act != 1 || !a->scale || !a->shift || !b->scale || !b->shift)
        return false;
    if (a->x != b->x || a->batch != b->batch || a->h != b->h || a->w_in != b->w_in || a->cin != b->cin || a->x_ld != b->x_ld) return false;
    if (a->h % 8 || a->w_in % 8) return false;
    if ((a->cin != 256 && a->cin != 128) || b->cout != 128 || a->cout != 128) return false;      // where the fused form measured ahead (conv_mpk.h)
    const long long npix = (long long)a->batch * a->h * a->w_in;
    if (npix * b->y_ld * 2 >= (1ll << 31) || (npix / 4) * a->y_ld * 2 >= (1ll << 31) || npix >= (1ll << 26)) return false;
    if (a->y_ld < a->cout || b->y_ld < b->cout || (a->y_ld * 2) % 16 || (b->y_ld * 2) % 16 || (a->x_ld * 2) % 16) return false;
    return true;
}

template <int CGW, int C1W, int CK>
int launch_mpk(const odamd_conv_desc* a, const odamd_conv_desc* b, hipStream_t st) {
    MpkP p = {};
    p.x = (const char*)a->x;
    p.w2 = (const char*)b->w; p.scale2 = b->scale; p.shift2 = b->shift; p.y2 = (char*)b->y;
    p.w1 = (const char*)a->w; p.scale1 = a->scale; p.shift1 = a->shift; p.y1 = (char*)a->y;
    p.B = a->batch; p.H = a->h; p.W = a->w_in; p.cbytes = a->cin * 2; p.x_ld = a->x_ld;
    p.y2_ld = b->y_ld; p.y1_ld = a->y_ld;
    p.patches_x = p.W / 8; p.patches_y = p.H / 8;
    p.npatches = p.B * p.patches_x * p.patches_y;
    p.npairs = (p.npatches + 1) / 2;
    const size_t npix = (size_t)p.B * p.H * p.W;
    p.y2bytes = (unsigned)((npix - 1) * p.y2_ld * 2 + (size_t)b->cout * 2);
    p.y1bytes = (unsigned)((npix / 4 - 1) * p.y1_ld * 2 + (size_t)a->cout * 2);
    const size_t staging = (size_t)(128 + 32 + 64 * CGW + 128 * C1W) * (CK + 16);
    const size_t slices = (size_t)4 * 32 * ((CGW > C1W ? CGW : C1W) * 128 + 16);
    const size_t lds = staging > slices ? staging : slices;
    auto k = conv_mpk_kernel<CGW, C1W, CK>;
    if (odamd_ensure_lds((const void*)k, lds) != hipSuccess) return (int)hipGetLastError();
    int grid = lds <= 80 * 1024 ? 512 : 256;
    if (grid > p.npairs) grid = p.npairs;
    k<<<grid, 256, lds, st>>>(p);
    return (int)hipGetLastError();
}

int dispatch_mpk(const odamd_conv_desc* a, const odamd_conv_desc* b, hipStream_t st) { return launch_mpk<2, 1, 128>(a, b, st); }

// ---- two chained 3x3 convolutions with 32 channels between and after them in one launch (conv_pair.h) ----------
// `a`: the first convolution, `b`: the second, which reads exactly what `a` writes.
bool chain_pair_ok(const odamd_conv_desc* a, const odamd_conv_desc* b) {
    if (!a || !b) return false;
    if (a->dtype != ODAMD_DTYPE_BF16 || b->dtype != ODAMD_DTYPE_BF16) return false;
    if (a->ksize != 3 || b->ksize != 3 || a->stride != 1 || b->stride != 1) return false;
    if (a->in_pool2 || b->in_pool2 || a->out_nchw_f32 || b->out_nchw_f32 || a->in_upzero2 || b->in_upzero2 || a->accumulate ||
        b->accumulate || !a->w_packed || !b->w_packed || !a->scale || !a->shift || !b->scale || !b->shift)
        return false;
    if (b->x != a->y || b->x_ld != a->y_ld || a->batch != b->batch || a->h != b->h || a->w_in != b->w_in) return false;
    if ((a->cin != 32 && a->cin != 64) || a->cout != 32 || b->cin != 32 || b->cout != 32) return false;
    if (a->x_ld % 8 || a->y_ld % 8 || b->y_ld % 8 || a->x_ld < a->cin || a->y_ld < 32 || b->y_ld < 32) return false;
    if (((uintptr_t)a->x & 15) || ((uintptr_t)a->y & 15) || ((uintptr_t)b->y & 15) || ((uintptr_t)a->w & 15) || ((uintptr_t)b->w & 15))
        return false;
    const long long npix = (long long)a->batch * a->h * a->w_in;
    if (npix * a->y_ld * 2 >= (1ll << 31) || npix * b->y_ld * 2 >= (1ll << 31)) return false;
    if ((long long)a->h * a->w_in >= (1ll << 24)) return false;
    return true;
}

int launch_chain_pair(const odamd_conv_desc* a, const odamd_conv_desc* b, hipStream_t st) {
    PairP p = {};
    p.x = (const char*)a->x; p.w1 = (const char*)a->w; p.w2 = (const char*)b->w;
    p.sc1 = a->scale; p.sh1 = a->shift; p.sc2 = b->scale; p.sh2 = b->shift;
    p.y1 = (char*)a->y; p.y2 = (char*)b->y;
    p.B = a->batch; p.H = a->h; p.W = a->w_in; p.x_ld = a->x_ld; p.y1_ld = a->y_ld; p.y2_ld = b->y_ld;
    p.act1 = a->act; p.act2 = b->act;
    p.tiles_x = (p.W + 15) / 16; p.tiles_y = (p.H + 15) / 16;
    // vertical runs of tiles per workgroup (the weights are copied once per run), short enough to leave ~1000 workgroups
    int seg = p.tiles_y;
    while (seg > 1 && (long long)p.B * p.tiles_x * ((p.tiles_y + seg - 1) / seg) < 1024) seg = (seg + 1) / 2;
    p.seg = seg;
    p.runs_y = (p.tiles_y + seg - 1) / seg;
    const long long nblocks = (long long)p.B * p.tiles_x * p.runs_y;
    if (nblocks >= (1ll << 30)) return ODAMD_E_SHAPE;
    p.nblocks = (int)nblocks;
    const size_t npix = (size_t)p.B * p.H * p.W;
    p.y1bytes = (unsigned)((npix - 1) * p.y1_ld * 2 + 64);
    p.y2bytes = (unsigned)((npix - 1) * p.y2_ld * 2 + 64);
    const int ks1 = a->cin / 16;
    const size_t lds = (size_t)9 * ks1 * 1024 + 18 * 1024 + (size_t)PR_IN_BLOCKS * 2 * ks1 * 256 + 512;
    if (ks1 == 2) {
        auto k = conv_pair_kernel<2>;
        if (odamd_ensure_lds((const void*)k, lds) != hipSuccess) return (int)hipGetLastError();
        k<<<p.nblocks, 256, lds, st>>>(p);
    } else {
        auto k = conv_pair_kernel<4>;
        if (odamd_ensure_lds((const void*)k, lds) != hipSuccess) return (int)hipGetLastError();
        k<<<p.nblocks, 256, lds, st>>>(p);
    }
    return (int)hipGetLastError();
}

template <int DT>
int dispatch_conv(const ConvP& p, int ks, int stride, hipStream_t st) {
    const int cin_bytes = p.Cin * Elem<DT>::SIZE;
    // large plain 1x1 layers: persistent streaming kernel
    if (ks == 1 && stride == 1 && !p.pool && !p.upz && !p.nchw && cin_bytes <= 256 &&
        (long long)p.B * p.H * p.W >= 256 * 128 && (long long)p.B * p.H * p.W < (1ll << 30) / (p.x_ld * Elem<DT>::SIZE) * 128) {
        if (p.Cout <= 32) return launch_stream<DT, 1, 4, 1>(p, st);
        if (p.Cout <= 64) return launch_stream<DT, 2, 2, 2>(p, st);
        return launch_stream<DT, 4, 2, 2>(p, st);
    }
    if (ks == 1 && stride == 1) {
        if (p.pool) {
            if (cin_bytes % 64 == 0) return dispatch_cg<DT, 1, 1, 64, true>(p, st);
            return dispatch_cg<DT, 1, 1, 32, true>(p, st);
        }
        if (cin_bytes % 128 == 0) return dispatch_cg<DT, 1, 1, 128, false>(p, st);
        if (cin_bytes % 64 == 0) return dispatch_cg<DT, 1, 1, 64, false>(p, st);
        return dispatch_cg<DT, 1, 1, 32, false>(p, st);
    }
    if (ks == 3 && stride == 1) {
        if (DT == ODAMD_DTYPE_BF16 && cin_bytes % 64 == 0 && p.Cout <= 64 && !p.upz) {
            if (p.Cout <= 32) return launch_conv<DT, 3, 1, 1, 8, 4, 1, 64, false>(p, st);
            return launch_conv<DT, 3, 1, 2, 8, 4, 1, 64, false>(p, st);
        }
        return dispatch_cg<DT, 3, 1, 32, false>(p, st);
    }
    if (ks == 3 && stride == 2) return dispatch_cg<DT, 3, 2, 32, false>(p, st);
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

// dgrad_s2_direct_kernel serves this launch: the data gradient of a stride-2 3x3 convolution (zero-upsampled input)
// in bf16 with dY of 32 or 64 channels and dX of at most 32 (whole 16-byte pieces), plain NHWC, no affine, no activation
bool dgrad_s2_direct_ok(const odamd_conv_desc* d, const ConvP& p) {
    if (d->dtype != ODAMD_DTYPE_BF16 || !d->in_upzero2 || d->ksize != 3 || d->stride != 1 || d->in_pool2 || d->out_nchw_f32) return false;
    if (d->scale || d->shift || d->act || d->w_packed) return false;
    if ((d->cin != 32 && d->cin != 64) || d->cout > 32 || d->cout % 8 || d->x_ld % 8 || d->y_ld % 8) return false;
    if (((uintptr_t)d->y & 15)) return false;
    const size_t npix = (size_t)p.B * p.Ho * p.Wo;
    return fast_epilogue_ok(npix, p.y_ld, 2, p.Cout, 8, p.accum) && npix < ((size_t)1 << 30);
}

int launch_dgrad_s2_direct(const odamd_conv_desc* d, const ConvP& c, hipStream_t st) {
    DgS2P p = {};
    p.dy = c.x; p.w = c.w; p.dx = c.y;
    p.B = c.B; p.Hr = c.Hr; p.Wr = c.Wr; p.dy_ld = c.x_ld; p.dx_ld = c.y_ld; p.Cout = c.Cout; p.accum = c.accum;
    p.blocks_x = (p.Wr + 7) / 8; p.blocks_y = (p.Hr + 3) / 4;
    p.nblocks = p.B * p.blocks_x * p.blocks_y;
    const size_t npix = (size_t)c.B * c.Ho * c.Wo;
    p.ybytes = (unsigned)((npix - 1) * c.y_ld * 2 + (size_t)c.Cout * 2);
    const int ksteps = d->cin / 16;
    const size_t wrs = (size_t)9 * d->cin * 2 + 16;
    const size_t lds = ((32 * wrs + 255) & ~(size_t)255) + 4 * 32 * (128 + 16);
    int grid = (p.nblocks + 3) / 4;
    if (grid > 1024) grid = 1024;
    if (ksteps == 2) dgrad_s2_direct_kernel<2><<<grid, 256, lds, st>>>(p);
    else dgrad_s2_direct_kernel<4><<<grid, 256, lds, st>>>(p);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

// Validates a descriptor and turns it into kernel parameters.
static int fill_conv_params(const odamd_conv_desc* d, ConvP* out, bool need_pointers) {
    if (!d) return ODAMD_E_ARG;
    if (need_pointers && (!d->x || !d->w || !d->y)) return ODAMD_E_ARG;
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
    p.H = d->in_pool2 ? d->h / 2 : (d->in_upzero2 ? d->h * 2 : d->h);
    p.W = d->in_pool2 ? d->w_in / 2 : (d->in_upzero2 ? d->w_in * 2 : d->w_in);
    p.upz = d->in_upzero2; p.accum = d->accumulate;
    if (d->in_upzero2 && (d->in_pool2 || d->stride != 1)) return ODAMD_E_ARG;
    if (d->accumulate && d->out_nchw_f32) return ODAMD_E_ARG;
    p.Cin = d->cin; p.Cout = d->cout; p.x_ld = d->x_ld; p.y_ld = d->y_ld;
    const int pad = (d->ksize - 1) / 2;
    p.Ho = (p.H + 2 * pad - d->ksize) / d->stride + 1;
    p.Wo = (p.W + 2 * pad - d->ksize) / d->stride + 1;
    p.act = d->act; p.pool = d->in_pool2; p.nchw = d->out_nchw_f32;
    if (!p.nchw) {
        if (d->y_ld < d->cout) return ODAMD_E_SHAPE;
        if (((uintptr_t)d->y & (4 * es - 1)) || (d->y_ld * es) % (4 * es)) return ODAMD_E_ALIGN;
    }
    *out = p;
    return ODAMD_OK;
}

int odamd_conv_weight_layout(const odamd_conv_desc* d, int* cout_groups, int* chunk_bytes) {
    ConvP p;
    if (!cout_groups || !chunk_bytes || fill_conv_params(d, &p, false) != ODAMD_OK) return 0;
    const S3Config s3 = s3_config(p, d->ksize, d->stride, d->dtype);
    if (s3.pg) {                // the whole weight tensor as one block of fragments
        *cout_groups = s3.cg;
        *chunk_bytes = p.Cin * 2;
        return 1;
    }
    const WresConfig wr = wres_config(p, d->ksize, d->stride, d->dtype);
    if (wr.kch) {               // the whole reduction of a block of 128 * cgw output channels as one block of fragments
        *cout_groups = 4 * wr.cgw;
        *chunk_bytes = p.Cin * 2;
        return 1;
    }
    const DmaConfig g = dma_config(p, d->ksize, d->stride, d->dtype);
    if (!g.pg) return 0;
    *cout_groups = g.cg;
    *chunk_bytes = g.ckb;
    return 1;
}

int odamd_conv_forward(const odamd_conv_desc* d, void* stream) {
    ConvP p;
    const int rc = fill_conv_params(d, &p, true);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (d->w_packed) {
        const S3Config s3 = s3_config(p, d->ksize, d->stride, d->dtype);
        if (s3.pg) return dispatch_s3(p, s3, st);
        const WresConfig wr = wres_config(p, d->ksize, d->stride, d->dtype);
        if (wr.kch) return dispatch_wres(p, wr, st);
        const DmaConfig g = dma_config(p, d->ksize, d->stride, d->dtype);
        if (!g.pg) return ODAMD_E_ARG;        // this layer has no packed-weight form (odamd_conv_weight_layout returned 0)
        return dispatch_dma(p, g, st);
    }
    if (dgrad_s2_direct_ok(d, p)) return launch_dgrad_s2_direct(d, p, st);
    return d->dtype == ODAMD_DTYPE_F32 ? dispatch_conv<ODAMD_DTYPE_F32>(p, d->ksize, d->stride, st)
                                       : dispatch_conv<ODAMD_DTYPE_BF16>(p, d->ksize, d->stride, st);
}

int odamd_conv_mp_supported(const odamd_conv_desc* pooled, const odamd_conv_desc* plain) {
    ConvP pa, pb;
    if (fill_conv_params(pooled, &pa, false) != ODAMD_OK || fill_conv_params(plain, &pb, false) != ODAMD_OK) return 0;
    return (mp_pair_ok(pooled, plain) || mpk_pair_ok(pooled, plain)) ? 1 : 0;
}

int odamd_conv_mp_forward(const odamd_conv_desc* pooled, const odamd_conv_desc* plain, void* stream) {
    ConvP pa, pb;
    int rc = fill_conv_params(pooled, &pa, true);
    if (rc) return rc;
    if ((rc = fill_conv_params(plain, &pb, true))) return rc;
    const bool wide = !mp_pair_ok(pooled, plain);
    if (wide && !mpk_pair_ok(pooled, plain)) return ODAMD_E_SHAPE;
    if (((uintptr_t)pooled->y & 15) || ((uintptr_t)plain->y & 15)) return ODAMD_E_ALIGN;
    return wide ? dispatch_mpk(pooled, plain, (hipStream_t)stream) : dispatch_mp(pooled, plain, (hipStream_t)stream);
}

int odamd_conv_chain_pair_supported(const odamd_conv_desc* first, const odamd_conv_desc* second) {
    ConvP pa, pb;
    if (fill_conv_params(first, &pa, false) != ODAMD_OK || fill_conv_params(second, &pb, false) != ODAMD_OK) return 0;
    return chain_pair_ok(first, second) ? 1 : 0;
}

int odamd_conv_chain_pair_forward(const odamd_conv_desc* first, const odamd_conv_desc* second, void* stream) {
    ConvP pa, pb;
    int rc = fill_conv_params(first, &pa, true);
    if (rc) return rc;
    if ((rc = fill_conv_params(second, &pb, true))) return rc;
    if (!chain_pair_ok(first, second)) return ODAMD_E_SHAPE;
    return launch_chain_pair(first, second, (hipStream_t)stream);
}

int odamd_conv_first_forward(const float* image, const float* w, const float* scale, const float* shift, void* y,
                             int batch, int h, int w_in, int cout, int y_ld, int act, int dtype, void* stream) {
    if (!image || !w || !y || batch < 1 || h < 1 || w_in < 1 || cout < 1 || (!scale != !shift)) return ODAMD_E_ARG;
    if (dtype != ODAMD_DTYPE_F32 && dtype != ODAMD_DTYPE_BF16) return ODAMD_E_ARG;
    const int es = dtype == ODAMD_DTYPE_F32 ? 4 : 2;
    if (y_ld < cout) return ODAMD_E_SHAPE;
    if (((uintptr_t)y & 15) || (y_ld * es) % 16) return ODAMD_E_ALIGN;
    if (cout > 1024) return ODAMD_E_SHAPE;
    dim3 grid((w_in + 63) / 64, (h + 15) / 16, batch);
    const size_t lds = ((size_t)3 * 18 * 67 + (size_t)((cout + 15) & ~15) * 32) * sizeof(float);
    if (dtype == ODAMD_DTYPE_F32)   // float32 keeps the scalar fma chain (bit-stable against the float32 parity fixtures)
        conv_first_valu_kernel<ODAMD_DTYPE_F32><<<dim3((w_in + 15) / 16, (h + 15) / 16, batch), 256, 0, (hipStream_t)stream>>>(
            image, w, scale, shift, (char*)y, h, w_in, cout, y_ld, act);
    else if (cout == 64)
        conv_first_wide_kernel<4><<<grid, 256, lds, (hipStream_t)stream>>>(image, w, scale, shift, (char*)y, h, w_in, y_ld, act);
    else if (cout == 32)
        conv_first_wide_kernel<2><<<grid, 256, lds, (hipStream_t)stream>>>(image, w, scale, shift, (char*)y, h, w_in, y_ld, act);
    else
        conv_first_kernel<ODAMD_DTYPE_BF16><<<grid, 256, lds, (hipStream_t)stream>>>(image, w, scale, shift, (char*)y, h, w_in, cout, y_ld, act);
    return (int)hipGetLastError();
}

// Workgroups of conv_first2_kernel the current device keeps resident at once (0 on error); queried once per device.
static int first2_resident_workgroups() {
    static int cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!cached[dev]) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_first2_kernel, 64 * first2::TH, first2::LDS_BYTES) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return 0;
        cached[dev] = std::max(per_cu, 1) * cus / 8 * 8;
    }
    return cached[dev];
}

int odamd_conv_first2_forward(const float* image, const float* w0, const float* scale0, const float* shift0,
                              const void* w1, const float* scale1, const float* shift1, void* y, int batch, int h,
                              int w_in, int y_ld, void* stream) {
    if (!image || !w0 || !scale0 || !shift0 || !w1 || !scale1 || !shift1 || !y || batch < 1 || h < 2 || w_in < 2)
        return ODAMD_E_ARG;
    if (y_ld < 32) return ODAMD_E_SHAPE;
    if (((uintptr_t)y & 15) || ((uintptr_t)w1 & 15) || (y_ld * 2) % 16) return ODAMD_E_ALIGN;
    const int ho = (h - 1) / 2 + 1, wo = (w_in - 1) / 2 + 1;
    static_assert(first2::LDS_BYTES <= 64 * 1024, "fits the default LDS limit of a workgroup");
    const size_t npix = (size_t)batch * ho * wo;
    const unsigned ybytes = fast_epilogue_ok(npix, y_ld, 2, 32, 32, 0) ? (unsigned)((npix - 1) * y_ld * 2 + 64) : 0u;
    // persistent workgroups: as many as the device holds at once (a multiple of 8, one share per XCD)
    const int tiles_x = (wo + first2::TW - 1) / first2::TW, tiles_y = (ho + first2::TH - 1) / first2::TH;
    const int segs_y = (tiles_y + first2::SEG - 1) / first2::SEG;
    const long long nitems_ll = (long long)batch * tiles_x * segs_y;
    if (nitems_ll > (1ll << 30)) return ODAMD_E_SHAPE;
    const int nitems = (int)nitems_ll, per_xcd = (nitems + 7) / 8;
    int resident = first2_resident_workgroups();
    if (resident < 8) return (int)hipErrorInvalidDevice;
    const int slots = std::min(resident / 8, per_xcd);
    const int rounds = (per_xcd + slots - 1) / slots;
    conv_first2_kernel<<<slots * 8, 64 * first2::TH, first2::LDS_BYTES, (hipStream_t)stream>>>(
        image, w0, scale0, shift0, (const char*)w1, scale1, shift1, (char*)y, h, w_in, ho, wo, y_ld, ybytes, nitems, tiles_x,
        tiles_y, segs_y, rounds);
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
    if ((size_t)h * w * 32 * 4 <= 64 * 1024 && vecs >= 4) {   // four vectors' planes (twice) fit LDS: 64-byte runs per pixel
        spp_cascade_kernel<4><<<dim3((vecs + 3) / 4, batch), 256, (size_t)h * w * 32 * 4, (hipStream_t)stream>>>(
            (const char*)x, (char*)y5, (char*)y9, (char*)y13, h, w, vecs, x_ld, y_ld, es, dtype);
        return (int)hipGetLastError();
    }
    if ((size_t)h * w * 32 <= 64 * 1024) {      // the plane (twice) fits LDS: cascade of 5x5 pools
        spp_cascade_kernel<1><<<dim3(vecs, batch), 256, (size_t)h * w * 32, (hipStream_t)stream>>>(
            (const char*)x, (char*)y5, (char*)y9, (char*)y13, h, w, vecs, x_ld, y_ld, es, dtype);
        return (int)hipGetLastError();
    }
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
    if (dtype != ODAMD_DTYPE_F32 && dtype != ODAMD_DTYPE_BF16) return ODAMD_E_ARG;
    // few planes into whole eight-channel pixels (the image): y has to start at a pixel, which a channel offset inside
    // an eight-channel row never does at this alignment
    if (c <= 8 && y_ld == 8 && ((uintptr_t)y & (dtype == ODAMD_DTYPE_F32 ? 31 : 15)) == 0) {
        const size_t npix = (size_t)batch * h * w;
        const unsigned blocks = (unsigned)((npix + 255) / 256 < 8192 ? (npix + 255) / 256 : 8192);
        if (dtype == ODAMD_DTYPE_F32)
            few_planes_to_nhwc8_kernel<ODAMD_DTYPE_F32><<<blocks, 256, 0, (hipStream_t)stream>>>(x, (char*)y, c, (size_t)h * w, x_c, npix);
        else
            few_planes_to_nhwc8_kernel<ODAMD_DTYPE_BF16><<<blocks, 256, 0, (hipStream_t)stream>>>(x, (char*)y, c, (size_t)h * w, x_c, npix);
        return (int)hipGetLastError();
    }
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

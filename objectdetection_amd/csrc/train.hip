// Training-step kernels for gfx950: train-mode BatchNorm + SiLU (forward and backward), pooling /
// upsample backward, convolution weight gradient on the matrix cores, detection loss
// (forward + gradient), fused Adam + EMA, weight packing.
//
// Layout conventions are those of conv.hip: NHWC activations with pixel stride `ld`, 16-byte
// channel vectors.  All reductions over pixels are done as per-workgroup partials in double
// followed by a tiny finalize kernel, so statistics and their gradients are bitwise reproducible
// run to run; only the weight gradient uses float atomics (one add per workgroup and element).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "odamd.h"

namespace {

constexpr int RED_MAX_BLOCKS = 1024;

__device__ __forceinline__ float silu_f(float z) { return z / (1.0f + expf(-z)); }
// d silu / dz = s (1 + z (1 - s)),  s = sigmoid(z)
__device__ __forceinline__ float silu_grad(float z) {
    const float s = 1.0f / (1.0f + expf(-z));
    return s * (1.0f + z * (1.0f - s));
}
// bf16 kernels round their results to 8 mantissa bits: hardware exp / reciprocal are plenty
template <int DT> __device__ __forceinline__ float sigmoid_t(float z) {
    if constexpr (DT == ODAMD_DTYPE_F32) return 1.0f / (1.0f + expf(-z));
    else return sigmoid_fast(z);
}
template <int DT> __device__ __forceinline__ float silu_t(float z) {
    if constexpr (DT == ODAMD_DTYPE_F32) return silu_f(z);
    else return z * sigmoid_t<DT>(z);
}
template <int DT> __device__ __forceinline__ float silu_grad_t(float z) {
    const float s = sigmoid_t<DT>(z);
    return s * (1.0f + z * (1.0f - s));
}

template <int DT> struct Vec;
template <> struct Vec<ODAMD_DTYPE_F32> {
    static constexpr int N = 4, ES = 4;
    static __device__ __forceinline__ void load(const char* p, float* v) {
        const float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(char* p, const float* v) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Vec<ODAMD_DTYPE_BF16> {
    static constexpr int N = 8, ES = 2;
    static __device__ __forceinline__ void load(const char* p, float* v) {
        const uint4 t = *(const uint4*)p;
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u); }
    }
    static __device__ __forceinline__ void store(char* p, const float* v) {
        uint4 t;
        t.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
        t.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
        t.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
        t.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
        *(uint4*)p = t;
    }
};

// ------------------------------------------------------------------------------------------
// Per-channel reductions over pixels.  Workgroup = vecs x rows threads: thread (v, r) owns the
// channels of 16-byte vector v and walks pixels r, r+rows, ... of the workgroup's pixel range.
// MODE 0: sum(y), sum(y^2)                                       (BN statistics)
// MODE 1: sum(dz), sum(dz * xhat), dz = dact * silu'(z)          (BN + SiLU backward)
// MODE 2: sum(y)                                                 (bias gradient)
// Partials land in part[block][2][C] as doubles.
// ------------------------------------------------------------------------------------------
template <int DT, int MODE>
__global__ void __launch_bounds__(256)
channel_reduce_kernel(const char* __restrict__ y, int y_ld, const char* __restrict__ dact, int d_ld, size_t npix, int Creal,
                      int vecs, int rows, const float* __restrict__ scale, const float* __restrict__ shift,
                      const float* __restrict__ mean, const float* __restrict__ invstd, double* __restrict__ part) {
    // The tensors hold whole 16-byte channel vectors; Creal channels are real, the rest of the last vector is
    // zero padding (irregular, pruned widths).  The partial table is laid out for the padded count C.
    const int C = vecs * Vec<DT>::N;
    constexpr int N = Vec<DT>::N, ES = Vec<DT>::ES;
    extern __shared__ __attribute__((aligned(16))) float red[];   // [rows][vecs][2][N]
    const int v = threadIdx.x % vecs, r = threadIdx.x / vecs;
    const size_t per = (npix + gridDim.x - 1) / gridDim.x;
    const size_t p0 = (size_t)blockIdx.x * per, p1 = min(npix, p0 + per);
    float s0[N], s1[N], sc[N], sh[N], mu[N], is[N], piv[N];
#pragma unroll
    for (int e = 0; e < N; ++e) {
        s0[e] = 0.f; s1[e] = 0.f;
        const int c = v * N + e;
        if (MODE == 1) {
            const bool real = c < Creal;
            sc[e] = real ? scale[c] : 0.f; sh[e] = real ? shift[c] : 0.f; mu[e] = real ? mean[c] : 0.f; is[e] = real ? invstd[c] : 0.f;
        }
    }
    // MODE 0 sums (y - pivot) and (y - pivot)^2 with pivot = the channel's value at pixel 0: the
    // variance then comes from numbers of the size of the spread, not of the mean (no cancellation)
    if (MODE == 0) {
        Vec<DT>::load(y + v * 16, piv);
        if (blockIdx.x == 0 && r == 0)
#pragma unroll
            for (int e = 0; e < N; ++e) part[(size_t)gridDim.x * 2 * C + v * N + e] = (double)piv[e];
    }
    // four pixels per trip: their loads are independent, so each thread keeps 4 (MODE 1: 8) 16-byte
    // loads in flight
    for (size_t pb = p0 + r; pb < p1; pb += (size_t)4 * rows) {
        float a[4][N], g[MODE == 1 ? 4 : 1][N];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t p = pb + (size_t)u * rows;
            ok[u] = p < p1;
            if (ok[u]) {
                Vec<DT>::load(y + (p * y_ld) * ES + v * 16, a[u]);
                if (MODE == 1) Vec<DT>::load(dact + (p * d_ld) * ES + v * 16, g[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!ok[u]) continue;
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < N; ++e) { const float d = a[u][e] - piv[e]; s0[e] += d; s1[e] += d * d; }
            } else if (MODE == 1) {
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    const float dz = g[u][e] * silu_grad_t<DT>((a[u][e] - mu[e]) * sc[e] + sh[e]);
                    s0[e] += dz;
                    s1[e] += dz * ((a[u][e] - mu[e]) * is[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < N; ++e) s0[e] += a[u][e];
            }
        }
    }
    float* mine = red + ((size_t)(r * vecs + v) * 2) * N;
#pragma unroll
    for (int e = 0; e < N; ++e) { mine[e] = s0[e]; mine[N + e] = s1[e]; }
    __syncthreads();
    // Fold the rows in double precision with all threads: output o = (v, sum index, e) of the 2*C per
    // workgroup is the column o of the [rows][2*C] float table; when 2*C < 256 the rows are cut into
    // `slices` runs that are added in a fixed order afterwards (a few threads walking all rows alone
    // took longer than streaming the activations for narrow layers).
    const int nout = vecs * 2 * N;
    double* dred = (double*)(red + (size_t)rows * nout);          // [slices][nout], behind the float table
    const int nthr = blockDim.x;                                   // vecs * rows, at most 256
    const int slices = nout >= nthr ? 1 : nthr / nout;
    const int rows_per = (rows + slices - 1) / slices;
    for (int t = threadIdx.x; t < nout * slices; t += nthr) {
        const int o = t % nout, sl = t / nout;
        double acc = 0.0;
        const int r1 = min(rows, (sl + 1) * rows_per);
        for (int rr = sl * rows_per; rr < r1; ++rr) acc += red[(size_t)rr * nout + o];
        if (slices == 1) {
            const int vv = o / (2 * N), which = (o / N) & 1, e = o % N;
            part[((size_t)blockIdx.x * 2 + which) * C + vv * N + e] = acc;
        } else {
            dred[sl * nout + o] = acc;
        }
    }
    if (slices > 1) {
        __syncthreads();
        for (int o = threadIdx.x; o < nout; o += nthr) {
            double acc = 0.0;
            for (int sl = 0; sl < slices; ++sl) acc += dred[sl * nout + o];
            const int vv = o / (2 * N), which = (o / N) & 1, e = o % N;
            part[((size_t)blockIdx.x * 2 + which) * C + vv * N + e] = acc;
        }
    }
}

// BN statistics finalize (model/layer.py:9,14 in train mode): batch mean / biased variance, running
// buffers updated with momentum and the unbiased variance, affine folded to (scale, shift).
// One wave per channel: the lanes stride over the partial blocks and a butterfly of double adds
// folds them (fixed order: the result does not depend on scheduling).
__device__ __forceinline__ void fold_partials(const double* __restrict__ part, int nblk, int C /* padded: table stride */, int c,
                                              double& s, double& q, bool second) {
    const int lane = threadIdx.x & 63;
    s = 0.0; q = 0.0;
    for (int b = lane; b < nblk; b += 64) {
        s += part[((size_t)b * 2) * C + c];
        if (second) q += part[((size_t)b * 2 + 1) * C + c];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s += __shfl_xor(s, m);
        if (second) q += __shfl_xor(q, m);
    }
}

__global__ void __launch_bounds__(256)
bn_stats_finalize_kernel(const double* __restrict__ part, int nblk, int C, int Cp, double npix, const float* gamma,
                         const float* beta, float eps, float momentum, float* running_mean,
                         float* running_var, float* mean, float* invstd, float* scale, float* shift) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double s, q;
    fold_partials(part, nblk, Cp, c, s, q, true);
    if ((threadIdx.x & 63) != 0) return;
    const double md = s / npix;                    // mean of (y - pivot)
    double var = q / npix - md * md;
    if (var < 0.0) var = 0.0;
    const double m = part[(size_t)nblk * 2 * Cp + c] + md;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m;
    invstd[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c];   // z = (y - mean)*scale + beta: subtracting the mean first keeps the cancellation exact
    if (running_mean) {
        const double unb = npix > 1.0 ? var * npix / (npix - 1.0) : var;
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

// Backward finalize: dgamma, dbeta (accumulated into the parameter gradients) and the two
// per-channel means the apply pass needs.
__global__ void __launch_bounds__(256)
bn_bwd_finalize_kernel(const double* __restrict__ part, int nblk, int C, int Cp, double npix, float* dgamma,
                       float* dbeta, float* mean_dz, float* mean_dzx) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double s, q;
    fold_partials(part, nblk, Cp, c, s, q, true);
    if ((threadIdx.x & 63) != 0) return;
    dbeta[c] += (float)s;
    dgamma[c] += (float)q;
    mean_dz[c] = (float)(s / npix);
    mean_dzx[c] = (float)(q / npix);
}

__global__ void __launch_bounds__(256)
bias_grad_finalize_kernel(const double* __restrict__ part, int nblk, int C, int Cp, float* dbias) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double s, q;
    fold_partials(part, nblk, Cp, c, s, q, false);
    if ((threadIdx.x & 63) != 0) return;
    dbias[c] += (float)s;
}

// act = silu((y - mean)*scale + beta), written at a channel offset of a wider buffer (shift holds beta).
// Thread (v, r) of a workgroup owns the channels of 16-byte vector v for good (its per-channel
// constants live in registers) and walks pixels r, r + rows, ... of the grid-strided pixel list, four
// pixels per trip so that their loads overlap.
template <int DT>
__global__ void __launch_bounds__(256)
bn_apply_silu_kernel(const char* __restrict__ y, int y_ld, char* __restrict__ out, int o_ld, size_t npix, int vecs, int rows,
                     const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean, int C) {
    constexpr int N = Vec<DT>::N, ES = Vec<DT>::ES;
    const int v = threadIdx.x % vecs, r = threadIdx.x / vecs;
    if (r >= rows) return;
    float sc[N], sh[N], mu[N];
#pragma unroll
    for (int e = 0; e < N; ++e) {     // padding channels of the last vector: scale 0, shift 0 -> silu(0) = 0 stays 0
        const int c = v * N + e;
        sc[e] = c < C ? scale[c] : 0.f; sh[e] = c < C ? shift[c] : 0.f; mu[e] = c < C ? mean[c] : 0.f;
    }
    const size_t step = (size_t)gridDim.x * rows;
    for (size_t pb = (size_t)blockIdx.x * rows + r; pb < npix; pb += 4 * step) {
        float a[4][N];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t p = pb + u * step;
            ok[u] = p < npix;
            if (ok[u]) Vec<DT>::load(y + (p * y_ld) * ES + v * 16, a[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!ok[u]) continue;
#pragma unroll
            for (int e = 0; e < N; ++e) a[u][e] = silu_t<DT>((a[u][e] - mu[e]) * sc[e] + sh[e]);
            Vec<DT>::store(out + ((pb + u * step) * o_ld) * ES + v * 16, a[u]);
        }
    }
}

// dy = gamma*invstd * (dz - mean(dz) - xhat * mean(dz*xhat)), written over y in place.
template <int DT>
__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(char* __restrict__ y, int y_ld, const char* __restrict__ dact, int d_ld, size_t npix, int vecs, int rows,
                    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
                    const float* __restrict__ invstd, const float* __restrict__ mean_dz,
                    const float* __restrict__ mean_dzx, int C) {
    constexpr int N = Vec<DT>::N, ES = Vec<DT>::ES;
    const int v = threadIdx.x % vecs, r = threadIdx.x / vecs;
    if (r >= rows) return;
    float sc[N], sh[N], mu[N], is[N], mdz[N], mdx[N];
#pragma unroll
    for (int e = 0; e < N; ++e) {     // padding channels: scale 0 -> their gradient stays 0
        const int c = v * N + e;
        const bool real = c < C;
        sc[e] = real ? scale[c] : 0.f; sh[e] = real ? shift[c] : 0.f; mu[e] = real ? mean[c] : 0.f;
        is[e] = real ? invstd[c] : 0.f; mdz[e] = real ? mean_dz[c] : 0.f; mdx[e] = real ? mean_dzx[c] : 0.f;
    }
    const size_t step = (size_t)gridDim.x * rows;
    for (size_t pb = (size_t)blockIdx.x * rows + r; pb < npix; pb += 2 * step) {
        float a[2][N], g[2][N];
        bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const size_t p = pb + u * step;
            ok[u] = p < npix;
            if (ok[u]) {
                Vec<DT>::load(y + (p * y_ld) * ES + v * 16, a[u]);
                Vec<DT>::load(dact + (p * d_ld) * ES + v * 16, g[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!ok[u]) continue;
#pragma unroll
            for (int e = 0; e < N; ++e) {
                const float dz = g[u][e] * silu_grad_t<DT>((a[u][e] - mu[e]) * sc[e] + sh[e]);
                const float xh = (a[u][e] - mu[e]) * is[e];
                a[u][e] = sc[e] * (dz - mdz[e] - xh * mdx[e]);   // scale = gamma * invstd
            }
            Vec<DT>::store(y + ((pb + u * step) * y_ld) * ES + v * 16, a[u]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// pooling / upsample backward (gradients are accumulated: dx += ...)
// ------------------------------------------------------------------------------------------
// MaxPool2d(2,2): gradient goes to the first maximum in row-major window order (torch's argmax).
template <int DT>
__global__ void __launch_bounds__(256)
maxpool2_bwd_kernel(const char* __restrict__ x, int x_ld, const char* __restrict__ dy, int dy_ld, char* __restrict__ dx,
                    int dx_ld, int B, int H, int W, int vecs) {
    constexpr int N = Vec<DT>::N, ES = Vec<DT>::ES;
    const int Ho = H / 2, Wo = W / 2;
    const size_t n = (size_t)B * Ho * Wo * vecs;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int v = i % vecs;
        size_t pix = i / vecs;
        const int ox = pix % Wo; pix /= Wo;
        const int oy = pix % Ho;
        const int b = pix / Ho;
        float g[N], in[4][N], acc[4][N];
        Vec<DT>::load(dy + ((((size_t)b * Ho + oy) * Wo + ox) * dy_ld) * ES + v * 16, g);
        size_t off[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            off[q] = (((size_t)b * H + 2 * oy + (q >> 1)) * W + 2 * ox + (q & 1));
            Vec<DT>::load(x + (off[q] * x_ld) * ES + v * 16, in[q]);
            Vec<DT>::load(dx + (off[q] * dx_ld) * ES + v * 16, acc[q]);
        }
#pragma unroll
        for (int e = 0; e < N; ++e) {
            int best = 0;
            float m = in[0][e];
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (in[q][e] > m) { m = in[q][e]; best = q; }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q][e] += (q == best) ? g[e] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) Vec<DT>::store(dx + (off[q] * dx_ld) * ES + v * 16, acc[q]);
    }
}

// Stride-1 max pool k (pad k/2): one thread per output window and channel finds the first maximum
// in row-major order and adds the window's gradient there.  float32 gradients only (atomics).
__global__ void __launch_bounds__(256)
maxpool_s1_bwd_kernel(const float* __restrict__ x, int x_ld, const float* __restrict__ dy, int dy_ld,
                      float* __restrict__ dx, int dx_ld, int B, int H, int W, int C, int k) {
    const size_t n = (size_t)B * H * W * C;
    const int rad = k / 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = i % C;
        size_t pix = i / C;
        const int ox = pix % W; pix /= W;
        const int oy = pix % H;
        const int b = pix / H;
        float m = -INFINITY;
        size_t arg = 0;
        for (int gy = max(oy - rad, 0); gy <= min(oy + rad, H - 1); ++gy)
            for (int gx = max(ox - rad, 0); gx <= min(ox + rad, W - 1); ++gx) {
                const size_t o = ((size_t)b * H + gy) * W + gx;
                const float t = x[o * x_ld + c];
                if (t > m) { m = t; arg = o; }
            }
        atomicAdd(&dx[arg * dx_ld + c], dy[(((size_t)b * H + oy) * W + ox) * dy_ld + c]);
    }
}

// bf16 variant: one workgroup per (image, 8-channel vector).  The input plane and a float32
// gradient plane live in LDS; windows scatter with LDS float adds, then the plane is added to dx.
__global__ void __launch_bounds__(256)
maxpool_s1_bwd_bf16_kernel(const char* __restrict__ x, int x_ld, const char* __restrict__ dy, int dy_ld,
                           char* __restrict__ dx, int dx_ld, int H, int W, int k) {
    extern __shared__ __attribute__((aligned(16))) float pl[];
    const int hw = H * W, v = blockIdx.x, b = blockIdx.y, rad = k / 2;
    float* grad = pl;                  // [hw][8]
    float* val = pl + (size_t)hw * 8;  // [hw][8]
    for (int p = threadIdx.x; p < hw; p += 256) {
        float t[8];
        Vec<ODAMD_DTYPE_BF16>::load(x + (((size_t)b * hw + p) * x_ld) * 2 + v * 16, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) { val[p * 8 + e] = t[e]; grad[p * 8 + e] = 0.0f; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < hw * 8; i += 256) {
        const int e = i & 7, p = i >> 3;
        const int oy = p / W, ox = p - oy * W;
        float m = -INFINITY;
        int arg = 0;
        for (int gy = max(oy - rad, 0); gy <= min(oy + rad, H - 1); ++gy)
            for (int gx = max(ox - rad, 0); gx <= min(ox + rad, W - 1); ++gx) {
                const float t = val[(gy * W + gx) * 8 + e];
                if (t > m) { m = t; arg = gy * W + gx; }
            }
        const uint16_t g = *(const uint16_t*)(dy + (((size_t)b * hw + p) * dy_ld) * 2 + v * 16 + e * 2);
        atomicAdd(&grad[arg * 8 + e], bf16_to_f32(g));
    }
    __syncthreads();
    for (int p = threadIdx.x; p < hw; p += 256) {
        float t[8];
        char* d = dx + (((size_t)b * hw + p) * dx_ld) * 2 + v * 16;
        Vec<ODAMD_DTYPE_BF16>::load(d, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] += grad[p * 8 + e];
        Vec<ODAMD_DTYPE_BF16>::store(d, t);
    }
}

// The three stride-1 max pools of an SPP block (k = 5, 9, 13 on the same input, model/layer.py:215-217) in one launch.
// The first maximum of a window in row-major order is the leftmost maximum of the first row that holds the maximum, so
// the search separates: pass 1 finds, per pixel and width, the leftmost maximum of its row segment (one left-to-right
// walk over 13 elements serves the three nested widths); pass 2 walks the rows of each window top-down over those
// records.  54 comparisons per element instead of 25 + 81 + 169.
// No atomics at all (float32 and bf16): the scatter "add the window's gradient at its
// arg-max" becomes a gather.  Pass 1 and 2 as above, but pass 2 only RECORDS every window's arg-max pixel (and its
// gradient) in LDS; in pass 3 every input element walks the windows that can have picked it, in a fixed order (k = 5, 9, 13,
// columns, rows), and adds the gradients of those whose arg-max it is.  Float addition in a fixed order: the result is
// bitwise repeatable, which LDS / global float atomics are not as soon as three windows meet in one element.
// One workgroup per (image, 16-byte channel vector); per element of the plane: value / sum (float), three gradients
// (storage type), three row-maximum columns (bytes: W <= 255) and three arg-max pixels (16 bits: H * W <= 65535).
template <int DT> struct SppStore;
template <> struct SppStore<ODAMD_DTYPE_F32> {
    typedef float T;
    static __device__ __forceinline__ float get(float v) { return v; }
};
template <> struct SppStore<ODAMD_DTYPE_BF16> {
    typedef uint16_t T;
    static __device__ __forceinline__ float get(uint16_t v) { return bf16_to_f32(v); }
};
template <int DT>
__host__ __device__ constexpr size_t spp_gather_bytes_per_pixel() {
    return (size_t)Vec<DT>::N * (sizeof(float) + 3 * sizeof(typename SppStore<DT>::T) + 3 + 3 * sizeof(uint16_t));
}

// NT threads: the passes are chains of dependent LDS reads, many waves hide their latency
template <int DT, int NT>
__global__ void __launch_bounds__(NT)
spp_bwd_gather_kernel(const char* __restrict__ x, int x_ld, const char* __restrict__ dy5, const char* __restrict__ dy9,
                      const char* __restrict__ dy13, int dy_ld, char* __restrict__ dx, int dx_ld, int H, int W) {
    constexpr int N = Vec<DT>::N, ES = Vec<DT>::ES;
    typedef typename SppStore<DT>::T S;
    extern __shared__ __attribute__((aligned(16))) float pl[];
    const int hw = H * W, v = blockIdx.x, b = blockIdx.y, ne = hw * N;
    float* val = pl;                                   // [hw][N]; the gradient sums from pass 3 on
    S* dyv = (S*)(val + ne);                           // [3][hw][N]
    uint16_t* argw = (uint16_t*)(dyv + 3 * ne);        // [3][hw][N]
    uint8_t* rcol = (uint8_t*)(argw + 3 * ne);         // [3][hw][N]
    const char* dys[3] = {dy5, dy9, dy13};
    for (int i = threadIdx.x; i < 4 * hw; i += NT) {   // one 16-byte vector per thread and turn: x, then the three gradients
        const int which = i / hw, p = i - which * hw;
        if (which == 0) {
            float t[N];
            Vec<DT>::load(x + (((size_t)b * hw + p) * x_ld) * ES + v * 16, t);
#pragma unroll
            for (int e = 0; e < N; ++e) val[p * N + e] = t[e];
        } else {
            const char* g = dys[which - 1] + (((size_t)b * hw + p) * dy_ld) * ES + v * 16;
            const uint4 raw = *(const uint4*)g;
            *(uint4*)(dyv + (which - 1) * ne + p * N) = raw;
        }
    }
    __syncthreads();
    // The loops below have compile-time trip counts and clamped coordinates (a position outside the plane reads a valid
    // slot and counts as -inf), so that the compiler issues a pass's LDS reads together instead of one per iteration.
    for (int i = threadIdx.x; i < ne; i += NT) {       // pass 1: leftmost row maxima of widths 5, 9, 13
        const int e = i % N, p = i / N;
        const int oy = p / W, ox = p - oy * W;
        float t[13];
#pragma unroll
        for (int d = -6; d <= 6; ++d) {
            const int gx = ox + d;
            const float q = val[(oy * W + min(max(gx, 0), W - 1)) * N + e];
            t[d + 6] = (gx >= 0 && gx < W) ? q : -INFINITY;
        }
        float m5 = -INFINITY, m9 = -INFINITY, m13 = -INFINITY;
        int c5 = ox, c9 = ox, c13 = ox;
#pragma unroll
        for (int d = -6; d <= 6; ++d) {
            const float q = t[d + 6];
            if (q > m13) { m13 = q; c13 = ox + d; }
            if (d >= -4 && d <= 4 && q > m9) { m9 = q; c9 = ox + d; }
            if (d >= -2 && d <= 2 && q > m5) { m5 = q; c5 = ox + d; }
        }
        rcol[i] = (uint8_t)c5; rcol[ne + i] = (uint8_t)c9; rcol[2 * ne + i] = (uint8_t)c13;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ne; i += NT) {       // pass 2: first row holding the window maximum
        const int e = i % N, p = i / N;
        const int oy = p / W, ox = p - oy * W;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            constexpr int RMAX = 6;
            const int rad = 2 + 2 * k;
            int col[2 * RMAX + 1];
            float t[2 * RMAX + 1];
#pragma unroll
            for (int d = -RMAX; d <= RMAX; ++d)
                if (d >= -rad && d <= rad) col[d + RMAX] = rcol[k * ne + (min(max(oy + d, 0), H - 1) * W + ox) * N + e];
#pragma unroll
            for (int d = -RMAX; d <= RMAX; ++d)
                if (d >= -rad && d <= rad) {
                    const int gy = oy + d;
                    const float q = val[(min(max(gy, 0), H - 1) * W + col[d + RMAX]) * N + e];
                    t[d + RMAX] = (gy >= 0 && gy < H) ? q : -INFINITY;
                }
            float m = -INFINITY;
            int arg = p;
#pragma unroll
            for (int d = -RMAX; d <= RMAX; ++d)
                if (d >= -rad && d <= rad && t[d + RMAX] > m) { m = t[d + RMAX]; arg = (oy + d) * W + col[d + RMAX]; }
            argw[k * ne + i] = (uint16_t)arg;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ne; i += NT) {       // pass 3: gather, fixed order
        const int e = i % N, p = i / N;
        const int py = p / W, px = p - py * W;
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            constexpr int RMAX = 6;
            const int rad = 2 + 2 * k;
            const uint16_t* aw = argw + k * ne + e;
            const S* dv = dyv + k * ne + e;
            // a window centred in column wx can only pick this element when the leftmost maximum of ITS segment of row py
            // is column px (one column in 2 rad + 1 on average): only those columns are walked down
            const uint8_t* rc = rcol + k * ne + e;
            int pick[2 * RMAX + 1];
#pragma unroll
            for (int d = -RMAX; d <= RMAX; ++d)
                if (d >= -rad && d <= rad) pick[d + RMAX] = rc[(py * W + min(max(px + d, 0), W - 1)) * N];
#pragma unroll
            for (int d = -RMAX; d <= RMAX; ++d)
                if (d >= -rad && d <= rad) {
                    const int wx = px + d;
                    if (wx < 0 || wx >= W || pick[d + RMAX] != px) continue;
                    for (int wy = max(py - rad, 0); wy <= min(py + rad, H - 1); ++wy) {
                        const int q = (wy * W + wx) * N;
                        if (aw[q] == (uint16_t)p) sum += SppStore<DT>::get(dv[q]);
                    }
                }
        }
        // every element writes its own slot; nobody reads `val` as values any more (pass 2 is behind the barrier)
        val[i] = sum;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < hw; p += NT) {
        float t[N];
        char* d = dx + (((size_t)b * hw + p) * dx_ld) * ES + v * 16;
        Vec<DT>::load(d, t);
#pragma unroll
        for (int e = 0; e < N; ++e) t[e] += val[p * N + e];
        Vec<DT>::store(d, t);
    }
}

// nearest x2 upsample: dx(y, x) += sum of the 2x2 block of dy.
template <int DT>
__global__ void __launch_bounds__(256)
upsample2_bwd_kernel(const char* __restrict__ dy, int dy_ld, char* __restrict__ dx, int dx_ld, int B, int H, int W, int vecs) {
    constexpr int N = Vec<DT>::N, ES = Vec<DT>::ES;
    const size_t n = (size_t)B * H * W * vecs;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int v = i % vecs;
        size_t pix = i / vecs;
        const int ix = pix % W; pix /= W;
        const int iy = pix % H;
        const int b = pix / H;
        float acc[N], g[N];
        char* d = dx + ((((size_t)b * H + iy) * W + ix) * dx_ld) * ES + v * 16;
        Vec<DT>::load(d, acc);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            Vec<DT>::load(dy + ((((size_t)b * 2 * H + 2 * iy + (q >> 1)) * 2 * W + 2 * ix + (q & 1)) * dy_ld) * ES + v * 16, g);
#pragma unroll
            for (int e = 0; e < N; ++e) acc[e] += g[e];
        }
        Vec<DT>::store(d, acc);
    }
}

// ------------------------------------------------------------------------------------------
// Convolution weight gradient:
//   dW[co][ci][ky][kx] = sum_pixels dY[pixel][co] * X[pixel*stride - pad + (ky,kx)][ci]
// GEMM view: M = output channels, N = input channels (per tap), K = pixels.  A workgroup owns one
// (co tile, ci tile) pair and every nsplit-th pixel patch; per patch it stages the dY patch
// [pixels][CO_T] and the X halo patch [halo pixels][CI_T] in LDS (a fragment read touches 32
// consecutive channels: conflict free) and accumulates all taps in registers.  The four waves are laid out
// WCO x WCI over the tile and, when the tile is smaller than four waves, WPX ways over the pixels of
// a patch.  Each wave keeps R x R MFMA tiles per tap.
// Every wave stores its sums to its own slice of `part` ([split][pixel group][taps][Cout][Cin]: the lanes of a
// store are consecutive input channels) and a second kernel folds the slices in a fixed order and adds the sums
// to dW in the torch weight layout [Cout][Cin][taps]: no atomics, the result is reproducible.  Workgroup ids are laid out so that the pairs of one split (which read the
// same patches) run on the same XCD and share its L2.
// ------------------------------------------------------------------------------------------
struct WgradP {
    const void* x; const void* dy; float* part;
    int B, H, W, Ho, Wo, Cin, Cout, x_ld, dy_ld;
    int TH, TW, npix, tiles_x, tiles_y, IH, IW, nsplit, ntiles, ncin, npairs;
    float inv_tw, inv_iw;      // 1 / TW, 1 / IW (div_small_t)
    size_t slice;
};

// The GEMM runs on v_mfma_f32_32x32x16_bf16 (float32 accumulation and float32 dW).  The MFMA wants, per lane, 8 consecutive K (= pixel) values of one channel, while the
// staged NHWC patches keep channels contiguous: the transposing LDS read ds_read_b64_tr_b16 delivers, to each group
// of 16 lanes, a block of 4 pixel rows x 16 channels column-major (lane i gets channel i of the four rows), every
// lane supplying the address of one row piece.  Two such reads make a lane's 8 pixels; the rows are addressed
// through the pixel table, so tap shifts and patch-row breaks need no alignment.  One K step covers 16 pixels of
// the patch.  Pixel rows in LDS are padded so that four consecutive rows of 64 bytes fall on different banks
// (wgrad_row_bytes).
typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8_t lds_tr8(const char* first, const char* second) {
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)first);
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)second);
    return bf16x8_t{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
// LDS bytes per staged pixel of a `channels`-wide tile whose consecutive K pixels are `step` rows apart
__host__ __device__ constexpr int wgrad_row_bytes(int channels, int step) {
    return step == 1 ? (channels == 32 ? 64 : channels * 2 + 64) : channels * 2 + 32;
}
__device__ __forceinline__ uint4 mask_bf16x8(uint4 v, int valid) {   // keep the first `valid` (1..7) elements
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (e >= valid) w[e >> 1] &= (e & 1) ? 0x0000FFFFu : 0xFFFF0000u;
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// float4 -> four (hi, lo) bf16 pairs: hi = bf16(v), lo = bf16(v - hi)
__device__ __forceinline__ void split_store(float4 v, uint16_t* hi, uint16_t* lo) {
    const float f[4] = {v.x, v.y, v.z, v.w};
    uint16_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = f32_to_bf16(f[e]);
        l[e] = f32_to_bf16(f[e] - bf16_to_f32(h[e]));
    }
    *(uint2*)hi = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    *(uint2*)lo = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
}

// c ? a : b by bit masks: a select between two values that live in registers, where the compiler would otherwise
// put the pair into a private-memory table and index it with c
__device__ __forceinline__ unsigned pick_u32(bool c, unsigned a, unsigned b) {
    const unsigned m = 0u - (unsigned)c;
    return (a & m) | (b & ~m);
}
__device__ __forceinline__ const char* pick_ptr(bool c, const char* a, const char* b) {
    const unsigned long long m = 0ull - (unsigned long long)c;
    return (const char*)(((unsigned long long)a & m) | ((unsigned long long)b & ~m));
}

// 16-byte pieces per thread that one staging batch of the weight-gradient kernel holds in registers
__host__ __device__ constexpr int wgrad_batch(int ks, bool split) { return split ? 6 : (ks == 3 ? 9 : 12); }
__device__ __forceinline__ int div_small_t(int v, float inv) { return (int)(((float)v + 0.5f) * inv); }   // exact for v < 2^21, divisor < 2^12

// SPLIT: x and dy are float32; every value is staged as two bf16 numbers hi = bf16(v), lo = bf16(v - hi)
// and a product is taken as hi*hi + hi*lo + lo*hi on three bf16 MFMAs with float32 accumulation (the
// dropped lo*lo term and the rounding of lo are ~2^-17 of the product): float32-grade weight gradients
// at five times the rate of the float32 MFMA.
template <int KS, int ST, int WCO, int WCI, int R, bool SPLIT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_wgrad_bf16_kernel(WgradP p) {
    constexpr int TAPS = KS * KS, PAD = (KS - 1) / 2;
    constexpr int CO_T = 32 * WCO * R, CI_T = 32 * WCI * R, WPX = 4 / (WCO * WCI);
    constexpr int NACC = TAPS * R * R;
    constexpr int EPP = SPLIT ? 4 : 8;                     // elements per 16-byte piece in global memory
    constexpr int GES = SPLIT ? 4 : 2;                     // bytes per element in global memory
    constexpr int PCO = CO_T / EPP, PCI = CI_T / EPP;      // pieces per staged pixel
    extern __shared__ __attribute__((aligned(16))) char wl8[];
    const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
    const int pair = slot % p.npairs, split = (slot / p.npairs) * 8 + xcd;
    if (split >= p.nsplit) return;
    const int npix = p.npix, nhalo = p.IH * p.IW;
    constexpr int PSA = wgrad_row_bytes(CO_T, 1), PSX = wgrad_row_bytes(CI_T, ST);   // LDS bytes per dY / x pixel
    int* tab = (int*)wl8;                                   // [npix] halo byte offset of tap (0,0)
    char* dyl = wl8 + (size_t)npix * 4;                     // [npix][PSA]    (SPLIT: hi parts)
    char* xl = dyl + (size_t)npix * PSA;                    // [IH*IW][PSX]   (SPLIT: hi parts)
    char* dyl_lo = xl + (size_t)nhalo * PSX;                // SPLIT only: lo parts, same shapes
    char* xl_lo = dyl_lo + (size_t)npix * PSA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kk = lane >> 5;
    // transposed reads: lane 4q + p of a 16-lane group addresses row q, channels 4p .. 4p+3 of the group's 16
    const int trq = (lane >> 2) & 3, trc = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    const int wco = wave % WCO, wci = (wave / WCO) % WCI, wpx = wave / (WCO * WCI);
    const int cb = pair / p.ncin, ib = pair % p.ncin;
    const int co0 = cb * CO_T, ci0 = ib * CI_T;
    const int nco = min(CO_T, p.Cout - co0), nci = min(CI_T, p.Cin - ci0);

    for (int q = tid; q < npix; q += 256) {
        const int ty = div_small_t(q, p.inv_tw), tx = q - ty * p.TW;
        tab[q] = ty < p.TH ? ((ty * ST) * p.IW + tx * ST) * PSX : 0;
    }
    int tapoff[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) tapoff[t] = ((t / KS) * p.IW + (t % KS)) * PSX;

    f32x16_t acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.0f;

    const int steps = npix >> 4;
    const int s0 = wpx * (steps / WPX), s1 = s0 + steps / WPX;
    // Staging: the 16-byte pieces of the dY patch and of the x halo patch form one list, NB pieces per thread and
    // batch; the global loads of a batch are all in flight before its first LDS store.  When a patch is a single
    // batch, the loads of the workgroup's NEXT patch are issued before the matrix work of the current one and land in
    // LDS after it.
    constexpr int NB = wgrad_batch(KS, SPLIT);
    const int nreal = p.TH * p.TW;                       // dY rows beyond the patch stay zero (cleared once below)
    const int nd = nreal * PCO, total = nd + nhalo * PCI;
    for (int i = tid; i < (npix - nreal) * (PSA / 16); i += 256) {
        *(uint4*)(dyl + nreal * PSA + i * 16) = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (SPLIT) *(uint4*)(dyl_lo + nreal * PSA + i * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    const bool single = total <= 256 * NB;
    uint4 v[NB];
    auto load_batch = [&](int t, int b0) __attribute__((always_inline)) {
        const int b = t / (p.tiles_x * p.tiles_y), tt = t % (p.tiles_x * p.tiles_y);
        const int oy0 = (tt / p.tiles_x) * p.TH, ox0 = (tt % p.tiles_x) * p.TW;
        const int iy0 = oy0 * ST - PAD, ix0 = ox0 * ST - PAD;
        const char* dyb = (const char*)p.dy + ((size_t)b * p.Ho * p.Wo) * p.dy_ld * GES;
        const char* xbp = (const char*)p.x + ((size_t)b * p.H * p.W) * p.x_ld * GES;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int i = b0 + tid + j * 256;
            asm volatile("" : "+v"(i));        // recompute the piece's coordinates per patch: kept across patches they cost ~10 registers a piece
            v[j] = make_uint4(0u, 0u, 0u, 0u);
            if (i >= total) continue;
            int valid;
            const char* src;
            if constexpr (KS == 1) {
                // one row of pixels (see wgrad_plan): patch t starts at pixel t * TW, no halo
                const bool isd = i < nd;
                const int k = isd ? i : i - nd;
                const int pix = isd ? k / PCO : k / PCI, part = isd ? k % PCO : k % PCI;
                const unsigned q = (unsigned)(t * p.TW + pix);
                valid = q < (unsigned)p.Wo ? (isd ? nco : nci) - part * EPP : 0;
                src = pick_ptr(isd, (const char*)p.dy, (const char*)p.x) +
                      ((size_t)q * pick_u32(isd, (unsigned)p.dy_ld, (unsigned)p.x_ld) + pick_u32(isd, (unsigned)co0, (unsigned)ci0) + (unsigned)(part * EPP)) * GES;
            } else if (i < nd) {
                const int pix = i / PCO, part = i % PCO;
                const int ty = div_small_t(pix, p.inv_tw), tx = pix - ty * p.TW;
                const int oy = oy0 + ty, ox = ox0 + tx;
                valid = (oy < p.Ho && ox < p.Wo) ? nco - part * EPP : 0;
                src = dyb + ((size_t)(oy * p.Wo + ox) * p.dy_ld + co0 + part * EPP) * GES;
            } else {
                const int k = i - nd, pix = k / PCI, part = k % PCI;
                const int hy = div_small_t(pix, p.inv_iw);
                const int gy = iy0 + hy, gx = ix0 + pix - hy * p.IW;
                valid = (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? nci - part * EPP : 0;
                src = xbp + ((size_t)(gy * p.W + gx) * p.x_ld + ci0 + part * EPP) * GES;
            }
            if (valid > 0) {
                v[j] = *(const uint4*)src;
                if (valid < EPP) {                 // ragged channel tail: the rest of the piece belongs to a neighbour
                    if constexpr (SPLIT) {
                        if (valid < 2) v[j].y = 0u;
                        if (valid < 3) v[j].z = 0u;
                        v[j].w = 0u;
                    } else {
                        v[j] = mask_bf16x8(v[j], valid);
                    }
                }
            }
        }
    };
    // 1x1 layers with single-batch patches: the element offset of a piece from the patch's first pixel does not depend
    // on the patch and is kept; the patch adds a uniform offset and the row-end test.  (Selections between two fields of
    // `p` by a lane-dependent condition would make the compiler keep `p` in private memory: copies first.)
    const unsigned dy_ld = (unsigned)p.dy_ld, x_ld = (unsigned)p.x_ld;
    const char* const dy_base = (const char*)p.dy;
    const char* const x_base = (const char*)p.x;
    const bool ragged = (nco % EPP) != 0 || (nci % EPP) != 0;      // workgroup-uniform: a channel tile ends inside a 16-byte piece
    unsigned flat_off[KS == 1 ? NB : 1];
    if constexpr (KS == 1) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int i = tid + j * 256;
            const bool isd = i < nd;
            const int k = isd ? i : i - nd;
            const int pix = isd ? k / PCO : k / PCI, part = isd ? k % PCO : k % PCI;
            flat_off[j] = (unsigned)pix * pick_u32(isd, dy_ld, x_ld) + pick_u32(isd, (unsigned)co0, (unsigned)ci0) + (unsigned)(part * EPP);
        }
    }
    auto load_flat = [&](int t) __attribute__((always_inline)) {
        if constexpr (KS == 1) {
            const int q0 = t * p.TW, lim = p.Wo - q0;                  // uniform
            const unsigned ody = (unsigned)q0 * dy_ld, ox = (unsigned)q0 * x_ld;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                int i = tid + j * 256;
                asm volatile("" : "+v"(i));
                const bool isd = i < nd;
                const int k = isd ? i : i - nd;
                const int pix = isd ? k / PCO : k / PCI, part = isd ? k % PCO : k % PCI;
                const int valid = (i < total && pix < lim) ? (int)pick_u32(isd, (unsigned)nco, (unsigned)nci) - part * EPP : 0;
                v[j] = make_uint4(0u, 0u, 0u, 0u);
                if (valid > 0) {
                    v[j] = *(const uint4*)(pick_ptr(isd, dy_base, x_base) + ((size_t)pick_u32(isd, ody, ox) + flat_off[j]) * GES);
                    if (ragged && valid < EPP) {
                        if constexpr (SPLIT) {
                            if (valid < 2) v[j].y = 0u;
                            if (valid < 3) v[j].z = 0u;
                            v[j].w = 0u;
                        } else {
                            v[j] = mask_bf16x8(v[j], valid);
                        }
                    }
                }
            }
        }
    };
    auto store_batch = [&](int b0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int i = b0 + tid + j * 256;
            asm volatile("" : "+v"(i));
            if (i >= total) continue;
            const bool isd = i < nd;
            const int k = isd ? i : i - nd;
            const int pix = isd ? k / PCO : k / PCI, part = isd ? k % PCO : k % PCI;
            const int off = pix * (isd ? PSA : PSX) + part * (SPLIT ? 8 : 16);
            if constexpr (SPLIT) {
                const float4 f = make_float4(__uint_as_float(v[j].x), __uint_as_float(v[j].y), __uint_as_float(v[j].z), __uint_as_float(v[j].w));
                split_store(f, (uint16_t*)((isd ? dyl : xl) + off), (uint16_t*)((isd ? dyl_lo : xl_lo) + off));
            } else {
                *(uint4*)((isd ? dyl : xl) + off) = v[j];
            }
        }
    };
    auto prefetch = [&](int t) __attribute__((always_inline)) {
        if constexpr (KS == 1) load_flat(t);
        else load_batch(t, 0);
    };
    if (single && split < p.ntiles) prefetch(split);
    for (int t = split; t < p.ntiles; t += p.nsplit) {
        __syncthreads();                                   // the previous patch has been consumed
        if (single) {
            store_batch(0);
        } else {
            for (int b0 = 0; b0 < total; b0 += 256 * NB) { load_batch(t, b0); store_batch(b0); }
        }
        __syncthreads();
        if (single && t + p.nsplit < p.ntiles) prefetch(t + p.nsplit);
        for (int s = s0; s < s1; ++s) {
            const int qa = 16 * s + 8 * kk + trq;                    // this lane's row of the first block; the second is 4 further
            const int t0 = tab[qa], t1 = tab[qa + 4];
            const char* ap = dyl + qa * PSA + (wco * R * 32 + trc) * 2;
            const char* bp = xl + (wci * R * 32 + trc) * 2;
            bf16x8_t a[R], al[SPLIT ? R : 1];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                a[r] = lds_tr8(ap + r * 64, ap + 4 * PSA + r * 64);
                if constexpr (SPLIT) al[r] = lds_tr8(ap + (dyl_lo - dyl) + r * 64, ap + (dyl_lo - dyl) + 4 * PSA + r * 64);
            }
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                for (int rc = 0; rc < R; ++rc) {
                    bf16x8_t bv, bl;
                    bv = lds_tr8(bp + t0 + tapoff[tp] + rc * 64, bp + t1 + tapoff[tp] + rc * 64);
                    if constexpr (SPLIT) bl = lds_tr8(bp + (xl_lo - xl) + t0 + tapoff[tp] + rc * 64, bp + (xl_lo - xl) + t1 + tapoff[tp] + rc * 64);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        f32x16_t& c = acc[(tp * R + r) * R + rc];
                        if constexpr (SPLIT) {
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[r], bv, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r], bl, c, 0, 0, 0);
                        }
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r], bv, c, 0, 0, 0);
                    }
                }
        }
    }
    // The WPX waves that split the pixels of the patches add their sums through LDS (in wave order, two accumulator
    // tiles per round), so that a workgroup writes ONE slice.
    if constexpr (WPX > 1) {
        constexpr int CH = 2, GROUPS = WCO * WCI;
        float* red = (float*)wl8;                               // [4 waves][CH][16][64 lanes]
#pragma unroll
        for (int a0 = 0; a0 < NACC; a0 += CH) {
            __syncthreads();
            if (wpx > 0) {
#pragma unroll
                for (int a = a0; a < (a0 + CH < NACC ? a0 + CH : NACC); ++a)
#pragma unroll
                    for (int e = 0; e < 16; ++e) red[((wave * CH + (a - a0)) * 16 + e) * 64 + lane] = acc[a][e];
            }
            __syncthreads();
            if (wpx == 0) {
#pragma unroll
                for (int w = 1; w < WPX; ++w)
#pragma unroll
                    for (int a = a0; a < (a0 + CH < NACC ? a0 + CH : NACC); ++a)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[a][e] += red[(((wave + GROUPS * w) * CH + (a - a0)) * 16 + e) * 64 + lane];
            }
        }
        if (wpx > 0) return;
    }
    float* out = p.part + (size_t)split * p.slice;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int rc = 0; rc < R; ++rc) {
            const int ci = ci0 + (wci * R + rc) * 32 + col;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + (wco * R + r) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kk;
                if (co < p.Cout && ci < p.Cin) {       // [tap][Cout][Cin]: a half wave writes 32 consecutive floats
                    float* o = out + (size_t)co * p.Cin + ci;
#pragma unroll
                    for (int tp = 0; tp < TAPS; ++tp) o[(size_t)tp * p.Cout * p.Cin] = acc[(tp * R + r) * R + rc][e];
                }
            }
        }
}

// First layer (3 input channels, 3x3, stride 1): the 27 (tap, channel) pairs are the N dimension of
// ONE 32x32 MFMA tile (im2col in the LDS address of each lane) instead of nine tiles that are 3/32
// full.  x is the NHWC image padded to 8 channels.  Four waves split the pixels of a patch; slices and
// fold as in the general kernels.
template <int DT>
__global__ void __launch_bounds__(256, 5) conv_wgrad_first_kernel(WgradP p) {
    constexpr int ES = DT == ODAMD_DTYPE_F32 ? 4 : 2;
    constexpr int KSTEP = DT == ODAMD_DTYPE_F32 ? 2 : 16;   // pixels per MFMA
    constexpr int KL = KSTEP / 2;                            // pixels per lane half
    extern __shared__ __attribute__((aligned(16))) char wl8[];
    const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
    const int pair = slot % p.npairs, split = (slot / p.npairs) * 8 + xcd;
    if (split >= p.nsplit) return;
    const int npix = p.npix, nhalo = p.IH * p.IW;
    int* tab = (int*)wl8;                                    // [npix] halo pixel index of tap (0,0)
    char* dyl = wl8 + (size_t)npix * 4;                      // [npix][32] elements
    char* xl = dyl + (size_t)npix * 32 * ES;                 // [nhalo][8] elements
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kk = lane >> 5;
    const int co0 = pair * 32;
    const int nco = min(32, p.Cout - co0);
    // lane's im2col column: n = tap * Cin + c
    const int tap = col / p.Cin, c = col - tap * p.Cin;
    const bool live = col < 9 * p.Cin;
    const int lane_off = live ? ((tap / 3) * p.IW + (tap % 3)) * 8 + c : 0;
    for (int q = tid; q < npix; q += 256) {
        const int ty = div_small_t(q, p.inv_tw), tx = q - ty * p.TW;
        tab[q] = ty < p.TH ? ty * p.IW + tx : 0;     // padding pixels: dY is zero there, any staged (finite) x will do
    }
    f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const int steps = npix / KSTEP;
    const int s0 = wave * (steps / 4), s1 = s0 + steps / 4;
    // staging as in conv_wgrad_bf16_kernel: one list of 16-byte pieces, batches of NB per thread, the next patch's
    // loads in flight during the matrix work when a patch is one batch
    constexpr int NB = 6;
    constexpr int PPP = 32 * ES / 16;                    // 16-byte pieces per dY pixel
    constexpr int XPP = 8 * ES / 16;                     // pieces per image pixel (8 channels)
    constexpr int per = 16 / ES;
    const int nreal = p.TH * p.TW;
    const int nd = nreal * PPP, total = nd + nhalo * XPP;
    const bool single = total <= 256 * NB;
    for (int i = tid; i < (npix - nreal) * PPP; i += 256) *(uint4*)(dyl + (size_t)nreal * 32 * ES + i * 16) = make_uint4(0u, 0u, 0u, 0u);
    uint4 v[NB];
    auto load_batch = [&](int t, int b0) __attribute__((always_inline)) {
        const int b = t / (p.tiles_x * p.tiles_y), tt = t % (p.tiles_x * p.tiles_y);
        const int oy0 = (tt / p.tiles_x) * p.TH, ox0 = (tt % p.tiles_x) * p.TW;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int i = b0 + tid + j * 256;
            asm volatile("" : "+v"(i));
            v[j] = make_uint4(0u, 0u, 0u, 0u);
            if (i >= total) continue;
            if (i < nd) {
                const int pix = i / PPP, part = i % PPP;
                const int ty = div_small_t(pix, p.inv_tw), tx = pix - ty * p.TW;
                const int oy = oy0 + ty, ox = ox0 + tx;
                const int valid = (oy < p.Ho && ox < p.Wo) ? nco - part * per : 0;
                if (valid > 0) {
                    v[j] = *(const uint4*)((const char*)p.dy + ((((size_t)b * p.Ho + oy) * p.Wo + ox) * p.dy_ld + co0 + part * per) * ES);
                    if (valid < per) {
                        if constexpr (DT == ODAMD_DTYPE_F32) {
                            if (valid < 2) v[j].y = 0u;
                            if (valid < 3) v[j].z = 0u;
                            v[j].w = 0u;
                        } else {
                            v[j] = mask_bf16x8(v[j], valid);
                        }
                    }
                }
            } else {
                const int k = i - nd, pix = k / XPP, part = k % XPP;
                const int hy = div_small_t(pix, p.inv_iw);
                const int gy = oy0 - 1 + hy, gx = ox0 - 1 + pix - hy * p.IW;
                if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                    v[j] = *(const uint4*)((const char*)p.x + ((((size_t)b * p.H + gy) * p.W + gx) * p.x_ld) * ES + part * 16);
            }
        }
    };
    auto store_batch = [&](int b0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int i = b0 + tid + j * 256;
            asm volatile("" : "+v"(i));
            if (i >= total) continue;
            if (i < nd) *(uint4*)(dyl + ((size_t)(i / PPP) * 32 + (i % PPP) * per) * ES) = v[j];
            else *(uint4*)(xl + ((size_t)((i - nd) / XPP) * 8) * ES + ((i - nd) % XPP) * 16) = v[j];
        }
    };
    if (single && split < p.ntiles) load_batch(split, 0);
    for (int t = split; t < p.ntiles; t += p.nsplit) {
        __syncthreads();
        if (single) {
            store_batch(0);
        } else {
            for (int b0 = 0; b0 < total; b0 += 256 * NB) { load_batch(t, b0); store_batch(b0); }
        }
        __syncthreads();
        if (single && t + p.nsplit < p.ntiles) load_batch(t + p.nsplit, 0);
        for (int s = s0; s < s1; ++s) {
            const int q0 = KSTEP * s + KL * kk;
            if constexpr (DT == ODAMD_DTYPE_F32) {
                const float a = ((const float*)dyl)[q0 * 32 + col];
                const float bv = live ? ((const float*)xl)[tab[q0] * 8 + lane_off] : 0.0f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
            } else {
                const int4 t0 = *(const int4*)(tab + q0), t1 = *(const int4*)(tab + q0 + 4);
                const int xo[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                bf16x8_t a, bv;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    a[e] = (short)((const uint16_t*)dyl)[(q0 + e) * 32 + col];
                    bv[e] = live ? (short)((const uint16_t*)xl)[xo[e] * 8 + lane_off] : (short)0;
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bv, acc, 0, 0, 0);
            }
        }
    }
    float* out = p.part + ((size_t)split * 4 + wave) * p.slice;
    if (live) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co0 + (e & 3) + 8 * (e >> 2) + 4 * kk;
            if (co < p.Cout) out[((size_t)co * p.Cin + c) * 9 + tap] = acc[e];
        }
    }
}

// dw[i] += sum over slices (fixed order).  Workgroup: 32 elements x G slice groups (G = 32 when there
// are hundreds of slices, so that no thread walks more than a few dozen dependent additions).
// A slice is [taps][cc] (cc = Cout * Cin); element (tap, j) belongs to dw[j * taps + tap] (taps = 1: same order).
template <int G>
__global__ void __launch_bounds__(32 * G) wgrad_fold_kernel(const float* __restrict__ part, size_t slice, int nslices,
                                                            float* __restrict__ dw, int taps, int cc) {
    __shared__ float red[G][32];
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const size_t i = (size_t)blockIdx.x * 32 + e;
    float s = 0.f;
    if (i < slice) {
        int k = g;
        for (; k + 3 * G < nslices; k += 4 * G) {        // four independent loads in flight
            const float a = part[(size_t)k * slice + i], b = part[(size_t)(k + G) * slice + i];
            const float c = part[(size_t)(k + 2 * G) * slice + i], d = part[(size_t)(k + 3 * G) * slice + i];
            s += a; s += b; s += c; s += d;
        }
        for (; k < nslices; k += G) s += part[(size_t)k * slice + i];
    }
    red[g][e] = s;
    __syncthreads();
    if (g == 0 && i < slice) {
        float t = red[0][e];
#pragma unroll
        for (int k = 1; k < G; ++k) t += red[k][e];
        const int tap = taps == 1 ? 0 : (int)(i / (size_t)cc);
        dw[(i - (size_t)tap * cc) * taps + tap] += t;
    }
}

// ------------------------------------------------------------------------------------------
// Detection loss (train_class.py:410-474): forward value and gradient wrt the decoded predictions.
// ------------------------------------------------------------------------------------------
// value + 4 partial derivatives (wrt predicted cx, cy, w, h): forward-mode differentiation keeps
// the CIoU backward an exact transcription of the forward formula.
struct D4 {
    float v, d[4];
};
__device__ __forceinline__ D4 dconst(float v) { return {v, {0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ D4 dvar(float v, int i) { D4 r = dconst(v); r.d[i] = 1.f; return r; }
__device__ __forceinline__ D4 operator+(D4 a, D4 b) { D4 r; r.v = a.v + b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ __forceinline__ D4 operator-(D4 a, D4 b) { D4 r; r.v = a.v - b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ __forceinline__ D4 operator*(D4 a, D4 b) { D4 r; r.v = a.v * b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ __forceinline__ D4 operator/(D4 a, D4 b) {
    D4 r; r.v = a.v / b.v;
    for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
    return r;
}
__device__ __forceinline__ D4 dscale(D4 a, float s) { D4 r; r.v = a.v * s; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * s; return r; }
__device__ __forceinline__ D4 dmax(D4 a, D4 b) { return a.v >= b.v ? a : b; }
__device__ __forceinline__ D4 dmin(D4 a, D4 b) { return a.v <= b.v ? a : b; }
__device__ __forceinline__ D4 dclamp0(D4 a) { return a.v > 0.f ? a : dconst(0.f); }
__device__ __forceinline__ D4 datan(D4 a) { D4 r; r.v = atanf(a.v); const float g = 1.f / (1.f + a.v * a.v); for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * g; return r; }

// CIoU of train_class.py:449-460 for one (prediction, label) pair; returns value and d/d(cx,cy,w,h).
__device__ D4 ciou_d4(const float* pr, const float* lb) {
    const D4 cx = dvar(pr[0], 0), cy = dvar(pr[1], 1), w = dvar(pr[2], 2), h = dvar(pr[3], 3);
    const D4 px = cx - dscale(w, 0.5f), py = cy - dscale(h, 0.5f);
    const float lw = lb[2], lh = lb[3], lx = lb[0] - lw / 2, ly = lb[1] - lh / 2;
    const D4 Lx = dconst(lx), Ly = dconst(ly), Lx2 = dconst(lx + lw), Ly2 = dconst(ly + lh);
    const D4 x1 = dmax(px, Lx), y1 = dmax(py, Ly), x2 = dmin(px + w, Lx2), y2 = dmin(py + h, Ly2);
    const D4 inter = dclamp0(x2 - x1) * dclamp0(y2 - y1);
    const D4 uni = w * h + dconst(lw * lh) - inter;
    const D4 iou = inter / uni;
    const D4 ex1 = dmin(px, Lx), ey1 = dmin(py, Ly), ex2 = dmax(px + w, Lx2), ey2 = dmax(py + h, Ly2);
    const D4 dxl = px - Lx, dyl = py - Ly, ew = ex2 - ex1, eh = ey2 - ey1;
    const D4 dist = (dxl * dxl + dyl * dyl) / (ew * ew + eh * eh);
    const D4 da = dconst(atanf(lw / lh)) - datan(w / h);
    const D4 v = dscale(da * da, 4.0f / (3.1415926f * 3.1415926f));
    const float alpha = v.v / (v.v + 1.0f - iou.v + 0.00001f);   // no gradient through alpha
    return iou - dist - dscale(v, alpha);
}

struct LossP {
    const float* pred;        // (B, R, K)
    float* dpred;             // (B, R, K), zero on entry
    const int* screen;        // (P) row index within its image
    const int* screen_img;    // (P) image index
    const float* label;       // (P, K)
    int B, R, K, P;
    float smooth;
    double* part;             // [0..nblk) confidence partials, then [nblk..2 nblk) frame, [2 nblk..3 nblk) class
    int nblk;
    float* out;               // (4) loss, frame, confidence, class
    const int* P_dev;         // when set: positives = min(*P_dev, P) is read on the device (P = capacity)
};
__device__ __forceinline__ int loss_npos(const LossP& p) { return p.P_dev ? min(*p.P_dev, p.P) : p.P; }

__device__ __forceinline__ double block_sum(double v, double* sm) {
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    const double r = sm[0];
    __syncthreads();
    return r;
}

// Marks the positive rows: dpred[b][r][4] = 1 (a flag consumed and overwritten by the next kernel).
__global__ void loss_mark_kernel(LossP p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < loss_npos(p)) p.dpred[((size_t)p.screen_img[i] * p.R + p.screen[i]) * p.K + 4] = 1.0f;
}

// Focal objectness loss over every row (train_class.py:470-474, alpha = 0.5, gamma = 2, x100).
__global__ void __launch_bounds__(256) loss_conf_kernel(LossP p) {
    __shared__ double sm[256];
    const size_t n = (size_t)p.B * p.R;
    double acc = 0.0;
    const float gscale = 100.0f / (float)n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float x = p.pred[i * p.K + 4];
        const float y = p.dpred[i * p.K + 4] != 0.0f ? 1.0f - p.smooth : p.smooth;
        const float bce = fmaxf(x, 0.0f) - x * y + log1pf(expf(-fabsf(x)));
        const float e = expf(-bce), om = 1.0f - e;
        const float a = 0.5f * y + 0.5f * (1.0f - y);
        acc += (double)(a * om * om * bce);
        // d/dx: a * (2 om e bce + om^2) * (sigmoid(x) - y)
        const float sg = 1.0f / (1.0f + expf(-x));
        p.dpred[i * p.K + 4] = gscale * a * (2.0f * om * e * bce + om * om) * (sg - y);
    }
    const double s = block_sum(acc, sm);
    if (threadIdx.x == 0) p.part[blockIdx.x] = s;
}

// Box (CIoU) and class terms over the positive rows (train_class.py:424-425).
__global__ void __launch_bounds__(256) loss_pos_kernel(LossP p) {
    __shared__ double sm[256];
    double fa = 0.0, ca = 0.0;
    const int C = p.K - 5;
    const int P = loss_npos(p);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
        const size_t row = ((size_t)p.screen_img[i] * p.R + p.screen[i]) * p.K;
        const float* pr = p.pred + row;
        const float* lb = p.label + (size_t)i * p.K;
        float* g = p.dpred + row;
        const D4 c = ciou_d4(pr, lb);
        fa += (double)c.v;
        // a row may be listed more than once (the reference's assignment keeps duplicates): add
        for (int k = 0; k < 4; ++k) atomicAdd(&g[k], -c.d[k] / (float)P);   // frame = 1 - mean(ciou)
        if (C == 1) {                                                  // BCEWithLogitsLoss, mean over P
            const float x = pr[5], y = lb[5];
            ca += (double)(fmaxf(x, 0.0f) - x * y + log1pf(expf(-fabsf(x))));
            atomicAdd(&g[5], (1.0f / (1.0f + expf(-x)) - y) / (float)P);
        } else {                                                       // CrossEntropyLoss with soft targets
            float m = pr[5];
            for (int k = 1; k < C; ++k) m = fmaxf(m, pr[5 + k]);
            float se = 0.0f, ts = 0.0f;
            for (int k = 0; k < C; ++k) { se += expf(pr[5 + k] - m); ts += lb[5 + k]; }
            const float lse = m + logf(se);
            float l = 0.0f;
            for (int k = 0; k < C; ++k) {
                l -= lb[5 + k] * (pr[5 + k] - lse);
                atomicAdd(&g[5 + k], (expf(pr[5 + k] - lse) * ts - lb[5 + k]) / (float)P);
            }
            ca += (double)l;
        }
    }
    const double f = block_sum(fa, sm);
    const double c = block_sum(ca, sm);
    if (threadIdx.x == 0) { p.part[p.nblk + blockIdx.x] = f; p.part[2 * p.nblk + blockIdx.x] = c; }
}

// One workgroup: thread t adds the partials t, t + 256, ... and a tree adds the threads (a fixed order; one thread
// walking a thousand partials alone took 50 us on the path between the forward and the backward pass).
__global__ void __launch_bounds__(256) loss_finalize_kernel(LossP p, int nconf, int npos) {
    __shared__ double sm[256];
    double c = 0.0, f = 0.0, k = 0.0;
    for (int i = threadIdx.x; i < nconf; i += 256) c += p.part[i];
    for (int i = threadIdx.x; i < npos; i += 256) { f += p.part[p.nblk + i]; k += p.part[2 * p.nblk + i]; }
    c = block_sum(c, sm);
    f = block_sum(f, sm);
    k = block_sum(k, sm);
    if (threadIdx.x != 0) return;
    const float conf = (float)(100.0 * c / ((double)p.B * p.R));
    float frame = 1.0f, cls = 1.0f;                                     // no positives: constants
    const int P = loss_npos(p);
    if (P > 0) { frame = (float)(1.0 - f / P); cls = (float)(k / P); }
    p.out[0] = frame + conf + cls;
    p.out[1] = frame;
    p.out[2] = conf;
    p.out[3] = cls;
}

// ------------------------------------------------------------------------------------------
// Adam (coupled L2, train_class.py:161-162) + EMA (train_class.py:485-493) over flat buffers.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                float* __restrict__ ema, size_t n, float step_size, float om_beta1, float beta2, float om_beta2,
                float bc2_sqrt, float eps, float wd, float ema_decay, float om_ema_decay, float grad_scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float w = p[i];
        const float gr = g[i] * grad_scale + wd * w;
        float mm = m[i];
        mm = mm + om_beta1 * (gr - mm);                                 // exp_avg.lerp_(grad, 1 - beta1)
        const float vv = beta2 * v[i] + om_beta2 * (gr * gr);
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        w = w - step_size * (mm / denom);
        m[i] = mm;
        v[i] = vv;
        p[i] = w;
        if (ema) ema[i] = ema_decay * ema[i] + om_ema_decay * w;
    }
}

__global__ void __launch_bounds__(256) ema_only_kernel(float* __restrict__ ema, const float* __restrict__ src, size_t n, float d, float omd) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        ema[i] = d * ema[i] + omd * src[i];
}

// ------------------------------------------------------------------------------------------
// Weight packing: float32 (cout, cin, k, k) master weights -> forward layout [cout][ky][kx][cin_phys]
// and data-gradient layout [cin][ky'][kx'][cout_phys] with the taps flipped, in the compute dtype.
// One launch covers every convolution of the network through a descriptor table.
// ------------------------------------------------------------------------------------------
struct PackDesc {
    const float* w;       // source
    void* fwd;            // destination, forward layout (may be null)
    void* bwd;            // destination, data-gradient layout (may be null)
    int cout, cin, k, cin_phys, cout_phys;
    int n_seg;            // cin is spread over segments of the consumer's input buffer
    int seg_off[8], seg_len[8];
    int fwd_f32;          // 1: the forward copy stays float32 whatever the dtype (first-layer kernel)
    int fwd_frag;         // 0: plain [cout][tap][cin_phys]; else groups | chunk_bytes << 8 of odamd_conv_weight_layout (bf16)
    int bwd_frag;         // same for the data-gradient copy (plain: [cin_phys][flipped tap][cout_phys])
    int reserved;
};

// Index of element (row, tap, column) in the fragment-ordered layout [rows / 32G][chunk][G][tap][k-step][2][32][8]
// (rows = output channels of the launch, columns = its input channels, `cols` of them).
__device__ __forceinline__ size_t frag_index(int frag, int row, int tap, int col, int taps, int cols) {
    const int groups = frag & 255, chunk = frag >> 8;
    const int ksn = chunk / 32, nch = cols * 2 / chunk, cpc = chunk / 2;
    const int cb = row / (32 * groups), g = (row >> 5) % groups, r = row & 31;
    const int ch = col / cpc, s = (col % cpc) >> 4, half = (col >> 3) & 1, e = col & 7;
    return ((((((size_t)cb * nch + ch) * groups + g) * taps + tap) * ksn + s) * 2 + half) * 256 + r * 8 + e;
}

// One destination of one layer.  A workgroup takes whole destination rows, a thread a column of the row, and walks
// the taps innermost: the index arithmetic (divisions by the layer's channel and chunk counts, the segment search)
// happens once per (row, column) instead of once per element, a thread reads the k*k contiguous source weights of its
// (output, input) channel pair, and the lanes of a wave write neighbouring columns.
// BWD: rows are physical input channels, columns output channels, taps flipped.
template <bool BWD>
__device__ __forceinline__ void pack_weights_pass(const PackDesc& d, int dtype) {
    const int kk = d.k * d.k;
    const int frag = BWD ? d.bwd_frag : d.fwd_frag;
    const int rows = BWD ? d.cin_phys : d.cout;
    const int cols = BWD ? d.cout_phys : d.cin_phys;
    char* dst = (char*)(BWD ? d.bwd : d.fwd);
    const bool f32 = dtype == ODAMD_DTYPE_F32 || (!BWD && d.fwd_f32);
    const int groups = frag & 255, chunk = frag >> 8;
    const int ksn = frag ? chunk / 32 : 1, nch = frag ? cols * 2 / chunk : 1, cpc = frag ? chunk / 2 : 1;
    // physical -> logical input channel, -1 for padding.  All eight slots with constant indices: a loop to n_seg
    // indexes the descriptor dynamically, which puts every thread's copy of it into private memory.
    auto logical = [&](int phys) {
        int first = 0, ci = -1;
#pragma unroll
        for (int sgm = 0; sgm < 8; ++sgm) {
            const int off = d.seg_off[sgm], len = sgm < d.n_seg ? d.seg_len[sgm] : 0;
            if (ci < 0 && phys >= off && phys < off + len) ci = first + phys - off;
            first += len;
        }
        return ci;
    };
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int row_ci = BWD ? logical(row) : 0;
        if (BWD && row_ci < 0) continue;             // padding stays zero
        for (int col = threadIdx.x; col < cols; col += 256) {
            const int co = BWD ? col : row;
            const int ci = BWD ? row_ci : logical(col);
            if (co >= d.cout || ci < 0) continue;
            size_t base, tap_stride;
            if (frag) {                              // [rows / 32G][chunk][G][tap][k-step][2][32][8], see frag_index
                const int cb = row / (32 * groups), g = (row >> 5) % groups, r = row & 31;
                const int ch = col / cpc, ks = (col - ch * cpc) >> 4, half = (col >> 3) & 1, e = col & 7;
                tap_stride = (size_t)ksn * 512;
                base = (((size_t)cb * nch + ch) * groups + g) * kk * tap_stride + (size_t)(ks * 2 + half) * 256 + r * 8 + e;
            } else {                                 // [row][tap][col]
                tap_stride = (size_t)cols;
                base = (size_t)row * kk * cols + col;
            }
            const float* src = d.w + ((size_t)co * d.cin + ci) * kk;
            for (int tap = 0; tap < kk; ++tap) {
                const float v = src[BWD ? kk - 1 - tap : tap];
                const size_t o = base + (size_t)tap * tap_stride;
                if (f32) ((float*)dst)[o] = v; else ((uint16_t*)dst)[o] = f32_to_bf16(v);
            }
        }
    }
}

__global__ void __launch_bounds__(256) pack_weights_kernel(const PackDesc* __restrict__ table, int dtype) {
    const PackDesc& d = table[blockIdx.y];           // uniform: scalar loads, no per-thread copy
    if (d.fwd) pack_weights_pass<false>(d, dtype);
    if (d.bwd) pack_weights_pass<true>(d, dtype);
}

int red_geometry(int C, int dtype, int* vecs, int* rows) {
    const int per = dtype == ODAMD_DTYPE_F32 ? 4 : 8;
    if (C < 1) return ODAMD_E_ARG;
    *vecs = (C + per - 1) / per;          // the tensors hold whole 16-byte vectors; a ragged tail is zero padding
    if (*vecs > 256) return ODAMD_E_SHAPE;
    *rows = 256 / *vecs;
    return ODAMD_OK;
}

int red_blocks(size_t npix, int rows) {
    size_t b = (npix + (size_t)rows * 16 - 1) / ((size_t)rows * 16);
    if (b > RED_MAX_BLOCKS) b = RED_MAX_BLOCKS;
    if (b < 1) b = 1;
    return (int)b;
}

// grid of the (vector, row) elementwise kernels: about eight pixels per thread, at most 4096 workgroups
int ew_rows_grid(size_t npix, int rows) {
    size_t g = (npix + (size_t)rows * 8 - 1) / ((size_t)rows * 8);
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

int ew_grid(size_t n) {
    size_t g = (n + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

namespace {
struct WgradPlan {
    WgradP p;
    int wco, wci, r;
    size_t lds;
    int grid, nslices;
    bool first;            // Cin <= 3, 3x3 stride 1: im2col kernel
};

// Tile shape, patch shape and split count of one weight-gradient launch.
int wgrad_plan(int batch, int h, int w, int cin, int cout, int ksize, int stride, int dtype, WgradPlan& o) {
    if (batch < 1 || h < 1 || w < 1 || cin < 1 || cout < 1) return ODAMD_E_ARG;
    if (!((ksize == 1 && stride == 1) || (ksize == 3 && (stride == 1 || stride == 2)))) return ODAMD_E_ARG;
    WgradP& p = o.p;
    p = {};
    p.B = batch; p.H = h; p.W = w; p.Cin = cin; p.Cout = cout;
    if (ksize == 1) {          // no spatial structure: the pixels of the whole batch are one row
        const size_t flat = (size_t)batch * h * w;
        if (flat >= ((size_t)1 << 30)) return ODAMD_E_SHAPE;
        p.B = batch = 1; p.H = h = 1; p.W = w = (int)flat;
    }
    const int pad = (ksize - 1) / 2, taps = ksize * ksize;
    p.Ho = (h + 2 * pad - ksize) / stride + 1;
    p.Wo = (w + 2 * pad - ksize) / stride + 1;
    o.first = ksize == 3 && stride == 1 && cin <= 3;
    if (o.first) { o.r = 1; o.wco = 1; o.wci = 1; }
    else if (ksize == 3) { o.r = 1; o.wco = cout > 32 ? 2 : 1; o.wci = cin > 32 ? 2 : 1; }
    else if (cout <= 32 && cin <= 32) { o.r = 1; o.wco = 1; o.wci = 1; }
    else { o.r = 2; o.wco = cout > 64 ? 2 : 1; o.wci = cin > 64 ? 2 : 1; }
    const int co_t = 32 * o.wco * o.r, ci_t = o.first ? 8 : 32 * o.wci * o.r;
    // the pixel count of a patch is padded to whole K steps (16 pixels) for every pixel group of waves
    const int es = dtype == ODAMD_DTYPE_F32 ? 4 : 2;
    const int round = o.first ? (dtype == ODAMD_DTYPE_F32 ? 8 : 64) : 16 * (4 / (o.wco * o.wci));
    const int maxpix = dtype == ODAMD_DTYPE_F32 ? 128 : 256;
    const size_t lds_target = 80 * 1024;          // two workgroups per CU
    auto lds_bytes = [&]() -> size_t {
        if (o.first) return ((size_t)p.npix * co_t + ((size_t)p.IH * p.IW + 1) * ci_t) * es + (size_t)p.npix * sizeof(int);
        // bf16 staging with padded rows; float32 inputs are staged as (hi, lo) bf16 pairs
        return ((size_t)p.npix * wgrad_row_bytes(co_t, 1) + ((size_t)p.IH * p.IW + 1) * wgrad_row_bytes(ci_t, stride)) * (es / 2) +
               (size_t)p.npix * sizeof(int);
    };
    // a patch whose pieces fit one staging batch is prefetched while the previous one is multiplied
    const int epp = dtype == ODAMD_DTYPE_F32 ? 4 : 8;
    const size_t one_batch = (size_t)256 * wgrad_batch(ksize, dtype == ODAMD_DTYPE_F32);
    auto pieces = [&]() -> size_t { return (size_t)p.TH * p.TW * (co_t / epp) + (size_t)p.IH * p.IW * (ci_t / epp); };
    if (ksize == 1) {
        // one row of pixels: a patch is a run of TW of them, as many as fit
        p.TH = 1; p.IH = 1;
        for (p.TW = maxpix;; p.TW -= round) {
            p.npix = p.TW; p.IW = p.TW;
            o.lds = lds_bytes();
            if ((o.lds <= lds_target && pieces() <= one_batch) || p.TW <= round) break;
        }
    } else {
        // patch width: the largest divisor of the row length in [8, 40], else 32 with guarded edges
        p.TW = 0;
        for (int d = p.Wo < 40 ? p.Wo : 40; d >= 8; --d)
            if (p.Wo % d == 0) { p.TW = d; break; }
        if (p.TW == 0) p.TW = p.Wo < 32 ? p.Wo : 32;
        // patch height: as many rows as fit, at most 128 pixels (float32) or 256 (bf16)
        for (p.TH = (maxpix / p.TW > 1 ? maxpix / p.TW : 1);; --p.TH) {
            if (p.TH > p.Ho) continue;
            p.npix = (p.TH * p.TW + round - 1) / round * round;
            p.IH = (p.TH - 1) * stride + ksize;
            p.IW = (p.TW - 1) * stride + ksize;
            o.lds = lds_bytes();
            if ((o.lds <= lds_target && (o.first || pieces() <= one_batch)) || p.TH == 1) break;
        }
    }
    p.inv_tw = 1.0f / (float)p.TW; p.inv_iw = 1.0f / (float)p.IW;
    if (!o.first && o.lds < 32 * 1024) o.lds = 32 * 1024;          // the waves' end-of-kernel reduction area
    if (o.lds > 160 * 1024) return ODAMD_E_SHAPE;
    p.tiles_x = (p.Wo + p.TW - 1) / p.TW;
    p.tiles_y = (p.Ho + p.TH - 1) / p.TH;
    p.ntiles = batch * p.tiles_x * p.tiles_y;
    p.ncin = o.first ? 1 : (cin + ci_t - 1) / ci_t;
    p.npairs = p.ncin * ((cout + co_t - 1) / co_t);
    p.slice = (size_t)cout * cin * taps;
    // about two workgroups per CU; the first-layer kernel stages ~13 KB per patch and needs eight per CU to keep
    // enough loads in flight
    int nmax = ((o.first ? 2048 : 512) + p.npairs - 1) / p.npairs;
    nmax = (nmax + 7) & ~7;
    if (nmax > p.ntiles) nmax = p.ntiles;
    int nsplit = nmax;
    if (!o.first) {
        // Splits trade patches per workgroup (serial: stage, barrier, multiply) against slices (written once and read
        // once by the fold): pick the multiple of 8 with the smallest estimate, in microseconds.
        const double staged_kb = ((double)p.npix * co_t + (double)p.IH * p.IW * ci_t) * es / 1024.0;
        const double t_patch = 1.0 + staged_kb / 9.0;      // one workgroup's patch on a CU it shares with another
        double best = 1e30;
        for (int n = 8; n <= nmax; n += 8) {
            const double serial = (double)((p.ntiles + n - 1) / n) * t_patch * (p.npairs * n > 256 ? 1.0 : 0.6);
            const double slices = 2.0 * n * (double)p.slice * sizeof(float) / 6.0e6 + 0.01 * n;
            if (serial + slices < best) { best = serial + slices; nsplit = n; }
        }
    }
    p.nsplit = nsplit;
    o.grid = 8 * p.npairs * ((nsplit + 7) / 8);
    o.nslices = o.first ? nsplit * 4 : nsplit;
    return ODAMD_OK;
}

template <int DT, int KS, int ST>
void wgrad_launch(const WgradPlan& o, hipStream_t st) {
    const int key = o.wco * 100 + o.wci * 10 + o.r;
#define ODAMD_WG(WCO, WCI, R)                                                                                     \
    case WCO * 100 + WCI * 10 + R: {                                                                              \
        auto k = conv_wgrad_bf16_kernel<KS, ST, WCO, WCI, R, DT == ODAMD_DTYPE_F32>;                               \
        (void)odamd_ensure_lds((const void*)k, o.lds);                                                            \
        k<<<o.grid, 256, o.lds, st>>>(o.p);                                                                       \
        break;                                                                                                    \
    }
    switch (key) {
        ODAMD_WG(1, 1, 1) ODAMD_WG(2, 1, 1) ODAMD_WG(1, 2, 1) ODAMD_WG(2, 2, 1)
        default:
            if constexpr (KS == 1) {
                switch (key) {
                    ODAMD_WG(1, 1, 2) ODAMD_WG(2, 1, 2) ODAMD_WG(1, 2, 2) ODAMD_WG(2, 2, 2)
                    default: break;
                }
            }
            break;
    }
#undef ODAMD_WG
}
}  // namespace

extern "C" {

size_t odamd_bn_workspace_bytes(int channels) {
    const int padded = (channels + 7) & ~7;       // the partial table is laid out for whole 16-byte vectors
    return ((size_t)RED_MAX_BLOCKS * 2 + 1) * padded * sizeof(double);
}

int odamd_bn_train_forward(const void* y, int y_ld, void* out, int out_ld, size_t npix, int channels, const float* gamma,
                           const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                           float* mean, float* invstd, float* scale, float* shift, int dtype, void* workspace,
                           size_t workspace_bytes, void* stream) {
    if (!y || !out || !gamma || !beta || !mean || !invstd || !scale || !shift || !workspace || npix < 1) return ODAMD_E_ARG;
    if (dtype != ODAMD_DTYPE_F32 && dtype != ODAMD_DTYPE_BF16) return ODAMD_E_ARG;
    if (workspace_bytes < odamd_bn_workspace_bytes(channels)) return ODAMD_E_WORKSPACE;
    int vecs, rows, rc;
    if ((rc = red_geometry(channels, dtype, &vecs, &rows))) return rc;
    if (((uintptr_t)y & 15) || ((uintptr_t)out & 15)) return ODAMD_E_ALIGN;
    const int nblk = red_blocks(npix, rows);
    hipStream_t st = (hipStream_t)stream;
    const int per = dtype == ODAMD_DTYPE_F32 ? 4 : 8;
    const size_t lds = (size_t)rows * vecs * 2 * per * sizeof(float) + 256 * sizeof(double);
    double* part = (double*)workspace;
    if (dtype == ODAMD_DTYPE_F32)
        channel_reduce_kernel<ODAMD_DTYPE_F32, 0><<<nblk, vecs * rows, lds, st>>>((const char*)y, y_ld, nullptr, 0, npix, channels, vecs, rows, nullptr, nullptr, nullptr, nullptr, part);
    else
        channel_reduce_kernel<ODAMD_DTYPE_BF16, 0><<<nblk, vecs * rows, lds, st>>>((const char*)y, y_ld, nullptr, 0, npix, channels, vecs, rows, nullptr, nullptr, nullptr, nullptr, part);
    bn_stats_finalize_kernel<<<(channels + 3) / 4, 256, 0, st>>>(part, nblk, channels, vecs * per, (double)npix, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
    if (dtype == ODAMD_DTYPE_F32)
        bn_apply_silu_kernel<ODAMD_DTYPE_F32><<<ew_rows_grid(npix, rows), vecs * rows, 0, st>>>((const char*)y, y_ld, (char*)out, out_ld, npix, vecs, rows, scale, shift, mean, channels);
    else
        bn_apply_silu_kernel<ODAMD_DTYPE_BF16><<<ew_rows_grid(npix, rows), vecs * rows, 0, st>>>((const char*)y, y_ld, (char*)out, out_ld, npix, vecs, rows, scale, shift, mean, channels);
    return (int)hipGetLastError();
}

int odamd_bn_silu_backward(void* y_dy, int y_ld, const void* dact, int dact_ld, size_t npix, int channels,
                           const float* scale, const float* shift, const float* mean, const float* invstd, float* dgamma,
                           float* dbeta, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (!y_dy || !dact || !scale || !shift || !mean || !invstd || !dgamma || !dbeta || !workspace || npix < 1) return ODAMD_E_ARG;
    if (dtype != ODAMD_DTYPE_F32 && dtype != ODAMD_DTYPE_BF16) return ODAMD_E_ARG;
    if (workspace_bytes < odamd_bn_workspace_bytes(channels) + 2 * channels * sizeof(float)) return ODAMD_E_WORKSPACE;
    int vecs, rows, rc;
    if ((rc = red_geometry(channels, dtype, &vecs, &rows))) return rc;
    const int nblk = red_blocks(npix, rows);
    hipStream_t st = (hipStream_t)stream;
    const int per = dtype == ODAMD_DTYPE_F32 ? 4 : 8;
    const size_t lds = (size_t)rows * vecs * 2 * per * sizeof(float) + 256 * sizeof(double);
    double* part = (double*)workspace;
    float* mean_dz = (float*)((char*)workspace + odamd_bn_workspace_bytes(channels));
    float* mean_dzx = mean_dz + channels;
    if (dtype == ODAMD_DTYPE_F32)
        channel_reduce_kernel<ODAMD_DTYPE_F32, 1><<<nblk, vecs * rows, lds, st>>>((const char*)y_dy, y_ld, (const char*)dact, dact_ld, npix, channels, vecs, rows, scale, shift, mean, invstd, part);
    else
        channel_reduce_kernel<ODAMD_DTYPE_BF16, 1><<<nblk, vecs * rows, lds, st>>>((const char*)y_dy, y_ld, (const char*)dact, dact_ld, npix, channels, vecs, rows, scale, shift, mean, invstd, part);
    bn_bwd_finalize_kernel<<<(channels + 3) / 4, 256, 0, st>>>(part, nblk, channels, vecs * per, (double)npix, dgamma, dbeta, mean_dz, mean_dzx);
    if (dtype == ODAMD_DTYPE_F32)
        bn_bwd_apply_kernel<ODAMD_DTYPE_F32><<<ew_rows_grid(npix, rows), vecs * rows, 0, st>>>((char*)y_dy, y_ld, (const char*)dact, dact_ld, npix, vecs, rows, scale, shift, mean, invstd, mean_dz, mean_dzx, channels);
    else
        bn_bwd_apply_kernel<ODAMD_DTYPE_BF16><<<ew_rows_grid(npix, rows), vecs * rows, 0, st>>>((char*)y_dy, y_ld, (const char*)dact, dact_ld, npix, vecs, rows, scale, shift, mean, invstd, mean_dz, mean_dzx, channels);
    return (int)hipGetLastError();
}

int odamd_bias_grad(const void* dy, int dy_ld, size_t npix, int channels, float* dbias, int dtype, void* workspace,
                    size_t workspace_bytes, void* stream) {
    if (!dy || !dbias || !workspace || npix < 1) return ODAMD_E_ARG;
    if (workspace_bytes < odamd_bn_workspace_bytes(channels)) return ODAMD_E_WORKSPACE;
    int vecs, rows, rc;
    if ((rc = red_geometry(channels, dtype, &vecs, &rows))) return rc;
    const int nblk = red_blocks(npix, rows);
    hipStream_t st = (hipStream_t)stream;
    const int per = dtype == ODAMD_DTYPE_F32 ? 4 : 8;
    const size_t lds = (size_t)rows * vecs * 2 * per * sizeof(float) + 256 * sizeof(double);
    double* part = (double*)workspace;
    if (dtype == ODAMD_DTYPE_F32)
        channel_reduce_kernel<ODAMD_DTYPE_F32, 2><<<nblk, vecs * rows, lds, st>>>((const char*)dy, dy_ld, nullptr, 0, npix, channels, vecs, rows, nullptr, nullptr, nullptr, nullptr, part);
    else if (dtype == ODAMD_DTYPE_BF16)
        channel_reduce_kernel<ODAMD_DTYPE_BF16, 2><<<nblk, vecs * rows, lds, st>>>((const char*)dy, dy_ld, nullptr, 0, npix, channels, vecs, rows, nullptr, nullptr, nullptr, nullptr, part);
    else
        return ODAMD_E_ARG;
    bias_grad_finalize_kernel<<<(channels + 3) / 4, 256, 0, st>>>(part, nblk, channels, vecs * per, dbias);
    return (int)hipGetLastError();
}

int odamd_maxpool2_backward(const void* x, int x_ld, const void* dy, int dy_ld, void* dx, int dx_ld, int batch, int h,
                            int w, int c, int dtype, void* stream) {
    if (!x || !dy || !dx || batch < 1 || h < 2 || w < 2 || c < 1) return ODAMD_E_ARG;
    const int per = dtype == ODAMD_DTYPE_F32 ? 4 : 8;
    if (dtype != ODAMD_DTYPE_F32 && dtype != ODAMD_DTYPE_BF16) return ODAMD_E_ARG;
    if ((c % per) || (x_ld % per) || (dy_ld % per) || (dx_ld % per) || ((uintptr_t)x & 15) || ((uintptr_t)dy & 15) || ((uintptr_t)dx & 15))
        return ODAMD_E_ALIGN;             // the kernel moves whole 16-byte channel vectors
    const int vecs = c / per;
    const size_t n = (size_t)batch * (h / 2) * (w / 2) * vecs;
    if (dtype == ODAMD_DTYPE_F32)
        maxpool2_bwd_kernel<ODAMD_DTYPE_F32><<<ew_grid(n), 256, 0, (hipStream_t)stream>>>((const char*)x, x_ld, (const char*)dy, dy_ld, (char*)dx, dx_ld, batch, h, w, vecs);
    else
        maxpool2_bwd_kernel<ODAMD_DTYPE_BF16><<<ew_grid(n), 256, 0, (hipStream_t)stream>>>((const char*)x, x_ld, (const char*)dy, dy_ld, (char*)dx, dx_ld, batch, h, w, vecs);
    return (int)hipGetLastError();
}

int odamd_maxpool_s1_backward(const void* x, int x_ld, const void* dy, int dy_ld, void* dx, int dx_ld, int batch, int h,
                              int w, int c, int ksize, int dtype, void* stream) {
    if (!x || !dy || !dx || batch < 1 || h < 1 || w < 1 || c < 1 || ksize < 1 || !(ksize & 1)) return ODAMD_E_ARG;
    if (dtype == ODAMD_DTYPE_F32) {
        maxpool_s1_bwd_kernel<<<ew_grid((size_t)batch * h * w * c), 256, 0, (hipStream_t)stream>>>(
            (const float*)x, x_ld, (const float*)dy, dy_ld, (float*)dx, dx_ld, batch, h, w, c, ksize);
    } else if (dtype == ODAMD_DTYPE_BF16) {
        if ((c & 7) || (x_ld & 7) || (dy_ld & 7) || (dx_ld & 7) || ((uintptr_t)x & 15) || ((uintptr_t)dy & 15) || ((uintptr_t)dx & 15))
            return ODAMD_E_ALIGN;
        const size_t lds = (size_t)h * w * 16 * sizeof(float);
        if (lds > 64 * 1024) return ODAMD_E_SHAPE;     // planes of up to 1024 pixels (the 20x20 .. 32x32 SPP maps)
        maxpool_s1_bwd_bf16_kernel<<<dim3(c / 8, batch), 256, lds, (hipStream_t)stream>>>(
            (const char*)x, x_ld, (const char*)dy, dy_ld, (char*)dx, dx_ld, h, w, ksize);
    } else {
        return ODAMD_E_ARG;
    }
    return (int)hipGetLastError();
}

int odamd_spp_backward(const void* x, int x_ld, const void* dy5, const void* dy9, const void* dy13, int dy_ld, void* dx,
                       int dx_ld, int batch, int h, int w, int c, int dtype, void* stream) {
    if (!x || !dy5 || !dy9 || !dy13 || !dx || batch < 1 || h < 1 || w < 1 || c < 1) return ODAMD_E_ARG;
    if (dtype != ODAMD_DTYPE_F32 && dtype != ODAMD_DTYPE_BF16) return ODAMD_E_ARG;
    const int per = dtype == ODAMD_DTYPE_F32 ? 4 : 8;
    const bool aligned = !((c % per) || (x_ld % per) || (dy_ld % per) || (dx_ld % per) || ((uintptr_t)x & 15) || ((uintptr_t)dy5 & 15) ||
                           ((uintptr_t)dy9 & 15) || ((uintptr_t)dy13 & 15) || ((uintptr_t)dx & 15));
    // one launch, no atomics, bitwise repeatable: planes whose per-element records fit the LDS of a compute unit
    // (up to about 32 x 32 pixels: the SPP map of inputs up to 1024 x 1024)
    const size_t lds = (size_t)h * w * (dtype == ODAMD_DTYPE_F32 ? spp_gather_bytes_per_pixel<ODAMD_DTYPE_F32>()
                                                                  : spp_gather_bytes_per_pixel<ODAMD_DTYPE_BF16>());
    if (aligned && w <= 255 && (size_t)h * w <= 65535 && lds <= 160 * 1024 && batch <= 65535) {
        const dim3 grid(c / per, batch);
        if (dtype == ODAMD_DTYPE_F32) {
            constexpr int NT = 512;
            auto k = spp_bwd_gather_kernel<ODAMD_DTYPE_F32, NT>;
            if (odamd_ensure_lds((const void*)k, lds) != hipSuccess) return (int)hipGetLastError();
            k<<<grid, NT, lds, (hipStream_t)stream>>>((const char*)x, x_ld, (const char*)dy5, (const char*)dy9, (const char*)dy13, dy_ld,
                                                        (char*)dx, dx_ld, h, w);
        } else {
            constexpr int NT = 1024;
            auto k = spp_bwd_gather_kernel<ODAMD_DTYPE_BF16, NT>;
            if (odamd_ensure_lds((const void*)k, lds) != hipSuccess) return (int)hipGetLastError();
            k<<<grid, NT, lds, (hipStream_t)stream>>>((const char*)x, x_ld, (const char*)dy5, (const char*)dy9, (const char*)dy13, dy_ld,
                                                        (char*)dx, dx_ld, h, w);
        }
        return (int)hipGetLastError();
    }
    // larger planes (or unaligned tensors): the three pools one after the other with float atomics (the sums are then
    // repeatable only up to the order of the additions)
    const void* dys[3] = {dy5, dy9, dy13};
    const int ks[3] = {5, 9, 13};
    for (int i = 0; i < 3; ++i) {
        const int rc = odamd_maxpool_s1_backward(x, x_ld, dys[i], dy_ld, dx, dx_ld, batch, h, w, c, ks[i], dtype, stream);
        if (rc) return rc;
    }
    return ODAMD_OK;
}

int odamd_upsample2_backward(const void* dy, int dy_ld, void* dx, int dx_ld, int batch, int h, int w, int c, int dtype,
                             void* stream) {
    if (!dy || !dx || batch < 1 || h < 1 || w < 1 || c < 1) return ODAMD_E_ARG;
    if (dtype != ODAMD_DTYPE_F32 && dtype != ODAMD_DTYPE_BF16) return ODAMD_E_ARG;
    const int per = dtype == ODAMD_DTYPE_F32 ? 4 : 8;
    if ((c % per) || (dy_ld % per) || (dx_ld % per) || ((uintptr_t)dy & 15) || ((uintptr_t)dx & 15)) return ODAMD_E_ALIGN;
    const int vecs = c / per;
    const size_t n = (size_t)batch * h * w * vecs;
    if (dtype == ODAMD_DTYPE_F32)
        upsample2_bwd_kernel<ODAMD_DTYPE_F32><<<ew_grid(n), 256, 0, (hipStream_t)stream>>>((const char*)dy, dy_ld, (char*)dx, dx_ld, batch, h, w, vecs);
    else
        upsample2_bwd_kernel<ODAMD_DTYPE_BF16><<<ew_grid(n), 256, 0, (hipStream_t)stream>>>((const char*)dy, dy_ld, (char*)dx, dx_ld, batch, h, w, vecs);
    return (int)hipGetLastError();
}

size_t odamd_conv_wgrad_workspace_bytes(int batch, int h, int w, int cin, int cout, int ksize, int stride, int dtype) {
    WgradPlan o;
    if (dtype != ODAMD_DTYPE_F32 && dtype != ODAMD_DTYPE_BF16) return 0;
    if (wgrad_plan(batch, h, w, cin, cout, ksize, stride, dtype, o) != ODAMD_OK) return 0;
    return (size_t)o.nslices * o.p.slice * sizeof(float);
}

int odamd_conv_wgrad(const void* x, int x_ld, const void* dy, int dy_ld, float* dw, int batch, int h, int w, int cin,
                     int cout, int ksize, int stride, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !dy || !dw || !workspace) return ODAMD_E_ARG;
    if (dtype != ODAMD_DTYPE_F32 && dtype != ODAMD_DTYPE_BF16) return ODAMD_E_ARG;
    const int per = dtype == ODAMD_DTYPE_F32 ? 4 : 8;    // elements per 16-byte piece
    if (((uintptr_t)x & 15) || ((uintptr_t)dy & 15) || ((uintptr_t)workspace & 15) || (x_ld % per) || (dy_ld % per)) return ODAMD_E_ALIGN;
    WgradPlan o;
    const int rc = wgrad_plan(batch, h, w, cin, cout, ksize, stride, dtype, o);
    if (rc != ODAMD_OK) return rc;
    if (workspace_bytes < (size_t)o.nslices * o.p.slice * sizeof(float)) return ODAMD_E_WORKSPACE;
    // the 1x1 kernels form element offsets inside the whole flat pixel row in 32 bits
    if (ksize == 1 && ((size_t)o.p.W * (size_t)x_ld >= ((size_t)1 << 32) || (size_t)o.p.Wo * (size_t)dy_ld >= ((size_t)1 << 32))) return ODAMD_E_SHAPE;
    o.p.x = x; o.p.dy = dy; o.p.part = (float*)workspace;
    o.p.x_ld = x_ld; o.p.dy_ld = dy_ld;
    hipStream_t st = (hipStream_t)stream;
    if (o.first) {
        if ((x_ld * (dtype == ODAMD_DTYPE_F32 ? 4 : 2)) % 16 || x_ld < 8) return ODAMD_E_ALIGN;   // image padded to 8 channels
        if (dtype == ODAMD_DTYPE_F32) conv_wgrad_first_kernel<ODAMD_DTYPE_F32><<<o.grid, 256, o.lds, st>>>(o.p);
        else conv_wgrad_first_kernel<ODAMD_DTYPE_BF16><<<o.grid, 256, o.lds, st>>>(o.p);
    } else if (dtype == ODAMD_DTYPE_F32) {
        if (ksize == 1) wgrad_launch<ODAMD_DTYPE_F32, 1, 1>(o, st);
        else if (stride == 1) wgrad_launch<ODAMD_DTYPE_F32, 3, 1>(o, st);
        else wgrad_launch<ODAMD_DTYPE_F32, 3, 2>(o, st);
    } else {
        if (ksize == 1) wgrad_launch<ODAMD_DTYPE_BF16, 1, 1>(o, st);
        else if (stride == 1) wgrad_launch<ODAMD_DTYPE_BF16, 3, 1>(o, st);
        else wgrad_launch<ODAMD_DTYPE_BF16, 3, 2>(o, st);
    }
    int err = (int)hipGetLastError();
    if (err) return err;
    const int taps = o.first ? 1 : ksize * ksize, cc = o.first ? (int)o.p.slice : cout * cin;   // the first-layer kernel writes the torch order
    if (o.nslices >= 256)
        wgrad_fold_kernel<32><<<(unsigned)((o.p.slice + 31) / 32), 1024, 0, st>>>(o.p.part, o.p.slice, o.nslices, dw, taps, cc);
    else
        wgrad_fold_kernel<8><<<(unsigned)((o.p.slice + 31) / 32), 256, 0, st>>>(o.p.part, o.p.slice, o.nslices, dw, taps, cc);
    return (int)hipGetLastError();
}

size_t odamd_loss_workspace_bytes(void) { return (size_t)3 * 1024 * sizeof(double); }

static int loss_launch(const float* pred, float* dpred, const int* screen, const int* screen_img, const float* label,
                       int batch, int rows, int kdim, int n_pos, const int* n_pos_dev, float label_smooth, float* out4,
                       void* workspace, size_t workspace_bytes, void* stream) {
    if (!pred || !dpred || !out4 || !workspace || batch < 1 || rows < 1 || kdim < 6 || n_pos < 0) return ODAMD_E_ARG;
    if (n_pos > 0 && (!screen || !screen_img || !label)) return ODAMD_E_ARG;
    if (workspace_bytes < odamd_loss_workspace_bytes()) return ODAMD_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    LossP p = {pred, dpred, screen, screen_img, label, batch, rows, kdim, n_pos, label_smooth, (double*)workspace, 1024, out4,
               n_pos_dev};
    int rc = (int)hipMemsetAsync(dpred, 0, (size_t)batch * rows * kdim * sizeof(float), st);
    if (rc) return rc;
    if (n_pos > 0) loss_mark_kernel<<<(n_pos + 255) / 256, 256, 0, st>>>(p);
    size_t nb = ((size_t)batch * rows + 255) / 256;
    const int nconf = (int)(nb > 1024 ? 1024 : nb);
    loss_conf_kernel<<<nconf, 256, 0, st>>>(p);
    int npos = 0;
    if (n_pos > 0) {
        npos = (n_pos + 255) / 256;
        if (npos > 1024) npos = 1024;
        loss_pos_kernel<<<npos, 256, 0, st>>>(p);
    }
    loss_finalize_kernel<<<1, 256, 0, st>>>(p, nconf, npos);
    return (int)hipGetLastError();
}

int odamd_detection_loss(const float* pred, float* dpred, const int* screen, const int* screen_img, const float* label,
                         int batch, int rows, int kdim, int n_pos, float label_smooth, float* out4, void* workspace,
                         size_t workspace_bytes, void* stream) {
    return loss_launch(pred, dpred, screen, screen_img, label, batch, rows, kdim, n_pos, nullptr, label_smooth, out4,
                       workspace, workspace_bytes, stream);
}

int odamd_detection_loss_dev(const float* pred, float* dpred, const int* screen, const int* screen_img, const float* label,
                             int batch, int rows, int kdim, const int* n_pos_dev, int capacity, float label_smooth,
                             float* out4, void* workspace, size_t workspace_bytes, void* stream) {
    if (!n_pos_dev || capacity < 1) return ODAMD_E_ARG;
    return loss_launch(pred, dpred, screen, screen_img, label, batch, rows, kdim, capacity, n_pos_dev, label_smooth, out4,
                       workspace, workspace_bytes, stream);
}

int odamd_adam_ema_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema, size_t n, double lr,
                        double beta1, double beta2, double eps, double weight_decay, int step, double ema_decay,
                        double grad_scale, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 1 || step < 1) return ODAMD_E_ARG;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    adam_ema_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, ema, n, (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps, (float)weight_decay, (float)ema_decay, (float)(1.0 - ema_decay), (float)grad_scale);
    return (int)hipGetLastError();
}

int odamd_ema_update(float* ema, const float* src, size_t n, double decay, void* stream) {
    if (!ema || !src || n < 1) return ODAMD_E_ARG;
    ema_only_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(ema, src, n, (float)decay, (float)(1.0 - decay));
    return (int)hipGetLastError();
}

int odamd_pack_weights(const void* table_dev, int n_entries, int dtype, void* stream) {
    if (!table_dev || n_entries < 1) return ODAMD_E_ARG;
    if (dtype != ODAMD_DTYPE_F32 && dtype != ODAMD_DTYPE_BF16) return ODAMD_E_ARG;
    // 256 workgroups per layer: the largest layers (2.4 M weights) bound the launch, 36 elements per thread then
    pack_weights_kernel<<<dim3(256, n_entries), 256, 0, (hipStream_t)stream>>>((const PackDesc*)table_dev, dtype);
    return (int)hipGetLastError();
}

size_t odamd_pack_desc_bytes(void) { return sizeof(PackDesc); }

}  // extern "C"

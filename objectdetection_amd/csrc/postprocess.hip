// Box decode, confidence filter, score sort and greedy NMS for gfx950.
//
// All arithmetic here is float32 with one rounding per operation (this file is compiled with
// -ffp-contract=off) so that, given identical inputs, the kept-box indices equal those of the
// numpy code in the reference (predict.py:33-41, :67-92) bit for bit.
//
// Kernels (all HBM/latency bound, no matrix work):
//   decode_kernel         heads (NCHW-flat) -> (B,R,K) decoded rows            elementwise stream
//   decode_bwd_kernel     autograd of the above
//   filter_kernel<SRC>    rows -> {row box, compacted 64-bit sort key, count}  ballot compaction
//   bitonic_*             per-image descending sort of the keys (LDS + global stages)
//   gather_sorted_kernel  sorted keys -> sorted boxes / row ids
//   iou_mask_kernel       64x64 tiles of the suppression bitmask, column boxes staged in LDS
//   nms_scan_kernel       one workgroup per image walks the mask in score order
//   emit_rows_kernel      kept rows -> (cx,cy,w,h,...) scaled to the source image
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include <stdlib.h>
#include "odamd.h"

#include <mutex>
#include <utility>

// Scratch of odamd_nms_rows_static, the entry point that allocates nothing.  Every GPU the library
// runs on holds its own copy of this array, at its own address, until the process ends.  In host
// code the name `scratch` is only a handle for hipGetSymbolAddress: the device address it returns
// (kept per device in static_scratch()) is what kernels, memsets and copies are given.
constexpr size_t STATIC_SCRATCH_BYTES = (size_t)ODAMD_STATIC_SCRATCH_MIB << 20;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

namespace {

typedef unsigned long long u64;

// Device address of `scratch` on the current device (looked up once per device).
int static_scratch(void** out) {
    constexpr int MAX_DEVICES = 64;
    static std::mutex mu;
    static void* ptr[MAX_DEVICES] = {};
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= MAX_DEVICES) return ODAMD_E_ARG;
    std::lock_guard<std::mutex> lock(mu);
    if (!ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        if (!p) return ODAMD_E_ARG;
        ptr[dev] = p;
    }
    *out = ptr[dev];
    return ODAMD_OK;
}

// 1 / (1 + e^-x): the library exponential and the hardware reciprocal (1 ulp; the IEEE division
// sequence is ten more vector instructions per value and these kernels are issue-bound without it).
// Every kernel of this file uses this one form, so the fused and the materialised paths agree bit for bit.
__device__ __forceinline__ float sigmoid_f32(float x) { return __builtin_amdgcn_rcpf(1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------
struct DecodeGeom {
    const float* head[3];
    int h[3], w[3];
    int rows_before[3];  // rows of earlier scales within one image
    float stride[3];
    float anchor[3][4][2];
    int n_anchor, kdim, rows, deploy;
};

// One decoded element.  k: channel within the row, (a, y, x): anchor and grid cell.
__device__ __forceinline__ float decode_value(const DecodeGeom& g, int s, int k, int a, int y, int x, float raw) {
    if (k < 4) {
        const float t = sigmoid_f32(raw);
        if (k == 0) return (2.0f * t - 0.5f + (float)x) * g.stride[s];
        if (k == 1) return (2.0f * t - 0.5f + (float)y) * g.stride[s];
        return (4.0f * (t * t)) * g.anchor[s][a][k - 2];
    }
    return g.deploy ? sigmoid_f32(raw) : raw;
}

// The same element from t = sigmoid_f32(raw) evaluated by the caller (once, for all lanes alike).
__device__ __forceinline__ float decode_from_sigmoid(const DecodeGeom& g, int s, int k, int a, int y, int x, float raw,
                                                     float t) {
    if (k < 4) {
        if (k == 0) return (2.0f * t - 0.5f + (float)x) * g.stride[s];
        if (k == 1) return (2.0f * t - 0.5f + (float)y) * g.stride[s];
        return (4.0f * (t * t)) * g.anchor[s][a][k - 2];
    }
    return g.deploy ? t : raw;
}

__global__ void __launch_bounds__(256) decode_kernel(DecodeGeom g, float* __restrict__ out) {
    const int s = blockIdx.y, b = blockIdx.z;
    const int hw = g.h[s] * g.w[s];
    const int n = g.n_anchor * hw * g.kdim;
    const float* src = g.head[s] + (size_t)b * n;
    float* dst = out + ((size_t)b * g.rows + g.rows_before[s]) * g.kdim;
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < n; f += gridDim.x * blockDim.x) {
        const int row = f / g.kdim, k = f - row * g.kdim;
        const int a = row / hw, cell = row - a * hw;
        const int y = cell / g.w[s], x = cell - y * g.w[s];
        dst[f] = decode_value(g, s, k, a, y, x, src[f]);
    }
}

// deploy.forward on an already decoded tensor: sigmoid of channels 4: in place.
__global__ void __launch_bounds__(256) deploy_sigmoid_kernel(float* __restrict__ pred, size_t n, int kdim) {
    for (size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x; f < n; f += (size_t)gridDim.x * blockDim.x)
        if ((int)(f % kdim) >= 4) pred[f] = sigmoid_f32(pred[f]);
}

// d raw = d out * d(decode)/d raw.  Channels >= 4 pass through (deploy == 0 in training).
__global__ void __launch_bounds__(256) decode_bwd_kernel(DecodeGeom g, const float* __restrict__ gout,
                                                         float* gh0, float* gh1, float* gh2) {
    const int s = blockIdx.y, b = blockIdx.z;
    const int hw = g.h[s] * g.w[s];
    const int n = g.n_anchor * hw * g.kdim;
    const float* src = g.head[s] + (size_t)b * n;
    const float* go = gout + ((size_t)b * g.rows + g.rows_before[s]) * g.kdim;
    float* dst = (s == 0 ? gh0 : s == 1 ? gh1 : gh2) + (size_t)b * n;
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < n; f += gridDim.x * blockDim.x) {
        const int row = f / g.kdim, k = f - row * g.kdim;
        float gv = go[f];
        if (k < 4) {
            const float t = sigmoid_f32(src[f]);
            const float dt = t * (1.0f - t);
            if (k < 2) {
                gv = gv * (2.0f * g.stride[s]) * dt;
            } else {
                const int a = row / hw;
                gv = gv * (8.0f * t * g.anchor[s][a][k - 2]) * dt;
            }
        }
        dst[f] = gv;
    }
}

// ------------------------------------------------------------------------------------------
// predict_class._image_process (predict.py:60-65): cv2.resize(image, (S, S)) [bilinear] -> float / 255
// -> (1, 3, S, S).  The resize follows OpenCV's 8-bit INTER_LINEAR scheme: source position
// (d + 0.5) * scale - 0.5, clamped taps, weights rounded to 11 fractional bits, horizontal pass in
// integers, vertical pass ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2, result a byte.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void resize_tap(int d, float scale, int n_src, int* s0, int* s1, int* w0, int* w1) {
    float f = ((float)d + 0.5f) * scale - 0.5f;
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.0f; }
    if (s >= n_src - 1) { s = n_src - 1; f = 0.0f; }
    *s0 = s;
    *s1 = s + 1 < n_src ? s + 1 : s;
    *w0 = (int)rintf((1.0f - f) * 2048.0f);
    *w1 = (int)rintf(f * 2048.0f);
}

__global__ void __launch_bounds__(256)
preprocess_u8_kernel(const uint8_t* __restrict__ img, int H0, int W0, float* __restrict__ out, int S) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= S * S) return;
    img += (size_t)blockIdx.y * H0 * W0 * 3;        // image blockIdx.y of a batch of equally sized images
    out += (size_t)blockIdx.y * 3 * S * S;
    const int dy = i / S, dx = i - dy * S;
    int x0, x1, a0, a1, y0, y1, b0, b1;
    if (W0 == S) { x0 = x1 = dx; a0 = 2048; a1 = 0; } else resize_tap(dx, (float)W0 / (float)S, W0, &x0, &x1, &a0, &a1);
    if (H0 == S) { y0 = y1 = dy; b0 = 2048; b1 = 0; } else resize_tap(dy, (float)H0 / (float)S, H0, &y0, &y1, &b0, &b1);
    const uint8_t* r0 = img + (size_t)y0 * W0 * 3;
    const uint8_t* r1 = img + (size_t)y1 * W0 * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int v;
        if (W0 == S && H0 == S) {
            v = r0[x0 * 3 + c];
        } else {
            const int h0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
            const int h1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
            v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
        out[(size_t)c * S * S + i] = (float)v / 255.0f;
    }
}

// ------------------------------------------------------------------------------------------
// Label -> grid-row assignment of torch_dataset (train_class.py:537-543, :578-607) for a batch.
// A grid row can express boxes between decode(-4) and decode(+4) of its cell and anchor; a label
// is assigned to every row whose range strictly contains its (cx, cy, w, h).  The reference's
// conflict branch never runs (its `label_set` stays empty), so a row matched by several labels is
// listed once per label and every copy carries the LAST matching label.  Output order = the
// reference's: images in order, rows ascending within an image.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void row_range(const DecodeGeom& g, int r, float t_lo, float t_hi, float* lo, float* hi) {
    int s = 0;
    if (r >= g.rows_before[2]) s = 2; else if (r >= g.rows_before[1]) s = 1;
    const int lr = r - g.rows_before[s], hw = g.h[s] * g.w[s];
    const int a = lr / hw, cell = lr - a * hw;
    const int y = cell / g.w[s], x = cell - y * g.w[s];
    lo[0] = (2.0f * t_lo - 0.5f + (float)x) * g.stride[s];  hi[0] = (2.0f * t_hi - 0.5f + (float)x) * g.stride[s];
    lo[1] = (2.0f * t_lo - 0.5f + (float)y) * g.stride[s];  hi[1] = (2.0f * t_hi - 0.5f + (float)y) * g.stride[s];
    lo[2] = (4.0f * (t_lo * t_lo)) * g.anchor[s][a][0];     hi[2] = (4.0f * (t_hi * t_hi)) * g.anchor[s][a][0];
    lo[3] = (4.0f * (t_lo * t_lo)) * g.anchor[s][a][1];     hi[3] = (4.0f * (t_hi * t_hi)) * g.anchor[s][a][1];
}

// matches of row r among the labels of its image: count and last matching label
__device__ __forceinline__ int row_matches(const float* __restrict__ lab, int nl, int kdim, const float* lo, const float* hi,
                                           int* last) {
    int cnt = 0;
    for (int l = 0; l < nl; ++l) {
        const float* q = lab + (size_t)l * kdim;
        if (lo[0] < q[0] && q[0] < hi[0] && lo[1] < q[1] && q[1] < hi[1] && lo[2] < q[2] && q[2] < hi[2] &&
            lo[3] < q[3] && q[3] < hi[3]) { ++cnt; *last = l; }
    }
    return cnt;
}

__device__ __forceinline__ int block_scan_excl(int v, int* sm, int* total) {   // 256 threads
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const int t = (int)threadIdx.x >= d ? sm[threadIdx.x - d] : 0;
        __syncthreads();
        sm[threadIdx.x] += t;
        __syncthreads();
    }
    const int incl = sm[threadIdx.x];
    *total = sm[255];
    __syncthreads();
    return incl - v;
}

// EMIT == false: totals[b] = entries of image b.  EMIT == true: writes the entries (needs totals).
template <bool EMIT>
__global__ void __launch_bounds__(256)
assign_labels_kernel(DecodeGeom g, float t_lo, float t_hi, const float* __restrict__ labels, const int* __restrict__ n_label,
                     int lmax, int* __restrict__ totals, int* __restrict__ screen, int* __restrict__ screen_img,
                     float* __restrict__ label_out, int capacity, int* __restrict__ n_pos, int batch) {
    __shared__ int sm[256];
    const int b = blockIdx.x;
    const int nl = min(n_label[b], lmax);
    const float* lab = labels + (size_t)b * lmax * g.kdim;
    int base = 0;
    if (EMIT) {
        for (int i = 0; i < b; ++i) base += totals[i];
        if (b == batch - 1 && threadIdx.x == 0) {
            const int all = base + totals[b];
            n_pos[0] = min(all, capacity);
            n_pos[1] = all;
        }
    }
    int running = 0;
    for (int r0 = 0; r0 < g.rows; r0 += 256) {
        const int r = r0 + threadIdx.x;
        int cnt = 0, last = 0;
        if (r < g.rows) {
            float lo[4], hi[4];
            row_range(g, r, t_lo, t_hi, lo, hi);
            cnt = row_matches(lab, nl, g.kdim, lo, hi, &last);
        }
        int total;
        const int off = block_scan_excl(cnt, sm, &total);
        if (EMIT) {
            for (int c = 0; c < cnt; ++c) {
                const int pos = base + running + off + c;
                if (pos < capacity) {
                    screen[pos] = r;
                    screen_img[pos] = b;
                    for (int k = 0; k < g.kdim; ++k) label_out[(size_t)pos * g.kdim + k] = lab[(size_t)last * g.kdim + k];
                }
            }
        }
        running += total;
    }
    if (!EMIT && threadIdx.x == 0) totals[b] = running;
}

// ------------------------------------------------------------------------------------------
// filter: rows -> row boxes + compacted sort keys
// ------------------------------------------------------------------------------------------
// Sort key: high word = score bits mapped so that unsigned order == float order, low word = row
// index.  A descending sort of the keys is "descending score, then descending row index".
// Key 0 never occurs for a real candidate and marks padding.
__device__ __forceinline__ u64 make_key(float score, int row) {
    if (score == 0.0f) score = 0.0f;  // -0 -> +0, numpy orders them as equal
    uint32_t u = __float_as_uint(score);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    // np.argsort puts every NaN (either sign, any payload) behind all numbers, so the reversed order
    // (predict.py:84) starts with them: one key above +inf for all of them
    if (score != score) u = 0xFFFFFFFFu;
    return ((u64)u << 32) | (uint32_t)row;
}

constexpr int FILTER_THREADS = 256;
// The per-image candidate counters the filter kernel adds to sit in separate 128-byte lines: atomics
// of different images then go to different L2 channels instead of queueing on one line.
constexpr int COUNT_STRIDE = 32;

__global__ void compact_counts_kernel(const int* __restrict__ padded, int batch, int n_pad, int* __restrict__ count) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) count[b] = padded[b * COUNT_STRIDE];
}

// Where the rows of a filter launch come from: up to three contiguous row ranges ("scales").
// SRC 0/2 use one range (the (B,R,K) tensor), SRC 1 one range per raw head.
struct FilterSrc {
    const float* base[3];   // first row of the range in image 0
    int rows_in[3];         // rows per image in the range
    int row_first[3];       // global row index of the range's first row
    int stage_first[3];     // first workgroup (blockIdx.x) of the range
};

// SRC == 0: rows are the deployed (B,R,K) tensor.  SRC == 1: rows are RAW head rows; decode and
// sigmoid are evaluated per surviving row.  SRC == 2: rows are already (x_min,y_min,w,h,..) and
// every row is a candidate (the stand-alone _nms of predict.py:82-92).  SRC == 3: the validation
// metric's candidates (train_class.py:340-342,78): (cx,cy,w,h,..) rows of the NON-deployed tensor,
// kept when the raw objectness exceeds the threshold, no positivity test; ties in the score go to
// the lower row (a stable descending sort), so the row is stored complemented in the key.
// SRC == 4: the same rule on rows that are already (x_min,y_min,w,h,..), every row a candidate
// (the stand-alone nms of train_class.py:77-83).
// A workgroup copies `rows_stage` whole rows of one range into LDS with coalesced loads (row
// stride padded to an odd number of words, so the per-row walk is bank-conflict free); then one
// lane per row evaluates predict.py:68-70,83: objectness first, so a rejected row costs one exp.
template <int SRC>
__global__ void __launch_bounds__(FILTER_THREADS)
filter_kernel(FilterSrc fs, DecodeGeom g, int rows, int kdim, int rows_stage, float conf_thr,
              float4* __restrict__ rowbox, u64* __restrict__ keys, int n_pad, int* __restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.y;
    const int bx = blockIdx.x;
    const int s = bx >= fs.stage_first[2] ? 2 : (bx >= fs.stage_first[1] ? 1 : 0);
    const int lrow0 = (bx - fs.stage_first[s]) * rows_stage;
    const int nrow = min(rows_stage, fs.rows_in[s] - lrow0);
    if (nrow <= 0) return;
    const int row0 = fs.row_first[s] + lrow0;
    const int ldk = kdim | 1;
    {   // coalesced copy into the padded rows; 16-byte loads when the range start is aligned
        const float* src = fs.base[s] + ((size_t)b * fs.rows_in[s] + lrow0) * kdim;
        const int total = nrow * kdim;
        const int nvec = (((uintptr_t)src & 15) == 0) ? total >> 2 : 0;
        // six loads per thread are issued before the first of them is used: one 16-byte load in flight per
        // thread does not cover the latency of HBM
        for (int v0 = threadIdx.x; v0 < nvec; v0 += 6 * FILTER_THREADS) {
            float4 t[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int v = v0 + q * FILTER_THREADS;
                t[q] = v < nvec ? ((const float4*)src)[v] : make_float4(0, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int v = v0 + q * FILTER_THREADS;
                if (v >= nvec) break;
                if (ldk == kdim) {   // odd row length: the LDS image is the memory image, one 16-byte store
                    ((float4*)lds)[v] = t[q];
                    continue;
                }
                const int f = v << 2;
                int r = f / kdim, k = f - r * kdim;
                const float e[4] = {t[q].x, t[q].y, t[q].z, t[q].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lds[r * ldk + k] = e[i];
                    if (++k == kdim) { k = 0; ++r; }
                }
            }
        }
        for (int f = (nvec << 2) + threadIdx.x; f < total; f += FILTER_THREADS) {
            const int r = f / kdim;
            lds[r * ldk + (f - r * kdim)] = src[f];
        }
    }
    __syncthreads();
    // One wave evaluates the (at most 64) staged rows.  Which one rotates with the workgroup index: the
    // hardware places wave w of every workgroup on SIMD w % 4, a fixed choice would pile all the
    // evaluation work of a CU onto one of its four SIMDs.
    const int eval_wave = (blockIdx.x + blockIdx.y) & 3;
    for (int r = (int)threadIdx.x - 64 * eval_wave; r >= 0 && r < ((nrow + 63) & ~63) && r < 64; r += FILTER_THREADS) {
        bool pass = false;
        u64 key = 0;
        if (r < nrow) {
            const float* p = lds + r * ldk;
            float conf = p[4];
            if (SRC == 1) conf = sigmoid_f32(conf);
            if (SRC == 2 || SRC == 4 || conf > conf_thr) {
                float cx = p[0], cy = p[1], bw = p[2], bh = p[3];
                if (SRC == 1) {
                    const int lrow = lrow0 + r;
                    const int hw = g.h[s] * g.w[s];
                    const int a = lrow / hw, cell = lrow - a * hw;
                    const int y = cell / g.w[s], x = cell - y * g.w[s];
                    cx = decode_value(g, s, 0, a, y, x, cx);
                    cy = decode_value(g, s, 1, a, y, x, cy);
                    bw = decode_value(g, s, 2, a, y, x, bw);
                    bh = decode_value(g, s, 3, a, y, x, bh);
                }
                const float xmin = (SRC == 2 || SRC == 4) ? cx : cx - bw / 2.0f;
                const float ymin = (SRC == 2 || SRC == 4) ? cy : cy - bh / 2.0f;
                pass = SRC >= 2 || ((xmin > 0.0f) && (ymin > 0.0f));
                if (pass) {
                    // the sigmoid is monotone, so the largest class probability is the sigmoid of the
                    // largest class logit: one exp per surviving row instead of one per class
                    float m = p[5];
                    bool cls_nan = m != m;      // np.max (predict.py:83) hands a NaN on, fmaxf drops it
                    for (int k = 6; k < kdim; ++k) {
                        m = fmaxf(m, p[k]);
                        cls_nan = cls_nan || p[k] != p[k];
                    }
                    if (cls_nan) m = __builtin_nanf("");
                    if (SRC == 1) m = sigmoid_f32(m);
                    key = make_key(conf * m, SRC >= 3 ? ~(row0 + r) : row0 + r);
                    if (SRC >= 3) {   // extents as the corner form x_max - x_min carries them (train_class.py:79)
                        bw = (xmin + bw) - xmin;
                        bh = (ymin + bh) - ymin;
                    }
                    rowbox[(size_t)b * rows + row0 + r] = make_float4(xmin, ymin, bw, bh);
                }
            }
        }
        // wave-level compaction: one atomic per wave, lanes take consecutive slots
        const u64 bal = __ballot(pass);
        if (bal) {
            const int lane = threadIdx.x & 63;
            int base = 0;
            if (lane == 0) base = atomicAdd(&count[b * COUNT_STRIDE], __popcll(bal));
            base = __shfl(base, 0);
            const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
            if (pass && pos < n_pad) keys[(size_t)b * n_pad + pos] = key;
        }
    }
}

// ------------------------------------------------------------------------------------------
// bitonic sort, descending, n_pad keys per image (power of two), padding keys are 0
// ------------------------------------------------------------------------------------------
constexpr int SORT_CHUNK = 4096;  // keys per workgroup in the LDS stages (32 KiB)
constexpr int SORT_THREADS = 512;

// Descending network: in block (i & k) == 0 the larger key goes to the lower index.
__device__ __forceinline__ void cmpex(u64& lo, u64& hi, bool desc) {
    const bool sw = desc ? (lo < hi) : (lo > hi);
    if (sw) { const u64 t = lo; lo = hi; hi = t; }
}

// The keys of an image sit compacted at the front and every real key is above the padding, so sorting the
// first P keys, P = the power of two that holds them (at least one chunk), sorts the whole array: the network
// on that prefix is the network of the whole array cut off after stage k = P, and (i & P) == 0 makes its last
// stage descending.  Workgroups beyond P and stages beyond k = P return at once.
__device__ __forceinline__ int sort_extent(const int* __restrict__ count, int b, int n_pad, int chunk) {
    if (!count) return n_pad;
    const int n = min(count[b], n_pad);
    if (n <= chunk) return chunk;
    return min(n_pad, 1 << (32 - __clz(n - 1)));
}

// All stages with k <= chunk (FULL) or the j < chunk tail of stage k_outer (!FULL), in LDS.
template <bool FULL>
__global__ void __launch_bounds__(SORT_THREADS) bitonic_lds_kernel(u64* __restrict__ keys, const int* __restrict__ count, int n_pad, int chunk, int k_outer) {
    __shared__ u64 s[SORT_CHUNK];
    u64* base = keys + (size_t)blockIdx.y * n_pad + (size_t)blockIdx.x * chunk;
    const int g0 = blockIdx.x * chunk;
    const int extent = sort_extent(count, blockIdx.y, n_pad, chunk);
    if (g0 >= extent || (!FULL && k_outer > extent)) return;
    for (int i = threadIdx.x; i < chunk; i += SORT_THREADS) s[i] = base[i];
    __syncthreads();
    for (int k = FULL ? 2 : k_outer; k <= (FULL ? chunk : k_outer); k <<= 1) {
        for (int j = min(k, chunk) >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < chunk / 2; t += SORT_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const bool desc = (((g0 + i) & k) == 0);
                u64 a = s[i], c = s[i + j];
                cmpex(a, c, desc);
                s[i] = a;
                s[i + j] = c;
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < chunk; i += SORT_THREADS) base[i] = s[i];
}

// One compare-exchange stage (k, j) with j >= chunk, straight in global memory.
__global__ void __launch_bounds__(256) bitonic_global_kernel(u64* __restrict__ keys, const int* __restrict__ count, int n_pad, int chunk, int k, int j) {
    u64* base = keys + (size_t)blockIdx.y * n_pad;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pad / 2) return;
    const int extent = sort_extent(count, blockIdx.y, n_pad, chunk);
    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
    if (k > extent || i >= extent) return;
    const bool desc = ((i & k) == 0);
    u64 a = base[i], c = base[i + j];
    const bool sw = desc ? (a < c) : (a > c);
    if (sw) { base[i] = c; base[i + j] = a; }
}

__global__ void __launch_bounds__(256)
gather_sorted_kernel(const u64* __restrict__ keys, const float4* __restrict__ rowbox, const int* __restrict__ count,
                     int rows, int n_pad, bool complemented, float4* __restrict__ sbox, int* __restrict__ srow) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad || i >= count[b]) return;
    int row = (int)(uint32_t)keys[(size_t)b * n_pad + i];
    if (complemented) row = ~row;
    srow[(size_t)b * n_pad + i] = row;
    sbox[(size_t)b * n_pad + i] = rowbox[(size_t)b * rows + row];
}

// ------------------------------------------------------------------------------------------
// IoU bitmask + greedy scan
// ------------------------------------------------------------------------------------------
// predict.py:35-41 for boxes (x_min, y_min, w, h); returns true when the candidate is
// suppressed, i.e. NOT (iou < thr).  NaN (0/0 for two empty boxes) suppresses, as in numpy.
__device__ __forceinline__ bool suppresses(const float4 a, const float4 c, float thr) {
    // np.maximum / np.minimum hand a NaN operand on (fmaxf / fminf drop it): a NaN corner makes the
    // intersection and with it the IoU NaN, which suppresses.  NaN extents reach the IoU through the areas.
    if (a.x != a.x || a.y != a.y || c.x != c.x || c.y != c.y) return true;
    const float x1 = fmaxf(a.x, c.x);
    const float y1 = fmaxf(a.y, c.y);
    const float x2 = fminf(a.x + a.z, c.x + c.z);
    const float y2 = fminf(a.y + a.w, c.y + c.w);
    const float inter = fmaxf(x2 - x1, 0.0f) * fmaxf(y2 - y1, 0.0f);
    const float uni = a.z * a.w + c.z * c.w - inter;
    const float iou = inter / uni;
    return !(iou < thr);
}

// The validation metric's rule (train_class.py:79-80: corner boxes with x_max = x_min + w go to
// torchvision.ops.nms with threshold 1 - iou_threshold): areas from the corner differences, a
// candidate is suppressed iff IoU > threshold, so a NaN IoU survives.
__device__ __forceinline__ bool suppresses_corner(const float4 a, const float4 c, float thr) {
    const float ax2 = a.x + a.z, ay2 = a.y + a.w, cx2 = c.x + c.z, cy2 = c.y + c.w;
    const float inter = fmaxf(fminf(ax2, cx2) - fmaxf(a.x, c.x), 0.0f) * fmaxf(fminf(ay2, cy2) - fmaxf(a.y, c.y), 0.0f);
    const float iou = inter / ((ax2 - a.x) * (ay2 - a.y) + (cx2 - c.x) * (cy2 - c.y) - inter);
    return iou > thr;
}

constexpr int MASK_WAVES = 4;
typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef float f32x8s __attribute__((ext_vector_type(8)));

// Bare v_max_f32 / v_min_f32 (no NaN canonicalisation): only used where NaN cannot decide.
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// The same with the first operand in a scalar register (a value that is uniform over the wave).
__device__ __forceinline__ float vmax_s(float s, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(s), "v"(b)); return r; }
__device__ __forceinline__ float vmin_s(float s, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "s"(s), "v"(b)); return r; }

// w = 2*w + bit(lane) with the bit taken from a lane mask: one v_addc_co_u32 (carry-in = the mask).
__device__ __forceinline__ uint32_t shift_in(uint32_t w, u64 lane_mask) {
    uint32_t r;
    u64 carry_out;
    asm("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(r), "=s"(carry_out) : "v"(w), "s"(lane_mask));
    return r;
}

// Per-box values derived once: corners, area and whether the cheap overlap test is valid for it.
struct BoxEx { float x2, y2, area, special; };
__device__ __forceinline__ BoxEx box_ex(const float4 c) {
    BoxEx e;
    e.x2 = c.x + c.z;
    e.y2 = c.y + c.w;
    e.area = c.z * c.w;
    // "regular" box: positive extent that survives rounding, positive finite area.  For two
    // regular boxes without overlap the reference arithmetic gives inter = +0, union > 0, IoU = +0.
    const bool regular = (e.x2 > c.x) && (e.y2 > c.y) && (e.area > 1.0e-30f) && (e.area < 3.0e38f) &&
                         (fabsf(c.x) < 3.0e38f) && (fabsf(c.y) < 3.0e38f) && (e.x2 < 3.0e38f) && (e.y2 < 3.0e38f);
    e.special = regular ? 0.0f : 1.0f;
    return e;
}

// predict_class.iou (predict.py:33-41): IoU of box i of `a` (or its only box) against box i of `b`.
__global__ void __launch_bounds__(256) iou_pairs_kernel(const float4* __restrict__ a, int na, const float4* __restrict__ b,
                                                        int nb, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = a[na == 1 ? 0 : i], c = b[nb == 1 ? 0 : i];
    const float x1 = fmaxf(p.x, c.x), y1 = fmaxf(p.y, c.y);
    const float x2 = fminf(p.x + p.z, c.x + c.z), y2 = fminf(p.y + p.w, c.y + c.w);
    const float inter = fmaxf(x2 - x1, 0.0f) * fmaxf(y2 - y1, 0.0f);
    // np.maximum / np.minimum hand a NaN corner on where fmaxf / fminf drop it
    const bool nan_corner = p.x != p.x || p.y != p.y || c.x != c.x || c.y != c.y;
    out[i] = nan_corner ? __builtin_nanf("") : inter / (p.z * p.w + c.z * c.w - inter);
}

// Column operands of the mask kernel's common path per sorted box: (x, y, x2, y2) in `colpk` and
// k_lo*area in `colk`.  A wave reads the records of its column tile with
// scalar loads: the values are uniform over the wave, so they cost neither vector registers nor
// LDS bandwidth (64 lanes reading one LDS address still occupy the LDS pipe for a full pass).
__device__ __forceinline__ void guard_factors(float thr, float& k_lo, float& k_hi) {
    const double t_lo = (double)thr * (1.0 - 3.814697265625e-6), t_hi = (double)thr * (1.0 + 3.814697265625e-6);
    k_lo = (float)(t_lo / (1.0 + t_lo));
    k_hi = (float)(t_hi / (1.0 + t_hi));
}

__global__ void __launch_bounds__(256)
pack_cols_kernel(const float4* __restrict__ sbox, const int* __restrict__ count, int n_pad, float thr,
                 float4* __restrict__ colpk, float* __restrict__ colk) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = count ? min(count[b], n_pad) : n_pad;
    if (i >= n) return;
    float k_lo, k_hi;
    guard_factors(thr, k_lo, k_hi);
    const float4 c = sbox[(size_t)b * n_pad + i];
    const BoxEx e = box_ex(c);
    colpk[(size_t)b * n_pad + i] = make_float4(c.x, c.y, e.x2, e.y2);
    // an irregular box is marked by a NaN here (only ever read where the whole tile is regular, or as this mark)
    colk[(size_t)b * n_pad + i] = e.special != 0.0f ? __builtin_nanf("") : k_lo * e.area;
}

// Grid: (GX, GY, image).  The number of candidates is only known on the device, so a fixed, small
// grid walks the tiles that exist: workgroup (bx, by) takes row tiles by, by+GY, ... of this chunk
// and, for each, column-tile groups bx, bx+GX, ... at or right of the diagonal.
// Wave w of a workgroup owns the 64x64 tile (row tile, column tile 4*group + w): lane i holds row
// box i in registers, the 64 column boxes sit in the wave's LDS slice and are read as broadcasts.
// Rows already suppressed by earlier chunks (bit set in `removed`) are skipped.
// SCOL: the common path takes its column operands from `colpk` (pack_cols_kernel) with scalar loads.
template <bool SCOL>
__global__ void __launch_bounds__(64 * MASK_WAVES)
iou_mask_kernel(const float4* __restrict__ sbox, const float4* __restrict__ colpk, const float* __restrict__ colk,
                const int* __restrict__ count, const u64* __restrict__ removed, int n_pad, int row_begin, int row_end, float thr, int corner_rule,
                u64* __restrict__ mask, size_t mask_stride_img) {
    __shared__ float4 cols[MASK_WAVES][64];
    __shared__ float4 colx[MASK_WAVES][64];
    __shared__ u64 col_special[MASK_WAVES];
    const bool thr_ok = thr > 1.0e-30f && thr < 1.0e30f;
    float k_lo, k_hi;
    guard_factors(thr, k_lo, k_hi);
    const int b = blockIdx.z;
    const int n = count ? min(count[b], n_pad) : n_pad;
    const int words = n_pad >> 6;
    const int nwords = (n + 63) >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rt_end = (min(row_end, n) + 63) >> 6;
    const int ngroups = (nwords + MASK_WAVES - 1) / MASK_WAVES;
    const float4* boxes = sbox + (size_t)b * n_pad;
    for (int rt = (row_begin >> 6) + blockIdx.y; rt < rt_end; rt += gridDim.y)
    for (int grp = rt / MASK_WAVES + blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int ct = __builtin_amdgcn_readfirstlane(grp * MASK_WAVES + wave);  // global column tile (wave-uniform)
    const bool active = (ct >= rt) && (ct < nwords);
    // every wave stages and reads only its own LDS slice: a wave-level barrier orders its accesses
    __builtin_amdgcn_wave_barrier();
    if (active) {
        const float4 c = (ct * 64 + lane < n) ? boxes[ct * 64 + lane] : make_float4(0, 0, 0, 0);
        const BoxEx e = box_ex(c);
        cols[wave][lane] = c;
        // the column's share of the two decision thresholds (see below): k * area
        colx[wave][lane] = make_float4(e.x2, e.y2, k_lo * e.area, k_hi * e.area);
        const u64 sp = __ballot(e.special != 0.0f);
        if (lane == 0) col_special[wave] = sp;
    }
    __builtin_amdgcn_wave_barrier();
    if (!active) continue;
    const int i = rt * 64 + lane;
    u64 word = 0;
    bool live = i < n;
    if (live && removed) live = !((removed[(size_t)b * words + rt] >> lane) & 1ull);
    const u64 live_mask = __ballot(live);
    if (live_mask) {
        const float4 a = live ? boxes[i] : make_float4(0, 0, 0, 0);
        const BoxEx ae = box_ex(a);
        // Division-free decision with a guard band.  For regular boxes (positive extents, area between
        // 1e-30 and 3e38) IoU >= thr is inter * (1 + thr) >= thr * (areaA + areaC); with
        // k_lo = t/(1+t) at t = thr*(1 - 2^-18) and k_hi likewise at thr*(1 + 2^-18):
        //   inter < k_lo*areaA + k_lo*areaC  =>  fl(inter/union) <  thr   (survives)
        //   inter > k_hi*areaA + k_hi*areaC  =>  fl(inter/union) >= thr   (suppressed)
        // (the band is an order of magnitude wider than the ~10 roundings involved); a pair inside the
        // band, an irregular box or an unusable threshold takes the exact reference arithmetic.
        const float a_lo = k_lo * ae.area, a_hi = k_hi * ae.area;
        const u64 row_special = __ballot(ae.special != 0.0f || !live) | (thr_ok ? 0ull : ~0ull);
        const u64 cs_ = col_special[wave];            // same for the whole wave: keep it in scalar registers
        const u64 cspec = ((u64)__builtin_amdgcn_readfirstlane((uint32_t)(cs_ >> 32)) << 32) |
                          __builtin_amdgcn_readfirstlane((uint32_t)cs_);
        uint32_t half[2] = {0u, 0u};
        if ((__ballot(live && ae.special != 0.0f) | cspec) == 0ull && thr_ok) {
            // Common case, every box of the tile regular.  Pairs that suppress are rare (a few in 10^5), so
            // a column only tests "surely below the threshold" for all rows at once; the wave leaves
            // this path for a column only when some row fails that test.
            const float a_keep = live ? a_lo : __builtin_inff();   // idle lanes never ask for the second look
            // Eight columns at a time without a branch between them (eight independent dependency
            // chains keep the vector pipe busy); one test decides whether any of them needs the second look.
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                uint32_t w32 = 0;
#pragma unroll 1
                for (int j0 = 0; j0 < 32; j0 += 8) {
                    u64 need[8];
                    float inter[8];
                    f32x4s sc[SCOL ? 8 : 1];
                    f32x8s sk;
                    if (SCOL) {   // 8 columns: 8 x 16 bytes + 32 bytes, all wave-uniform addresses
                        const size_t c0 = (size_t)b * n_pad + (size_t)ct * 64 + hh * 32 + j0;
                        sk = *(const f32x8s*)(colk + c0);
#pragma unroll
                        for (int u = 0; u < 8; ++u) sc[u] = ((const f32x4s*)colpk)[c0 + u];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int j = hh * 32 + j0 + u;
                        float x1, y1, x2, y2, ez;
                        if (SCOL) {
                            x1 = vmax_s(sc[u][0], a.x); y1 = vmax_s(sc[u][1], a.y);
                            x2 = vmin_s(sc[u][2], ae.x2); y2 = vmin_s(sc[u][3], ae.y2);
                            ez = sk[u];
                        } else {
                            const float4 c = cols[wave][j];
                            const float4 e = colx[wave][j];
                            x1 = vmax(a.x, c.x); y1 = vmax(a.y, c.y);
                            x2 = vmin(ae.x2, e.x); y2 = vmin(ae.y2, e.y);
                            ez = e.z;
                        }
                        // one clamp is enough for the keep test: with dx clamped, a negative dy makes the product
                        // non-positive and the (positive) threshold decides "keep" as it must
                        const float dy = y2 - y1;
                        const float prod = vmax(x2 - x1, 0.0f) * dy;
                        inter[u] = prod;
                        need[u] = __ballot(!(prod < a_keep + ez));
                    }
                    if ((need[0] | need[1] | need[2] | need[3] | need[4] | need[5] | need[6] | need[7]) == 0ull) continue;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (need[u] == 0ull) continue;
                        const int j = hh * 32 + j0 + u;
                        const float4 c = cols[wave][j];
                        // need => product >= threshold > 0 => dy > 0 (dx > 0): the product is the intersection
                        u64 sup = __ballot(inter[u] > a_hi + colx[wave][j].w) & need[u];
                        const u64 amb = need[u] & ~sup;                 // inside the guard band: exact arithmetic
                        if (amb) {
                            const bool mine = (amb >> lane) & 1ull;
                            sup |= __ballot(mine && (corner_rule ? suppresses_corner(a, c, thr) : suppresses(a, c, thr)));
                        }
                        w32 |= (uint32_t)((sup >> lane) & 1ull) << (j0 + u);
                    }
                }
                half[hh] = w32;
            }
        } else {
#pragma unroll
        for (int hh = 1; hh >= 0; --hh) {
            uint32_t w32 = 0;
#pragma unroll 8
            for (int jj = 31; jj >= 0; --jj) {      // descending: the shift-in below builds bit jj last-in-lowest
                const int j = hh * 32 + jj;
                const float4 c = cols[wave][j];
                const float4 e = colx[wave][j];
                const float x1 = vmax(a.x, c.x), y1 = vmax(a.y, c.y);
                const float x2 = vmin(ae.x2, e.x), y2 = vmin(ae.y2, e.y);
                const float inter = vmax(x2 - x1, 0.0f) * vmax(y2 - y1, 0.0f);
                const u64 keep_sure = __ballot(inter < a_lo + e.z);
                u64 sup = __ballot(inter > a_hi + e.w);
                // lanes that need the exact arithmetic: undecided, irregular row box, or the whole
                // column when its box is irregular
                u64 amb = (~(keep_sure | sup) | row_special | (((cspec >> j) & 1ull) ? ~0ull : 0ull)) & live_mask;
                if (amb) {
                    const bool mine = (amb >> lane) & 1ull;
                    const u64 exact = __ballot(mine && (corner_rule ? suppresses_corner(a, c, thr) : suppresses(a, c, thr)));
                    sup = (sup & ~amb) | exact;
                }
                w32 = shift_in(w32, sup);
            }
            half[hh] = w32;
        }
        }
        word = ((u64)half[1] << 32) | half[0];
        const int ncol = n - ct * 64;                                     // valid columns of this tile
        if (ncol < 64) word &= (1ull << ncol) - 1ull;
        if (ct == rt) word &= lane == 63 ? 0ull : ~((2ull << lane) - 1ull);  // strictly later boxes only
        if (!live) word = 0;
    }
    mask[(size_t)b * mask_stride_img + (size_t)(i - row_begin) * words + ct] = word;
    }
}

// Between two chunks of the greedy pass: rows [0, r1) of the sorted candidate list are decided (kept ones are in the
// output already), and every later candidate whose bit in `removed` is set will never be looked at again.  The
// survivors move to the front of the other copy of the list (order preserved, so the greedy result is the same), the
// count shrinks and `removed` starts again at zero: the next chunk's mask has neither dead rows (a wave of 64 rows with
// a few live lanes costs what a full one costs) nor dead columns.  One workgroup per image.
constexpr int COMPACT_THREADS = 1024;
constexpr int COMPACT_MAX_WORDS = 4096;       // n_pad <= 2^18

__global__ void __launch_bounds__(COMPACT_THREADS)
compact_live_kernel(const float4* __restrict__ sbox_in, const int* __restrict__ srow_in, float4* __restrict__ sbox_out,
                    int* __restrict__ srow_out, int* __restrict__ count, u64* __restrict__ removed, int n_pad, int r1) {
    __shared__ u64 live[COMPACT_MAX_WORDS];
    __shared__ int base[COMPACT_MAX_WORDS];
    __shared__ int part[COMPACT_THREADS];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(count[b], n_pad);
    const int words = n_pad >> 6, nw = (n + 63) >> 6, w1 = r1 >> 6;
    u64* rem = removed + (size_t)b * words;
    constexpr int PER = COMPACT_MAX_WORDS / COMPACT_THREADS;
    int local = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int w = tid * PER + k;
        u64 bits = 0;
        if (w >= w1 && w < nw) {
            bits = ~rem[w];
            if (n - w * 64 < 64) bits &= (1ull << (n - w * 64)) - 1ull;
        }
        if (w < COMPACT_MAX_WORDS) { live[w] = bits; base[w] = local; }
        local += __popcll(bits);
    }
    part[tid] = local;
    __syncthreads();
    for (int d = 1; d < COMPACT_THREADS; d <<= 1) {
        const int t = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += t;
        __syncthreads();
    }
    const int total = part[COMPACT_THREADS - 1];
    const int mine = part[tid] - local;                // exclusive prefix of this thread's words
#pragma unroll
    for (int k = 0; k < PER; ++k) base[tid * PER + k] += mine;
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    const float4* si = sbox_in + (size_t)b * n_pad;
    const int* ri = srow_in + (size_t)b * n_pad;
    float4* so = sbox_out + (size_t)b * n_pad;
    int* ro = srow_out + (size_t)b * n_pad;
    for (int w = w1 + wave; w < nw; w += COMPACT_THREADS / 64) {
        const u64 bits = live[w];
        if ((bits >> lane) & 1ull) {
            const int pos = base[w] + __popcll(bits & ((1ull << lane) - 1ull));
            so[pos] = si[w * 64 + lane];
            ro[pos] = ri[w * 64 + lane];
        }
    }
    for (int w = tid; w < nw; w += COMPACT_THREADS) rem[w] = 0ull;
    if (tid == 0) count[b] = total;
}

// ------------------------------------------------------------------------------------------
// IoU mask with area windows.  A pair can only reach the threshold when the two areas are within
// a factor (1 + thr) / thr of each other, and the cheap keep test of iou_mask_kernel already proves
// it: inter <= min(ea_row, ea_col) (ea = product of the rounded extents, see below), so
//     ea_col < k_lo * area_row            or            ea_row < k_lo * area_col
// implies inter < k_lo * area_row + k_lo * area_col, the "surely survives" decision.  To make that
// usable for whole waves, the rows of a wave are chosen by size AND place: within blocks of AREA_BLOCK sorted
// positions the rows are listed by (half-octave area class, Morton code of the centre) (area_order_kernel); a wave
// takes 64 consecutive entries, i.e. rows of similar area from one region of the image, and for every column tile a
// ballot over the column boxes keeps only the columns that some row of the wave could meet at the threshold - by
// area, by shape, and by lying inside the bounding box of the wave's rows at all (a column outside it has
// intersection 0 with every row, which never reaches a positive threshold) - and
// packs them into the wave's LDS slice.  The surviving columns (about half on mixed-size candidates) run
// through the same common-case test as in iou_mask_kernel, eight at a time; everything else (irregular boxes, second look, exact
// arithmetic) is unchanged, and so are the bits written: the scan kernel cannot tell the difference.
// The price: the strict upper triangle is only exploited per wave (column tiles before the wave's first row are
// skipped, the rest is computed for all its rows and masked per lane).  Measured at 11 000 candidates per image: the
// bounding box takes 10 % off the kernel (what is left is the per-tile work: staging, ballots, packing); blocks of
// 4096 rows (seven waves per size class and image region instead of two) gain nothing more and cost a longer sort.
//   inter <= ea: dx = fl(min(ax2, cx2) - max(ax, cx)) <= fl(ax2 - ax) = ew by monotone rounding, same for dy,
//   and for dx, dy > 0 fl(dx * dy) <= fl(ew * eh); with dx <= 0 or dy <= 0 the product is <= 0.
//   Likewise inter <= fl(min(ew_row, ew_col) * min(eh_row, eh_col)), which also separates shapes:
//   a column is dropped when that bound, taken with the widest and the tallest row of the wave, stays
//   below fl(min_row(k_lo * area_row) + k_lo * area_col) <= the pair's own threshold.
// ------------------------------------------------------------------------------------------
constexpr int AREA_BLOCK = 1024;          // granule of the row order (chunks are made of whole blocks)

// bits 0..9 of v spread to the even bit positions
__device__ __forceinline__ uint32_t spread10(uint32_t v) {
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

// Row order of the windowed mask kernel: within blocks of BLOCK sorted positions, ascending (area class, Morton code
// of the box centre in 16-pixel cells, position).  The order only decides which rows share a wave: any order is correct.
template <int BLOCK>
__global__ void __launch_bounds__(256)
area_order_kernel(const float4* __restrict__ sbox, const int* __restrict__ count, int n_pad, int* __restrict__ perm) {
    __shared__ u64 s[BLOCK];
    const int b = blockIdx.y, base = blockIdx.x * BLOCK;
    const int n = count ? min(count[b], n_pad) : n_pad;
    for (int t = threadIdx.x; t < BLOCK; t += 256) {
        const int gi = base + t;
        u64 key = ~0ull;
        if (gi < n) {
            const float4 c = sbox[(size_t)b * n_pad + gi];
            const BoxEx e = box_ex(c);
            // positive floats order like their bit patterns: exponent and first mantissa bit = half-octave classes;
            // irregular boxes go last
            const uint32_t cls = e.special != 0.0f ? 0x1FFu : (__float_as_uint(e.area) >> 22);
            const float cx = c.x + 0.5f * c.z, cy = c.y + 0.5f * c.w;
            const uint32_t qx = (uint32_t)fminf(fmaxf(cx * 0.0625f, 0.0f), 1023.0f);   // NaN -> 0
            const uint32_t qy = (uint32_t)fminf(fmaxf(cy * 0.0625f, 0.0f), 1023.0f);
            const uint32_t morton = spread10(qx) | (spread10(qy) << 1);
            key = ((u64)cls << 52) | ((u64)morton << 32) | (uint32_t)gi;
        }
        s[t] = key;
    }
    __syncthreads();
    if (base < n) {
        for (int k = 2; k <= BLOCK; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = threadIdx.x; t < BLOCK / 2; t += 256) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int q = i | j;
                    const u64 x = s[i], y = s[q];
                    if ((x > y) == ((i & k) == 0)) { s[i] = y; s[q] = x; }
                }
                __syncthreads();
            }
    }
    for (int t = threadIdx.x; t < BLOCK; t += 256)
        perm[(size_t)b * n_pad + base + t] = s[t] == ~0ull ? -1 : (int)(uint32_t)s[t];
}

__global__ void __launch_bounds__(64 * MASK_WAVES)
iou_mask_win_kernel(const float4* __restrict__ sbox, const float4* __restrict__ colpk, const float* __restrict__ colk,
                    const int* __restrict__ perm, const int* __restrict__ count, const u64* __restrict__ removed,
                    int n_pad, int row_begin, int row_end, float thr, int corner_rule, u64* __restrict__ mask,
                    size_t mask_stride_img) {
    constexpr int order_block = AREA_BLOCK;
    __shared__ float4 cols[MASK_WAVES][64];   // the tile's column boxes as they are (second look, exact arithmetic)
    // the surviving columns of the wave's tile, packed: (x, y, x2, y2), k_lo * area, original column
    __shared__ float4 pk_box[MASK_WAVES][72];
    __shared__ float pk_k[MASK_WAVES][72];
    __shared__ int pk_j[MASK_WAVES][72];
    const bool thr_ok = thr > 1.0e-30f && thr < 1.0e30f;
    float k_lo, k_hi;
    guard_factors(thr, k_lo, k_hi);
    const int b = blockIdx.z;
    const int n = min(count[b], n_pad);
    const int words = n_pad >> 6;
    const int nwords = (n + 63) >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rg_end = (min(row_end, n) + 63) >> 6;
    const int ngroups = (nwords + MASK_WAVES - 1) / MASK_WAVES;
    const float4* boxes = sbox + (size_t)b * n_pad;
    for (int rg = (row_begin >> 6) + blockIdx.y; rg < rg_end; rg += gridDim.y) {
        const int i = perm[(size_t)b * n_pad + rg * 64 + lane];         // sorted position of this lane's row, or -1
        bool live = i >= 0;
        if (live) live = !((removed[(size_t)b * words + (i >> 6)] >> (i & 63)) & 1ull);
        const u64 live_mask = __ballot(live);
        const float4 a = live ? boxes[i] : make_float4(0, 0, 0, 0);
        const BoxEx ae = box_ex(a);
        const float a_lo = k_lo * ae.area, a_hi = k_hi * ae.area;
        const u64 row_special = __ballot(ae.special != 0.0f || !live) | (thr_ok ? 0ull : ~0ull);
        const bool rows_regular = __ballot(live && ae.special != 0.0f) == 0ull && thr_ok;
        // the wave's window: smallest row threshold share, largest row "effective area"
        float wmin = live ? a_lo : __builtin_inff();
        float wmax = live ? (ae.x2 - a.x) * (ae.y2 - a.y) : -__builtin_inff();
        float wmax_w = live ? ae.x2 - a.x : -__builtin_inff();     // widest and tallest row (rounded extents)
        float wmax_h = live ? ae.y2 - a.y : -__builtin_inff();
        // bounding box of the wave's rows (only used when every live row is regular) and its first row
        float bx0 = live ? a.x : __builtin_inff(), by0 = live ? a.y : __builtin_inff();
        float bx1 = live ? ae.x2 : -__builtin_inff(), by1 = live ? ae.y2 : -__builtin_inff();
        int i_min = live ? i : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            wmin = fminf(wmin, __shfl_xor(wmin, o));
            wmax = fmaxf(wmax, __shfl_xor(wmax, o));
            wmax_w = fmaxf(wmax_w, __shfl_xor(wmax_w, o));
            wmax_h = fmaxf(wmax_h, __shfl_xor(wmax_h, o));
            bx0 = fminf(bx0, __shfl_xor(bx0, o));
            by0 = fminf(by0, __shfl_xor(by0, o));
            bx1 = fmaxf(bx1, __shfl_xor(bx1, o));
            by1 = fmaxf(by1, __shfl_xor(by1, o));
            i_min = min(i_min, __shfl_xor(i_min, o));
        }
        // column tiles before the wave's first live row hold no later box; a wave without live rows still writes
        // its rows' (zero) words from the block's first tile on, as the scan of those rows' tiles reads them
        const int blk0 = (rg * 64 / order_block) * (order_block / 64);
        const int tile0 = live_mask ? max(blk0, i_min >> 6) : blk0;
        for (int grp = tile0 / MASK_WAVES + blockIdx.x; grp < ngroups; grp += gridDim.x) {
            const int ct = __builtin_amdgcn_readfirstlane(grp * MASK_WAVES + wave);   // column tile (wave-uniform)
            if (ct >= nwords) continue;
            // every wave stages and reads only its own LDS slice: a wave-level barrier orders its accesses
            __builtin_amdgcn_wave_barrier();
            const bool cvalid = ct * 64 + lane < n;
            const size_t cidx = (size_t)b * n_pad + (size_t)ct * 64 + lane;
            // the column's record from pack_cols_kernel: (x, y, x2, y2) and k_lo * area (NaN = irregular box)
            const float4 cp = cvalid ? colpk[cidx] : make_float4(0, 0, 0, 0);
            const float ck = cvalid ? colk[cidx] : __builtin_nanf("");
            cols[wave][lane] = cvalid ? boxes[ct * 64 + lane] : make_float4(0, 0, 0, 0);
            const u64 cspec = __ballot(ck != ck);
            // columns some row of this wave could meet at the threshold (negated tests: NaN keeps the column);
            // third test: inter <= min(ew_row, ew_col) * min(eh_row, eh_col), so boxes of similar area but
            // different shape are dropped as well
            const float cew = cp.z - cp.x, ceh = cp.w - cp.y;
            const bool keep_col = cvalid && !(cew * ceh < wmin) && !(wmax < ck) &&
                                  !(fminf(wmax_w, cew) * fminf(wmax_h, ceh) < wmin + ck) &&
                                  !(cp.z < bx0) && !(cp.x > bx1) && !(cp.w < by0) && !(cp.y > by1);
            const u64 compat = __ballot(keep_col);
            const int ncompat = __popcll(compat);
            if (keep_col) {
                const int pos = __popcll(compat & ((1ull << lane) - 1ull));
                pk_box[wave][pos] = cp;
                pk_k[wave][pos] = ck;
                pk_j[wave][pos] = lane;
            }
            if (lane < 8) {   // a short last batch meets records that can never ask for the second look
                pk_box[wave][ncompat + lane] = make_float4(0, 0, 0, 0);
                pk_k[wave][ncompat + lane] = __builtin_inff();
                pk_j[wave][ncompat + lane] = 0;
            }
            __builtin_amdgcn_wave_barrier();
            u64 word = 0;
            if (live_mask) {
                uint32_t half[2] = {0u, 0u};
                if (rows_regular && cspec == 0ull) {
                    // Common case, every box of the tile regular (all 64 columns exist: a missing one is irregular).
                    const float a_keep = live ? a_lo : __builtin_inff();   // idle lanes never ask for the second look
#pragma unroll 1
                    for (int q0 = 0; q0 < ncompat; q0 += 8) {   // eight surviving columns per branch
                        u64 need[8];
                        float inter[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const float4 cb = pk_box[wave][q0 + u];
                            const float x1 = vmax(a.x, cb.x), y1 = vmax(a.y, cb.y);
                            const float x2 = vmin(ae.x2, cb.z), y2 = vmin(ae.y2, cb.w);
                            const float dy = y2 - y1;
                            const float prod = vmax(x2 - x1, 0.0f) * dy;
                            inter[u] = prod;
                            need[u] = __ballot(!(prod < a_keep + pk_k[wave][q0 + u]));
                        }
                        if ((need[0] | need[1] | need[2] | need[3] | need[4] | need[5] | need[6] | need[7]) == 0ull) continue;
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            if (need[u] == 0ull) continue;
                            const int j = __builtin_amdgcn_readfirstlane(pk_j[wave][q0 + u]);
                            const float4 cc = cols[wave][j];
                            // need => product >= threshold > 0 => dy > 0 (dx > 0): the product is the intersection
                            u64 sup = __ballot(inter[u] > a_hi + k_hi * (cc.z * cc.w)) & need[u];
                            const u64 amb = need[u] & ~sup;                 // inside the guard band: exact arithmetic
                            if (amb) {
                                const bool mine = (amb >> lane) & 1ull;
                                sup |= __ballot(mine && (corner_rule ? suppresses_corner(a, cc, thr) : suppresses(a, cc, thr)));
                            }
                            const uint32_t bit = (uint32_t)((sup >> lane) & 1ull);
                            if (j < 32) half[0] |= bit << j;
                            else half[1] |= bit << (j - 32);
                        }
                    }
                } else {
#pragma unroll
                    for (int hh = 1; hh >= 0; --hh) {
                        uint32_t w32 = 0;
#pragma unroll 8
                        for (int jq = 31; jq >= 0; --jq) {      // descending: the shift-in below builds bit jq last-in-lowest
                            const int j = hh * 32 + jq;
                            // a tile with an irregular box (or the last, partial one)
                            const float4 cc = cols[wave][j];
                            const BoxEx ce = box_ex(cc);
                            const float x1 = vmax(a.x, cc.x), y1 = vmax(a.y, cc.y);
                            const float x2 = vmin(ae.x2, ce.x2), y2 = vmin(ae.y2, ce.y2);
                            const float inter = vmax(x2 - x1, 0.0f) * vmax(y2 - y1, 0.0f);
                            const u64 keep_sure = __ballot(inter < a_lo + k_lo * ce.area);
                            u64 sup = __ballot(inter > a_hi + k_hi * ce.area);
                            u64 amb = (~(keep_sure | sup) | row_special | (((cspec >> j) & 1ull) ? ~0ull : 0ull)) & live_mask;
                            if (amb) {
                                const bool mine = (amb >> lane) & 1ull;
                                const u64 exact = __ballot(mine && (corner_rule ? suppresses_corner(a, cc, thr) : suppresses(a, cc, thr)));
                                sup = (sup & ~amb) | exact;
                            }
                            w32 = shift_in(w32, sup);
                        }
                        half[hh] = w32;
                    }
                }
                word = ((u64)half[1] << 32) | half[0];
                const int ncol = n - ct * 64;                                     // valid columns of this tile
                if (ncol < 64) word &= (1ull << ncol) - 1ull;
                const int lim = i - ct * 64;                                      // strictly later boxes only
                if (lim >= 63) word = 0;
                else if (lim >= 0) word &= ~((2ull << lim) - 1ull);
                if (!live) word = 0;
            }
            if (i >= 0) mask[(size_t)b * mask_stride_img + (size_t)(i - row_begin) * words + ct] = word;
        }
    }
}

constexpr int SCAN_GROUP = 256;                       // threads that share a tile's update
constexpr int SCAN_GROUPS = 2;                        // thread groups; group g owns the tiles t with t % SCAN_GROUPS == g
constexpr int SCAN_THREADS = SCAN_GROUP * SCAN_GROUPS;
constexpr int SCAN_PAIRS = SCAN_GROUP / 2;            // column-word pairs one group covers from registers
constexpr int SCAN_SPARSE_ROWS = 20;                  // up to this many rows with in-tile bits: loop over them alone

// One workgroup per image.  `removed` (n_pad bits) lives in LDS for the chunk and in global
// memory between chunks.  Per 64-row tile: the first wave of the owning group resolves the tile serially
// from its 64 diagonal words (a chain of scalar instructions), then the group's threads OR the kept rows'
// words into the later part of `removed`: thread (pair, half) holds the two column words 2*pair, 2*pair+1 of
// the 32 rows of its half (32 16-byte loads: a wave never has more loads in flight than its counter of
// outstanding loads can express, so issuing them does not stall).  The greedy order makes the tiles a serial
// chain; what bounds it is the chain itself plus the update of `removed`, so everything else sits beside it:
// a group issues the loads of ITS next tile at the start of the iteration in which the other group resolves.
__global__ void __launch_bounds__(SCAN_THREADS)
nms_scan_kernel(const u64* __restrict__ mask, size_t mask_stride_img, const int* __restrict__ count,
                const int* __restrict__ srow, u64* __restrict__ removed_g, int n_pad, int row_begin, int row_end,
                int* __restrict__ keep_idx, int* __restrict__ keep_count, int max_keep) {
    extern __shared__ __attribute__((aligned(16))) u64 removed[];
    __shared__ u64 kept_bits;
    __shared__ int kept_total;
    const int b = blockIdx.x;
    const int n = count ? min(count[b], n_pad) : n_pad;
    if (row_begin >= n) return;
    const int words = n_pad >> 6;
    const int nwords = (n + 63) >> 6;
    const u64* m = mask + (size_t)b * mask_stride_img;
    u64* rg = removed_g + (size_t)b * words;
    for (int w = threadIdx.x; w < nwords; w += SCAN_THREADS) removed[w] = rg[w];
    if (threadIdx.x == 0) kept_total = keep_count[b];
    __syncthreads();
    const int group = __builtin_amdgcn_readfirstlane(threadIdx.x / SCAN_GROUP);       // wave-uniform
    const int w0 = threadIdx.x % SCAN_GROUP;
    const int lane = threadIdx.x & 63;
    const bool resolver = __builtin_amdgcn_readfirstlane(w0) < 64;                    // first wave of the group
    const int pair = w0 % SCAN_PAIRS, half = __builtin_amdgcn_readfirstlane(w0 / SCAN_PAIRS);
    const bool pair_ok = 2 * pair + 1 < words && 2 * pair < nwords;                    // both words exist in a row
    const int t_begin = row_begin >> 6;
    const int t_end = (min(row_end, n) + 63) >> 6;
    // this group's next tile: the first one at or after t_begin with t % SCAN_GROUPS == group
    int mine = t_begin + (group - t_begin % SCAN_GROUPS + SCAN_GROUPS) % SCAN_GROUPS;
    u64 diag_next = 0;
    int srow_next = 0;
    ulonglong2 v[32];
    auto fetch_diag = [&](int t) {                     // diagonal words and row indices of tile t (wave-uniform t)
        if (t >= t_end || !resolver) return;
        const int i = t * 64 + lane;
        diag_next = (i < n) ? m[(size_t)(i - row_begin) * words + t] : 0ull;
        srow_next = (i < n) ? srow[(size_t)b * n_pad + i] : 0;
    };
    auto fetch_rows = [&](int t) {                     // this thread's two column words of its 32 rows of tile t
        if (t >= t_end) return;
        if (pair_ok && 2 * pair + 1 > t) {
            const u64* m0 = m + (size_t)(t * 64 + half * 32 - row_begin) * words + 2 * pair;
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = *(const ulonglong2*)(m0 + (size_t)u * words);
        }
    };
    fetch_diag(mine);
    fetch_rows(mine);
    bool rows_pending = false;
    for (int t = t_begin; t < t_end; ++t) {
        const bool owner = t == mine;                  // uniform within a group
        if (rows_pending) {                            // never the owner: it fetched SCAN_GROUPS - 1 tiles ago
            fetch_rows(mine);
            rows_pending = false;
        }
        if (owner && resolver) {
            const u64 diag = ~diag_next;              // complemented once, in all lanes: a step then only ANDs
            const uint32_t dlo = (uint32_t)diag, dhi = (uint32_t)(diag >> 32);
            u64 rem = removed[t];
            if (n - t * 64 < 64) rem |= ~((1ull << (n - t * 64)) - 1ull);
            // wave-uniform state in scalar registers: the 64-step chain below is scalar work.  A row's word
            // only has bits of strictly later rows, so bit q is final once step q has looked at it: what is
            // left of `alive` after the last step is the set of kept rows.
            u64 alive = ~(((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(rem >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rem));
            // rows whose word is empty inside the tile cannot change `alive`: with few others, visit only those
            u64 todo = __ballot(diag != ~0ull) & alive;
            if (__popcll(todo) <= SCAN_SPARSE_ROWS) {
                while (todo) {
                    const int q = __builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    const u64 dq = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)dhi, q) << 32) |
                                   (uint32_t)__builtin_amdgcn_readlane((int)dlo, q);
                    if ((alive >> q) & 1ull) alive &= dq;
                }
            } else
#pragma unroll
            for (int q = 0; q < 64; ++q) {
                const u64 dq = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)dhi, q) << 32) |
                               (uint32_t)__builtin_amdgcn_readlane((int)dlo, q);
                // alive &= alive[q] ? dq : ~0, as the three scalar instructions it is (the compiler's own
                // sequence shifts and tests 32-bit halves: five dependent instructions a step)
                u64 pick;
                asm volatile("s_bitcmp1_b64 %0, %3\n\ts_cselect_b64 %1, %2, -1\n\ts_and_b64 %0, %0, %1"
                             : "+s"(alive), "=&s"(pick) : "s"(dq), "i"(q) : "scc");
            }
            const u64 kept = alive;
            const int base = kept_total;
            if (lane == 0) {
                kept_bits = kept;
                kept_total = base + __popcll(kept);
                removed[t] |= ~kept;  // every row of a finished tile is decided
            }
            if ((kept >> lane) & 1ull) {
                const int pos = base + __popcll(kept & ((1ull << lane) - 1ull));
                if (pos < max_keep) keep_idx[(size_t)b * max_keep + pos] = srow_next;
            }
        }
        __syncthreads();
        if (owner) {
            const u64 kept_all = kept_bits;
            const u64 kept = ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(kept_all >> 32)) << 32) |
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)kept_all);
            const u64* mt = m + (size_t)(t * 64 - row_begin) * words;
            // the rows' words were loaded SCAN_GROUPS tiles ago (the loads do not depend on the outcome: rows
            // that were not kept are masked out)
            if (pair_ok && 2 * pair + 1 > t) {
                const uint32_t kh = half ? (uint32_t)(kept >> 32) : (uint32_t)kept;    // this half's 32 rows
                u64 acc0 = 0, acc1 = 0;
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    const u64 on = 0ull - (u64)((kh >> u) & 1u);                        // uniform: all ones for a kept row
                    acc0 |= v[u].x & on;
                    acc1 |= v[u].y & on;
                }
                if (2 * pair > t && acc0) atomicOr((unsigned long long*)&removed[2 * pair], (unsigned long long)acc0);
                if (2 * pair + 1 < nwords && acc1) atomicOr((unsigned long long*)&removed[2 * pair + 1], (unsigned long long)acc1);
            }
            // further column words (more than 16384 candidates): loaded here, and only for the rows that were kept (the
            // outcome is known and wave-uniform by now; with dense candidates most rows of a tile are suppressed), in
            // batches of eight loads
            for (int w = w0 + 2 * SCAN_PAIRS; w < nwords; w += SCAN_GROUP) {
                if (w <= t) continue;
                u64 acc = 0, todo = kept;
#pragma unroll 1
                while (todo) {
                    u64 x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int q = todo ? __builtin_ctzll(todo) : -1;      // uniform
                        todo = todo ? (todo & (todo - 1ull)) : 0ull;
                        x[u] = q >= 0 ? mt[(size_t)q * words + w] : 0ull;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc |= x[u];
                }
                if (acc) removed[w] |= acc;
            }
            mine += SCAN_GROUPS;
            fetch_diag(mine);                          // two loads; the rows follow at the top of the next iteration
            rows_pending = true;
        }
        __syncthreads();
    }
    for (int w = threadIdx.x; w < nwords; w += SCAN_THREADS) rg[w] = removed[w];
    if (threadIdx.x == 0) keep_count[b] = kept_total;
}

// Kept rows -> output rows (predict.py:74-79).  SRC as in filter_kernel.  One wave per kept row:
// the row's (scale, anchor, cell) is wave-uniform, lanes walk the K columns.  A wave takes
// EMIT_ROWS consecutive kept rows per trip and issues all their loads before the first use, so
// that enough reads are in flight to cover the latency of HBM.
constexpr int EMIT_ROWS = 8;

template <int SRC>
__global__ void __launch_bounds__(256)
emit_rows_kernel(const float* __restrict__ pred, DecodeGeom g, int rows, int kdim, const int* __restrict__ keep_idx,
                 const int* __restrict__ keep_count, int max_keep, const float* __restrict__ scale_xy,
                 float* __restrict__ out) {
    const int b = blockIdx.y;
    const int nk = min(keep_count[b], max_keep);
    const float xs = scale_xy ? scale_xy[2 * b] : 1.0f;
    const float ys = scale_xy ? scale_xy[2 * b + 1] : 1.0f;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int j0 = wave * EMIT_ROWS; j0 < nk; j0 += gridDim.x * 4 * EMIT_ROWS) {
        const float* src[EMIT_ROWS];
        int s[EMIT_ROWS], a[EMIT_ROWS], y[EMIT_ROWS], x[EMIT_ROWS];
#pragma unroll
        for (int u = 0; u < EMIT_ROWS; ++u) {
            const int j = min(j0 + u, nk - 1);   // the tail repeats the last row (loads only)
            const int row = keep_idx[(size_t)b * max_keep + j];
            s[u] = a[u] = y[u] = x[u] = 0;
            if (SRC != 1) {
                src[u] = pred + ((size_t)b * rows + row) * kdim;
            } else {
                s[u] = row >= g.rows_before[2] ? 2 : (row >= g.rows_before[1] ? 1 : 0);
                const int lrow = row - g.rows_before[s[u]];
                const int hw = g.h[s[u]] * g.w[s[u]];
                a[u] = lrow / hw;
                const int cell = lrow - a[u] * hw;
                y[u] = cell / g.w[s[u]];
                x[u] = cell - y[u] * g.w[s[u]];
                src[u] = g.head[s[u]] + ((size_t)b * g.n_anchor * hw + lrow) * kdim;
            }
        }
        for (int k0 = 0; k0 < kdim; k0 += 64) {
            const int k = k0 + lane;
            const bool live = k < kdim;
            float v[EMIT_ROWS], p0[EMIT_ROWS], p1[EMIT_ROWS];
#pragma unroll
            for (int u = 0; u < EMIT_ROWS; ++u) {
                v[u] = live ? src[u][k] : 0.0f;
                // the box columns also need their partner column (centre <-> extent)
                p0[u] = (k0 == 0 && SRC != 2 && k < 4) ? src[u][k < 2 ? k + 2 : k] : 0.0f;
                p1[u] = (k0 == 0 && SRC >= 3 && k < 4) ? src[u][k < 2 ? k : k - 2] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < EMIT_ROWS; ++u) {
                if (!live || j0 + u >= nk) continue;
                float r = v[u];
                if (SRC == 1) r = decode_from_sigmoid(g, s[u], k, a[u], y[u], x[u], r, sigmoid_f32(r));
                if (SRC >= 3 && k < 4) {
                    // train_class.py:340,79,82: x_min = cx - w/2 ; w' = (x_min + w) - x_min
                    const float wh = p0[u];
                    const float lo = SRC == 3 ? p1[u] - wh / 2.0f : p1[u];
                    r = k < 2 ? lo : (lo + wh) - lo;
                } else if (SRC != 2 && k < 4) {
                    // x_min = cx - w/2 ; cx' = x_min + w/2 ; scaled.  Same roundings as the reference.
                    if (k < 2) {
                        float wh = p0[u];
                        if (SRC == 1) wh = decode_value(g, s[u], k + 2, a[u], y[u], x[u], wh);
                        const float half = wh / 2.0f;
                        r = ((r - half) + half) * (k == 0 ? xs : ys);
                    } else {
                        r = r * (k == 2 ? xs : ys);
                    }
                }
                out[((size_t)b * max_keep + j0 + u) * kdim + k] = r;
            }
        }
    }
}

// emit_rows_kernel<1> (rows decoded from the raw heads) in a leaner form, used when every grid is
// narrower than 4096 cells: the row's (scale, anchor, cell) is worked out once per row with scalar
// values (anchor by comparisons, the cell split by a float reciprocal), the class columns are a plain
// elementwise pass (one sigmoid per element, no divergence) and the 4 box columns of the wave's 4 rows
// are handled together by 16 lanes.  Same arithmetic per element as decode_kernel + emit_rows_kernel<0>.
struct EmitGeom { float inv_w[3]; float inv_cls; };   // 1 / grid width per scale, 1 / (kdim - 4)

constexpr int EMIT_HEADS_ROWS = 8;   // kept rows per wave and trip

__global__ void __launch_bounds__(256)
emit_heads_kernel(DecodeGeom g, EmitGeom eg, int kdim, const int* __restrict__ keep_idx, const int* __restrict__ keep_count,
                  int max_keep, const float* __restrict__ scale_xy, float* __restrict__ out) {
    // per wave: the source pointer of its rows (lanes 0..7 work the row geometry out, one row each; doing
    // it with wave-uniform scalar code costs ~100 scalar instructions per row and the CU has one scalar unit)
    __shared__ const float* row_src[4][EMIT_HEADS_ROWS];
    const int b = blockIdx.y;
    const int nk = min(keep_count[b], max_keep);
    const float xs = scale_xy ? scale_xy[2 * b] : 1.0f;
    const float ys = scale_xy ? scale_xy[2 * b + 1] : 1.0f;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wv;
    const int ncls = kdim - 4;          // objectness + classes: the elementwise part of a row
    for (int j0 = wave * EMIT_HEADS_ROWS; j0 < nk; j0 += gridDim.x * 4 * EMIT_HEADS_ROWS) {
        // lanes 0..31: (row u = lane / 4, box column k = lane % 4); lanes 0..7 of them also publish the row pointer
        const int u4 = (lane >> 2) & (EMIT_HEADS_ROWS - 1), kb = lane & 3;
        const int ju = lane < EMIT_HEADS_ROWS ? lane : u4;
        const int j = min(j0 + ju, nk - 1);     // the tail repeats the last row (loads only)
        const int row = keep_idx[(size_t)b * max_keep + j];
        const int s = row >= g.rows_before[2] ? 2 : (row >= g.rows_before[1] ? 1 : 0);
        const int lrow = row - g.rows_before[s];
        const int gw = s == 2 ? g.w[2] : (s == 1 ? g.w[1] : g.w[0]);
        const int hw = (s == 2 ? g.h[2] : (s == 1 ? g.h[1] : g.h[0])) * gw;
        const float* head = s == 2 ? g.head[2] : (s == 1 ? g.head[1] : g.head[0]);
        const float* src = head + ((size_t)b * g.n_anchor * hw + lrow) * kdim;
        __builtin_amdgcn_wave_barrier();
        if (lane < EMIT_HEADS_ROWS) row_src[wv][lane] = src;
        __builtin_amdgcn_wave_barrier();
        // ---- objectness and class columns of the 8 rows as one flat element list --------------------
        const int total = EMIT_HEADS_ROWS * ncls;
        for (int e0 = 0; e0 < total; e0 += 64) {
            const int e = e0 + lane;
            // floor(e / ncls): (e + 0.5) / ncls stays 0.5 / ncls away from an integer (e < 2^16, ncls < 2^12)
            const int u = (int)(((float)e + 0.5f) * eg.inv_cls);
            const int k = 4 + e - u * ncls;
            if (e < total && j0 + u < nk) {
                const float v = row_src[wv][u][k];
                out[((size_t)b * max_keep + j0 + u) * kdim + k] = g.deploy ? sigmoid_f32(v) : v;
            }
        }
        // ---- box columns: this lane's own row (lanes 8.. recomputed it for row lane / 4) ----------------
        if (lane < 4 * EMIT_HEADS_ROWS) {
            // geometry of row u4 (for lanes < 8 the values above belong to row `lane`, so redo them for u4)
            const int j2 = min(j0 + u4, nk - 1);
            const int row2 = keep_idx[(size_t)b * max_keep + j2];
            const int s2 = row2 >= g.rows_before[2] ? 2 : (row2 >= g.rows_before[1] ? 1 : 0);
            const int lrow2 = row2 - g.rows_before[s2];
            const int gw2 = s2 == 2 ? g.w[2] : (s2 == 1 ? g.w[1] : g.w[0]);
            const int hw2 = (s2 == 2 ? g.h[2] : (s2 == 1 ? g.h[1] : g.h[0])) * gw2;
            const int a2 = (lrow2 >= hw2) + (lrow2 >= 2 * hw2) + (lrow2 >= 3 * hw2);
            const int cell = lrow2 - a2 * hw2;
            const float inv = s2 == 2 ? eg.inv_w[2] : (s2 == 1 ? eg.inv_w[1] : eg.inv_w[0]);
            // floor(cell / w): (cell + 0.5) / w stays 0.5 / w away from an integer, far more than the
            // rounding error of the float product for cell < 2^21 and w < 2^12
            const int y = (int)(((float)cell + 0.5f) * inv);
            const int x = cell - y * gw2;
            const float st = s2 == 2 ? g.stride[2] : (s2 == 1 ? g.stride[1] : g.stride[0]);
            float an = 0.0f;
#pragma unroll
            for (int ss = 0; ss < 3; ++ss)
#pragma unroll
                for (int aa = 0; aa < 4; ++aa)
                    if (ss == s2 && aa == a2) an = kb == 2 ? g.anchor[ss][aa][0] : g.anchor[ss][aa][1];
            const float gc = kb == 0 ? (float)x : (float)y;
            const float t = sigmoid_f32(row_src[wv][u4][kb]);
            // decode_value: (2t - 0.5 + cell) * stride for the centre, 4 t^2 * anchor for the extent
            const float val = kb < 2 ? (2.0f * t - 0.5f + gc) * st : (4.0f * (t * t)) * an;
            const float wh = __shfl_down(val, 2);      // the extent that belongs to this centre column
            float r;
            if (kb < 2) {   // x_min = cx - w/2 ; cx' = x_min + w/2 ; scaled.  Same roundings as the reference.
                const float half = wh / 2.0f;
                r = ((val - half) + half) * (kb == 0 ? xs : ys);
            } else {
                r = val * (kb == 2 ? xs : ys);
            }
            if (j0 + u4 < nk) out[((size_t)b * max_keep + j0 + u4) * kdim + kb] = r;
        }
    }
}

// Index of the first maximum of v[0..n) (torch.argmax on a row without NaN).
__device__ __forceinline__ int first_argmax(const float* v, int n) {
    int best = 0;
    float m = v[0];
    for (int k = 1; k < n; ++k)
        if (v[k] > m) { m = v[k]; best = k; }
    return best;
}

constexpr int MATCH_MAX_PRED = 1024;

// True/false positive counting of the validation metric (train_class.py:347-360), one workgroup
// per image.  kept: (B, max_keep, K) rows (x_min,y_min,w,h,conf,cls..) after NMS; labels:
// (B, lmax, K) rows (cx,cy,w,h,conf,cls..).  A label is found when any kept row of its class
// overlaps it with IoU > thr (train_class.py:67-74 arithmetic); one found label = one true positive.
// out: (B, 3) int32 = (tp, fn, fp) with fn = labels - tp and fp = kept - tp.
__global__ void __launch_bounds__(256)
metric_match_kernel(const float* __restrict__ kept, const int* __restrict__ keep_count, int max_keep,
                    const float* __restrict__ labels, const int* __restrict__ n_label, int lmax, int kdim, float thr,
                    int* __restrict__ out) {
    __shared__ float4 pbox[MATCH_MAX_PRED];
    __shared__ int pcls[MATCH_MAX_PRED];
    __shared__ int tp;
    const int b = blockIdx.x;
    const int nk = min(keep_count[b], max_keep);
    const int nl = min(n_label[b], lmax);
    if (threadIdx.x == 0) tp = 0;
    for (int j = threadIdx.x; j < nk; j += 256) {
        const float* r = kept + ((size_t)b * max_keep + j) * kdim;
        pbox[j] = make_float4(r[0], r[1], r[2], r[3]);
        pcls[j] = first_argmax(r + 5, kdim - 5);
    }
    __syncthreads();
    for (int l = threadIdx.x; l < nl; l += 256) {
        const float* r = labels + ((size_t)b * lmax + l) * kdim;
        const float lw = r[2], lh = r[3];
        const float lx = r[0] - lw / 2.0f, ly = r[1] - lh / 2.0f;       // train_class.py:341
        const int cls = first_argmax(r + 5, kdim - 5);
        bool found = false;
        for (int j = 0; j < nk; ++j) {
            const float4 p = pbox[j];
            const float x1 = fmaxf(p.x, lx), y1 = fmaxf(p.y, ly);
            const float x2 = fminf(p.x + p.z, lx + lw), y2 = fminf(p.y + p.w, ly + lh);
            const float inter = fmaxf(x2 - x1, 0.0f) * fmaxf(y2 - y1, 0.0f);
            const float iou = inter / (p.z * p.w + lw * lh - inter);
            found = found || (iou > thr && pcls[j] == cls);
        }
        if (found) atomicAdd(&tp, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[3 * b] = tp;
        out[3 * b + 1] = nl - tp;
        out[3 * b + 2] = nk - tp;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
int next_pow2(int v) {
    int p = 64;
    while (p < v) p <<= 1;
    return p;
}

int fill_geom(const odamd_decode_desc* d, DecodeGeom* g, bool need_heads = true) {
    if (!d || d->n_anchor < 1 || d->n_anchor > 4 || d->batch < 1 || d->kdim < 6) return ODAMD_E_ARG;
    int rows = 0;
    for (int s = 0; s < 3; ++s) {
        if ((need_heads && !d->head[s]) || d->h[s] < 1 || d->w[s] < 1) return ODAMD_E_ARG;
        g->head[s] = d->head[s];
        g->h[s] = d->h[s];
        g->w[s] = d->w[s];
        g->stride[s] = d->stride[s];
        g->rows_before[s] = rows;
        rows += d->n_anchor * d->h[s] * d->w[s];
        for (int a = 0; a < 4; ++a)
            for (int c = 0; c < 2; ++c) g->anchor[s][a][c] = d->anchor_px[s][a][c];
    }
    g->n_anchor = d->n_anchor;
    g->kdim = d->kdim;
    g->rows = rows;
    g->deploy = d->deploy;
    return ODAMD_OK;
}

struct Workspace {
    u64* keys;
    float4* rowbox;
    float4* sbox;
    float4* sbox2;   // second copies of the sorted list: compact_live_kernel moves the survivors from one to the other
    int* srow2;
    float4* colpk;   // column operands of the mask kernel (pack_cols_kernel)
    float* colk;
    int* srow;
    int* perm;       // row order of the windowed mask kernel (area_order_kernel)
    u64* removed;
    u64* mask;
    int* count;
    int* count_pad;  // the filter kernel's counters, one 128-byte line per image
    size_t mask_stride_img;  // words per image
    int chunk_rows;
    size_t total;
};

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

Workspace carve(char* base, int batch, int rows, int n_pad) {
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes); return p; };
    w.keys = (u64*)take((size_t)batch * n_pad * 8);
    w.rowbox = (float4*)take((size_t)batch * rows * 16);
    w.sbox = (float4*)take((size_t)batch * n_pad * 16);
    w.colpk = (float4*)take((size_t)batch * n_pad * 16);
    w.colk = (float*)take((size_t)batch * n_pad * 4);
    w.srow = (int*)take((size_t)batch * n_pad * 4);
    w.sbox2 = (float4*)take((size_t)batch * n_pad * 16);
    w.srow2 = (int*)take((size_t)batch * n_pad * 4);
    w.perm = (int*)take((size_t)batch * n_pad * 4);
    w.removed = (u64*)take((size_t)batch * (n_pad / 64) * 8);
    w.count = (int*)take((size_t)batch * 4);
    w.count_pad = (int*)take((size_t)batch * COUNT_STRIDE * 4);
    // mask chunk: whole 64-row tiles, at most 64 MiB per image and 1 GiB per batch (16 MiB per image from 64 images up):
    // fewer, larger chunks for small batches of dense images, whose chunk launches are otherwise latency bound
    long long per_image = (1ll << 30) / batch;
    if (per_image > (64ll << 20)) per_image = 64ll << 20;
    if (per_image < (16ll << 20)) per_image = 16ll << 20;
    long long chunk = per_image * 8 / n_pad;
    chunk = (chunk / 64) * 64;
    if (chunk < 64) chunk = 64;
    if (chunk > n_pad) chunk = n_pad;
    w.chunk_rows = (int)chunk;
    w.mask_stride_img = (size_t)chunk * (n_pad / 64);
    w.mask = (u64*)take((size_t)batch * w.mask_stride_img * 8);
    w.total = off;
    return w;
}

int sort_keys(u64* keys, const int* count, int batch, int n_pad, hipStream_t st) {
    const int chunk = n_pad < SORT_CHUNK ? n_pad : SORT_CHUNK;
    dim3 grid_l(n_pad / chunk, batch);
    bitonic_lds_kernel<true><<<grid_l, SORT_THREADS, 0, st>>>(keys, count, n_pad, chunk, 0);
    for (int k = chunk * 2; k <= n_pad; k <<= 1) {
        for (int j = k >> 1; j >= chunk; j >>= 1)
            bitonic_global_kernel<<<dim3((n_pad / 2 + 255) / 256, batch), 256, 0, st>>>(keys, count, n_pad, chunk, k, j);
        bitonic_lds_kernel<false><<<grid_l, SORT_THREADS, 0, st>>>(keys, count, n_pad, chunk, k);
    }
    return (int)hipGetLastError();
}

int run_nms(const Workspace& w, int* count, int batch, int rows, int n_pad, float iou_thr, int corner_rule, int* out_idx,
            int* out_count, int max_keep, hipStream_t st) {
    const int words = n_pad / 64;
    // rows grouped by size inside 1024-row blocks need chunks made of whole blocks
    const bool windowed = n_pad % AREA_BLOCK == 0 && w.chunk_rows % AREA_BLOCK == 0;
    float4* sbox = w.sbox;
    float4* sbox_alt = w.sbox2;
    int* srow = w.srow;
    int* srow_alt = w.srow2;
    // an image has at most `rows` candidates: chunks beyond them would be empty launches.  Every pass handles rows
    // [0, chunk) of the CURRENT list: after a pass the decided rows and the suppressed candidates are dropped
    // (compact_live_kernel: `count` shrinks, `removed` restarts at zero), so with nothing suppressed at all pass k sees
    // the original rows [k * chunk, (k + 1) * chunk) - the host's loop bound is that worst case, passes over an empty list
    // return at once.
    const int r_end = rows < n_pad ? rows : n_pad;
    const int r1 = w.chunk_rows < n_pad ? w.chunk_rows : n_pad;
    for (int r0 = 0; r0 < r_end; r0 += w.chunk_rows) {
        pack_cols_kernel<<<dim3((n_pad + 255) / 256, batch), 256, 0, st>>>(sbox, count, n_pad, iou_thr, w.colpk, w.colk);
        // the pass handles rows [0, r1) only: their blocks alone are ordered
        if (windowed) area_order_kernel<AREA_BLOCK><<<dim3(r1 / AREA_BLOCK, batch), 256, 0, st>>>(sbox, count, n_pad, w.perm);
        const int gx = (words + MASK_WAVES - 1) / MASK_WAVES, gy = r1 / 64;
        // the windowed kernel does its per-row-group set-up once per workgroup row: more rows, fewer column strides
        // (about 256 workgroups per image either way)
        const int ly = windowed ? 64 : 32;
        const int lx = windowed ? (gy >= 64 ? 4 : 256 / gy) : 8;
        dim3 grid(gx < lx ? gx : lx, gy < ly ? gy : ly, batch);
        if (windowed)
            iou_mask_win_kernel<<<grid, 64 * MASK_WAVES, 0, st>>>(sbox, w.colpk, w.colk, w.perm, count, w.removed, n_pad,
                                                                   0, r1, iou_thr, corner_rule, w.mask,
                                                                   w.mask_stride_img);
        else
            iou_mask_kernel<true><<<grid, 64 * MASK_WAVES, 0, st>>>(sbox, w.colpk, w.colk, count, w.removed, n_pad, 0,
                                                                     r1, iou_thr, corner_rule, w.mask, w.mask_stride_img);
        nms_scan_kernel<<<batch, SCAN_THREADS, (size_t)words * 8, st>>>(w.mask, w.mask_stride_img, count, srow,
                                                                        w.removed, n_pad, 0, r1, out_idx, out_count,
                                                                        max_keep);
        if (r0 + w.chunk_rows < r_end) {
            compact_live_kernel<<<batch, COMPACT_THREADS, 0, st>>>(sbox, srow, sbox_alt, srow_alt, count, w.removed, n_pad, r1);
            std::swap(sbox, sbox_alt);
            std::swap(srow, srow_alt);
        }
    }
    return (int)hipGetLastError();
}

template <int SRC>
int postprocess_impl(const float* pred, const DecodeGeom& g, int batch, int rows, int kdim, float conf_thr,
                     float iou_thr, const float* scale_xy, int n_pad, float* out_rows, int* out_idx, int* out_count,
                     int* cand_count, int max_keep, void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (batch < 1 || rows < 1 || kdim < 6 || max_keep < 1 || !out_idx || !out_count || !workspace) return ODAMD_E_ARG;
    if (n_pad == 0) n_pad = next_pow2(rows);
    if (n_pad < 64 || (n_pad & (n_pad - 1)) || n_pad > (1 << 18)) return ODAMD_E_ARG;
    if (((uintptr_t)workspace & 255) != 0) return ODAMD_E_ALIGN;
    Workspace w = carve((char*)workspace, batch, rows, n_pad);
    if (w.total > workspace_bytes) return ODAMD_E_WORKSPACE;
    int rc;
    if ((rc = (int)hipMemsetAsync(w.keys, 0, (size_t)batch * n_pad * 8, st))) return rc;
    if ((rc = (int)hipMemsetAsync(w.removed, 0, (size_t)batch * (n_pad / 64) * 8, st))) return rc;
    if ((rc = (int)hipMemsetAsync(w.count_pad, 0, (size_t)batch * COUNT_STRIDE * 4, st))) return rc;
    if ((rc = (int)hipMemsetAsync(out_count, 0, (size_t)batch * 4, st))) return rc;
    // rows staged per workgroup: one wave's worth (about 22 KiB at 85 columns, so seven workgroups
    // per CU keep enough loads in flight), fewer when rows are very wide
    const int ldk = kdim | 1;
    int rows_stage = (48 * 1024) / (ldk * 4);
    if (rows_stage > 64) rows_stage = 64;
    if (rows_stage < 1) return ODAMD_E_ARG;  // more than ~12k classes
    FilterSrc fs = {};
    int nstage = 0;
    if (SRC == 1) {
        for (int s = 0; s < 3; ++s) {
            fs.base[s] = g.head[s];
            fs.rows_in[s] = g.n_anchor * g.h[s] * g.w[s];
            fs.row_first[s] = g.rows_before[s];
            fs.stage_first[s] = nstage;
            nstage += (fs.rows_in[s] + rows_stage - 1) / rows_stage;
        }
    } else {
        fs.base[0] = pred;
        fs.rows_in[0] = rows;
        nstage = (rows + rows_stage - 1) / rows_stage;
        fs.stage_first[1] = fs.stage_first[2] = nstage;   // ranges 1, 2 are empty
    }
    dim3 fgrid(nstage, batch);
    filter_kernel<SRC><<<fgrid, FILTER_THREADS, (size_t)rows_stage * ldk * 4, st>>>(
        fs, g, rows, kdim, rows_stage, conf_thr, w.rowbox, w.keys, n_pad, w.count_pad);
    compact_counts_kernel<<<(batch + 255) / 256, 256, 0, st>>>(w.count_pad, batch, n_pad, w.count);
    if ((rc = sort_keys(w.keys, w.count, batch, n_pad, st))) return rc;
    gather_sorted_kernel<<<dim3((n_pad + 255) / 256, batch), 256, 0, st>>>(w.keys, w.rowbox, w.count, rows, n_pad,
                                                                           SRC >= 3, w.sbox, w.srow);
    if (cand_count)     // before the greedy pass: it shrinks the list (and its count) as it goes
        if ((rc = (int)hipMemcpyAsync(cand_count, w.count, (size_t)batch * 4, hipMemcpyDeviceToDevice, st))) return rc;
    if ((rc = run_nms(w, w.count, batch, rows, n_pad, iou_thr, SRC >= 3, out_idx, out_count, max_keep, st))) return rc;
    if (out_rows) {
        const int gx = (max_keep + 4 * EMIT_ROWS - 1) / (4 * EMIT_ROWS);
        bool lean = SRC == 1;
        EmitGeom eg = {};
        for (int s = 0; s < 3 && lean; ++s) {
            lean = g.w[s] < 4096 && (long long)g.h[s] * g.w[s] < (1 << 21);
            eg.inv_w[s] = 1.0f / (float)g.w[s];
        }
        lean = lean && kdim - 4 < 4096;
        eg.inv_cls = 1.0f / (float)(kdim - 4);
        if (lean)
            emit_heads_kernel<<<dim3(gx > 512 ? 512 : gx, batch), 256, 0, st>>>(g, eg, kdim, out_idx, out_count, max_keep,
                                                                               scale_xy, out_rows);
        else
            emit_rows_kernel<SRC><<<dim3(gx > 512 ? 512 : gx, batch), 256, 0, st>>>(pred, g, rows, kdim, out_idx,
                                                                                       out_count, max_keep, scale_xy,
                                                                                       out_rows);
    }
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int odamd_decode(const odamd_decode_desc* d, float* out, void* stream) {
    DecodeGeom g;
    int rc = fill_geom(d, &g);
    if (rc) return rc;
    if (!out) return ODAMD_E_ARG;
    int maxn = 0;
    for (int s = 0; s < 3; ++s) maxn = max(maxn, d->n_anchor * d->h[s] * d->w[s] * d->kdim);
    int gx = (maxn + 255) / 256;
    if (gx > 2048) gx = 2048;
    decode_kernel<<<dim3(gx, 3, d->batch), 256, 0, (hipStream_t)stream>>>(g, out);
    return (int)hipGetLastError();
}

int odamd_preprocess_u8(const uint8_t* image_hwc, int h, int w, float* out_chw, int size, void* stream) {
    if (!image_hwc || !out_chw || h < 1 || w < 1 || size < 1 || h > 32768 || w > 32768) return ODAMD_E_ARG;
    preprocess_u8_kernel<<<(size * size + 255) / 256, 256, 0, (hipStream_t)stream>>>(image_hwc, h, w, out_chw, size);
    return (int)hipGetLastError();
}

int odamd_preprocess_u8_batch(const uint8_t* images_hwc, int batch, int h, int w, float* out_chw, int size, void* stream) {
    if (!images_hwc || !out_chw || batch < 1 || batch > 65535 || h < 1 || w < 1 || size < 1 || size > 32768) return ODAMD_E_ARG;
    preprocess_u8_kernel<<<dim3((size * size + 255) / 256, batch), 256, 0, (hipStream_t)stream>>>(images_hwc, h, w, out_chw, size);
    return (int)hipGetLastError();
}

int odamd_assign_labels(const odamd_decode_desc* geometry, float sig_lo, float sig_hi, const float* labels,
                        const int* n_label, int lmax, int* screen, int* screen_img, float* label_out, int capacity,
                        int* n_pos, int* totals, void* stream) {
    DecodeGeom g;
    int rc = fill_geom(geometry, &g, false);
    if (rc) return rc;
    if (!labels || !n_label || !screen || !screen_img || !label_out || !n_pos || !totals || lmax < 1 || capacity < 1)
        return ODAMD_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    assign_labels_kernel<false><<<geometry->batch, 256, 0, st>>>(g, sig_lo, sig_hi, labels, n_label, lmax, totals, nullptr,
                                                                 nullptr, nullptr, capacity, n_pos, geometry->batch);
    assign_labels_kernel<true><<<geometry->batch, 256, 0, st>>>(g, sig_lo, sig_hi, labels, n_label, lmax, totals, screen,
                                                                screen_img, label_out, capacity, n_pos, geometry->batch);
    return (int)hipGetLastError();
}

int odamd_decode_backward(const odamd_decode_desc* d, const float* grad_out, float* const grad_head[3], void* stream) {
    DecodeGeom g;
    int rc = fill_geom(d, &g);
    if (rc) return rc;
    if (!grad_out || !grad_head || !grad_head[0] || !grad_head[1] || !grad_head[2] || d->deploy) return ODAMD_E_ARG;
    int maxn = 0;
    for (int s = 0; s < 3; ++s) maxn = max(maxn, d->n_anchor * d->h[s] * d->w[s] * d->kdim);
    // four elements per thread of the largest map: the grid is the same for all three maps, and most workgroups of a
    // one-element-per-thread grid belonged to the two small ones and had nothing to do
    int gx = (maxn + 1023) / 1024;
    if (gx > 2048) gx = 2048;
    decode_bwd_kernel<<<dim3(gx, 3, d->batch), 256, 0, (hipStream_t)stream>>>(g, grad_out, grad_head[0], grad_head[1],
                                                                              grad_head[2]);
    return (int)hipGetLastError();
}

int odamd_deploy_sigmoid(float* pred, size_t rows, int kdim, void* stream) {
    if (!pred || rows < 1 || kdim < 5) return ODAMD_E_ARG;
    const size_t n = rows * (size_t)kdim;
    size_t gx = (n + 255) / 256;
    if (gx > 4096) gx = 4096;
    deploy_sigmoid_kernel<<<(int)gx, 256, 0, (hipStream_t)stream>>>(pred, n, kdim);
    return (int)hipGetLastError();
}

size_t odamd_postprocess_workspace_bytes(int batch, int rows, int kdim, int n_pad) {
    (void)kdim;
    if (batch < 1 || rows < 1) return 0;
    if (n_pad == 0) n_pad = next_pow2(rows);
    return carve(nullptr, batch, rows, n_pad).total;
}

int odamd_postprocess(const float* pred, int batch, int rows, int kdim, float conf_thr, float iou_thr,
                      const float* scale_xy, int n_pad, float* out_rows, int* out_idx, int* out_count, int* cand_count,
                      int max_keep, void* workspace, size_t workspace_bytes, void* stream) {
    if (!pred) return ODAMD_E_ARG;
    DecodeGeom g = {};
    return postprocess_impl<0>(pred, g, batch, rows, kdim, conf_thr, iou_thr, scale_xy, n_pad, out_rows, out_idx,
                               out_count, cand_count, max_keep, workspace, workspace_bytes, (hipStream_t)stream);
}

int odamd_postprocess_heads(const odamd_decode_desc* d, float conf_thr, float iou_thr, const float* scale_xy, int n_pad,
                            float* out_rows, int* out_idx, int* out_count, int* cand_count, int max_keep,
                            void* workspace, size_t workspace_bytes, void* stream) {
    DecodeGeom g;
    int rc = fill_geom(d, &g);
    if (rc) return rc;
    g.deploy = 1;
    return postprocess_impl<1>(nullptr, g, d->batch, g.rows, d->kdim, conf_thr, iou_thr, scale_xy, n_pad, out_rows,
                               out_idx, out_count, cand_count, max_keep, workspace, workspace_bytes,
                               (hipStream_t)stream);
}

int odamd_nms_rows(const float* rows_xywh, int batch, int rows, int kdim, float iou_thr, int n_pad, float* out_rows,
                   int* out_idx, int* out_count, int max_keep, void* workspace, size_t workspace_bytes, void* stream) {
    if (!rows_xywh) return ODAMD_E_ARG;
    DecodeGeom g = {};
    return postprocess_impl<2>(rows_xywh, g, batch, rows, kdim, 0.0f, iou_thr, nullptr, n_pad, out_rows, out_idx,
                               out_count, nullptr, max_keep, workspace, workspace_bytes, (hipStream_t)stream);
}

int odamd_nms_rows_static(const float* rows_xywh, int batch, int rows, int kdim, float iou_thr, float* out_rows,
                          int* out_idx, int* out_count, int max_keep, void* stream) {
    if (!rows_xywh || batch < 1 || rows < 1) return ODAMD_E_ARG;
    int n_pad = next_pow2(rows);
    if (n_pad < 64) n_pad = 64;
    if (n_pad > (1 << 18)) return ODAMD_E_LIMIT;
    // the same workspace layout as odamd_nms_rows, carved out of this GPU's static array
    const size_t need = carve(nullptr, batch, rows, n_pad).total;
    if (need > STATIC_SCRATCH_BYTES) return ODAMD_E_LIMIT;
    void* ws = nullptr;
    const int rc = static_scratch(&ws);
    if (rc) return rc;
    DecodeGeom g = {};
    return postprocess_impl<2>(rows_xywh, g, batch, rows, kdim, 0.0f, iou_thr, nullptr, n_pad, out_rows, out_idx,
                               out_count, nullptr, max_keep, ws, STATIC_SCRATCH_BYTES, (hipStream_t)stream);
}

int odamd_metric_nms(const float* pred, int batch, int rows, int kdim, float conf_thr, float nms_thr, int n_pad,
                     float* out_rows, int* out_idx, int* out_count, int* cand_count, int max_keep, void* workspace,
                     size_t workspace_bytes, void* stream) {
    if (!pred) return ODAMD_E_ARG;
    DecodeGeom g = {};
    return postprocess_impl<3>(pred, g, batch, rows, kdim, conf_thr, nms_thr, nullptr, n_pad, out_rows, out_idx,
                               out_count, cand_count, max_keep, workspace, workspace_bytes, (hipStream_t)stream);
}

int odamd_metric_nms_rows(const float* rows_xywh, int batch, int rows, int kdim, float nms_thr, int n_pad,
                          float* out_rows, int* out_idx, int* out_count, int max_keep, void* workspace,
                          size_t workspace_bytes, void* stream) {
    if (!rows_xywh) return ODAMD_E_ARG;
    DecodeGeom g = {};
    return postprocess_impl<4>(rows_xywh, g, batch, rows, kdim, 0.0f, nms_thr, nullptr, n_pad, out_rows, out_idx,
                               out_count, nullptr, max_keep, workspace, workspace_bytes, (hipStream_t)stream);
}

int odamd_metric_match(const float* kept_rows, const int* keep_count, int max_keep, const float* labels,
                       const int* n_label, int lmax, int batch, int kdim, float iou_thr, int* counts, void* stream) {
    if (!kept_rows || !keep_count || !labels || !n_label || !counts || batch < 1 || kdim < 6 || lmax < 1 ||
        max_keep < 1)
        return ODAMD_E_ARG;
    if (max_keep > MATCH_MAX_PRED) return ODAMD_E_SHAPE;
    metric_match_kernel<<<batch, 256, 0, (hipStream_t)stream>>>(kept_rows, keep_count, max_keep, labels, n_label, lmax,
                                                               kdim, iou_thr, counts);
    return (int)hipGetLastError();
}

int odamd_iou_pairs(const float* a_xywh, int na, const float* b_xywh, int nb, float* out, int n, void* stream) {
    if (!a_xywh || !b_xywh || !out || n < 1 || na < 1 || nb < 1) return ODAMD_E_ARG;
    if ((na != 1 && na < n) || (nb != 1 && nb < n)) return ODAMD_E_SHAPE;
    if (((uintptr_t)a_xywh & 15) || ((uintptr_t)b_xywh & 15)) return ODAMD_E_ALIGN;
    iou_pairs_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>((const float4*)a_xywh, na, (const float4*)b_xywh,
                                                                       nb, out, n);
    return (int)hipGetLastError();
}

int odamd_iou_mask(const float* boxes_xywh, int n, float iou_thr, uint64_t* mask, void* stream) {
    if (!boxes_xywh || !mask || n < 64 || (n & 63)) return ODAMD_E_ARG;
    if (((uintptr_t)boxes_xywh & 15) != 0) return ODAMD_E_ALIGN;
    const int words = n / 64;
    const int gx = (words + MASK_WAVES - 1) / MASK_WAVES;
    dim3 grid(gx < 8 ? gx : 8, words < 64 ? words : 64, 1);
    iou_mask_kernel<false><<<grid, 64 * MASK_WAVES, 0, (hipStream_t)stream>>>((const float4*)boxes_xywh, nullptr, nullptr,
                                                                              nullptr, nullptr, n, 0, n, iou_thr, 0, (u64*)mask, 0);
    return (int)hipGetLastError();
}

}  // extern "C"

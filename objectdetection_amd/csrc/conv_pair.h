// Two consecutive 3x3 stride-1 convolutions in one launch (bf16, NHWC, 32 channels between and after them):
// the chains of elan / elan_h (model/layer.py:74-81, 109-118) on the wide maps, where each layer alone is a short,
// HBM- and latency-bound launch that writes 64 bytes of every 256-byte pixel and reads them straight back.
// Included by conv.hip inside its anonymous namespace.
//
// A workgroup (4 waves) owns a 16 x 16 output tile (and, one after the other, `seg` vertically adjacent ones):
//   1. the 20 x 20 input patch travels global -> LDS by LDS-DMA, 4 channel parts x 16 slots per instruction, in the
//      [slot / 16][part][slot % 16] order of conv_s3_kernel (a fragment read covers 256 bytes of distinct banks);
//   2. the first convolution is computed on the 18 x 18 halo-extended tile (12 pixel groups of 32, three per wave),
//      affine + SiLU, pixels outside the image forced to zero (they are the second convolution's zero padding);
//      v_permlane32_swap gives every lane 8 consecutive channels of its pixel, which go to LDS (over the patch, after a
//      barrier) in the same slot order and, for the 16 x 16 interior, to the first layer's output in HBM;
//   3. the second convolution reads that LDS image (8 pixel groups, two per wave) and stores its tile.
// Both weight tensors (packed by the caller in fragment order, odamd_conv_weight_layout's form with one block and one
// chunk) stay in LDS for the workgroup's life, the folded BatchNorm constants as well.

#ifdef ODAMD_LAB_STAMPS
#define PAIR_STAMP(i) do { if (lane == 0 && blockIdx.x < 1024 && k == 1) g_lab_stamps[(blockIdx.x * 8 + wave) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PAIR_STAMP(i) do { } while (0)
#endif

struct PairP {
    const char* x; const char* w1; const char* w2;
    const float* sc1; const float* sh1; const float* sc2; const float* sh2;
    char* y1; char* y2;
    int B, H, W, x_ld, y1_ld, y2_ld, act1, act2;
    int tiles_x, tiles_y, seg, runs_y, nblocks;
    unsigned y1bytes, y2bytes;
};

constexpr int PR_MW = 18, PR_IW = 20, PR_MID = 324, PR_IN_BLOCKS = 25, PR_MID_BLOCKS = 21;

template <int NGRP, int KS>
__device__ __forceinline__ void pair_mma(f32x16_t (&acc)[NGRP], const int (&baddr)[NGRP][9], const char* wl_lane,
                                         const char* src) {
    constexpr int DT = ODAMD_DTYPE_BF16;
    u32x4 a[2], bb[2][NGRP];
    auto rd = [&](int g, int slot) {
        const int tap = g / KS, s = g % KS;
        a[slot] = *(const u32x4*)(wl_lane + g * 1024);
#pragma unroll
        for (int i = 0; i < NGRP; ++i) bb[slot][i] = *(const u32x4*)(src + baddr[i][tap] + s * 512);
    };
    rd(0, 0);
    static_for<0, 9 * KS>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g + 1 < 9 * KS) rd(g + 1, (g + 1) & 1);
#pragma unroll
        for (int i = 0; i < NGRP; ++i) mma_step<DT>(acc[i], a[g & 1], bb[g & 1][i]);
    });
}

// Half q2 (channels 16 q2 .. 16 q2 + 15) of an accumulator tile: affine (+ SiLU), bf16, half-wave exchange.  Lane (r, h)
// returns channels 16 q2 + 8 h .. + 7 of pixel r.  `aff`: [scale 32][shift 32] floats in LDS.
__device__ __forceinline__ u32x4 pair_unit(const f32x16_t& ac, int q2, int h, const float* aff, bool act) {
    constexpr int DT = ODAMD_DTYPE_BF16;
    uint32_t pk[4];
#pragma unroll
    for (int g = 0; g < 2; ++g) {           // elements 8 q2 + 4 g .. + 3: channels 16 q2 + 8 g + 4 h .. + 3
        const int c = 16 * q2 + 8 * g + 4 * h;
        const float4 sc = *(const float4*)(aff + c), sh = *(const float4*)(aff + 32 + c);
        float t0 = ac[8 * q2 + 4 * g] * sc.x + sh.x, t1 = ac[8 * q2 + 4 * g + 1] * sc.y + sh.y;
        float t2 = ac[8 * q2 + 4 * g + 2] * sc.z + sh.z, t3 = ac[8 * q2 + 4 * g + 3] * sc.w + sh.w;
        if (act) { t0 = silu<DT>(t0); t1 = silu<DT>(t1); t2 = silu<DT>(t2); t3 = silu<DT>(t3); }
        pk[2 * g] = f32x2_to_bf16x2(t0, t1);
        pk[2 * g + 1] = f32x2_to_bf16x2(t2, t3);
    }
    const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
    return u32x4{s0[0], s1[0], s0[1], s1[1]};
}

// KS1: input channels of the first convolution / 16 (2 or 4)
template <int KS1>
__global__ void __launch_bounds__(256) conv_pair_kernel(PairP p) {
    constexpr int PARTS1 = 2 * KS1, PQ = PARTS1 / 4;
    constexpr int NPIECE = PR_IN_BLOCKS * PQ, MAXJ = (NPIECE + 3) / 4;
    constexpr int W1_BYTES = 9 * KS1 * 1024, W2_BYTES = 18 * 1024, BUF_BYTES = PR_IN_BLOCKS * PARTS1 * 256;
    static_assert(BUF_BYTES >= PR_MID_BLOCKS * 4 * 256, "the intermediate tile reuses the patch buffer");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* wl1 = lds;
    char* wl2 = lds + W1_BYTES;
    char* buf = wl2 + W2_BYTES;
    float* aff = (float*)(buf + BUF_BYTES);       // [layer][scale 32 | shift 32]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    const int lid = xcd_remap(blockIdx.x, p.nblocks);
    const int tx = lid % p.tiles_x, rest = lid / p.tiles_x;
    const int ry = rest % p.runs_y, b = rest / p.runs_y;
    const int ox0 = tx * 16;

    // ---- weights (fragment order: piece (tap, s) of 1 KiB, lane (r, h) holds channels 16 s + 8 h .. + 7 of output
    // channel r: a plain copy) and the affine constants ------------------------------------------------------------
    for (int i = wave; i < 9 * KS1; i += 4) dma16(p.w1 + i * 1024 + lane * 16, wl1 + i * 1024);
    for (int i = wave; i < 18; i += 4) dma16(p.w2 + i * 1024 + lane * 16, wl2 + i * 1024);
    if (tid < 128) {
        const float* src = tid < 32 ? p.sc1 : tid < 64 ? p.sh1 : tid < 96 ? p.sc2 : p.sh2;
        aff[tid] = src[tid & 31];
    }

    // ---- this wave's pieces of a patch (tile independent): 4 parts of 16 consecutive slots each ---------------------
    int pyx[MAXJ];
#pragma unroll
    for (int jj = 0; jj < MAXJ; ++jj) {
        const int i = wave + jj * 4;
        const int g16 = i / PQ, pq = i - g16 * PQ;
        const int part = pq * 4 + (lane >> 4);
        const int q = 16 * g16 + (lane & 15);
        const int iy = q / PR_IW, ix = q - iy * PR_IW;
        pyx[jj] = iy << 12 | ix << 4 | part;
    }
    const char* zl = (const char*)g_zero_line;
    const char* xb = p.x + (size_t)b * p.H * p.W * p.x_ld * 2;
    auto issue_patch = [&](int oy0) {
#pragma unroll
        for (int jj = 0; jj < MAXJ; ++jj) {
            const int i = wave + jj * 4;
            if (NPIECE % 4 == 0 || i < NPIECE) {
                const int g16 = i / PQ, pq = i - g16 * PQ;
                const int gy = oy0 - 2 + (pyx[jj] >> 12), gx = ox0 - 2 + ((pyx[jj] >> 4) & 255);
                const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                dma16(inside ? xb + ((size_t)(gy * p.W + gx) * p.x_ld) * 2 + (pyx[jj] & 15) * 16 : zl,
                      buf + (g16 * PARTS1 + pq * 4) * 256);
            }
        }
    };

    // ---- per-lane fragment addresses -------------------------------------------------------------------------------
    // first convolution: pixel q of the 18-wide intermediate tile reads patch slot (my + ky) * 20 + mx + kx
    int b1[3][9], my_[3], mx_[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = (wave * 3 + i) * 32 + r;
        const int my = q / PR_MW, mx = q - my * PR_MW;
        my_[i] = q < PR_MID ? my : -100;            // a lane past the tile: outside every image
        mx_[i] = mx;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int slot = q < PR_MID ? (my + tap / 3) * PR_IW + mx + tap % 3 : 0;
            b1[i][tap] = ((slot >> 4) * PARTS1 + h) * 256 + (slot & 15) * 16;
        }
    }
    // second convolution: pixel (ty, tx) of the 16 x 16 tile reads intermediate slot (ty + ky) * 18 + tx + kx
    int b2[2][9];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = (wave * 2 + i) * 32 + r;
        const int ty = q >> 4, txx = q & 15;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int slot = (ty + tap / 3) * PR_MW + txx + tap % 3;
            b2[i][tap] = ((slot >> 4) * 4 + h) * 256 + (slot & 15) * 16;
        }
    }
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(p.y1, 0, p.y1bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(p.y2, 0, p.y2bytes, 0x00020000);

    const int t_first = ry * p.seg;
    issue_patch(t_first * 16);
    for (int k = 0; k < p.seg; ++k) {
        const int tyi = t_first + k;
        if (tyi >= p.tiles_y) break;
        const int oy0 = tyi * 16;
        PAIR_STAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PAIR_STAMP(1);
        __syncthreads();                                  // patch (and, the first time, weights and constants) in LDS

        f32x16_t acc1[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[i][e] = 0.0f;
        PAIR_STAMP(2);
        pair_mma<3, KS1>(acc1, b1, wl1 + lane * 16, buf);
        PAIR_STAMP(3);
        __syncthreads();                                  // every patch read is done: the buffer becomes the intermediate tile
        PAIR_STAMP(4);

#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int q = (wave * 3 + i) * 32 + r;
            const int gy = oy0 - 1 + my_[i], gx = ox0 - 1 + mx_[i];
            const bool in_img = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const bool interior = in_img && my_[i] >= 1 && my_[i] <= 16 && mx_[i] >= 1 && mx_[i] <= 16;
            const int pix = (b * p.H + gy) * p.W + gx;
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                u32x4 v = pair_unit(acc1[i], q2, h, aff, p.act1);
                if (!in_img) v = u32x4{0u, 0u, 0u, 0u};
                if (q < PR_MID) *(u32x4*)(buf + ((q >> 4) * 4 + 2 * q2 + h) * 256 + (q & 15) * 16) = v;
                __builtin_amdgcn_raw_buffer_store_b128(v, rs1, interior ? (pix * p.y1_ld + 16 * q2 + 8 * h) * 2 : -1, 0, 0);
            }
        }
        PAIR_STAMP(5);
        __syncthreads();                                  // the intermediate tile is complete
        PAIR_STAMP(6);

        f32x16_t acc2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[i][e] = 0.0f;
        pair_mma<2, 2>(acc2, b2, wl2 + lane * 16, buf);
        PAIR_STAMP(7);
        __syncthreads();                                  // the intermediate tile has been read: the next patch may land
        PAIR_STAMP(8);
        if (k + 1 < p.seg && tyi + 1 < p.tiles_y) issue_patch(oy0 + 16);
        PAIR_STAMP(9);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = (wave * 2 + i) * 32 + r;
            const int oy = oy0 + (q >> 4), ox = ox0 + (q & 15);
            const bool ok = oy < p.H && ox < p.W;
            const int pix = (b * p.H + oy) * p.W + ox;
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const u32x4 v = pair_unit(acc2[i], q2, h, aff + 64, p.act2);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs2, ok ? (pix * p.y2_ld + 16 * q2 + 8 * h) * 2 : -1, 0, 0);
            }
        }
        PAIR_STAMP(10);
    }
}

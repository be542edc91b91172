// LDS-DMA convolution (bf16, 1x1 / 3x3 stride 1, 3x3 stride 2, NHWC in and out) for the deep, small-map layers.
// Included by conv.hip inside its anonymous namespace.
//
// Same GEMM view as conv_igemm_kernel (D[cout][pixel], A = weights, B = pixels, one
// v_mfma_f32_32x32x16_bf16 per 16 input channels), different data movement:
//   * nothing is staged through registers: every operand goes global -> LDS with
//     global_load_lds_dwordx4 (1 KiB per wave instruction), NBUF LDS buffers deep, one raw s_barrier
//     per input-channel chunk and a counted s_waitcnt vmcnt, so the loads of chunk c+1 (and c+2)
//     are in flight while the matrix cores work on chunk c;
//   * the weights are packed by the caller in the order the MFMA fragments are read
//     ([cout block][chunk][cout group][tap][k-step][lane][8 bf16], see odamd_conv_weight_layout), so a
//     chunk's weights are ONE contiguous run in memory, their LDS image is a plain copy and every
//     A-fragment read is lane-linear (conflict free);
//   * the input halo patch is staged as [16-byte channel part][pixel] planes: an LDS-DMA piece is 64
//     consecutive patch pixels of one part (its per-lane source address does the gather, pixels outside
//     the image read a zero line), and a B-fragment read touches consecutive 16-byte slots.
// A 1x1 convolution has no spatial structure: the host passes the whole batch as one 1 x (B*H*W) image.
// Stride 2 keeps the even and the odd columns of a patch row apart ([IWH even][IWH odd] slots per row), so that the
// 32 consecutive output pixels of a B fragment still read consecutive slots for every tap.

#ifdef ODAMD_LAB_STAMPS
__device__ unsigned long long g_lab_stamps[1024 * 8 * 16];
#define LAB_STAMP(i) do { if (lane == 0 && blockIdx.x < 1024) g_lab_stamps[(blockIdx.x * 8 + wave) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define LAB_STAMP(i) do { } while (0)
#endif

struct DmaP {
    const char* x; const char* w; const float* scale; const float* shift; char* y;
    int B, H, W, Cin, Cout, x_ld, y_ld;
    int Ho, Wo;   // output size (= H, W for stride 1)
    int IWP;      // slots per patch row: IW, or 2 * IWH for stride 2
    int IWH;      // stride 2: even columns of a patch row (the odd ones follow)
    int TH, TW, tiles_x, tiles_y, ncb, IH, IW;
    int npad;     // staged pixels per part plane: IH * IW rounded up to 64
    int nchunk, act, nblocks;
    float inv_tw, inv_iw;
    float inv_ncb, inv_tiles, inv_tiles_x;   // reciprocals for the block-index split (div_small)
    int fast_epi; unsigned ybytes;           // branch-free epilogue (see nhwc_epilogue_fast)
};

__device__ __forceinline__ void dma16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define ODAMD_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        ODAMD_W(0) ODAMD_W(1) ODAMD_W(2) ODAMD_W(3) ODAMD_W(4) ODAMD_W(5) ODAMD_W(6) ODAMD_W(7) ODAMD_W(8) ODAMD_W(9)
        ODAMD_W(10) ODAMD_W(11) ODAMD_W(12) ODAMD_W(13) ODAMD_W(14) ODAMD_W(15) ODAMD_W(16) ODAMD_W(17) ODAMD_W(18)
        ODAMD_W(19) ODAMD_W(20) ODAMD_W(21) ODAMD_W(22) ODAMD_W(23) ODAMD_W(24) ODAMD_W(25) ODAMD_W(26) ODAMD_W(27)
        ODAMD_W(28) ODAMD_W(29) ODAMD_W(30) ODAMD_W(31) ODAMD_W(32)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef ODAMD_W
}

// s_waitcnt vmcnt(N) with a compile-time N
template <int N>
__device__ __forceinline__ void wait_vmcnt_imm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// upper bound of staged input pixels per workgroup (the host tile chooser enforces it)
__host__ __device__ constexpr int dma_max_in_pix(int ks, int pg, int st = 1) {
    return ks == 1 ? 32 * pg : st == 2 ? (pg >= 8 ? 1280 : 704) : (pg >= 8 ? 384 : 256);
}

template <int KS, int PG, int CG, int WP, int WC, int CKB, int NBUF, int ST = 1>
__global__ void __launch_bounds__(64 * WP * WC) conv_dma_kernel(DmaP p) {
    constexpr int DT = ODAMD_DTYPE_BF16;
    constexpr int TAPS = KS * KS, NW = WP * WC, PGW = PG / WP, CGW = CG / WC;
    constexpr int PARTS = CKB / 16, KSTEPS = CKB / 32;
    constexpr int W_PIECES = CG * TAPS * KSTEPS;                     // 1 KiB each
    constexpr int MAX_NPJ = (dma_max_in_pix(KS, PG, ST) + 63) / 64;
    constexpr int MAXJ = (PARTS * MAX_NPJ + NW - 1) / NW;            // input pieces per wave and chunk
    constexpr int MAXW = (W_PIECES + NW - 1) / NW;                   // weight pieces per wave and chunk
    constexpr int PAD = (KS - 1) / 2;
    static_assert(PG % WP == 0 && CG % WC == 0, "wave grid");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int in_bytes = PARTS * p.npad * 16;
    const int buf_bytes = in_bytes + W_PIECES * 1024;

    // block index -> (image, tile, cout block); integer divisions by run-time values cost ~300 dependent
    // cycles each on this path, the float reciprocal form three instructions (exact for these ranges)
    const int lid = xcd_remap(blockIdx.x, p.nblocks);
    const int t1 = div_small(lid, p.inv_ncb);
    const int cb = lid - t1 * p.ncb;
    const int b = div_small(t1, p.inv_tiles);
    const int tile = t1 - b * (p.tiles_x * p.tiles_y);
    const int trow = div_small(tile, p.inv_tiles_x);
    const int oy0 = trow * p.TH, ox0 = (tile - trow * p.tiles_x) * p.TW;
    const int co0 = cb * 32 * CG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wp = wave / WC, wc = wave % WC;
    LAB_STAMP(0);

    // ---- this wave's share of a chunk's DMA pieces ------------------------------------------------
    const int n_in = PARTS * ((p.IH * p.IWP + 63) >> 6);
    const int iy0 = oy0 * ST - PAD, ix0 = ox0 * ST - PAD;
    int goff[MAXJ];            // byte offset of this lane's source pixel within the image, -1: zero line
    int loff[MAXJ];            // LDS offset of the piece within a buffer (wave uniform)
    int n_mine = 0;
#pragma unroll
    for (int jj = 0; jj < MAXJ; ++jj) {
        const int i = wave + jj * NW;
        const int j = i / PARTS, part = i - j * PARTS;
        const int q = 64 * j + lane;
        const int iy = div_small(q, p.inv_iw);
        int ix = q - iy * p.IWP;
        if (ST == 2) ix = ix < p.IWH ? 2 * ix : 2 * (ix - p.IWH) + 1;     // slot -> patch column
        const int gy = iy0 + iy, gx = ix0 + ix;
        const bool inside = i < n_in && iy < p.IH && ix < p.IW && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        goff[jj] = inside ? (gy * p.W + gx) * p.x_ld * 2 + part * 16 : -1;
        loff[jj] = (part * p.npad + 64 * j) * 16;
        n_mine += i < n_in;
    }
#pragma unroll
    for (int jj = 0; jj < MAXW; ++jj) n_mine += (W_PIECES % NW == 0) || wave + jj * NW < W_PIECES;
    const char* xb = p.x + (size_t)b * p.H * p.W * p.x_ld * 2;
    const char* wb = p.w + (size_t)cb * p.nchunk * (W_PIECES * 1024) + lane * 16;
    const char* zl = (const char*)g_zero_line;

    // DMA piece `k` (0 .. MAXJ + MAXW - 1) of this wave for chunk c into buffer `buf`; k is a compile-time
    // constant at every call site, so that the pieces can be spread between the MFMA groups of a chunk
    auto issue_piece = [&](int k, int c, char* buf) {
        if (k < MAXJ) {
            if (wave + k * NW < n_in) dma16(goff[k] >= 0 ? xb + goff[k] + c * CKB : zl, buf + loff[k]);
        } else {
            const int i = wave + (k - MAXJ) * NW;
            if ((W_PIECES % NW == 0) || i < W_PIECES)
                dma16(wb + (size_t)c * (W_PIECES * 1024) + i * 1024, buf + in_bytes + i * 1024);
        }
    };
    auto issue = [&](int c, int bi) {
        char* buf = lds + bi * buf_bytes;
#pragma unroll
        for (int k = 0; k < MAXJ + MAXW; ++k) issue_piece(k, c, buf);
    };

    // the first chunks are on their way while the rest of the set-up runs
    const int n = p.nchunk;
    LAB_STAMP(1);
#pragma unroll
    for (int c = 0; c < NBUF - 1; ++c)
        if (c < n) issue(c, c);
    LAB_STAMP(2);

    // ---- per-lane fragment offsets -----------------------------------------------------------------
    int b_off[PGW], opix[PGW];
#pragma unroll
    for (int i = 0; i < PGW; ++i) {
        const int q = (wp * PGW + i) * 32 + r;
        int ty = div_small(q, p.inv_tw), tx = q - ty * p.TW;
        const int oy = oy0 + ty, ox = ox0 + tx;
        opix[i] = (ty < p.TH && oy < p.Ho && ox < p.Wo) ? oy * p.Wo + ox : -1;
        if (ty >= p.TH) { ty = 0; tx = 0; }
        b_off[i] = ((ty * ST * p.IWP + tx) + h * p.npad) * 16;
    }
    const int a_off = in_bytes + wc * CGW * TAPS * KSTEPS * 1024 + lane * 16;

    Affine<DT, CGW> aff;
    aff.load(p.scale, p.shift, lane, co0 + wc * CGW * 32, p.Cout);

    f32x16_t acc[PGW][CGW];
#pragma unroll
    for (int i = 0; i < PGW; ++i)
#pragma unroll
        for (int j = 0; j < CGW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    constexpr int NG = TAPS * KSTEPS;             // MFMA groups per chunk: one (tap, k-step) each
    constexpr int NPIECE = MAXJ + MAXW;
    int bi = 0;                                   // buffer of chunk c
    for (int c = 0; c < n; ++c) {
        // chunks up to c + NBUF - 2 have been issued; chunk c must have landed
        const int ahead = min(NBUF - 2, n - 1 - c);
        wait_vmcnt(ahead * n_mine);
        if (c == 0) LAB_STAMP(3);
        if (c == 1) LAB_STAMP(6);
        __builtin_amdgcn_s_barrier();             // every wave's pieces of chunk c are in LDS; buffer of chunk c-1 is free
        if (c == 0) LAB_STAMP(4);
        if (c == 1) LAB_STAMP(7);
        const bool more = c + NBUF - 1 < n;
        char* nbuf = lds + (bi == 0 ? NBUF - 1 : bi - 1) * buf_bytes;
        const char* buf = lds + bi * buf_bytes;
        // fragments of group g+1 are read while the matrix cores work on group g; the DMA pieces of
        // chunk c + NBUF - 1 are issued between the groups, so their issue time hides behind the MFMAs
        u32x4 a[2][CGW], bb[2][PGW];
        auto read_group = [&](int g, int slot) {
            const int tap = g / KSTEPS, s = g % KSTEPS;
            const int tap_off = ST == 2 ? ((tap / KS) * p.IWP + ((tap % KS) & 1) * p.IWH + ((tap % KS) >> 1)) * 16
                                        : ((tap / KS) * p.IW + (tap % KS)) * 16;
#pragma unroll
            for (int j = 0; j < CGW; ++j) a[slot][j] = *(const u32x4*)(buf + a_off + ((j * TAPS + tap) * KSTEPS + s) * 1024);
#pragma unroll
            for (int i = 0; i < PGW; ++i) bb[slot][i] = *(const u32x4*)(buf + b_off[i] + tap_off + s * 2 * p.npad * 16);
        };
        read_group(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) read_group(g + 1, (g + 1) & 1);
            if (more) {
#pragma unroll
                for (int k = g * NPIECE / NG; k < (g + 1) * NPIECE / NG; ++k) issue_piece(k, c + NBUF - 1, nbuf);
            }
#pragma unroll
            for (int i = 0; i < PGW; ++i)
#pragma unroll
                for (int j = 0; j < CGW; ++j) mma_step<DT>(acc[i][j], a[g & 1][j], bb[g & 1][i]);
        }
        bi = bi + 1 == NBUF ? 0 : bi + 1;
        if (c == 0) LAB_STAMP(5);
        if (c == 1) LAB_STAMP(8);
    }
    LAB_STAMP(9);
    __builtin_amdgcn_s_barrier();                 // all fragment reads are done: LDS becomes the epilogue slices
    char* stage = lds + wave * (32 * (CGW * 128 + 16));
    const int cw0 = co0 + wc * CGW * 32;
    if (p.fast_epi) {
        if (cw0 < p.Cout) {
            if (p.act) nhwc_epilogue_fast<DT, PGW, CGW, true>(acc, opix, stage, lane, cw0, b * p.Ho * p.Wo, aff, p.y, p.ybytes, p.y_ld);
            else nhwc_epilogue_fast<DT, PGW, CGW, false>(acc, opix, stage, lane, cw0, b * p.Ho * p.Wo, aff, p.y, p.ybytes, p.y_ld);
        }
    } else {
        nhwc_epilogue<DT, PGW, CGW>(acc, opix, stage, lane, cw0, (size_t)b * p.Ho * p.Wo, aff, p.Cout, p.y, p.y_ld, p.act, 0);
    }
    LAB_STAMP(10);
}

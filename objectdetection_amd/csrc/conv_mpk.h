// The two 1x1 convolutions at the head of an `mp` block in one launch for 256 input channels (bf16, NHWC): cbs1 on the
// 2x2 max-pooled input and cbs2 on the input itself (model/layer.py:139-145).  conv_mp_kernel keeps both weight slabs and
// whole pixel rows in LDS, which stops at 128 channels; here the input channels are walked in chunks of CK bytes per
// pixel and the weights of a chunk are streamed with it (they come from L2), so that two workgroups fit a CU.
// Included by conv.hip inside its anonymous namespace.
//
// Measured alternatives at batch 64, 80 x 80, 256 -> 128 + 128 (two separate launches: 156 us): this form 106 us; both
// weight slabs resident in LDS (135 KB: one workgroup of four waves per CU, the input read exactly once) 119 us with one
// staging buffer and 153 us with two - a single wave per SIMD hides neither the LDS round trips nor the barriers; the
// same streaming form for 512 input channels (256 + 256 outputs, 16 chunks) 134 us against 108 us for the two launches,
// so the dispatcher only takes 256 channels.
//
// A workgroup (4 waves) walks PAIRS of 8 x 8 input patches (128 pixels -> 32 pooled pixels = one matrix pixel group; any
// map with H % 8 == 0 and W % 8 == 0, e.g. 40 x 40).  Per chunk:
//   A. the chunk's 16-byte pieces of the 128 pixels and of all (Cout2 + Cout1) weight rows, fetched into registers one
//      chunk ahead (the loads fly during the matrix work of the chunk before), go to LDS rows of CK + 16 bytes;
//   B. the 32 pooled pixels of the chunk: max of four 16-byte vectors, second LDS area;
//   C. matrix work: plain branch 2 pixel groups x CGW channel groups per wave (wave grid 2 x 2), pooled branch 1 pixel
//      group x C1W channel groups per wave, v_mfma_f32_32x32x16_bf16 as everywhere, k-steps in ascending order (the sums
//      equal those of the separate launches bit for bit).
// After the last chunk both NHWC epilogues run through wave-private LDS slices that overlay the staging areas.

struct MpkP {
    const char* x;
    const char* w2; const float* scale2; const float* shift2; char* y2;     // plain branch (full resolution)
    const char* w1; const float* scale1; const float* shift1; char* y1;     // pooled branch (half resolution)
    int B, H, W, cbytes, x_ld, y2_ld, y1_ld;
    int patches_x, patches_y, npatches, npairs;
    unsigned y2bytes, y1bytes;
};

// CGW: channel groups (of 32) per wave in the plain branch (Cout2 = 64 * CGW); C1W: the same in the pooled branch
// (Cout1 = 128 * C1W); CK: bytes of input channels per pixel and chunk (64 or 128).
template <int CGW, int C1W, int CK>
__global__ void __launch_bounds__(256) conv_mpk_kernel(MpkP p) {
    constexpr int DT = ODAMD_DTYPE_BF16;
    constexpr int PARTS = CK / 16, KS = CK / 32, RSB = CK + 16;
    constexpr int COUT2 = 64 * CGW, COUT1 = 128 * C1W, WROWS = COUT2 + COUT1;
    constexpr int NIN = 128 * PARTS / 256;             // staged input pieces per thread and chunk
    constexpr int NWT = WROWS * PARTS / 256;           // staged weight pieces per thread and chunk
    constexpr int RPJ = 256 / PARTS;                   // rows (pixels / weight rows) covered by one round of the 256 threads
    static_assert(NIN >= 1 && NWT >= 1 && (COUT2 % RPJ) == 0, "piece split");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* lds_in = lds;                                // [128][RSB]
    char* lds_pool = lds_in + 128 * RSB;               // [32][RSB]
    char* lds_w2 = lds_pool + 32 * RSB;                // [COUT2][RSB]
    char* lds_w1 = lds_w2 + COUT2 * RSB;               // [COUT1][RSB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wp = wave >> 1, wc = wave & 1;
    const int nch = p.cbytes / CK;

    // ---- staging geometry of this thread ----------------------------------------------------------------------------
    const int part = tid % PARTS, row0 = tid / PARTS;  // piece j covers row row0 + RPJ * j
    int in_rel[NIN];                                   // byte offset of pixel (row0 + RPJ j) & 63 within its patch
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
        const int q = (row0 + RPJ * j) & 63;
        in_rel[j] = ((q >> 3) * p.W + (q & 7)) * p.x_ld * 2 + part * 16;
    }
    auto patch_origin = [&](int P, int& b, int& y0, int& x0) {
        const int px = P % p.patches_x, rest = P / p.patches_x;
        b = rest / p.patches_y;
        y0 = (rest - b * p.patches_y) * 8;
        x0 = px * 8;
    };
    u32x4 v_in[NIN], v_w[NWT];
    const char* xsrc[2];                               // origins of the two patches of the pair being fetched
    auto set_pair = [&](int t) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            int b, y0, x0;
            patch_origin(min(2 * t + k, p.npatches - 1), b, y0, x0);      // a pair past the end repeats the last patch
            xsrc[k] = p.x + ((size_t)(b * p.H + y0) * p.W + x0) * p.x_ld * 2;
        }
    };
    auto fetch = [&](int c) {
#pragma unroll
        for (int j = 0; j < NIN; ++j) v_in[j] = *(const u32x4*)(xsrc[(row0 + RPJ * j) >> 6] + in_rel[j] + c * CK);
#pragma unroll
        for (int j = 0; j < NWT; ++j) {
            const int row = row0 + RPJ * j;            // plain rows first: row < COUT2 depends on j alone
            const char* src = RPJ * j < COUT2 ? p.w2 + (size_t)row * p.cbytes : p.w1 + (size_t)(row - COUT2) * p.cbytes;
            v_w[j] = *(const u32x4*)(src + c * CK + part * 16);
        }
    };
    // pooled vector of this thread: pooled pixel pq = 16 patch + 4 row + col (threads past 32 * PARTS idle)
    const int pq = tid / PARTS;
    const bool pool_mine = tid < 32 * PARTS;
    const int pool_src = ((pq >> 4) * 64 + 2 * ((pq >> 2) & 3) * 8 + 2 * (pq & 3)) * RSB + part * 16;
    const int pool_dst = pq * RSB + part * 16;

    // ---- fragment offsets ---------------------------------------------------------------------------------------------
    int a_off[CGW], b_off[2], a1_off[C1W];
#pragma unroll
    for (int j = 0; j < CGW; ++j) a_off[j] = ((wc * CGW + j) * 32 + r) * RSB + 16 * h;
#pragma unroll
    for (int i = 0; i < 2; ++i) b_off[i] = ((wp * 2 + i) * 32 + r) * RSB + 16 * h;
#pragma unroll
    for (int j = 0; j < C1W; ++j) a1_off[j] = ((wave * C1W + j) * 32 + r) * RSB + 16 * h;
    const int b1_off = r * RSB + 16 * h;

    Affine<DT, CGW> aff2;
    aff2.load(p.scale2, p.shift2, lane, wc * CGW * 32, COUT2);
    Affine<DT, C1W> aff1;
    aff1.load(p.scale1, p.shift1, lane, wave * C1W * 32, COUT1);

    int t = blockIdx.x;
    if (t < p.npairs) { set_pair(t); fetch(0); }
    for (; t < p.npairs; t += gridDim.x) {
        f32x16_t acc[2][CGW], acc1[1][C1W];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < CGW; ++j) acc[i][j][e] = 0.0f;
#pragma unroll
            for (int j = 0; j < C1W; ++j) acc1[0][j][e] = 0.0f;
        }
        for (int c = 0; c < nch; ++c) {
            __syncthreads();                 // the fragment reads of the chunk before (or the epilogue slices) are done
#pragma unroll
            for (int j = 0; j < NIN; ++j) *(u32x4*)(lds_in + (row0 + RPJ * j) * RSB + part * 16) = v_in[j];
#pragma unroll
            for (int j = 0; j < NWT; ++j) *(u32x4*)(lds_w2 + (row0 + RPJ * j) * RSB + part * 16) = v_w[j];    // lds_w1 follows lds_w2
            __syncthreads();
            // the next chunk (of this pair, or the first one of the workgroup's next pair) flies during the rest of this one
            if (c + 1 < nch) fetch(c + 1);
            else if (t + (int)gridDim.x < p.npairs) { set_pair(t + gridDim.x); fetch(0); }
            if (pool_mine) {
                const char* s0 = lds_in + pool_src;
                const u32x4 m = vmax16(vmax16(*(const u32x4*)s0, *(const u32x4*)(s0 + RSB), DT),
                                       vmax16(*(const u32x4*)(s0 + 8 * RSB), *(const u32x4*)(s0 + 9 * RSB), DT), DT);
                *(u32x4*)(lds_pool + pool_dst) = m;
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                u32x4 a[CGW], bb[2];
#pragma unroll
                for (int j = 0; j < CGW; ++j) a[j] = *(const u32x4*)(lds_w2 + a_off[j] + 32 * s);
#pragma unroll
                for (int i = 0; i < 2; ++i) bb[i] = *(const u32x4*)(lds_in + b_off[i] + 32 * s);
                const u32x4 bp = *(const u32x4*)(lds_pool + b1_off + 32 * s);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < CGW; ++j) mma_step<DT>(acc[i][j], a[j], bb[i]);
#pragma unroll
                for (int j = 0; j < C1W; ++j) mma_step<DT>(acc1[0][j], *(const u32x4*)(lds_w1 + a1_off[j] + 32 * s), bp);
            }
        }
        __syncthreads();                     // every wave has finished reading the staging areas: they become epilogue slices
        int opix[2], opix1[1];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = (wp * 2 + i) * 32 + r;
            const int P = 2 * t + (q >> 6);
            int b, y0, x0;
            patch_origin(min(P, p.npatches - 1), b, y0, x0);
            opix[i] = P < p.npatches ? (b * p.H + y0 + ((q >> 3) & 7)) * p.W + x0 + (q & 7) : -1;
        }
        {
            const int P = 2 * t + (r >> 4);
            int b, y0, x0;
            patch_origin(min(P, p.npatches - 1), b, y0, x0);
            opix1[0] = P < p.npatches ? (b * (p.H >> 1) + (y0 >> 1) + ((r >> 2) & 3)) * (p.W >> 1) + (x0 >> 1) + (r & 3) : -1;
        }
        constexpr int SLICE = 32 * ((CGW > C1W ? CGW : C1W) * 128 + 16);
        char* stage = lds + wave * SLICE;
        nhwc_epilogue_fast<DT, 2, CGW, true>(acc, opix, stage, lane, wc * CGW * 32, 0, aff2, p.y2, p.y2bytes, p.y2_ld);
        nhwc_epilogue_fast<DT, 1, C1W, true>(acc1, opix1, stage, lane, wave * C1W * 32, 0, aff1, p.y1, p.y1bytes, p.y1_ld);
    }
}

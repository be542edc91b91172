// The two 1x1 convolutions at the head of an `mp` block in one launch (bf16, NHWC): cbs1 on the 2x2 max-pooled
// input and cbs2 on the input itself (model/layer.py:139-145) read the same tensor, the largest one of the block.
// Included by conv.hip inside its anonymous namespace.
//
// Persistent workgroups (4 waves) keep both weight slabs in LDS and walk 8 x 16 patches of the input (128 pixels, whole
// patches only: H % 8 == 0, W % 16 == 0).  Per patch:
//   A. the patch (staged through registers one patch ahead, as in conv1x1_stream_kernel) -> LDS [128 pixels][row];
//   B. its 4 x 8 pooled pixels -> a second LDS area (max of four 16-byte vectors per thread);
//   C. cbs2: 4 pixel groups x CG2 channel groups over the patch, cbs1: 1 pixel group x CG1 channel groups over the
//      pooled pixels, same v_mfma_f32_32x32x16_bf16 fragments as everywhere (A = weights, B = pixels);
//   D. both NHWC epilogues (affine + SiLU, 16-byte stores of whole channel runs) through wave-private LDS slices that
//      overlay the patch.
// The input is read from HBM once instead of twice, and the pooled branch no longer waits on four dependent loads per
// staged vector.

struct MpP {
    const char* x;
    const char* w2; const float* scale2; const float* shift2; char* y2;     // plain branch (full resolution)
    const char* w1; const float* scale1; const float* shift1; char* y1;     // pooled branch (half resolution)
    int B, H, W, Cin, x_ld, y2_ld, y1_ld, Cout2, Cout1;
    int tiles_x, tiles_y, ntiles;
    int w2_off, w1_off, pool_off;       // LDS offsets
    unsigned y2bytes, y1bytes;
};

// CG2 / CG1: 32-channel groups of the plain / pooled branch; NPF: staged 16-byte pieces per thread (128 * parts / 256).
template <int CG2, int CG1, int NPF>
__global__ void __launch_bounds__(256) conv_mp_kernel(MpP p) {
    constexpr int DT = ODAMD_DTYPE_BF16;
    constexpr int WC = CG2 >= 2 ? 2 : 1, WP = 4 / WC;                  // wave grid of the plain branch: WP x WC
    constexpr int PGW = 4 / WP, CGW = CG2 / WC;
    static_assert(CG2 % WC == 0 && CG1 >= 1 && CG1 <= 4, "wave split");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int cbytes = p.Cin * 2;               // multiple of 32, <= 256
    const int parts = cbytes >> 4;
    const int RSB = cbytes + 16;                // LDS bytes per staged pixel / weight row
    char* lds_in = lds;                         // [128][RSB], overlaid by the epilogue slices
    char* lds_pool = lds + p.pool_off;          // [32][RSB]
    char* lds_w2 = lds + p.w2_off;              // [32 * CG2][RSB]
    char* lds_w1 = lds + p.w1_off;              // [32 * CG1][RSB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wp = wave / WC, wc = wave % WC;

    // weights: once
    for (int i = tid; i < 32 * (CG2 + CG1) * parts; i += 256) {
        const int row = i / parts, part = i - row * parts;
        const bool second = row >= 32 * CG2;
        const int co = second ? row - 32 * CG2 : row;
        const char* src = (second ? p.w1 : p.w2) + (size_t)co * cbytes + part * 16;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (co < (second ? p.Cout1 : p.Cout2)) v = *(const u32x4*)src;
        *(u32x4*)((second ? lds_w1 + co * RSB : lds_w2 + co * RSB) + part * 16) = v;
    }
    int a_off[CGW], b_off[PGW];
#pragma unroll
    for (int j = 0; j < CGW; ++j) a_off[j] = ((wc * CGW + j) * 32 + r) * RSB + 16 * h;
#pragma unroll
    for (int i = 0; i < PGW; ++i) b_off[i] = ((wp * PGW + i) * 32 + r) * RSB + 16 * h;
    // staged piece j of this thread: patch pixel q = (tid + 256 j) / parts = 16 py + px, 16-byte part % parts
    int g_off[NPF], l_off[NPF];
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
        const int i = tid + j * 256;
        const int q = i / parts, part = i - q * parts;
        g_off[j] = ((q >> 4) * p.W + (q & 15)) * p.x_ld * 2 + part * 16;
        l_off[j] = q * RSB + part * 16;
    }
    // pooled vector(s) of this thread: pooled pixel pq = 8 qy + qx, part
    constexpr int NPL = (NPF + 3) / 4;
    int pl_src[NPL], pl_dst[NPL];
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
        const int i = tid + j * 256;
        const int pq = i / parts, part = i - pq * parts;
        pl_src[j] = i < 32 * parts ? ((2 * (pq >> 3)) * 16 + 2 * (pq & 7)) * RSB + part * 16 : -1;
        pl_dst[j] = pq * RSB + part * 16;
    }
    Affine<DT, CGW> aff2;
    aff2.load(p.scale2, p.shift2, lane, (wc * CGW) * 32, p.Cout2);
    Affine<DT, 1> aff1;
    aff1.load(p.scale1, p.shift1, lane, min(wave, CG1 - 1) * 32, p.Cout1);

    u32x4 v_in[NPF];
    auto origin = [&](int t, int& b, int& y0, int& x0) {
        const int tx = t % p.tiles_x, rest = t / p.tiles_x;
        b = rest / p.tiles_y;
        y0 = (rest - b * p.tiles_y) * 8;
        x0 = tx * 16;
    };
    auto fetch = [&](int t) {
        int b, y0, x0;
        origin(t, b, y0, x0);
        const char* xb = p.x + ((size_t)(b * p.H + y0) * p.W + x0) * p.x_ld * 2;
#pragma unroll
        for (int j = 0; j < NPF; ++j) v_in[j] = *(const u32x4*)(xb + g_off[j]);
    };
    int t = blockIdx.x;
    if (t < p.ntiles) fetch(t);
    for (; t < p.ntiles; t += gridDim.x) {
        __syncthreads();                     // epilogue slices (overlaying lds_in) of the previous patch are done
#pragma unroll
        for (int j = 0; j < NPF; ++j) *(u32x4*)(lds_in + l_off[j]) = v_in[j];
        __syncthreads();
        if (t + (int)gridDim.x < p.ntiles) fetch(t + gridDim.x);   // in flight during the rest of this patch
        // ---- B: 2x2 max pool of the patch ------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            if (pl_src[j] >= 0) {
                const char* s0 = lds_in + pl_src[j];
                const u32x4 m = vmax16(vmax16(*(const u32x4*)s0, *(const u32x4*)(s0 + RSB), DT),
                                       vmax16(*(const u32x4*)(s0 + 16 * RSB), *(const u32x4*)(s0 + 17 * RSB), DT), DT);
                *(u32x4*)(lds_pool + pl_dst[j]) = m;
            }
        }
        __syncthreads();
        // ---- C: both GEMMs -----------------------------------------------------------------------------
        f32x16_t acc[PGW][CGW], acc1[1][1];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            acc1[0][0][e] = 0.0f;
#pragma unroll
            for (int i = 0; i < PGW; ++i)
#pragma unroll
                for (int j = 0; j < CGW; ++j) acc[i][j][e] = 0.0f;
        }
        const bool pooled_wave = wave < CG1;                                  // wave-uniform
        const int a1_off = (min(wave, CG1 - 1) * 32 + r) * RSB + 16 * h, b1_off = r * RSB + 16 * h;
        for (int c0 = 0; c0 < cbytes; c0 += 32) {
            u32x4 a[CGW], bb[PGW];
#pragma unroll
            for (int j = 0; j < CGW; ++j) a[j] = *(const u32x4*)(lds_w2 + a_off[j] + c0);
#pragma unroll
            for (int i = 0; i < PGW; ++i) bb[i] = *(const u32x4*)(lds_in + b_off[i] + c0);
#pragma unroll
            for (int i = 0; i < PGW; ++i)
#pragma unroll
                for (int j = 0; j < CGW; ++j) mma_step<DT>(acc[i][j], a[j], bb[i]);
            if (pooled_wave) mma_step<DT>(acc1[0][0], *(const u32x4*)(lds_w1 + a1_off + c0), *(const u32x4*)(lds_pool + b1_off + c0));
        }
        __syncthreads();                     // every wave has finished reading lds_in
        // ---- D: epilogues --------------------------------------------------------------------------------
        int b, y0, x0;
        origin(t, b, y0, x0);
        int opix[PGW];
#pragma unroll
        for (int i = 0; i < PGW; ++i) {
            const int q = (wp * PGW + i) * 32 + r;
            opix[i] = ((b * p.H + y0 + (q >> 4)) * p.W + x0 + (q & 15));
        }
        char* stage = lds + wave * (32 * (CGW * 128 + 16));
        nhwc_epilogue_fast<DT, PGW, CGW, true>(acc, opix, stage, lane, wc * CGW * 32, 0, aff2, p.y2, p.y2bytes, p.y2_ld);
        if (pooled_wave) {
            int opix1[1] = {(b * (p.H >> 1) + (y0 >> 1) + (r >> 3)) * (p.W >> 1) + (x0 >> 1) + (r & 7)};
            nhwc_epilogue_fast<DT, 1, 1, true>(acc1, opix1, stage, lane, wave * 32, 0, aff1, p.y1, p.y1bytes, p.y1_ld);
        }
    }
}

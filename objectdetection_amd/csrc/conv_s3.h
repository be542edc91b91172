// Persistent 3x3 convolution (bf16, NHWC in and out, stride 1 or 2) for the wide, shallow layers: 32 or 64 input
// channels, 32 or 64 output channels, maps of 80 x 80 and larger.  Included by conv.hip inside its anonymous namespace.
//
// These layers are bound by HBM (two to four MFMA k-steps per tap), and a workgroup per tile spends more
// time filling and draining its pipeline than computing.  Here a workgroup (4 waves) stays resident and walks tiles:
//   * the whole weight tensor (fragment order, see odamd_conv_weight_layout) is copied to LDS once per workgroup;
//   * the input halo patch of tile t+1 travels global -> LDS (global_load_lds_dwordx4) while the matrix cores work on
//     tile t.  One DMA instruction copies 4 channel parts (16 bytes each) of 16 consecutive patch slots, i.e. runs of
//     64 contiguous bytes per pixel (a wave instruction that gathers 64 different pixels costs the texture path one
//     cache line per lane); in LDS the lanes land as [slot / 16][part][slot % 16], so the 16 lanes of one
//     ds_read_b128 pass, which read one part of 16 consecutive slots, cover 256 bytes of distinct banks wherever the
//     run starts.  Stride 2 keeps even and odd patch columns apart so that a fragment still reads consecutive slots. NBUF patch buffers (tiles up to t + NBUF - 1 in flight), one
//     s_barrier per tile and a counted s_waitcnt vmcnt;
//   * the epilogue needs no LDS: v_permlane32_swap exchanges the two half-waves' packed bf16 pairs so that every lane
//     ends up with 8 consecutive channels of its pixel, stored with one 16-byte buffer store.
// Tiles are dealt like those of conv_first2_kernel: runs of vertically adjacent tiles per workgroup, contiguous item
// ranges per XCD, so halo rows and columns are re-read from the L2 that just saw them.

#ifdef ODAMD_LAB_STAMPS
#define S3_STAMP(i) do { if (lane == 0 && blockIdx.x < 1024 && j == 9) g_lab_stamps[(blockIdx.x * 8 + wave) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define S3_STAMP(i) do { } while (0)
#endif

struct S3P {
    const char* x; const char* w; const float* scale; const float* shift; char* y;
    int H, W, Ho, Wo, x_ld, y_ld, act;
    int TH, TW, twl;            // output tile; TW = 1 << twl (8, 16 or 32), a pixel group is 32 / TW rows of TW pixels
    int IH, IW, IWH, npad;      // patch rows / columns, columns per parity plane (stride 2), slots per part plane
    int nitems, tiles_x, tiles_y, segs_y, rounds;
    unsigned ybytes;
};

constexpr int S3_SEG = 8;       // vertically adjacent tiles a workgroup takes in one go

struct S3Tile { int b, oy0, ox0; bool valid; };
__device__ __forceinline__ S3Tile s3_item(int i, const S3P& p) {
    const int w = blockIdx.x;
    const int per_xcd = (p.nitems + 7) >> 3, slots = gridDim.x >> 3;   // the host launches a multiple of 8 workgroups
    const int idx = (w >> 3) + i * slots;
    const int item = (w & 7) * per_xcd + idx;
    S3Tile t;
    t.valid = idx < per_xcd && item < p.nitems;
    const int it = t.valid ? item : 0;
    const int tx = it % p.tiles_x, rest = it / p.tiles_x, sy = rest % p.segs_y;
    t.b = rest / p.segs_y;
    t.oy0 = sy * S3_SEG * p.TH;
    t.ox0 = tx * p.TW;
    return t;
}
__device__ __forceinline__ S3Tile s3_step(const S3Tile& first, int k, const S3P& p) {
    S3Tile t = first;
    t.oy0 = first.oy0 + k * p.TH;
    t.valid = first.valid && t.oy0 < p.Ho;
    return t;
}

// KSTEPS: input channels / 16; CG: output channels / 32; ST: stride; PG: pixel groups (32 pixels) per tile;
// NPAD_MAX: upper bound of p.npad (sizes the per-wave piece lists); NBUF: patch buffers.
template <int KSTEPS, int CG, int ST, int PG, int NPAD_MAX, int NBUF, bool ACT>
__global__ void __launch_bounds__(256) conv_s3_kernel(S3P p) {
    constexpr int DT = ODAMD_DTYPE_BF16;
    constexpr int NW = 4, PGW = PG / NW, TAPS = 9, PARTS = 2 * KSTEPS;
    constexpr int W_PIECES = CG * TAPS * KSTEPS;                 // 1 KiB each
    constexpr int MAXJ = PARTS * (NPAD_MAX / 64) / NW;           // input pieces per wave and tile
    static_assert(PG % NW == 0 && (PARTS * (NPAD_MAX / 64)) % NW == 0, "wave split");
    static_assert(S3_SEG % 2 == 0, "the main loop is unrolled by two tiles");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* wl = lds;                                              // weights
    const int in_bytes = PARTS * p.npad * 16;
    char* bufs = lds + W_PIECES * 1024;                          // NBUF patch buffers

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int IWP = ST == 2 ? 2 * p.IWH : p.IW;                  // slots per patch row

    // ---- weights: one contiguous run, copied as it is -----------------------------------------------
    for (int i = wave; i < W_PIECES; i += NW) dma16(p.w + i * 1024 + lane * 16, wl + i * 1024);

    // ---- this wave's pieces of a patch: 4 parts of 16 consecutive slots each ------------------------
    // slot -> patch pixel (iy, ix) does not depend on the tile; -1: a slot no fragment read uses
    constexpr int PQ = PARTS / 4;                                // pieces per 16 slots
    int pyx[MAXJ], loff[MAXJ];
#pragma unroll
    for (int jj = 0; jj < MAXJ; ++jj) {
        const int i = wave + jj * NW;
        const int g16 = i / PQ, pq = i - g16 * PQ;
        const int part = pq * 4 + (lane >> 4);
        const int q = 16 * g16 + (lane & 15);
        const int iy = q / IWP, c = q - iy * IWP;
        int ix = c;
        if (ST == 2) ix = c < p.IWH ? 2 * c : 2 * (c - p.IWH) + 1;
        pyx[jj] = (iy < p.IH && ix < p.IW) ? (iy << 12 | ix << 4 | part) : -1;       // the host sets npad = NPAD_MAX
        loff[jj] = (g16 * PARTS + pq * 4) * 256;
    }
    const char* zl = (const char*)g_zero_line;
    // ---- tile sequence of this workgroup, D tiles ahead of the one being computed ------------------
    constexpr int D = NBUF - 1, NST = PGW * CG * 2;              // NST: store instructions of this wave per tile
    int gen_i = 0, gen_k = 0;
    S3Tile gen_item = s3_item(0, p);
    auto next_tile = [&]() {
        const S3Tile t = s3_step(gen_item, gen_k, p);
        if (++gen_k == S3_SEG) {
            gen_k = 0;
            ++gen_i;
            gen_item = s3_item(gen_i, p);
            gen_item.valid = gen_item.valid && gen_i < p.rounds;
        }
        return t;
    };
    // One DMA piece of tile t (a tile that does not exist copies the zero line: every iteration of the main loop
    // issues the same number of vector-memory instructions, which keeps it free of branches and its waits immediate).
    auto issue_piece = [&](int jj, const S3Tile& t, char* buf) {
        const char* xb = p.x + (size_t)t.b * p.H * p.W * p.x_ld * 2;
        const int gy = t.oy0 * ST - 1 + (pyx[jj] >> 12), gx = t.ox0 * ST - 1 + ((pyx[jj] >> 4) & 255);
        const bool inside = t.valid && pyx[jj] >= 0 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        dma16(inside ? xb + ((size_t)(gy * p.W + gx) * p.x_ld) * 2 + (pyx[jj] & 15) * 16 : zl, buf + loff[jj]);
    };
    S3Tile q[D];          // q[0]: the tile computed next
#pragma unroll
    for (int i = 0; i < D; ++i) {
        q[i] = next_tile();
#pragma unroll
        for (int jj = 0; jj < MAXJ; ++jj) issue_piece(jj, q[i], bufs + i * in_bytes);
    }

    // ---- per-lane fragment offsets and output pixels (relative to the tile) ------------------------
    // b_addr[i][tap]: LDS offset (within a buffer) of part h of the slot that tap reads for (group i, lane r)
    int b_addr[PGW][TAPS], ty_[PGW];
    const int tx_ = r & (p.TW - 1);
#pragma unroll
    for (int i = 0; i < PGW; ++i) {
        const int g = wave * PGW + i;
        ty_[i] = ((g * 32) >> p.twl) + (r >> p.twl);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const int slot = ST == 2 ? (2 * ty_[i] + ky) * IWP + (kx & 1) * p.IWH + (kx >> 1) + tx_ : (ty_[i] + ky) * IWP + kx + tx_;
            b_addr[i][tap] = ((slot >> 4) * PARTS + h) * 256 + (slot & 15) * 16;
        }
    }
    const int a_off = lane * 16;

    // affine constants of the channels this lane holds after the MFMA: group j, element e -> channel
    // 32 j + 8 (e >> 2) + 4 h + (e & 3)
    float scl[CG][16], shf[CG][16];
#pragma unroll
    for (int j = 0; j < CG; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = 32 * j + 8 * (e >> 2) + 4 * h + (e & 3);
            scl[j][e] = p.scale[c];
            shf[j][e] = p.shift[c];
        }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.ybytes, 0x00020000);

    // Epilogue unit u = (pixel group i, channel group jc, half q) of tile t: affine (+ SiLU) of 8 accumulator
    // elements, bf16, half-wave exchange, one 16-byte store.  The packed pairs pk[2g'], pk[2g'+1] of a lane hold
    // channels 8g + 4h .. + 3 (g = 2q + g'); after v_permlane32_swap the lower half-wave holds channels 16q .. 16q + 7
    // of its pixel and the upper half-wave channels 16q + 8 .. 16q + 15.  A tile that does not exist stores nothing
    // (offset -1 is dropped by the range check of the buffer resource).
    constexpr int NU = PGW * CG * 2;
    auto epi_unit = [&](int u, const f32x16_t (&ac)[PGW][CG], const S3Tile& t) {
        const int i = u / (CG * 2), jc = (u / 2) % CG, qh = u % 2;
        const int oy = t.oy0 + ty_[i], ox = t.ox0 + tx_;
        const bool ok = t.valid && ty_[i] < p.TH && oy < p.Ho && ox < p.Wo;
        const int pix = (t.b * p.Ho + oy) * p.Wo + ox;
        uint32_t pk[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int e = 8 * qh + 2 * m;
            float t0 = ac[i][jc][e] * scl[jc][e] + shf[jc][e], t1 = ac[i][jc][e + 1] * scl[jc][e + 1] + shf[jc][e + 1];
            if (ACT) { t0 = silu<DT>(t0); t1 = silu<DT>(t1); }
            pk[m] = f32x2_to_bf16x2(t0, t1);
        }
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
        const u32x4 out = u32x4{s0[0], s1[0], s0[1], s1[1]};
        const int voff = ok ? (pix * p.y_ld + 32 * jc + 16 * qh + 8 * h) * 2 : -1;
        __builtin_amdgcn_raw_buffer_store_b128(out, rsrc, voff, 0, 0);
    };

    // ---- main loop: the matrix instructions of tile j, with the DMA pieces of tile j + D and the epilogue of tile
    // j - 1 spread between them (one basic block: the scheduler fills the MFMA shadows with that work) --------------
    constexpr int NG = TAPS * KSTEPS;
    // vector-memory instructions issued after the last piece of a tile until its wait: those of D - 1 whole iterations
    // (stores after the last piece within its own iteration are left out, which only makes the wait stricter)
    // That count assumes D - 1 whole iterations behind the tile's pieces; the first D tiles of a workgroup have
    // fewer (and the stores of the tile "before the first" fail the range check and retire at once, ahead of older
    // loads), so they wait for the pieces alone: stricter is always safe.  The choice is a branch at the head of a
    // tile, where the wait and the barrier end the scheduling region anyway.
    constexpr int WAITN = (D - 1) * (MAXJ + NST);
    int bi = 0, warm = 0;
    S3Tile prev = {0, 0, 0, false};
    auto body = [&](f32x16_t (&ac)[PGW][CG], const f32x16_t (&ap)[PGW][CG], int j) {
        const S3Tile cur = q[0];
        S3_STAMP(0);
        if (warm >= D) wait_vmcnt_imm<WAITN>();     // this wave's pieces of the tile have landed
        else wait_vmcnt_imm<(D - 1) * MAXJ>();
        ++warm;
        S3_STAMP(1);
        __builtin_amdgcn_s_barrier();         // ... and everybody else's; the buffer of the previous tile is free
        S3_STAMP(2);
        const char* buf = bufs + bi * in_bytes;
        char* nbuf = bufs + (bi == 0 ? NBUF - 1 : bi - 1) * in_bytes;
#pragma unroll
        for (int i = 0; i + 1 < D; ++i) q[i] = q[i + 1];
        q[D - 1] = next_tile();
        const S3Tile inc = q[D - 1];
        bi = bi + 1 == NBUF ? 0 : bi + 1;
#pragma unroll
        for (int i = 0; i < PGW; ++i)
#pragma unroll
            for (int jc = 0; jc < CG; ++jc)
#pragma unroll
                for (int e = 0; e < 16; ++e) ac[i][jc][e] = 0.0f;
        u32x4 a[2][CG], bb[2][PGW];
        auto read_group = [&](int g, int slot) {
            const int tap = g / KSTEPS, s2 = g % KSTEPS;
#pragma unroll
            for (int jc = 0; jc < CG; ++jc) a[slot][jc] = *(const u32x4*)(wl + a_off + ((jc * TAPS + tap) * KSTEPS + s2) * 1024);
#pragma unroll
            for (int i = 0; i < PGW; ++i) bb[slot][i] = *(const u32x4*)(buf + b_addr[i][tap] + s2 * 512);
        };
        read_group(0, 0);
        static_for<0, NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g + 1 < NG) read_group(g + 1, (g + 1) & 1);
            static_for<g * MAXJ / NG, (g + 1) * MAXJ / NG>([&](auto kc) { issue_piece(decltype(kc)::value, inc, nbuf); });
            static_for<g * NU / NG, (g + 1) * NU / NG>([&](auto uc) { epi_unit(decltype(uc)::value, ap, prev); });
#pragma unroll
            for (int i = 0; i < PGW; ++i)
#pragma unroll
                for (int jc = 0; jc < CG; ++jc) mma_step<DT>(ac[i][jc], a[g & 1][jc], bb[g & 1][i]);
        });
        prev = cur;
        S3_STAMP(5);
    };
    f32x16_t accA[PGW][CG], accB[PGW][CG];
#pragma unroll
    for (int i = 0; i < PGW; ++i)
#pragma unroll
        for (int jc = 0; jc < CG; ++jc)
#pragma unroll
            for (int e = 0; e < 16; ++e) accB[i][jc][e] = 0.0f;
    const int njobs = p.rounds * S3_SEG;      // even
    for (int j = 0; j < njobs; j += 2) {
        body(accA, accB, j);
        body(accB, accA, j + 1);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) epi_unit(u, accB, prev);
}

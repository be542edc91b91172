// Persistent 1x1 convolution with the weights resident in REGISTERS (bf16, stride 1, NHWC in and out) for
// the deep layers: 256 / 512 / 1024 input channels, output channels in multiples of 128.
// Included by conv.hip inside its anonymous namespace.
//
// GEMM view as everywhere (D[cout][pixel], A = weights, B = pixels, v_mfma_f32_32x32x16_bf16).  A 1x1 layer with
// K >= 256 is a real GEMM: a workgroup per output tile re-loads a weight slab as large as its input tile, so the
// vector memory path moves twice the bytes and the matrix cores wait.  Here
//   * a workgroup (4 waves, one per SIMD, up to 512 registers each) owns 128 * CGW output channels for its whole
//     life; every wave keeps the weights of its 32 * CGW channels for ALL input channels in registers (the A
//     fragments of all k-steps: 128 or 256 registers), loaded once from the fragment-ordered weight tensor
//     (odamd_conv_weight_layout);
//   * the workgroup walks 64-pixel tiles of the flat (batch * H * W) pixel list.  A tile's input travels global ->
//     LDS (global_load_lds_dwordx4) in units of 256 input channels through a ring of NBUF buffers, up to NBUF - 1
//     units ahead of the matrix cores; one s_barrier and one counted s_waitcnt vmcnt per unit.  A DMA instruction
//     copies 8 pixels x 128 contiguous bytes (whole cache lines); in LDS the lanes land as
//     [pixel / 8][16-byte part][pixel % 8] with one spare part per row of 8 pixels, which makes every ds_read_b128
//     pass of a B fragment (32 consecutive pixels of one part) hit 16 distinct bank groups;
//   * the epilogue is no separate phase: affine + SiLU + bf16 of a tile's results are spread between the matrix
//     instructions of the NEXT tile's first unit (two accumulator sets alternate), so vector, matrix and memory work
//     overlap inside one wave.  The bf16 results pass through a small wave-private LDS area so that every store
//     instruction writes whole 128-byte runs per pixel.
// Workgroups with the same pixel walk and different channel blocks are neighbours on one XCD (block b and b + 8),
// so the second reader of a tile finds it in that XCD's L2.

struct WresP {
    const char* x; const char* w; const float* scale; const float* shift; char* y;
    int npix, x_ld, y_ld, Cout;
    int ntiles;          // 64-pixel tiles
    int ncb;             // channel blocks of 128 * CGW
    int nwalk;           // pixel walkers: gridDim.x / ncb
    int rounds;          // tiles per walker, rounded up to an even number
    unsigned ybytes;
};

constexpr int WRES_ROWB = 33 * 128;            // LDS bytes per row of 8 pixels x 256 channels (+ one spare part)
constexpr int WRES_UNIT = 8 * WRES_ROWB;       // one unit: 64 pixels x 256 channels

// KCH: units (256 input channels) per tile; CGW: 32-channel groups per wave; NBUF: ring buffers.
template <int KCH, int CGW, int NBUF, bool ACT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) conv1x1_wres_kernel(WresP p) {
    constexpr int DT = ODAMD_DTYPE_BF16;
    constexpr int PGW = 2, KS16 = 16;              // pixel groups per tile (every wave computes both); k-steps per unit
    constexpr int MAXJ = 8;                        // DMA pieces per wave and unit (32 pieces of 1 KiB)
    constexpr int D = NBUF - 1;                    // units in flight ahead of the matrix cores
    constexpr int NU = PGW * CGW * 2;              // epilogue units (8 accumulator values each) per tile
    static_assert(D >= 1 && D <= 3, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* bufs = lds;                              // NBUF units
    float* aff = (float*)(lds + NBUF * WRES_UNIT); // [2][128 * CGW]: scale, shift of this workgroup's channels

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // workgroup -> (channel block, pixel walker): blocks b and b + 8 share an XCD
    const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const int cb = slot % p.ncb;
    const int walker = (slot / p.ncb) * 8 + xcd;
    const int cbase = cb * 128 * CGW;              // first channel of the workgroup
    const int cw0 = cbase + wave * 32 * CGW;       // first channel of the wave

    // affine constants of the workgroup's channels: requested first, written to LDS once everything else is on its way
    float aff_sc = 0.f, aff_sh = 0.f;
    if (tid < 128 * CGW) { aff_sc = p.scale[cbase + tid]; aff_sh = p.shift[cbase + tid]; }
    // ---- DMA pieces of a unit: piece i = wave + 4 jj covers pixels 8 (i / 4) .. + 7, parts 8 (i % 4) .. + 7 -------
    // (the pixel count is a multiple of 64: every tile that exists is whole)
    const char* psrc[MAXJ];   // this lane's source of piece jj for tile 0, unit 0
    int loff[MAXJ];           // LDS offset of the piece within a unit buffer (wave uniform)
#pragma unroll
    for (int jj = 0; jj < MAXJ; ++jj) {
        const int i = wave + 4 * jj;
        const int g = i >> 2, pg = i & 3;
        psrc[jj] = p.x + (size_t)(8 * g + (lane & 7)) * p.x_ld * 2 + (pg * 8 + (lane >> 3)) * 16;
        loff[jj] = g * WRES_ROWB + pg * 1024;
    }
    const size_t tile_bytes = (size_t)64 * p.x_ld * 2;
    // unit sequence of this workgroup: (tile, chunk), tiles walker, walker + nwalk, ...
    int gen_tile = walker, gen_ch = 0;
    struct Unit { int tile, ch; };
    auto next_unit = [&]() {
        Unit u = {gen_tile, gen_ch};
        if (++gen_ch == KCH) { gen_ch = 0; gen_tile += p.nwalk; }
        return u;
    };
    // A tile past the end copies the last tile (valid memory); its results are never stored.
    auto unit_offset = [&](const Unit& u) { return (size_t)min(u.tile, p.ntiles - 1) * tile_bytes + u.ch * 512; };
    auto issue_piece = [&](int jj, size_t uoff, char* buf) { dma16(psrc[jj] + uoff, buf + loff[jj]); };
    // The first units are requested before the weights: a small layer is a few tiles per workgroup, and the weight
    // loads (128 .. 256 KB per workgroup) would otherwise stand in front of its first tile.
    Unit q[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        q[i] = next_unit();
#pragma unroll
        for (int jj = 0; jj < MAXJ; ++jj) issue_piece(jj, unit_offset(q[i]), bufs + i * WRES_UNIT);
    }
    // ---- weights: the A fragments of every k-step, once -------------------------------------------------
    u32x4 A[KCH][CGW][KS16];
    {
        const char* wb = p.w + ((size_t)(cb * 4 * CGW + wave * CGW) * (KCH * KS16)) * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < CGW; ++j)
#pragma unroll
            for (int ch = 0; ch < KCH; ++ch)
#pragma unroll
                for (int s = 0; s < KS16; ++s) A[ch][j][s] = *(const u32x4*)(wb + ((size_t)(j * KCH + ch) * KS16 + s) * 1024);
        // The weights only ever feed matrix instructions, which read the accumulation half of the register file
        // directly; the results are post-processed by vector instructions, which cannot.  Steer the allocation
        // accordingly (otherwise every result is copied out of that half one register at a time).
#pragma unroll
        for (int j = 0; j < CGW; ++j)
#pragma unroll
            for (int ch = 0; ch < KCH; ++ch)
#pragma unroll
                for (int s = 0; s < KS16; ++s) asm volatile("" : "+a"(A[ch][j][s]));
    }
    // the weights must be in their registers before the first matrix instruction, and the loop's counted waits
    // only know DMA pieces and stores: drain everything once
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid < 128 * CGW) { aff[tid] = aff_sc; aff[128 * CGW + tid] = aff_sh; }
    __syncthreads();                               // the affine table is in LDS (and every load above has landed)

    // B fragment of (pixel group i, k-step s): part 2 s + h of pixel 32 i + r
    int b_off[PGW];
#pragma unroll
    for (int i = 0; i < PGW; ++i) {
        const int px = 32 * i + r;
        b_off[i] = ((px >> 3) * 33 + h) * 128 + (px & 7) * 16;
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.ybytes, 0x00020000);

    // Affine constants of channel slice sl = 2 jc + qh of this wave: channels 16 sl + 4 h + (0 .. 3) and + 8, scale then
    // shift.  (Read as raw 16-byte vectors like the fragments: a float-typed LDS read after an LDS-DMA instruction
    // makes the compiler drain the vector-memory queue first.)  Kept in registers where the weights leave room,
    // otherwise fetched from LDS one pair of epilogue units ahead of their use.
    constexpr bool AFF_REGS = KCH * CGW <= 2;
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    auto load_aff = [&](int sl, u32x4 (&v)[4]) {
        const char* ab = (const char*)aff + (wave * 32 * CGW + 16 * sl + 4 * h) * 4;
        v[0] = *(const u32x4*)ab; v[1] = *(const u32x4*)(ab + 32);
        v[2] = *(const u32x4*)(ab + 512 * CGW); v[3] = *(const u32x4*)(ab + 512 * CGW + 32);
    };
    float affr[AFF_REGS ? CGW * 2 : 1][16];
    u32x4 affc[4], affn[4];
    if constexpr (AFF_REGS) {
#pragma unroll
        for (int sl = 0; sl < CGW * 2; ++sl) {
            u32x4 v[4];
            load_aff(sl, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                affr[sl][e] = __uint_as_float(v[0][e]); affr[sl][4 + e] = __uint_as_float(v[1][e]);
                affr[sl][8 + e] = __uint_as_float(v[2][e]); affr[sl][12 + e] = __uint_as_float(v[3][e]);
            }
        }
    } else {
        load_aff(0, affn);
    }

    // Epilogue of a tile in NH half-slots and PGW flushes.  Half-slot hs = (pixel group i, 16-channel slice k = 2 jc + qh,
    // half q): affine + SiLU of 4 accumulator values, bf16, one 8-byte write into the wave's transposition area
    // [32 pixels][64 CGW channels].  After the last half-slot of a pixel group the flush reads the area back 16 bytes
    // per lane, 4 CGW lanes per pixel, and stores whole 128-byte (CGW = 2) or 64-byte runs per pixel: a store
    // instruction covers 8 or 16 pixel rows completely instead of 32-byte pieces of 32 rows.  LDS operations of one
    // wave execute in order, so the area needs no barrier.  The half-slots are dealt evenly to the k-steps of the
    // next tile's first unit, so every k-step carries the same vector work beside its matrix instructions.
    constexpr int SLICES = CGW * 2, NH = PGW * SLICES * 2;
    constexpr int TROW = CGW * 64 + 16;                        // bytes per pixel in the transposition area
    constexpr int LPP = CGW * 4, PPP = 64 / LPP, PASSES = 32 / PPP;
    static_assert(PGW * PASSES == NU, "one store instruction per 8 accumulator values");
    static_assert(KS16 % NH == 0 || NH % KS16 == 0, "half-slots spread evenly over the k-steps");
    char* tarea = (char*)(aff + 2 * 128 * CGW) + wave * (32 * TROW);
    auto epi_half = [&](int hs, const f32x16_t (&ac)[PGW][CGW], int tile) {
        const int i = hs / (SLICES * 2), k = (hs / 2) % SLICES, q2 = hs % 2;
        const int jc = k / 2, qh = k % 2;
        float sc[4], sh[4];
        if constexpr (AFF_REGS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[e] = affr[k][4 * q2 + e]; sh[e] = affr[k][8 + 4 * q2 + e]; }
        } else {
            if (q2 == 0) {   // the constants of the NEXT slice are fetched one slot ahead of their use
#pragma unroll
                for (int e = 0; e < 4; ++e) affc[e] = affn[e];
                load_aff((k + 1) % SLICES, affn);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[e] = __uint_as_float(affc[q2][e]); sh[e] = __uint_as_float(affc[2 + q2][e]); }
        }
        uint32_t pk[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int e = 8 * qh + 4 * q2 + 2 * m;
            float t0 = ac[i][jc][e] * sc[2 * m] + sh[2 * m], t1 = ac[i][jc][e + 1] * sc[2 * m + 1] + sh[2 * m + 1];
            if (ACT) { t0 = silu<DT>(t0); t1 = silu<DT>(t1); }
            pk[m] = f32x2_to_bf16x2(t0, t1);
        }
        // channels 16 k + 8 q2 + 4 h + (0 .. 3) of pixel r
        *(u32x2*)(tarea + r * TROW + (16 * k + 8 * q2 + 4 * h) * 2) = u32x2{pk[0], pk[1]};
        if (hs % (SLICES * 2) == SLICES * 2 - 1) {             // the pixel group is complete: flush it
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int px = ps * PPP + lane / LPP, part = lane % LPP;
                const u32x4 out = *(const u32x4*)(tarea + px * TROW + part * 16);
                const int pix = tile * 64 + 32 * i + px;
                const bool ok = (unsigned)pix < (unsigned)p.npix;      // also false for the tile "-1" before the first one
                const int voff = ok ? (pix * p.y_ld + cw0) * 2 + part * 16 : -1;
                __builtin_amdgcn_raw_buffer_store_b128(out, rsrc, voff, 0, 0);
            }
        }
    };

    // Counted wait at the head of a unit: every vector-memory instruction this wave issued after the unit's last
    // DMA piece may still be outstanding.  Those are the pieces of the D - 1 units issued since, plus the NU stores
    // of every first unit of a tile among the D - 1 units processed since (the stores of a tile go out during the
    // first unit of the next one).  During the first D units of a workgroup fewer stores have been issued than the
    // steady-state count assumes, so they wait for the pieces alone (stricter is always safe).  The choice is a
    // branch at the head of a unit, where the wait and the barrier end the scheduling region anyway.
    // (Stores that only exist to be counted must not be used to even this out: a store whose offset fails the range
    // check of the buffer resource retires at once, ahead of older loads.)
    int bi = 0, prev_tile = -1, warm = 0;
    // One tile: KCH units.  `ac` receives the tile, `ap` holds the previous tile's results for the epilogue.
    auto body = [&](f32x16_t (&ac)[PGW][CGW], const f32x16_t (&ap)[PGW][CGW]) {
        const int cur_tile = q[0].tile;
#pragma unroll
        for (int ch = 0; ch < KCH; ++ch) {
            int stores = 0;
#pragma unroll
            for (int k = 1; k < D; ++k) stores += ((ch - k) % KCH + KCH) % KCH == 0 ? NU : 0;
            wait_vmcnt((D - 1) * MAXJ + (warm >= D ? stores : 0));
            ++warm;
            __builtin_amdgcn_s_barrier();
            const char* buf = bufs + bi * WRES_UNIT;
            char* nbuf = bufs + (bi == 0 ? NBUF - 1 : bi - 1) * WRES_UNIT;
#pragma unroll
            for (int i = 0; i + 1 < D; ++i) q[i] = q[i + 1];
            q[D - 1] = next_unit();
            const size_t inc = unit_offset(q[D - 1]);
            bi = bi + 1 == NBUF ? 0 : bi + 1;
            u32x4 bb[2][PGW];
#pragma unroll
            for (int i = 0; i < PGW; ++i) bb[0][i] = *(const u32x4*)(buf + b_off[i]);
#pragma unroll
            for (int s = 0; s < KS16; ++s) {
                if (s + 1 < KS16) {
#pragma unroll
                    for (int i = 0; i < PGW; ++i) bb[(s + 1) & 1][i] = *(const u32x4*)(buf + b_off[i] + (s + 1) * 256);
                }
#pragma unroll
                for (int k = s * MAXJ / KS16; k < (s + 1) * MAXJ / KS16; ++k) issue_piece(k, inc, nbuf);
                if (ch == 0) {
#pragma unroll
                    for (int hs = s * NH / KS16; hs < (s + 1) * NH / KS16; ++hs) epi_half(hs, ap, prev_tile);
                }
#pragma unroll
                for (int i = 0; i < PGW; ++i)
#pragma unroll
                    for (int jc = 0; jc < CGW; ++jc) {
                        if (ch == 0 && s == 0) {   // the tile's first k-step starts from a constant zero (no register clearing)
                            const f32x16_t z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            ac[i][jc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A[ch][jc][s]),
                                                                                __builtin_bit_cast(bf16x8_t, bb[s & 1][i]), z, 0, 0, 0);
                        } else {
                            mma_step<DT>(ac[i][jc], A[ch][jc][s], bb[s & 1][i]);
                        }
                    }
            }
        }
        prev_tile = cur_tile;
    };
    f32x16_t accA[PGW][CGW], accB[PGW][CGW];
#pragma unroll
    for (int i = 0; i < PGW; ++i)
#pragma unroll
        for (int jc = 0; jc < CGW; ++jc)
#pragma unroll
            for (int e = 0; e < 16; ++e) accB[i][jc][e] = 0.0f;
    for (int j = 0; j < p.rounds; j += 2) {
        body(accA, accB);
        body(accB, accA);
    }
#pragma unroll
    for (int hs = 0; hs < NH; ++hs) epi_half(hs, accB, prev_tile);
}

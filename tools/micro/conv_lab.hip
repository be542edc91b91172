// Times odamd_conv_forward on single layers of the network (bf16, batch 64) with HIP events.
// The kernels are compiled into this binary straight from the library source, so ablation macros
// (-DLAB_...) and experimental variants can be tried without touching the shared library.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I objectdetection_amd/csrc -o tools/micro/conv_lab tools/micro/conv_lab.hip
#define ODAMD_LAB
#define LAB_DMA_VARIANTS \
    ODAMD_DMA_CASE(3, 8, 2, 4, 1, 32, 3) ODAMD_DMA_CASE(3, 8, 2, 4, 1, 64, 2) ODAMD_DMA_CASE(3, 4, 2, 2, 2, 32, 3) \
    ODAMD_DMA_CASE(3, 4, 2, 4, 1, 32, 3) ODAMD_DMA_CASE(3, 4, 4, 2, 2, 32, 2) ODAMD_DMA_CASE(3, 8, 4, 4, 2, 32, 2) \
    ODAMD_DMA_CASE(3, 8, 4, 4, 2, 32, 3) ODAMD_DMA_CASE(1, 4, 4, 2, 2, 128, 2) ODAMD_DMA_CASE(1, 8, 4, 2, 2, 128, 2) ODAMD_DMA_CASE(1, 4, 4, 2, 2, 64, 2) \
    ODAMD_DMA_CASE(1, 8, 4, 4, 2, 64, 3) ODAMD_DMA_CASE(1, 8, 8, 4, 2, 64, 2) ODAMD_DMA_CASE(1, 8, 4, 4, 2, 128, 2) \
    ODAMD_DMA_CASE(3, 8, 4, 2, 2, 32, 2) ODAMD_DMA_CASE(3, 8, 2, 4, 2, 32, 2)
#define LAB_S3_VARIANTS \
    ODAMD_S3_CASE(2, 1, 1, 4, 256, 3) ODAMD_S3_CASE(2, 1, 1, 4, 256, 4) ODAMD_S3_CASE(2, 1, 1, 8, 384, 3) \
    ODAMD_S3_CASE(2, 2, 2, 4, 640, 2) ODAMD_S3_CASE(4, 2, 1, 4, 256, 2) ODAMD_S3_CASE(4, 2, 1, 4, 192, 3) \
    ODAMD_S3_CASE(4, 2, 1, 4, 192, 2) ODAMD_S3_CASE(4, 1, 1, 4, 256, 2) ODAMD_S3_CASE(4, 1, 1, 4, 192, 3) \
    ODAMD_S3_CASE(4, 1, 1, 8, 384, 2)
#define LAB_DMA_S2_VARIANTS ODAMD_DMA_S2_CASE(8, 4, 2, 4, 32, 2) ODAMD_DMA_S2_CASE(8, 2, 4, 1, 32, 2) ODAMD_DMA_S2_CASE(4, 2, 4, 1, 32, 3) ODAMD_DMA_S2_CASE(4, 2, 4, 1, 64, 2) ODAMD_DMA_S2_CASE(4, 2, 2, 2, 32, 2)
#define LAB_WRES_VARIANTS ODAMD_WRES_CASE(1, 2, 4) ODAMD_WRES_CASE(1, 1, 4) ODAMD_WRES_CASE(1, 2, 2) ODAMD_WRES_CASE(2, 2, 2) ODAMD_WRES_CASE(2, 1, 4) ODAMD_WRES_CASE(2, 1, 2)
#include "../../objectdetection_amd/csrc/conv.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

struct Case { const char* name; int h, cin, cout, k, s, pool; };

static uint16_t bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
}

int main(int argc, char** argv) {
    const int B = 64;
    std::vector<Case> cases = {
        {"3x3 128->128 @20", 20, 128, 128, 3, 1, 0},  {"3x3 256->256 @20", 20, 256, 256, 3, 1, 0},
        {"3x3 256->128 @20", 20, 256, 128, 3, 1, 0},  {"1x1 512->512 @20", 20, 512, 512, 1, 1, 0},
        {"1x1 1024->256 @20", 20, 1024, 256, 1, 1, 0}, {"1x1 512->256 @20", 20, 512, 256, 1, 1, 0},
        {"1x1 256->128 @20", 20, 256, 128, 1, 1, 0},  {"3x3 128->128 @40", 40, 128, 128, 3, 1, 0},
        {"3x3 64->64 @40", 40, 64, 64, 3, 1, 0},      {"3x3 128->64 @40", 40, 128, 64, 3, 1, 0},
        {"1x1 512->512 @40", 40, 512, 512, 1, 1, 0},  {"1x1 256->256 @40", 40, 256, 256, 1, 1, 0},
        {"1x1 512->128 @40", 40, 512, 128, 1, 1, 0},  {"3x3s2 256->256 @40", 40, 256, 256, 3, 2, 0},
        {"3x3s2 128->128 @40", 40, 128, 128, 3, 2, 0}, {"1x1 256->256 @80", 80, 256, 256, 1, 1, 0},
        {"3x3 64->64 @80", 80, 64, 64, 3, 1, 0},      {"3x3 32->32 @80", 80, 32, 32, 3, 1, 0},
        {"1x1 128->128 @160", 160, 128, 128, 1, 1, 0}, {"3x3 32->32 @160", 160, 32, 32, 3, 1, 0},
        {"1x1p 128->64 @160", 160, 128, 64, 1, 1, 1},  {"3x3 32->32 @320", 320, 32, 32, 3, 1, 0},
        {"3x3s2 32->64 @320", 320, 32, 64, 3, 2, 0},
        {"1x1 256->128 @80", 80, 256, 128, 1, 1, 0},  {"1x1 512->256 @40", 40, 512, 256, 1, 1, 0},
        {"1x1 256->256 @20", 20, 256, 256, 1, 1, 0},
        {"3x3s2 64->64 @160", 160, 64, 64, 3, 2, 0},   {"3x3s2 128->128 @80", 80, 128, 128, 3, 2, 0},
        {"3x3s2 64->64 @80", 80, 64, 64, 3, 2, 0},
        {"3x3 64->32 @80", 80, 64, 32, 3, 1, 0},
        // preset l (base width 64)
        {"3x3 128->128 @160", 160, 128, 128, 3, 1, 0}, {"3x3 128->128 @320", 320, 128, 128, 3, 1, 0},          // 30, 31
        {"3x3 256->256 @80", 80, 256, 256, 3, 1, 0},   {"3x3 512->512 @40", 40, 512, 512, 3, 1, 0},            // 32, 33
        {"3x3s2 64->128 @640", 640, 64, 128, 3, 2, 0}, {"3x3s2 128->256 @320", 320, 128, 256, 3, 2, 0},        // 34, 35
        {"3x3s2 256->256 @160", 160, 256, 256, 3, 2, 0}, {"1x1 2048->2048 @40", 40, 2048, 2048, 1, 1, 0},      // 36, 37
        {"1x1 1024->1024 @80", 80, 1024, 1024, 1, 1, 0}, {"1x1 4096->1024 @20", 20, 4096, 1024, 1, 1, 0},      // 38, 39
        {"1x1 2048->1024 @20", 20, 2048, 1024, 1, 1, 0}, {"1x1 512->512 @160", 160, 512, 512, 1, 1, 0},        // 40, 41
        {"3x3 1024->1024 @20", 20, 1024, 1024, 3, 1, 0},                                                        // 42
    };
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    S3Config s3_env = {};       // LAB_S3="ksteps,cg,st,pg,npad,nbuf": the persistent 3x3 kernel instantiation to force
    if (const char* e = getenv("LAB_S3"))
        sscanf(e, "%d,%d,%d,%d,%d,%d", &s3_env.ksteps, &s3_env.cg, &s3_env.st, &s3_env.pg, &s3_env.npad_max, &s3_env.nbuf);
    const int ldmul = argc > 2 ? atoi(argv[2]) : 1;      // x_ld = ldmul * cin, y_ld = ldmul * cout (slices of wider tensors)
    size_t max_in = 0, max_out = 0, max_w = 0;
    for (size_t ci = 0; ci < cases.size(); ++ci) {
        const Case& c = cases[ci];
        if (argc > 1 && atoi(argv[1]) >= 0 && (int)ci != atoi(argv[1])) continue;      // buffers for the selected case only
        const int ho = c.pool ? c.h / 2 : (c.s == 2 ? c.h / 2 : c.h);
        max_in = std::max(max_in, (size_t)B * c.h * c.h * c.cin);
        max_out = std::max(max_out, (size_t)B * ho * ho * c.cout);
        max_w = std::max(max_w, (size_t)c.cout * c.k * c.k * c.cin);
    }
    uint16_t *x, *w, *y;
    float *sc, *sh;
    hipMalloc(&x, max_in * 2 * ldmul);
    hipMalloc(&y, max_out * 2 * ldmul);
    hipMalloc(&w, max_w * 2);
    hipMalloc(&sc, 4096 * 4);
    hipMalloc(&sh, 4096 * 4);
    {
        std::vector<uint16_t> hx(max_in * ldmul), hw(max_w);
        uint32_t seed = 12345u;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (int)((seed >> 8) % 2001u) - 1000; };
        for (auto& v : hx) v = bf(rnd() * 1e-3f);
        for (auto& v : hw) v = bf(rnd() * 3e-5f);
        hipMemcpy(x, hx.data(), max_in * 2 * ldmul, hipMemcpyHostToDevice);
        hipMemcpy(w, hw.data(), max_w * 2, hipMemcpyHostToDevice);
        std::vector<float> one(4096, 1.0f), zero(4096, 0.01f);
        hipMemcpy(sc, one.data(), 4096 * 4, hipMemcpyHostToDevice);
        hipMemcpy(sh, zero.data(), 4096 * 4, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    uint16_t *wp_dev, *y2;
    hipMalloc(&wp_dev, max_w * 2 * 2);
    hipMalloc(&y2, max_out * 2 * ldmul);
    std::vector<uint16_t> hw(max_w);
    hipMemcpy(hw.data(), w, max_w * 2, hipMemcpyDeviceToHost);
    double total = 0, total_new = 0;
    auto time_it = [&](odamd_conv_desc& d, int* rc_out) {
        int rc = 0;
        for (int i = 0; i < 5; ++i) rc |= odamd_conv_forward(&d, nullptr);
        hipDeviceSynchronize();
        const int iters = 50;
        hipEventRecord(e0, nullptr);
        for (int i = 0; i < iters; ++i) rc |= odamd_conv_forward(&d, nullptr);
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        *rc_out = rc | (int)hipGetLastError();
        return ms * 1e3 / iters;
    };
    for (size_t ci = 0; ci < cases.size(); ++ci) {
        if (only >= 0 && (int)ci != only) continue;
        const Case& c = cases[ci];
        odamd_conv_desc d = {};
        d.x = x; d.w = w; d.scale = sc; d.shift = sh; d.y = y;
        d.batch = B; d.h = c.h; d.w_in = c.h; d.cin = c.cin; d.cout = c.cout;
        d.x_ld = c.cin * ldmul; d.y_ld = c.cout * ldmul; d.ksize = c.k; d.stride = c.s; d.act = 1; d.dtype = ODAMD_DTYPE_BF16;
        d.in_pool2 = c.pool;
        int rc = 0, rc2 = 0;
        const double us = time_it(d, &rc);
        const int hi = c.pool ? c.h / 2 : c.h, ho = c.s == 2 ? hi / 2 : hi;
        const double flops = 2.0 * B * ho * ho * c.cout * c.k * c.k * c.cin;
        const double bytes = 2.0 * ((double)B * c.h * c.h * c.cin + (double)B * ho * ho * c.cout + (double)c.cout * c.k * c.k * c.cin);
        const double ideal = std::max(flops / 2.5e15, bytes / 8e12) * 1e6;
        total += us;
        printf("%2zu %-20s rc=%d %7.2f us %6.1f TF/s %6.0f GB/s ideal %6.2f frac %.2f |", ci, c.name, rc, us, flops / us * 1e-6,
               bytes / us * 1e-3, ideal, ideal / us);
        static const DmaConfig variants[] = {
            {0, 0, 0, 0, 0, 0, 0},          // the library's own choice of a packed-weight kernel, if it has one
            {-3, 0, 0, 0, 0, 0, 0},         // without the register-resident 1x1 kernel (the LDS-DMA kernel's own choice)
            {-4, 1, 0, 0, 0, 0, 0},         // register-resident 1x1 kernel with one channel group per wave
            {-5, 4, 0, 0, 0, 0, 0}, {-5, 2, 0, 0, 0, 0, 0},   // ... with 4 / 2 ring buffers
            {-1, 4, 256, 3, 0, 0, 0}, {-1, 4, 256, 4, 0, 0, 0}, {-1, 8, 384, 3, 0, 0, 0}, {-2, 4, 640, 2, 0, 0, 0},   // conv_s3 variants: pg, npad, nbuf
            {3, 8, 4, 4, 2, 32, 2, 2}, {3, 8, 4, 2, 4, 32, 2, 2}, {3, 8, 2, 4, 1, 32, 2, 2}, {3, 4, 2, 4, 1, 32, 3, 2}, {3, 4, 2, 4, 1, 64, 2, 2}, {3, 4, 4, 2, 2, 32, 2, 2}, {3, 4, 2, 2, 2, 32, 2, 2},
            {3, 8, 2, 4, 1, 64, 2}, {3, 8, 2, 4, 1, 32, 3}, {3, 8, 2, 4, 1, 32, 2}, {3, 4, 2, 2, 2, 32, 3}, {3, 4, 2, 4, 1, 32, 3},
            {3, 4, 4, 2, 2, 32, 2}, {3, 8, 4, 4, 2, 32, 2}, {3, 8, 4, 4, 2, 32, 3}, {3, 8, 4, 2, 2, 32, 2}, {3, 8, 2, 4, 2, 32, 2}, {1, 8, 4, 2, 2, 128, 2}, {1, 4, 4, 2, 2, 128, 2},
            {1, 4, 4, 2, 2, 64, 3}, {1, 4, 4, 2, 2, 64, 2}, {1, 8, 4, 4, 2, 64, 3}, {1, 8, 8, 4, 2, 64, 2}, {1, 8, 4, 4, 2, 128, 2}};
        double best = us;
        for (const DmaConfig& v : variants) {
            if (v.ks > 0 && (v.ks != c.k || c.s != (v.st == 2 ? 2 : 1) || c.pool)) continue;
            if (v.ks > 0 && (only < 0 || getenv("LAB_QUICK"))) continue;     // the full sweep only with a case selected
            g_lab_s3 = {};
            g_lab_no_wres = 0;
            g_lab_wres = {};
            if (v.ks == -3 || v.ks == -4 || v.ks == -5) {
                if (c.k != 1 || c.pool) continue;
                if (v.ks == -5 && ((c.cin != 256 && c.cin != 512) || (c.cin == 512 && v.pg == 4 && c.cout % 256 == 0))) continue;
                if (v.ks == -3) g_lab_no_wres = 1; else if (v.ks == -4) g_lab_wres = {0, 1, 0}; else g_lab_wres = {0, 0, v.pg};
            } else
            if (v.ks < 0) {
                if (c.k != 3 || c.cin != 32 || c.s != -v.ks || c.cout != 32 * c.s) continue;
                g_lab_s3 = {2, c.s, c.s, v.pg, v.cg, v.wp};
            } else
            g_lab_force = v;
            if (v.ks == 0) g_lab_s3 = s3_env;
            int G = 0, CB = 0;
            if (v.ks > 0) g_lab_no_wres = 1;
            if (!odamd_conv_weight_layout(&d, &G, &CB)) { g_lab_force = {}; g_lab_s3 = {}; g_lab_no_wres = 0; g_lab_wres = {}; continue; }
            const int T = c.k * c.k, ncb = (c.cout + 32 * G - 1) / (32 * G), nch = c.cin * 2 / CB, KSN = CB / 32;
            std::vector<uint16_t> pk((size_t)ncb * nch * G * T * KSN * 512, 0);
            size_t o = 0;
            for (int cb = 0; cb < ncb; ++cb) for (int ch = 0; ch < nch; ++ch) for (int g = 0; g < G; ++g)
                for (int t = 0; t < T; ++t) for (int s2 = 0; s2 < KSN; ++s2) for (int hh = 0; hh < 2; ++hh)
                    for (int r = 0; r < 32; ++r) for (int e = 0; e < 8; ++e, ++o) {
                        const int co = cb * 32 * G + g * 32 + r, ci2 = ch * CB / 2 + s2 * 16 + hh * 8 + e;
                        pk[o] = co < c.cout ? hw[((size_t)co * T + t) * c.cin + ci2] : 0;
                    }
            hipMemcpy(wp_dev, pk.data(), pk.size() * 2, hipMemcpyHostToDevice);
            odamd_conv_desc d2 = d;
            d2.w = wp_dev; d2.y = y2; d2.w_packed = 1;
            hipMemset(y2, 0, (size_t)B * ho * ho * c.cout * 2 * ldmul);
            hipMemset(y, 0, (size_t)B * ho * ho * c.cout * 2 * ldmul);
            { int rcx; time_it(d, &rcx); }
            int rc2 = 0;
            const double us2 = time_it(d2, &rc2);
            g_lab_force = {};
            g_lab_s3 = {};
            g_lab_no_wres = 0;
            g_lab_wres = {};
            const size_t n = (size_t)B * ho * ho * c.cout * ldmul;
            std::vector<uint16_t> a(n), b2(n);
            hipMemcpy(a.data(), y, n * 2, hipMemcpyDeviceToHost);
            hipMemcpy(b2.data(), y2, n * 2, hipMemcpyDeviceToHost);
            double maxdiff = 0;
            for (size_t i = 0; i < n; ++i) {
                uint32_t ua = (uint32_t)a[i] << 16, ub = (uint32_t)b2[i] << 16;
                float fa, fb;
                memcpy(&fa, &ua, 4); memcpy(&fb, &ub, 4);
                const double df = fabs((double)fa - fb);
                if (!(df <= maxdiff)) maxdiff = df;
            }
#ifdef ODAMD_LAB_STAMPS
            if (v.ks <= 0 && c.cin == 32) {
                static std::vector<unsigned long long> st(1024 * 8 * 16);
                hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_lab_stamps), st.size() * 8);
                const char* nm[6] = {"top", "wait", "barrier", "issue", "mma", "epi"};
                for (int blk : {0, 77, 300})
                    for (int wv : {0, 3}) {
                        const unsigned long long* q = &st[(blk * 8 + wv) * 16];
                        printf("\n      blk %3d wave %d:", blk, wv);
                        for (int i = 1; i < 6; ++i) printf(" %s=%lld", nm[i], (long long)(q[i] - q[i - 1]));
                        printf(" tile=%lld", (long long)(q[5] - q[0]));
                    }
                printf("\n     ");
            } else if (v.ks > 0)
            {
                static std::vector<unsigned long long> st(1024 * 8 * 16);
                hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_lab_stamps), st.size() * 8);
                const char* nm[11] = {"start", "setup", "issue0", "wait0", "bar0", "mma0", "wait1", "bar1", "mma1", "loopend", "epi"};
                for (int blk : {0, 77}) {
                    const unsigned long long* q = &st[(blk * 8 + 0) * 16];
                    printf("\n      [%d,%d,%d,%d,%d,%d] blk %3d:", v.pg, v.cg, v.wp, v.wc, v.ckb, v.nbuf, blk);
                    for (int i = 1; i < 11; ++i) printf(" %s=%lld", nm[i], (long long)(q[i] - q[i - 1]));
                    printf(" total=%lld", (long long)(q[10] - q[0]));
                }
                printf("\n     ");
            }
#endif
            printf(" [%s%s%d,%d,%d,%d,%d,%d]", v.st == 2 ? "s2:" : "", v.ks == -3 ? "dma:" : v.ks == -4 ? "wres1:" : v.ks == -5 ? "nbuf:" : "", v.pg, v.cg, v.wp, v.wc, v.ckb, v.nbuf);
            if (rc2) printf(" rc=%d", rc2); else printf(" %.1f%s", us2, maxdiff > 0.02 ? "(BAD)" : "");
            if (!rc2 && maxdiff <= 0.02 && us2 < best) best = us2;
        }
        total_new += best;
        printf(" best %.1f frac %.2f\n", best, ideal / best);
    }
    printf("total %.1f us  with dma %.1f us\n", total, total_new);

    return 0;
}

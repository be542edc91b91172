// Times odamd_conv_first2_forward (640x640 image, batch 64) with HIP events and, when built with
// -DODAMD_LAB_STAMPS, prints the cycle stamps of a few workgroups' phases.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I objectdetection_amd/csrc -o tools/micro/conv_lab_F tools/micro/first2_lab.hip
#define ODAMD_LAB
#include "../../objectdetection_amd/csrc/conv.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    const int B = 64, H = 640, W = 640, Ho = 320, Wo = 320;
    float *img, *w0, *sc0, *sh0, *sc1, *sh1;
    uint16_t *w1, *y;
    hipMalloc(&img, (size_t)B * 3 * H * W * 4);
    hipMalloc(&w0, 16 * 27 * 4); hipMalloc(&sc0, 64); hipMalloc(&sh0, 64);
    hipMalloc(&sc1, 128); hipMalloc(&sh1, 128);
    hipMalloc(&w1, 32 * 9 * 16 * 2);
    hipMalloc(&y, (size_t)B * Ho * Wo * 32 * 2);
    {
        std::vector<float> h((size_t)B * 3 * H * W);
        srand(1);
        for (auto& v : h) v = (rand() % 1000) * 1e-3f;
        hipMemcpy(img, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> ww(16 * 27);
        for (auto& v : ww) v = (rand() % 2001 - 1000) * 3e-4f;
        hipMemcpy(w0, ww.data(), ww.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> one(32, 1.0f), sm(32, 0.01f);
        hipMemcpy(sc0, one.data(), 64, hipMemcpyHostToDevice); hipMemcpy(sh0, sm.data(), 64, hipMemcpyHostToDevice);
        hipMemcpy(sc1, one.data(), 128, hipMemcpyHostToDevice); hipMemcpy(sh1, sm.data(), 128, hipMemcpyHostToDevice);
        std::vector<uint16_t> w1h(32 * 9 * 16, 0x3c00);
        hipMemcpy(w1, w1h.data(), w1h.size() * 2, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int rc = 0;
    for (int i = 0; i < 3; ++i) rc |= odamd_conv_first2_forward(img, w0, sc0, sh0, w1, sc1, sh1, y, B, H, W, 32, nullptr);
    hipDeviceSynchronize();
    const int iters = 20;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; ++i) rc |= odamd_conv_first2_forward(img, w0, sc0, sh0, w1, sc1, sh1, y, B, H, W, 32, nullptr);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters, bytes = (double)B * 3 * H * W * 4 + (double)B * Ho * Wo * 32 * 2;
    printf("first2 rc=%d %.1f us  %.0f GB/s (algorithmic)\n", rc | (int)hipGetLastError(), us, bytes / us * 1e-3);
#ifdef ODAMD_LAB_STAMPS
    static std::vector<unsigned long long> st(1024 * 8 * 16);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_lab_stamps), st.size() * 8);
    const char* nm[9] = {"start", "issue", "imgwait+lds", "barA", "B", "barB", "C", "barC", "epi"};
    {
        unsigned long long lo = ~0ull, hi = 0, life = 0; int late = 0, nb = 0;
        for (int blk = 0; blk < 1024; ++blk) { const unsigned long long* q = &st[(blk * 8) * 16]; if (!q[0]) continue; ++nb; if (q[0] < lo) lo = q[0]; if (q[10] > hi) hi = q[10]; life += q[10] - q[0]; }
        for (int blk = 0; blk < 1024; ++blk) { const unsigned long long* q = &st[(blk * 8) * 16]; if (q[0] && q[0] > lo + 50000) ++late; }
        printf("blocks %d span %llu cycles, mean life %llu, late starters %d\n", nb, hi - lo, life / (nb ? nb : 1), late);
    }
    for (int blk : {0, 3, 7})
        for (int wv : {0, 3}) {
            const unsigned long long* q = &st[(blk * 8 + wv) * 16];
            printf("blk %d wave %d:", blk, wv);
            for (int i = 1; i < 9; ++i) printf(" %s=%lld", nm[i], (long long)(q[i] - q[i - 1]));
            printf(" A=%lld tile=%lld\n", (long long)(q[2] - q[9]), (long long)(q[8] - q[9]));
        }
#endif
    return 0;
}

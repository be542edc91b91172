// Phase times of conv_pair_kernel (cycle stamps of the second tile of a workgroup) and its launch time.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DODAMD_LAB_STAMPS -I include -I objectdetection_amd/csrc -o tools/micro/pair_lab tools/micro/pair_lab.hip
// usage: pair_lab [h] [cin] [ld] [act]
#define ODAMD_LAB
#include "../../objectdetection_amd/csrc/conv.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    const int B = 64, H = argc > 1 ? atoi(argv[1]) : 160, cin = argc > 2 ? atoi(argv[2]) : 32, ld = argc > 3 ? atoi(argv[3]) : 128;
    const int act = argc > 4 ? atoi(argv[4]) : 1;
    const size_t n = (size_t)B * H * H * ld;
    uint16_t *buf, *w1, *w2;
    float *sc, *sh;
    hipMalloc(&buf, n * 2);
    hipMalloc(&w1, 32 * 9 * cin * 2);
    hipMalloc(&w2, 32 * 9 * 32 * 2);
    hipMalloc(&sc, 128);
    hipMalloc(&sh, 128);
    {
        std::vector<uint16_t> hx(n), hw(32 * 9 * 64);
        uint32_t seed = 777u;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (uint16_t)(0x3c00 + ((seed >> 12) & 0xff) + ((seed >> 31) << 15)); };
        for (auto& v : hx) v = rnd();
        for (auto& v : hw) v = (uint16_t)(rnd() - 0x0400);
        hipMemcpy(buf, hx.data(), n * 2, hipMemcpyHostToDevice);
        hipMemcpy(w1, hw.data(), 32 * 9 * cin * 2, hipMemcpyHostToDevice);
        hipMemcpy(w2, hw.data(), 32 * 9 * 32 * 2, hipMemcpyHostToDevice);
        std::vector<float> one(32, 1.0f), z(32, 0.0f);
        hipMemcpy(sc, one.data(), 128, hipMemcpyHostToDevice);
        hipMemcpy(sh, z.data(), 128, hipMemcpyHostToDevice);
    }
    odamd_conv_desc a = {}, b = {};
    a.x = buf; a.w = w1; a.scale = sc; a.shift = sh; a.y = buf + cin;
    a.batch = B; a.h = H; a.w_in = H; a.cin = cin; a.cout = 32; a.x_ld = ld; a.y_ld = ld; a.ksize = 3; a.stride = 1; a.act = act;
    a.dtype = ODAMD_DTYPE_BF16; a.w_packed = 1;
    b = a;
    b.x = a.y; b.w = w2; b.y = buf + cin + 32; b.cin = 32;
    if (!odamd_conv_chain_pair_supported(&a, &b)) { printf("unsupported\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int rc = 0;
    for (int i = 0; i < 3; ++i) rc |= odamd_conv_chain_pair_forward(&a, &b, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < 20; ++i) rc |= odamd_conv_chain_pair_forward(&a, &b, nullptr);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%dx%d b%d %d->32->32 ld %d act %d: rc %d  %.1f us per launch\n", H, H, B, cin, ld, act, rc, ms * 1e3 / 20);
#ifdef ODAMD_LAB_STAMPS
    static std::vector<unsigned long long> st(1024 * 8 * 16);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_lab_stamps), st.size() * 8);
    const char* nm[11] = {"top", "wait", "bar", "mma1", "bar", "epi1", "bar", "mma2", "bar", "issue", "epi2"};
    for (int blk : {0, 77, 300, 511})
        for (int wv : {0, 3}) {
            const unsigned long long* q = &st[(blk * 8 + wv) * 16];
            printf("  blk %3d wave %d:", blk, wv);
            for (int i = 1; i < 11; ++i) printf(" %s=%lld", nm[i], (long long)(q[i] - q[i - 1]));
            printf(" tile=%lld\n", (long long)(q[10] - q[0]));
        }
#endif
    return 0;
}

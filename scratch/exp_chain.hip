// Round-5 experiment: the one-launch bottom MLP (mlp_chain_kernel) alone, with its parts switched off one at a time.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../openrec_amd/csrc -I../include exp_chain.hip -o exp_chain
#include "exp_chain_kernel.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstring>
#include <algorithm>
void orx_set_error(const char*, ...) {}
void orx_prof_begin(orx_ctx*, int) {}
void orx_prof_end(orx_ctx*, int) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename F> static float time_us(F f, int reps = 30) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) f();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.0f / reps;
}

template <int DBG> static void launch(const ChainArgs& g0) {
    ChainArgs g = g0;
    int maxw = g.K0;
    for (int s = 0; s < g.n_stages; ++s) maxw = maxw > g.S[s].N ? maxw : g.S[s].N;
    g.pitch = ch_pitch(maxw);
    const size_t shm = (size_t)2 * CH_ROWS * g.pitch * sizeof(_Float16);
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void*)mlp_chain_kernel<DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024)); once = true; }
    hipLaunchKernelGGL(mlp_chain_kernel<DBG>, dim3((g.B + CH_ROWS - 1) / CH_ROWS), dim3(CH_NT), shm, 0, g);
}

int main() {
    const int B = 8192, F = 27, d = 128;
    const int w[4] = {13, 512, 256, 128};
    float* dense; CK(hipMalloc(&dense, (size_t)B * 13 * 4)); CK(hipMemset(dense, 0, (size_t)B * 13 * 4));
    float* Z; CK(hipMalloc(&Z, (size_t)B * F * d * 4));
    _Float16* dense16; CK(hipMalloc(&dense16, (size_t)B * 16 * 2));
    ChainArgs g; memset(&g, 0, sizeof(g));
    g.B = B; g.n_stages = 3; g.X32 = dense; g.ldx = 13; g.K0 = 13; g.xscale = 1.0f; g.X16out = dense16; g.ldx16 = 16;
    for (int l = 0; l < 3; ++l) {
        const int in = w[l], out = w[l + 1], ldw = in >= 64 ? (in + 63) & ~63 : (in + 7) & ~7;
        _Float16* W; CK(hipMalloc(&W, (size_t)out * ldw * 2 + 4096)); CK(hipMemset(W, 0, (size_t)out * ldw * 2 + 4096));
        float* b; CK(hipMalloc(&b, out * 4)); CK(hipMemset(b, 0, out * 4));
        _Float16* Y16 = nullptr; if (l < 2) CK(hipMalloc(&Y16, (size_t)B * out * 2));
        ChainStage& S = g.S[l];
        S.W = W; S.ldw = ldw; S.N = out; S.K = in; S.bias = b; S.act = 1; S.Y16 = Y16; S.ldy16 = out;
        if (l == 2) { S.Y32 = Z + (size_t)(F - 1) * d; S.ldy32 = (int64_t)F * d; }
    }
    printf("forward chain 13-512-256-128, B=%d\n", B);
    printf("  full                         %7.2f us\n", time_us([&] { launch<0>(g); }));
    printf("  no global stores             %7.2f us\n", time_us([&] { launch<1>(g); }));
    printf("  weights: L1 hits             %7.2f us\n", time_us([&] { launch<2>(g); }));
    printf("  weights: packed order        %7.2f us\n", time_us([&] { launch<16>(g); }));
    printf("  no MFMA                      %7.2f us\n", time_us([&] { launch<4>(g); }));
    printf("  no stores, L1 weights        %7.2f us\n", time_us([&] { launch<3>(g); }));
    printf("  no stores, packed            %7.2f us\n", time_us([&] { launch<17>(g); }));
    printf("  no stores, L1, no MFMA, no in%7.2f us\n", time_us([&] { launch<15>(g); }));
    // single stages
    for (int l = 0; l < 3; ++l) {
        ChainArgs h = g; h.n_stages = 1; h.S[0] = g.S[l]; h.K0 = l == 0 ? 13 : 13;   // (the input stage stays the 13-wide one; stage l reads LDS garbage beyond it)
        printf("  only stage %d (K=%d N=%d): full %7.2f us, no stores %7.2f us, packed+no stores %7.2f us\n", l, h.S[0].K, h.S[0].N,
               time_us([&] { launch<0>(h); }), time_us([&] { launch<1>(h); }), time_us([&] { launch<17>(h); }));
    }
    {   // where the time goes inside a workgroup: s_memtime (100 MHz) at the stage boundaries
        unsigned long long* st; CK(hipMalloc(&st, (size_t)256 * 8 * 8)); CK(hipMemset(st, 0, 256 * 64));
        ChainArgs h = g; h.xgb = reinterpret_cast<float*>(st);
        for (int rep = 0; rep < 3; ++rep) {
            launch<32>(h); CK(hipDeviceSynchronize());
            std::vector<unsigned long long> v(256 * 8);
            CK(hipMemcpy(v.data(), st, v.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int b = 0; b < 256; ++b) { t0 = std::min(t0, v[b * 8]); t1 = std::max(t1, v[b * 8 + 4]); }
            printf("  rep %d: first start -> last end %.2f us;", rep, (t1 - t0) * 0.01);
            for (int b : {0, 100, 255}) {
                printf("  wg %d: start +%.2f |", b, (v[b * 8] - t0) * 0.01);
                for (int i = 1; i < 5; ++i) printf(" %.2f", (v[b * 8 + i] - v[b * 8 + i - 1]) * 0.01);
            }
            printf("\n");
        }
    }
    return 0;
}

// Round-6 experiment: where the time of the SHORT-K products of the DLRM top MLP goes (8192 x 1024 x 512: 21.8-23.6 us in the step, as long as the
// 8192 x 1024 x 1024 product) -- gemm16_nt_dma_kernel<4,2,4,4,1,3> with its parts switched off, K = 512 beside K = 1024, with the epilogue forms of
// the step (fp16-only forward; backward with the fused activation backward: Y16 read, column sums).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../openrec_amd/csrc -I../include exp_k512.hip -o exp_k512
#include "../openrec_amd/csrc/kernels_gemm16.hip"
#include <cstdio>
#include <vector>
void orx_set_error(const char*, ...) {}
void orx_prof_begin(orx_ctx*, int) {}
void orx_prof_end(orx_ctx*, int) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <typename F> static float time_us(F f, int reps = 40) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) f();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.0f / reps;
}
template <int WM, int WN, int TM, int TN, int MINB, int NS, int DBG>
static void launch_dbg(const Nt16Args& g) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    constexpr size_t shm = (size_t)NS * (BM + BN) * 64 * 2;
    auto kern = gemm16_nt_dma_kernel<WM, WN, TM, TN, MINB, NS, false, DBG>;
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); once = true; }
    const unsigned nb = (unsigned)(((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN));
    hipLaunchKernelGGL(kern, dim3(nb), dim3(64 * WM * WN), shm, 0, g);
}
int main() {
    const int M = 8192, N = 1024;
    for (int K : {512, 1024, 256}) {
        std::vector<_Float16> hA((size_t)M * K), hB((size_t)N * K), hY((size_t)M * N);
        unsigned s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 9) & 0xffff) / 65536.0f - 0.5f; };
        for (auto& v : hA) v = (_Float16)rnd();
        for (auto& v : hB) v = (_Float16)(rnd() * 0.1f);
        for (auto& v : hY) v = (_Float16)rnd();
        _Float16 *A, *B, *C16, *Y; float* C; float* bias; float* gb;
        CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&B, hB.size() * 2)); CK(hipMalloc(&C16, (size_t)M * N * 2)); CK(hipMalloc(&Y, (size_t)M * N * 2)); CK(hipMalloc(&C, (size_t)M * N * 4));
        CK(hipMalloc(&bias, N * 4)); CK(hipMemset(bias, 0, N * 4)); CK(hipMalloc(&gb, (size_t)64 * N * 4));
        CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(Y, hY.data(), hY.size() * 2, hipMemcpyHostToDevice));
        const double gf = 2.0 * M * N * K * 1e-9;
        printf("== M %d N %d K %d  (%.1f GFLOP)\n", M, N, K, gf);
        auto rep = [&](const char* name, float us) { printf("  %-64s %7.2f us  %6.0f TFLOP/s\n", name, us, gf / us * 1e3); };
        Nt16Args fwd{A, K, B, K, nullptr, N, C16, N, bias, M, N, K, 1, nullptr, nullptr, 0, 0, nullptr};          // forward, lean (fp16 only)
        Nt16Args bwd{A, K, B, K, nullptr, N, C16, N, nullptr, M, N, K, 0, nullptr, Y, N, 1, gb};                   // input gradient + fused activation backward + column sums
        Nt16Args bwd32 = bwd; bwd32.C = C;                                                                          // ... with the fp32 copy as well
        rep("forward form (fp16 out, bias, relu), write-through stores", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 32>(fwd); }));
        rep("backward form (Y16 read, column sums, fp16 out), write-through", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 32>(bwd); }));
        rep("backward form + fp32 copy", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 32>(bwd32); }));
        { unsigned long long* mk; CK(hipMalloc(&mk, (size_t)M * N / 32 * 8)); CK(hipMemset(mk, 0x5a, (size_t)M * N / 32 * 8));
          Nt16Args v = bwd; v.mask_in = mk; rep("backward form, relu' from the mask words (one request per lane)", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 32>(v); }));
          Nt16Args w = fwd; w.mask_out = mk; rep("forward form + mask words written", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 32>(w); })); CK(hipFree(mk)); }
        { Nt16Args v = bwd; v.ldy = 0; rep("backward form, Y16 rows all the same 2 KB (ldy = 0: no HBM read)", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 32>(v); })); }
        { Nt16Args v = bwd; v.act_y = 0; rep("backward form, act_y = 0 (Y loaded, not applied)", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 32>(v); })); }
        { Nt16Args v = bwd; v.act_y = 0; v.gb = nullptr; rep("backward form, act_y = 0, no column sums", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 32>(v); })); }
        { Nt16Args v = fwd; v.bias = nullptr; rep("forward form without bias (relu only)", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 32>(v); })); }
        { Nt16Args v = bwd; v.gb = nullptr; rep("backward form WITHOUT the column sums", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 32>(v); })); }
        { Nt16Args v = fwd; v.bias = nullptr; v.act = 0; rep("plain fp16 store (no bias, no activation)", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 32>(v); })); }
        { Nt16Args v = bwd; rep("backward form, ordinary stores", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 0>(v); })); }
        { Nt16Args v = fwd; rep("forward form, ordinary stores", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 0>(v); })); }
        rep("no epilogue", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 1>(fwd); }));
        rep("no epilogue, no DMA in loop", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 3>(fwd); }));
        rep("no epilogue, no MFMA", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 5>(fwd); }));
        rep("no epilogue, neither (launch + prologue)", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 7>(fwd); }));
        rep("forward epilogue only (no MFMA, no DMA in loop)", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 6 + 32>(fwd); }));
        rep("backward epilogue only", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 6 + 32>(bwd); }));
        rep("128x64 tiles, 2 per CU: forward form", time_us([&] { launch_dbg<2, 2, 4, 2, 2, 3, 32>(fwd); }));
        rep("128x128 tiles 2 stages, 2 per CU: forward form", time_us([&] { launch_dbg<2, 2, 4, 4, 2, 2, 32>(fwd); }));
        rep("128x128 tiles 2 stages, 2 per CU: backward form", time_us([&] { launch_dbg<2, 2, 4, 4, 2, 2, 32>(bwd); }));
        {
            unsigned long long* dbg; CK(hipMalloc(&dbg, 256 * 16));
            Nt16Args gd = fwd; gd.C = reinterpret_cast<float*>(dbg);
            for (int r = 0; r < 3; ++r) launch_dbg<4, 2, 4, 4, 1, 3, 9>(gd);
            CK(hipDeviceSynchronize());
            unsigned long long ho[512]; CK(hipMemcpy(ho, dbg, sizeof(ho), hipMemcpyDeviceToHost));
            double c = 0, w = 0; for (int i = 0; i < 256; ++i) { c += ho[2 * i]; w += ho[2 * i + 1]; }
            c /= 256; w /= 256;
            printf("  main loop alone: %.0f cycles = %.2f us (%.2f GHz), %.0f cycles per K step\n", c, w / 100.0, c / (w * 10.0), c / ((K + 63) / 64));
            CK(hipFree(dbg));
        }
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C16)); CK(hipFree(Y)); CK(hipFree(C)); CK(hipFree(bias)); CK(hipFree(gb));
    }
    return 0;
}

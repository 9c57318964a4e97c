// Round-4 experiments: the LDS-DMA staged fp16 product (gemm16_nt_dma_kernel) against the register-staged one, with the parts
// of the kernel switched off one at a time.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../openrec_amd/csrc -I../include exp_dma.hip -o exp_dma
#include "../openrec_amd/csrc/kernels_gemm16.hip"
#include <cstdio>
#include <vector>
#include <cmath>
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

__global__ void flush_kernel(float4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = float4{1.f, 2.f, 3.f, 4.f};
}
static float4* g_flush = nullptr;
// one launch at a time, each behind a 512 MB write (caches cold) and, optionally, behind the launch that PRODUCES its operand
template <typename F, typename P> static float time_cold_us(F f, P producer, int reps = 8) {
    if (!g_flush) CK(hipMalloc(&g_flush, (size_t)512 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float tot = 0;
    for (int i = 0; i < reps + 2; ++i) {
        hipLaunchKernelGGL(flush_kernel, dim3(2048), dim3(256), 0, 0, g_flush, ((size_t)512 << 20) / 16);
        producer();
        CK(hipEventRecord(e0, 0)); f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 2) tot += ms;
    }
    return tot * 1000.0f / reps;
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

// MFMA rate and shader clock under load: every wavefront issues n x 16 independent-accumulator 16x16x32 products on random data
__global__ __launch_bounds__(512) void mfma_rate_kernel(const _Float16* src, int n, unsigned long long* out, float* sink) {
    h8 a = *reinterpret_cast<const h8*>(src + threadIdx.x * 8), b = *reinterpret_cast<const h8*>(src + 4096 + threadIdx.x * 8);
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float x = 0.f;
    for (int i = 0; i < 16; ++i) x += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
    if (x == 12345.678f) sink[0] = x;
}

int main() {
    {
        std::vector<_Float16> h(8192);
        unsigned s = 777; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((s >> 9) & 0xffff) / 65536.0f - 0.5f); }
        _Float16* d; unsigned long long* o; float* sink;
        CK(hipMalloc(&d, 16384)); CK(hipMalloc(&o, 256 * 16)); CK(hipMalloc(&sink, 4));
        CK(hipMemcpy(d, h.data(), 16384, hipMemcpyHostToDevice));
        for (int n : {64, 512}) {
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(mfma_rate_kernel, dim3(256), dim3(512), 0, 0, d, n, o, sink);
            CK(hipDeviceSynchronize());
            unsigned long long ho[512]; CK(hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost));
            double c = 0, w = 0; for (int i = 0; i < 256; ++i) { c += ho[2 * i]; w += ho[2 * i + 1]; }
            c /= 256; w /= 256;
            printf("mfma rate: %d x 16 products per wavefront, 2 wavefronts per SIMD: %.0f shader cycles, %.2f us -> %.2f GHz, %.1f cycles per product per SIMD\n",
                   n, c, w / 100.0, c / (w * 10.0), c / (n * 16.0 * 2.0));
        }
    }
    {   // ---- weight gradient: C[M][N] (slabs) = A16[K][M]^T * B16[K][N], K = 8192 samples
        const int K = 8192;
        orx_ctx ctx; ctx.num_cu = 256;
        const int shapes[][2] = {{1024, 1024}};
        for (auto& sh : shapes) {
            const int M = sh[0], N = sh[1];
            std::vector<_Float16> hA((size_t)K * M), hB((size_t)K * N);
            unsigned s = 4242;
            auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 9) & 0xffff) / 65536.0f - 0.5f; };
            for (auto& v : hA) v = (_Float16)rnd();
            for (auto& v : hB) v = (_Float16)(rnd() * 0.1f);
            _Float16 *A, *B; float* slab; float* C;
            int S, tiles, kchunk;
            orx_gemm16_tn_plan(&ctx, M, N, K, &S, &tiles, &kchunk);
            CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&B, hB.size() * 2)); CK(hipMalloc(&slab, (size_t)tiles * S * SLAB_STRIDE * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
            CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
            Tn16Args g{A, M, B, N, C, N, slab, M, N, K, kchunk, 1.0f};
            const double gf = 2.0 * M * N * K * 1e-9;
            printf("== weight gradient M %d N %d K %d (%.1f GFLOP), %d tiles x %d slices\n", M, N, K, gf, tiles, S);
            auto rep = [&](const char* name, float us) { printf("  %-44s %7.2f us  %6.0f TFLOP/s\n", name, us, gf / us * 1e3); };
            auto run_reg = [&] { hipFuncSetAttribute((const void*)gemm16_tn_kernel<2, 2, 4, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 288 * 2);
                                 hipLaunchKernelGGL((gemm16_tn_kernel<2, 2, 4, 4, 2>), dim3(tiles * S), dim3(256), 2 * 64 * 288 * 2, 0, g); };
            auto run_d2 = [&] { hipFuncSetAttribute((const void*)gemm16_tn_dma_kernel<2, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 256 * 2);
                                hipLaunchKernelGGL((gemm16_tn_dma_kernel<2, 2, false>), dim3(tiles * S), dim3(256), 2 * 64 * 256 * 2, 0, g); };
            auto run_d3 = [&] { hipFuncSetAttribute((const void*)gemm16_tn_dma_kernel<1, 3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 64 * 256 * 2);
                                hipLaunchKernelGGL((gemm16_tn_dma_kernel<1, 3, false>), dim3(tiles * S), dim3(256), 3 * 64 * 256 * 2, 0, g); };
            auto run_d2t = [&] { hipFuncSetAttribute((const void*)gemm16_tn_dma_kernel<2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 256 * 2);
                                 hipLaunchKernelGGL((gemm16_tn_dma_kernel<2, 2, true>), dim3(tiles * S), dim3(256), 2 * 64 * 256 * 2, 0, g); };
            const size_t nsl = (size_t)tiles * S * SLAB_STRIDE;
            std::vector<float> ref(nsl), got(nsl);
            CK(hipMemset(slab, 0, nsl * 4)); run_reg(); CK(hipDeviceSynchronize()); CK(hipMemcpy(ref.data(), slab, nsl * 4, hipMemcpyDeviceToHost));
            auto check = [&](const char* name) {
                CK(hipDeviceSynchronize()); CK(hipMemcpy(got.data(), slab, nsl * 4, hipMemcpyDeviceToHost));
                double worst = 0; for (size_t i = 0; i < nsl; ++i) worst = std::max(worst, (double)fabsf(got[i] - ref[i]));
                printf("  %-44s max |diff| vs register-staged %.3g\n", name, worst);
            };
            CK(hipMemset(slab, 0, nsl * 4)); run_d2(); check("dma 2 stages");
            CK(hipMemset(slab, 0, nsl * 4)); run_d3(); check("dma 3 stages");
            CK(hipMemset(slab, 0, nsl * 4)); run_d2t(); check("dma 2 stages, tail form");
            rep("reg", time_us(run_reg)); rep("dma 2 stages, 2 / CU", time_us(run_d2)); rep("dma 3 stages, 1 / CU", time_us(run_d3)); rep("dma 2 stages, tail form", time_us(run_d2t));
            auto none = [] {};
            rep("cold: reg", time_cold_us(run_reg, none)); rep("cold: dma 2 stages", time_cold_us(run_d2, none)); rep("cold: dma 3 stages", time_cold_us(run_d3, none));
            CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(slab)); CK(hipFree(C));
        }
    }
    const int M = 8192;
    const int shapes[][2] = {{1024, 1024}, {512, 1024}};
    orx_ctx ctx; ctx.num_cu = 256;
    for (auto& sh : shapes) {
        const int N = sh[0], K = sh[1];
        std::vector<_Float16> hA((size_t)M * K), hB((size_t)N * K);
        unsigned s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 9) & 0xffff) / 65536.0f - 0.5f; };
        for (auto& v : hA) v = (_Float16)rnd();
        for (auto& v : hB) v = (_Float16)(rnd() * 0.1f);
        _Float16 *A, *B, *C16; float* C; float* bias;
        CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&B, hB.size() * 2)); CK(hipMalloc(&C16, (size_t)M * N * 2)); CK(hipMalloc(&C, (size_t)M * N * 4));
        CK(hipMalloc(&bias, N * 4)); CK(hipMemset(bias, 0, N * 4));
        CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        Nt16Args g{A, K, B, K, nullptr, N, C16, N, bias, M, N, K, 1, nullptr, nullptr, 0, 0, nullptr};
        const double gf = 2.0 * M * N * K * 1e-9;
        printf("== M %d N %d K %d  (%.1f GFLOP), fp16-only epilogue (bias + relu)\n", M, N, K, gf);
        auto rep = [&](const char* name, float us) { printf("  %-44s %7.2f us  %6.0f TFLOP/s\n", name, us, gf / us * 1e3); };
        // correctness of the DMA kernels against the register-staged kernel (same operands, same k order per MFMA chain)
        launch_nt<4, 2, 4, 4, 1, 16>(&ctx, g); CK(hipDeviceSynchronize());
        std::vector<_Float16> ref((size_t)M * N), got((size_t)M * N);
        CK(hipMemcpy(ref.data(), C16, ref.size() * 2, hipMemcpyDeviceToHost));
        auto check = [&](const char* name) {
            CK(hipDeviceSynchronize()); CK(hipMemcpy(got.data(), C16, got.size() * 2, hipMemcpyDeviceToHost));
            double worst = 0; for (size_t i = 0; i < got.size(); ++i) worst = std::max(worst, (double)fabsf((float)got[i] - (float)ref[i]));
            printf("  %-44s max |diff| vs register-staged %.3g\n", name, worst);
        };
        CK(hipMemset(C16, 0, (size_t)M * N * 2)); launch_nt_dma<4, 2, 4, 4, 1, 3>(&ctx, g); check("dma 256x128 3 stages");
        CK(hipMemset(C16, 0, (size_t)M * N * 2)); launch_nt_dma<2, 2, 4, 4, 2, 2>(&ctx, g); check("dma 128x128 2 stages");
        rep("reg 256x128", time_us([&] { launch_nt<4, 2, 4, 4, 1, 16>(&ctx, g); }));
        rep("reg 128x128", time_us([&] { launch_nt<2, 2, 4, 4, 2, 8>(&ctx, g); }));
        rep("reg 128x64", time_us([&] { launch_nt<2, 2, 4, 2, 2, 16>(&ctx, g); }));
        rep("dma 256x128 3 stages", time_us([&] { launch_nt_dma<4, 2, 4, 4, 1, 3>(&ctx, g); }));
        rep("dma 256x128 2 stages", time_us([&] { launch_nt_dma<4, 2, 4, 4, 1, 2>(&ctx, g); }));
        rep("dma 128x128 2 stages, 2 / CU", time_us([&] { launch_nt_dma<2, 2, 4, 4, 2, 2>(&ctx, g); }));
        rep("dma 128x64 3 stages", time_us([&] { launch_nt_dma<2, 2, 4, 2, 2, 3>(&ctx, g); }));
        if (K == N) {
            // the operand of the timed launch is the output of the launch before it (as in the MLP), caches flushed before the pair
            _Float16* Y; CK(hipMalloc(&Y, (size_t)M * N * 2));
            Nt16Args g1 = g; g1.C16 = Y;
            Nt16Args g2 = g; g2.A = Y; g2.lda = N;
            auto none = [] {};
            rep("cold: reg 256x128", time_cold_us([&] { launch_nt<4, 2, 4, 4, 1, 16>(&ctx, g); }, none));
            rep("cold: dma 256x128 3 stages", time_cold_us([&] { launch_nt_dma<4, 2, 4, 4, 1, 3>(&ctx, g); }, none));
            rep("cold: dma 256x128 2 stages", time_cold_us([&] { launch_nt_dma<4, 2, 4, 4, 1, 2>(&ctx, g); }, none));
            rep("cold: dma 128x128 2 stages", time_cold_us([&] { launch_nt_dma<2, 2, 4, 4, 2, 2>(&ctx, g); }, none));
            rep("behind its producer: reg 256x128", time_cold_us([&] { launch_nt<4, 2, 4, 4, 1, 16>(&ctx, g2); }, [&] { launch_nt<4, 2, 4, 4, 1, 16>(&ctx, g1); }));
            rep("behind its producer: dma 256x128 3 stages", time_cold_us([&] { launch_nt_dma<4, 2, 4, 4, 1, 3>(&ctx, g2); }, [&] { launch_nt_dma<4, 2, 4, 4, 1, 3>(&ctx, g1); }));
            rep("behind its producer: dma 128x128 2 stages", time_cold_us([&] { launch_nt_dma<2, 2, 4, 4, 2, 2>(&ctx, g2); }, [&] { launch_nt_dma<2, 2, 4, 4, 2, 2>(&ctx, g1); }));
            rep("behind its producer: dma 3st, no epilogue", time_cold_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 1>(g2); }, [&] { launch_nt_dma<4, 2, 4, 4, 1, 3>(&ctx, g1); }));
            rep("behind its producer: dma 3st, no epi, no MFMA", time_cold_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 5>(g2); }, [&] { launch_nt_dma<4, 2, 4, 4, 1, 3>(&ctx, g1); }));
            rep("behind its producer: dma 3st, no epi, no DMA", time_cold_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 3>(g2); }, [&] { launch_nt_dma<4, 2, 4, 4, 1, 3>(&ctx, g1); }));
            CK(hipFree(Y));
        }
        rep("dma 256x128 3st, nontemporal stores", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 16>(g); }));
        rep("dma 128x64 3st, nontemporal stores", time_us([&] { launch_dbg<2, 2, 4, 2, 2, 3, 16>(g); }));
        rep("dma 256x128 3st, write-through stores", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 32>(g); }));
        rep("dma 128x64 3st, write-through stores", time_us([&] { launch_dbg<2, 2, 4, 2, 2, 3, 32>(g); }));
        { CK(hipMemset(C16, 0, (size_t)M * N * 2)); launch_dbg<4, 2, 4, 4, 1, 3, 32>(g); check("dma 256x128 3st write-through"); }
        rep("dma 256x128 3st, no epilogue", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 1>(g); }));
        rep("dma 256x128 3st, no epilogue, no DMA in loop", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 3>(g); }));
        rep("dma 256x128 3st, no epilogue, no MFMA", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 5>(g); }));
        rep("dma 256x128 3st, no epilogue, neither", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 7>(g); }));
        rep("dma 256x128 3st, epilogue only", time_us([&] { launch_dbg<4, 2, 4, 4, 1, 3, 6>(g); }));
        {   // the main loop alone, timed inside the kernel
            unsigned long long* dbg; CK(hipMalloc(&dbg, 256 * 16));
            Nt16Args gd = g; gd.C = reinterpret_cast<float*>(dbg);
            auto show = [&](const char* name) {
                CK(hipDeviceSynchronize());
                unsigned long long ho[512]; CK(hipMemcpy(ho, dbg, sizeof(ho), hipMemcpyDeviceToHost));
                double c = 0, w = 0; for (int i = 0; i < 256; ++i) { c += ho[2 * i]; w += ho[2 * i + 1]; }
                c /= 256; w /= 256;
                printf("  %-44s main loop %.0f cycles = %.2f us (%.2f GHz), %.0f cycles per K step\n", name, c, w / 100.0, c / (w * 10.0), c / ((K + 63) / 64));
            };
            for (int r = 0; r < 3; ++r) launch_dbg<4, 2, 4, 4, 1, 3, 9>(gd); show("dma 256x128 3st: loop");
            for (int r = 0; r < 3; ++r) launch_dbg<4, 2, 4, 4, 1, 3, 11>(gd); show("dma 256x128 3st: loop, no DMA");
            for (int r = 0; r < 3; ++r) launch_dbg<4, 2, 4, 4, 1, 3, 13>(gd); show("dma 256x128 3st: loop, no MFMA");
            CK(hipFree(dbg));
        }
        rep("dma 128x128 2st, no epilogue", time_us([&] { launch_dbg<2, 2, 4, 4, 2, 2, 1>(g); }));
        rep("dma 128x128 2st, no epilogue, no DMA in loop", time_us([&] { launch_dbg<2, 2, 4, 4, 2, 2, 3>(g); }));
        rep("dma 128x128 2st, no epilogue, no MFMA", time_us([&] { launch_dbg<2, 2, 4, 4, 2, 2, 5>(g); }));
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C16)); CK(hipFree(C)); CK(hipFree(bias));
    }
    return 0;
}

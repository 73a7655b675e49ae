// Diagnosis harness (not shipped): the GEMM kernel of ovo_amd/csrc/gemm.hip with one half of its main loop compiled out
//   -DOVO_GEMM_PROBE_NO_MMA  : DMA pipeline + barriers + epilogue only  (what the data movement alone costs)
//   -DOVO_GEMM_PROBE_NO_LOAD : fragment reads + MFMA on the prologue's stale tiles (what the math + LDS reads cost)
// build: hipcc --offload-arch=gfx950 -O3 -I ovo_amd/csrc -I include [-D...] tools/gemm_probe.hip ovo_amd/csrc/core.hip -o gpurun_out/probe_X
#include "../ovo_amd/csrc/gemm.hip"
#include <cstdio>
#include <vector>

int main(int argc, char **argv) {
    const int shapes[][3] = {{1154, 3072, 1024}, {1154, 1024, 1024}, {1154, 4096, 1024}, {1154, 1024, 4096}, {4096, 4096, 4096},
                             {65536, 336, 128}, {65536, 448, 128}, {65536, 112, 448}, {16384, 672, 224}, {4096, 1792, 448}};
    for (auto &sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        void *A, *W, *C;
        hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2);
        hipMemset(A, 0, (size_t)M * K * 2); hipMemset(W, 0, (size_t)N * K * 2);
        ovo_gemm_t g = {A, K, W, K, nullptr, C, N, nullptr, 0, M, N, K, 2, 2, 0, 1.0f};
        for (int i = 0; i < 3; ++i) ovo_gemm(&g, nullptr);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipDeviceSynchronize(); hipEventRecord(e0, nullptr);
        const int iters = 50;
        for (int i = 0; i < iters; ++i) ovo_gemm(&g, nullptr);
        hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("(%d, %d, %d)  %8.1f us  %6.0f TF\n", M, N, K, 1e3 * ms / iters, 2.0 * M * N * K / (1e3 * ms / iters) / 1e6);
        hipFree(A); hipFree(W); hipFree(C);
    }
    return 0;
}

// Diagnosis harness (not shipped): per-phase cycle stamps of one workgroup of the fused attention kernel.
// build: hipcc --offload-arch=gfx950 -O3 -DOVO_ATTN_TRACE -I ovo_amd/csrc -I include tools/attn_probe.hip ovo_amd/csrc/core.hip -o tools/bin/attn_probe
#include "../ovo_amd/csrc/attention.hip"
#include <cstdio>
#include <vector>
int main() {
    const int B = 2, H = 16, T = 577, hd = 64, D = H * hd;
    void *qkv, *out;
    hipMalloc(&qkv, (size_t)B * T * 3 * D * 2); hipMalloc(&out, (size_t)B * T * D * 2);
    hipMemset(qkv, 0x11, (size_t)B * T * 3 * D * 2);
    ovo_attention_t a = {};
    a.q = qkv; a.k = (char *)qkv + D * 2; a.v = (char *)qkv + 4 * D; a.o = out;
    a.q_sb = a.k_sb = a.v_sb = (int64_t)T * 3 * D; a.q_sh = a.k_sh = a.v_sh = hd; a.q_st = a.k_st = a.v_st = 3 * D;
    a.o_sb = (int64_t)T * D; a.o_sh = hd; a.o_st = D; a.B = B; a.H = H; a.Tq = T; a.Tk = T; a.hd = hd; a.scale = 0.125f;
    for (int i = 0; i < 5; ++i) ovo_attention(&a, nullptr);
    hipDeviceSynchronize();
    unsigned long long tr[256];
    hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_attn_trace), sizeof(tr));
    printf("fetch issue %llu\n", tr[1] - tr[0]);
    for (int t = 0; t < 10; ++t) {
        const int b = 2 + 4 * t;
        printf("tile %2d: barrier(+prev) %6llu  commit+barrier %6llu  fetch issue %6llu  compute %6llu\n", t,
               tr[b] - (t ? tr[b - 1] : tr[1]), tr[b + 1] - tr[b], tr[b + 2] - tr[b + 1], tr[b + 3] - tr[b + 2]);
    }
    for (int t = 2; t < 6; ++t) printf("tile %d compute split: S=K.Q %llu  softmax %llu  pack %llu  P.V %llu\n", t, tr[100 + 4 * t] - tr[4 + 4 * t], tr[101 + 4 * t] - tr[100 + 4 * t], tr[102 + 4 * t] - tr[101 + 4 * t], tr[5 + 4 * t] - tr[102 + 4 * t]);
    printf("total loop %llu cycles (s_memtime ticks)\n", tr[250] - tr[0]);
    return 0;
}

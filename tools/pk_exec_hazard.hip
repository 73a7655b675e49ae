// pk_exec_hazard.hip -- a packed-f32 VALU instruction (v_pk_add_f32: what -O3's SLP vectoriser makes of two adjacent f32 adds) directly followed by a write
// of EXEC (s_and_saveexec_b64: the head of a divergent `if`).  Found while looking for the cause of k_mlp_stream's non-deterministic two-workgroup variants
// (VERDICT r5 item 2): the LayerNorm statistics of a row block came out wrong in lanes 48 .. 63 only -- the lanes of the LAST 16-lane pass of the packed
// instruction that preceded `s_and_saveexec_b64 sN, <mask of lanes 0 .. 31>` -- one time in ~10^5, and only with a wave of ANOTHER workgroup busy on the same
// SIMD (MFMA chunk loop beside the LayerNorm code); with -fno-slp-vectorize (no v_pk_* in the LayerNorm code) it never happened.
//
// Here: `checker` waves run  v_pk_add_f32 acc, acc, one  ;  s_and_saveexec_b64 save, lowmask  ;  s_or_b64 exec, exec, save   in a long unrolled chain and
// compare the accumulators with the count; `partner` waves (other workgroups on the same CUs) run MFMAs or nothing.  Variants: the plain pair of
// v_add_f32 instead of the packed add, an s_nop between the packed add and the EXEC write.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_exec_hazard tools/pk_exec_hazard.hip && /tmp/pk_exec_hazard
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// MODE 0: v_pk_add_f32 then EXEC write; 1: two v_add_f32 then EXEC write; 2: v_pk_add_f32, s_nop 1, EXEC write; 3: v_pk_add_f32 then an EXEC write that
// changes nothing (mask = all lanes)
template <int MODE>
__global__ void __launch_bounds__(256, 2) k_hazard(uint32_t *err, int iters, int partner_every, int partner_mfma) {
    const int lane = threadIdx.x & 63;
    const bool partner = partner_every > 0 && (blockIdx.x % partner_every) != 0;
    if (partner) {                                                   // keep the matrix pipe (or just the issue port) of the shared SIMDs busy
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)1.0f; }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float v = (float)lane;
        for (int it = 0; it < iters * 4; ++it) {
            if (partner_mfma) {
#pragma unroll
                for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) v = v * 1.0000001f + 0.5f;
            }
        }
        if (acc[0] == 12345.678f || v == 12345.678f) err[3] = 1;
        return;
    }
    f32x2 acc = {0.f, 0.f};
    const f32x2 one = {1.0f, 1.0f};
    const unsigned long long lowmask = MODE == 3 ? ~0ull : 0x00000000ffffffffull;
    unsigned long long save;
    uint32_t bad = 0;
    for (int it = 0; it < iters; ++it) {
        acc = f32x2{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (MODE == 1) {
                float a0 = acc[0], a1 = acc[1];
                asm volatile("v_add_f32 %0, %0, %3\n\tv_add_f32 %1, %1, %3\n\ts_and_saveexec_b64 %2, %4\n\ts_or_b64 exec, exec, %2"
                             : "+v"(a0), "+v"(a1), "=&s"(save) : "v"(1.0f), "s"(lowmask) : "memory");
                acc[0] = a0; acc[1] = a1;
            }
            else if (MODE == 2)
                asm volatile("v_pk_add_f32 %0, %0, %2\n\ts_nop 1\n\ts_and_saveexec_b64 %1, %3\n\ts_or_b64 exec, exec, %1" : "+v"(acc), "=&s"(save) : "v"(one), "s"(lowmask) : "memory");
            else
                asm volatile("v_pk_add_f32 %0, %0, %2\n\ts_and_saveexec_b64 %1, %3\n\ts_or_b64 exec, exec, %1" : "+v"(acc), "=&s"(save) : "v"(one), "s"(lowmask) : "memory");
        }
        if (acc[0] != 32.0f || acc[1] != 32.0f) {
            ++bad;
            const uint32_t at = atomicAdd(err + 4, 1u);
            if (at < 16) { err[8 + at * 4] = (uint32_t)lane; err[9 + at * 4] = __float_as_uint(acc[0]); err[10 + at * 4] = __float_as_uint(acc[1]); err[11 + at * 4] = blockIdx.x; }
        }
    }
    if (bad) { atomicAdd(err, bad); atomicAdd(err + 1 + (lane >= 48 ? 1 : 0), bad); }
}

template <int MODE>
static void run(uint32_t *err, const char *tag, int partner_every, int partner_mfma, int iters, int launches) {
    CK(hipMemset(err, 0, 4096));
    for (int l = 0; l < launches; ++l) k_hazard<MODE><<<2048, 256>>>(err, iters, partner_every, partner_mfma);
    CK(hipDeviceSynchronize());
    uint32_t h[128];
    CK(hipMemcpy(h, err, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-64s partner %-22s: %8u wrong accumulators (lanes 0-47: %u, lanes 48-63: %u)\n", tag,
           partner_every == 0 ? "none" : (partner_mfma ? "MFMA, other workgroups" : "VALU, other workgroups"), h[0], h[1], h[2]);
    for (uint32_t i = 0; i < (h[4] < 6 ? h[4] : 6); ++i)
        printf("      workgroup %u lane %u: %g %g (expected 32 32)\n", h[11 + i * 4], h[8 + i * 4], *(float *)&h[9 + i * 4], *(float *)&h[10 + i * 4]);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000, launches = argc > 2 ? atoi(argv[2]) : 5;
    uint32_t *err;
    CK(hipMalloc(&err, 4096));
    for (int pm = 0; pm < 3; ++pm) {
        const int every = pm == 0 ? 0 : 2, mfma = pm == 2;
        run<0>(err, "v_pk_add_f32 ; s_and_saveexec_b64 (lanes 0-31)", every, mfma, iters, launches);
        run<1>(err, "v_add_f32 x 2 ; s_and_saveexec_b64 (lanes 0-31)", every, mfma, iters, launches);
        run<2>(err, "v_pk_add_f32 ; s_nop 1 ; s_and_saveexec_b64 (lanes 0-31)", every, mfma, iters, launches);
        run<3>(err, "v_pk_add_f32 ; s_and_saveexec_b64 (all lanes: EXEC unchanged)", every, mfma, iters, launches);
    }
    return 0;
}

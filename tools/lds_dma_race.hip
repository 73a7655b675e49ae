// lds_dma_race.hip -- the smallest form of the LDS-DMA double-buffer protocol of mlp_stream.hip / gemm8p.hip / winattn.hip / gemm_stream.hip, with
// nothing else in the kernel, to find out WHY k_mlp_stream's two-workgroups-per-CU variants were not deterministic:
//
//     for every chunk c:   s_waitcnt vmcnt(0);  s_barrier;  DMA(chunk c + 1 -> buffer (c + 1) & 1);  read buffer c & 1 (every wave reads EVERY piece)
//
// The source holds, in every 16-byte piece, {chunk << 16 | piece, ~that, chunk, piece}: a reader can tell a stale piece (the chunk that was in the
// buffer before: c - 2), an early overwrite (c + 2) and garbage apart.  Knobs (argv): threads per workgroup, workgroups per CU (extra dynamic LDS pads
// a workgroup so that only one fits), cycles of delay between the barrier and the first read, read-back of the wave's own pieces before the barrier,
// the wave-uniformity of the LDS destination (readfirstlane or not), an MFMA / ds_read filler that keeps the other workgroup's LDS port busy.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_dma_race tools/lds_dma_race.hip && /tmp/lds_dma_race
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

struct Args {
    const uint32_t *src;      // [chunks][PIECES][4]
    uint32_t *err;            // [0] mismatching pieces, [1] stale (chunk - 2), [2] early (chunk + 2), [3] other, [4..] first records {chunk, piece, seen0, seen2}
    int chunks, groups, delay, readback, filler, mode, perm_rounds;
};

constexpr int PIECES = 1920;                   // 16-byte pieces per chunk = 30 KB, as k_mlp_stream<128, 112, 448>
constexpr int BUF = PIECES * 16;

template <int NTHREADS>
__global__ void __launch_bounds__(NTHREADS, 512 / NTHREADS) k_race(Args a, int n_slots) {     // (<= 256 VGPRs: two 256-thread workgroups fit a CU)
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PPT = (PIECES + NTHREADS - 1) / NTHREADS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.src, 0, a.chunks * BUF, 0x00020000);
    auto dma = [&](int chunk, auto BUF_) {
        char *base = smem + decltype(BUF_)::value * BUF;
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            const int id0 = p * NTHREADS + wave * 64;
            if (id0 >= PIECES) continue;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(base + id0 * 16), 16, (p * NTHREADS + tid) * 16, chunk * BUF, 0, 0);
        }
        if (a.mode & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    uint32_t bad = 0, stale = 0, early = 0, bad_perm = 0;
    // mode & 4 / 8: a ds_bpermute_b32 self-check (what __shfl_xor compiles to: the LDS crossbar, no LDS memory) while LDS-DMA writes are landing on this CU --
    // 4: right after this wave's own DMA issue, 8: between the barrier and the DMA issue (only OTHER waves' / workgroups' DMA can be in flight)
    auto perm_check = [&](int rounds, unsigned salt) {
        for (int it = 0; it < rounds; ++it) {
            const unsigned v = (unsigned)lane * 2654435761u + salt + (unsigned)it * 40503u;
            unsigned r = v;
            r += (unsigned)__shfl_xor((int)r, 16, 64);
            r += (unsigned)__shfl_xor((int)r, 32, 64);
            const unsigned l0 = lane & 15;
            unsigned want = 0;
            for (int k = 0; k < 4; ++k) want += (l0 + 16u * k) * 2654435761u + salt + (unsigned)it * 40503u;
            if (r != want) ++bad_perm;
        }
    };
    const unsigned lds_alloc = __builtin_amdgcn_s_getreg((31 << 11) | 6);           // HW_REG_LDS_ALLOC: base [7:0], size [20:12] (granules)
    if (tid == 0 && (lds_alloc & 0xff)) atomicAdd(a.err + 5, 1u);
    if (tid == 0 && blockIdx.x == n_slots - 1) a.err[6] = lds_alloc;
    for (int grp = blockIdx.x; grp < a.groups; grp += n_slots) {
        __syncthreads();
        dma(0, std::integral_constant<int, 0>{});
        auto body = [&](auto PAR_, int c) {
            constexpr int PAR = decltype(PAR_)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (a.readback) {                                 // the issuing wave reads its own pieces back before it enters the barrier
                u32x4 v = *(const u32x4 *)(smem + PAR * BUF + ((PPT - 1) * NTHREADS + tid < PIECES ? (PPT - 1) * NTHREADS + tid : tid) * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(v) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            if (a.mode & 8) perm_check(a.perm_rounds, (unsigned)c * 977u + blockIdx.x);
            if (c + 1 < a.chunks) dma(c + 1, std::integral_constant<int, PAR ^ 1>{});
            __builtin_amdgcn_sched_barrier(0);
            if (a.mode & 4) perm_check(a.perm_rounds, (unsigned)c * 977u + blockIdx.x);
            if (a.delay > 0) {
                const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)a.delay) {}
            }
            const char *cur = smem + PAR * BUF;
            // every wave reads every piece, the pieces the OTHER waves brought in first (the last pieces issued are read first)
#pragma unroll 2
            for (int k = PIECES / 64 - 1; k >= 0; --k) {
                const int piece = ((k + wave * 7) % (PIECES / 64)) * 64 + lane;
                const u32x4 v = *(const u32x4 *)(cur + piece * 16);
                const uint32_t want = ((uint32_t)c << 16) | (uint32_t)piece;
                if (v[0] != want || v[1] != ~want || v[2] != (uint32_t)c || v[3] != (uint32_t)piece) {
                    ++bad;
                    if (v[2] + 2 == (uint32_t)c && v[3] == (uint32_t)piece) ++stale;
                    else if (v[2] == (uint32_t)c + 2 && v[3] == (uint32_t)piece) ++early;
                    const uint32_t at = atomicAdd(a.err + 4, 1u);
                    if (at < 64) { uint32_t *r = a.err + 8 + at * 4; r[0] = (uint32_t)c; r[1] = (uint32_t)piece | ((uint32_t)wave << 24); r[2] = v[0]; r[3] = v[2]; }
                }
            }
            if (a.filler) {                                    // keep the LDS port and the matrix pipe busy, as the products of a real chunk do
                typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
                typedef __attribute__((ext_vector_type(4))) float f32x4;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int it = 0; it < a.filler; ++it) {
                    for (int k = 0; k < PIECES / 64; k += 2) {
                        const bf16x8 x = *(const bf16x8 *)(cur + (k * 64 + lane) * 16), y = *(const bf16x8 *)(cur + ((k + 1) * 64 + lane) * 16);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc, 0, 0, 0);
                    }
                }
                if (acc[0] == 12345.678f) a.err[6] = 1;
            }
        };
        int c = 0;
        for (; c + 1 < a.chunks; c += 2) {
            body(std::integral_constant<int, 0>{}, c);
            body(std::integral_constant<int, 1>{}, c + 1);
        }
        if (c < a.chunks) body(std::integral_constant<int, 0>{}, c);
    }
    if (bad_perm) atomicAdd(a.err + 7, bad_perm);
    if (bad) { atomicAdd(a.err + 0, bad); atomicAdd(a.err + 1, stale); atomicAdd(a.err + 2, early); atomicAdd(a.err + 3, bad - stale - early); }
#endif
}

// A kernel with NO LDS-DMA and no LDS memory at all: ds_bpermute_b32 self-checks in a loop (a LayerNorm's wave reduction).  Launched on a second stream
// beside k_race: its waves share CUs with the DMA kernel's workgroups when registers allow.
__global__ void __launch_bounds__(256) k_perm_only(uint32_t *err, int rounds) {
    const int lane = threadIdx.x & 63;
    uint32_t bad = 0;
    for (int it = 0; it < rounds; ++it) {
        const unsigned salt = blockIdx.x * 7919u + (unsigned)it * 40503u;
        const unsigned v = (unsigned)lane * 2654435761u + salt;
        unsigned r = v;
        r += (unsigned)__shfl_xor((int)r, 16, 64);
        r += (unsigned)__shfl_xor((int)r, 32, 64);
        unsigned want = 0;
        for (int k = 0; k < 4; ++k) want += ((lane & 15) + 16u * k) * 2654435761u + salt;
        if (r != want) ++bad;
    }
    if (bad) atomicAdd(err, bad);
    if (threadIdx.x == 0) atomicAdd(err + 1, 1u);
}

template <int NTHREADS>
static void run(const Args &a0, int per_cu, int launches, const char *tag) {
    Args a = a0;
    // LDS per workgroup: two buffers, padded so that exactly `per_cu` workgroups fit a CU's 160 KB
    size_t lds = 2 * (size_t)BUF;
    if (per_cu == 1) lds = 100 * 1024;
    CK(hipFuncSetAttribute((const void *)k_race<NTHREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_race<NTHREADS>, NTHREADS, lds));
    const int slots = 256 * per_cu;
    CK(hipMemset(a.err, 0, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int l = 0; l < launches; ++l) k_race<NTHREADS><<<slots, NTHREADS, lds>>>(a, slots);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    uint32_t h[1024];
    CK(hipMemcpy(h, a.err, 4096, hipMemcpyDeviceToHost));
    printf("%-46s threads %3d per_cu %d (occupancy %d) delay %4d readback %d filler %d mode %2d : %8u WRONG ds_bpermute results; %8u bad pieces (stale %u, early %u, other %u) in %d launches, %.1f us each; workgroups with a non-zero LDS base %u (LDS_ALLOC of the last: %08x)\n",
           tag, NTHREADS, per_cu, occ, a.delay, a.readback, a.filler, a.mode, h[7], h[0], h[1], h[2], h[3], launches, 1e3 * ms / launches, h[5], h[6]);
    const uint32_t n = h[4] < 8 ? h[4] : 8;
    for (uint32_t i = 0; i < n; ++i)
        printf("      chunk %u piece %u (wave %u): word0 %08x chunk word %u\n", h[8 + i * 4], h[8 + i * 4 + 1] & 0xffffff, h[8 + i * 4 + 1] >> 24, h[8 + i * 4 + 2], h[8 + i * 4 + 3]);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int chunks = 14, groups = argc > 1 ? atoi(argv[1]) : 4096, launches = argc > 2 ? atoi(argv[2]) : 50;
    std::vector<uint32_t> src((size_t)chunks * PIECES * 4);
    for (int c = 0; c < chunks; ++c)
        for (int p = 0; p < PIECES; ++p) {
            uint32_t *w = src.data() + ((size_t)c * PIECES + p) * 4;
            w[0] = ((uint32_t)c << 16) | (uint32_t)p; w[1] = ~w[0]; w[2] = (uint32_t)c; w[3] = (uint32_t)p;
        }
    Args a;
    memset(&a, 0, sizeof(a));
    uint32_t *dsrc;
    CK(hipMalloc(&dsrc, src.size() * 4));
    CK(hipMemcpy(dsrc, src.data(), src.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&a.err, 4096));
    a.src = dsrc; a.chunks = chunks; a.groups = groups;
    struct Cfg { int threads, per_cu, delay, readback, filler, mode; const char *tag; };
    const Cfg cfgs[] = {
        {512, 1, 0, 0, 0, 0, "one 512-thread workgroup per CU (production)"},
        {512, 1, 0, 0, 4, 0, "  + filler"},
        {256, 1, 0, 0, 0, 0, "one 256-thread workgroup per CU"},
        {256, 1, 0, 0, 4, 0, "  + filler"},
        {256, 2, 0, 0, 0, 0, "TWO 256-thread workgroups per CU"},
        {256, 2, 0, 0, 4, 0, "  + filler"},
        {256, 2, 0, 0, 16, 0, "  + long filler"},
        {256, 2, 500, 0, 4, 0, "  + 500 cycles between barrier and first read"},
        {256, 2, 2000, 0, 4, 0, "  + 2000 cycles"},
        {256, 2, 0, 1, 4, 0, "  + read-back of own pieces before the barrier"},
        {256, 2, 0, 0, 4, 2, "  + wait for every DMA at its issue"},
        {512, 1, 0, 0, 4, 4, "ONE workgroup/CU, bpermute checks after own DMA issue"},
        {512, 1, 0, 0, 4, 8, "ONE workgroup/CU, bpermute checks before own DMA issue"},
        {256, 1, 0, 0, 4, 4, "ONE 256-thread workgroup/CU, checks after own DMA issue"},
        {256, 2, 0, 0, 4, 4, "TWO workgroups/CU, bpermute checks after own DMA issue"},
        {256, 2, 0, 0, 4, 8, "TWO workgroups/CU, checks before own DMA issue (others' DMA only)"},
        {256, 2, 0, 0, 4, 10, "TWO workgroups/CU, no DMA in flight under the checks of its own workgroup"},
    };
    a.perm_rounds = 64;
    for (const Cfg &c : cfgs) {
        a.delay = c.delay; a.readback = c.readback; a.filler = c.filler; a.mode = c.mode;
        if (c.threads == 512) run<512>(a, c.per_cu, launches, c.tag);
        else run<256>(a, c.per_cu, launches, c.tag);
    }
    // ---- two different kernels on two streams: the DMA kernel (one 512-thread workgroup per CU, 60 KB of LDS, <= 256 VGPRs) and the checker (no LDS)
    {
        hipStream_t s1, s2;
        CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
        uint32_t *perr;
        CK(hipMalloc(&perr, 64));
        for (int with_dma = 1; with_dma >= 0; --with_dma) {
            CK(hipMemset(perr, 0, 64)); CK(hipMemset(a.err, 0, 4096));
            a.delay = 0; a.readback = 0; a.filler = 4; a.mode = 0;
            const size_t lds = 2 * (size_t)BUF;
            CK(hipFuncSetAttribute((const void *)k_race<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CK(hipDeviceSynchronize());
            for (int l = 0; l < launches; ++l) {
                if (with_dma) k_race<256><<<256, 256, lds, s1>>>(a, 256);
                k_perm_only<<<4096, 256, 0, s2>>>(perr, 20000);
            }
            CK(hipDeviceSynchronize());
            uint32_t h[16];
            CK(hipMemcpy(h, perr, 64, hipMemcpyDeviceToHost));
            printf("two kernels on two streams: bpermute-only kernel %s the LDS-DMA kernel: %u WRONG ds_bpermute results in %u workgroups x 4 waves x 20000 checks\n",
                   with_dma ? "BESIDE" : "WITHOUT", h[0], h[1]);
        }
    }
    return 0;
}

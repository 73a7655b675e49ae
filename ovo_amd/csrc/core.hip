// core.hip -- error reporting, ABI version and the optional per-kernel-family event profiler of libovo_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <string.h>

#include <atomic>
#include <chrono>
#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

void ovo_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- profiler: hipEvent pairs around the launches of a kernel family, on the launch stream --------------
namespace {
struct Rec { hipEvent_t a, b; int kind; double work, bytes; int shape[3], flags; int cslot; double per_item, fixed; };
struct Prof {
    bool on = false;
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    int32_t *counts = nullptr;                                   // pinned: device-side item counts of launches whose work the host does not know
    int n_counts = 0;
    hipEvent_t get() {
        if (used == pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; pool.push_back(e); }
        return pool[used++];
    }
} g_prof;
}  // namespace

bool ovo_prof_enabled() { return g_prof.on; }
void ovo_prof_begin(int kind, double work, hipStream_t s) {
    if (!g_prof.on || g_prof.recs.size() >= (1u << 20)) return;
    Rec r; r.kind = kind; r.work = work; r.bytes = 0; r.shape[0] = r.shape[1] = r.shape[2] = 0; r.flags = 0; r.cslot = -1; r.per_item = r.fixed = 0; r.a = g_prof.get(); r.b = g_prof.get();
    if (!r.a || !r.b) return;
    (void)hipEventRecord(r.a, s);
    g_prof.recs.push_back(r);
}
void ovo_prof_shape(int a, int b, int c) {
    if (!g_prof.on || g_prof.recs.empty()) return;
    Rec &r = g_prof.recs.back();
    r.shape[0] = a; r.shape[1] = b; r.shape[2] = c;
}
void ovo_prof_flags(int flags) {
    if (!g_prof.on || g_prof.recs.empty()) return;
    g_prof.recs.back().flags = flags;
}
void ovo_prof_bytes(double bytes) {
    if (!g_prof.on || g_prof.recs.empty()) return;
    g_prof.recs.back().bytes = bytes;
}
static double g_last_bytes[OVO_PROF_KINDS];
void ovo_prof_end(hipStream_t s) {
    if (!g_prof.on || g_prof.recs.empty()) return;
    (void)hipEventRecord(g_prof.recs.back().b, s);
}
void ovo_prof_count(const int32_t *device_count, double per_item, double fixed, hipStream_t s) {
    constexpr int SLOTS = 1 << 16;
    if (!g_prof.on || g_prof.recs.empty() || !device_count) return;
    if (!g_prof.counts && hipHostMalloc((void **)&g_prof.counts, SLOTS * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) { g_prof.counts = nullptr; return; }
    if (g_prof.n_counts >= SLOTS) return;
    Rec &r = g_prof.recs.back();
    r.cslot = g_prof.n_counts++; r.per_item = per_item; r.fixed = fixed;
    (void)hipMemcpyAsync(g_prof.counts + r.cslot, device_count, sizeof(int32_t), hipMemcpyDeviceToHost, s);      // behind the end event: not in the timed pair
}

template <typename T>
static int host_wait(const T *flag, T value, int64_t timeout_us) {
    OVO_REQUIRE(flag, "null flag");
    const volatile T *f = flag;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        if (*f == value) { std::atomic_thread_fence(std::memory_order_acquire); return OVO_OK; }
        if ((spin & 255) == 255 &&
            std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_us) {
            ovo_set_error("ovo_host_wait: timed out after %lld us (flag %lld, expected %lld)", (long long)timeout_us, (long long)*f, (long long)value);
            return OVO_E_LAUNCH;
        }
        __builtin_ia32_pause();
    }
}

__global__ void k_marker(int id, int *sink) { if (sink && id < 0) *sink = id; }

extern "C" {
int ovo_marker(int id, ovo_stream_t stream) {
    k_marker<<<1, 1, 0, (hipStream_t)stream>>>(id, nullptr);
    OVO_CHECK_LAUNCH();
    return OVO_OK;
}
const char *ovo_hip_last_error(void) { return g_err; }
int ovo_hip_abi_version(void) { return 12; }

int ovo_profile_start(void) {
    g_prof.recs.clear();
    g_prof.used = 0;
    g_prof.n_counts = 0;
    g_prof.on = true;
    return OVO_OK;
}

int ovo_profile_stop(double *ms, double *work, int64_t *launches, int n_kinds) {
    g_prof.on = false;
    OVO_REQUIRE(ms && work && launches && n_kinds > 0 && n_kinds <= OVO_PROF_KINDS, "bad argument");
    for (int i = 0; i < n_kinds; ++i) { ms[i] = 0; work[i] = 0; launches[i] = 0; }
    for (int i = 0; i < OVO_PROF_KINDS; ++i) g_last_bytes[i] = 0;
    OVO_HIP(hipDeviceSynchronize());
    FILE *dump = getenv("OVO_PROF_DUMP") ? fopen(getenv("OVO_PROF_DUMP"), "a") : nullptr;   // diagnosis: one line per launch
    for (const Rec &r : g_prof.recs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess || r.kind >= n_kinds) continue;
        double wk = r.work, by = r.bytes;
        if (r.cslot >= 0 && g_prof.counts) wk = by = r.per_item * (double)g_prof.counts[r.cslot] + r.fixed;
        ms[r.kind] += t; work[r.kind] += wk; launches[r.kind] += 1; g_last_bytes[r.kind] += by;
        if (dump) fprintf(dump, "%d %d %d %d %.0f %.4f %d\n", r.kind, r.shape[0], r.shape[1], r.shape[2], wk, t, r.flags);
    }
    if (dump) fclose(dump);
    g_prof.recs.clear();
    g_prof.used = 0;
    return OVO_OK;
}

int ovo_profile_bytes(double *bytes, int n_kinds) {
    OVO_REQUIRE(bytes && n_kinds > 0 && n_kinds <= OVO_PROF_KINDS, "bad argument");
    for (int i = 0; i < n_kinds; ++i) bytes[i] = g_last_bytes[i];
    return OVO_OK;
}

// ---- pinned result blocks of the keyframe chain -------------------------------------------------------------------------
void *ovo_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        ovo_set_error("ovo_host_alloc: hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    memset(p, 0, bytes);
    return p;
}
void ovo_host_free(void *p) { if (p) (void)hipHostFree(p); }

int ovo_host_wait32(const int32_t *flag, int32_t value, int64_t timeout_us) { return host_wait(flag, value, timeout_us); }
int ovo_host_wait64(const int64_t *flag, int64_t value, int64_t timeout_us) { return host_wait(flag, value, timeout_us); }
}

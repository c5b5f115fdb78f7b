// Library-wide pieces of the C ABI: version, per-thread error string, optional HIP-event kernel timing.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <atomic>
#include <mutex>
#include <utility>
#include <vector>
#include "../../include/unimatch_hip.h"
#include "timing.h"

static thread_local char g_err[512] = "";

void um_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int um_version(void) { return UM_VERSION; }

extern "C" const char* um_last_error_string(void) { return g_err; }

// ---- timing: hipEvents recorded on the launch stream right around a kernel, read back on demand ----------
namespace {
struct TimingState {
    std::mutex mu;
    unsigned mask = 0;                     // bit k: kernel id k is timed
    std::vector<std::pair<hipEvent_t, hipEvent_t>> live[UM_K_COUNT];
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
};
TimingState g_t;
}  // namespace

bool um_timing_on() { return g_t.mask != 0; }

void* um_timing_begin(int kid, hipStream_t stream) {
    if (kid < 0 || kid >= UM_K_COUNT || !((g_t.mask >> kid) & 1u)) return nullptr;
    std::lock_guard<std::mutex> lk(g_t.mu);
    hipEvent_t a = g_t.get(), b = g_t.get();
    if (!a || !b) return nullptr;
    (void)hipEventRecord(a, stream);
    g_t.live[kid].push_back({a, b});
    return (void*)b;
}

void um_timing_end(void* token, hipStream_t stream) {
    if (token) (void)hipEventRecord((hipEvent_t)token, stream);
}

extern "C" int um_timing_enable(int kernel_mask) {
    std::lock_guard<std::mutex> lk(g_t.mu);
    g_t.mask = (unsigned)kernel_mask;
    return 0;
}

extern "C" int um_timing_collect(int kernel_id, double* total_ms, int* launches) {
    if (kernel_id < 0 || kernel_id >= UM_K_COUNT || !total_ms || !launches) {
        um_set_error("um_timing_collect: bad kernel id %d or null output", kernel_id);
        return UM_ERR_BAD_ARG;
    }
    std::lock_guard<std::mutex> lk(g_t.mu);
    double sum = 0.0;
    int n = 0;
    for (auto& pr : g_t.live[kernel_id]) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(pr.second);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, pr.first, pr.second);
        if (e == hipSuccess) {
            sum += ms;
            ++n;
        }
        g_t.pool.push_back(pr.first);
        g_t.pool.push_back(pr.second);
    }
    g_t.live[kernel_id].clear();
    *total_ms = sum;
    *launches = n;
    return 0;
}

// ---- launch census: which instantiation served each call (tests assert the measured configuration ran the kernels the
// bench times: non-split attention / FFN, gsv4; see UM_V_* in the header).  Off by default; relaxed counters, no ordering.
namespace {
std::atomic<int> g_census_on{0};
std::atomic<long> g_census[UM_V_COUNT];
}  // namespace

void um_census_hit(int variant) {
    if (variant >= 0 && variant < UM_V_COUNT && g_census_on.load(std::memory_order_relaxed))
        g_census[variant].fetch_add(1, std::memory_order_relaxed);
}

extern "C" int um_census_enable(int on) {
    g_census_on.store(on ? 1 : 0, std::memory_order_relaxed);
    if (on)
        for (auto& c : g_census) c.store(0, std::memory_order_relaxed);
    return 0;
}

extern "C" long um_census_count(int variant) {
    if (variant < 0 || variant >= UM_V_COUNT) return -1;
    return g_census[variant].load(std::memory_order_relaxed);
}

// ---- operand-range flag word (include/unimatch_hip.h, um_range_flags): one unsigned in pinned, mapped host memory; kernels receive
// its device address in their argument structs and OR a role bit into it (system-scope atomic) in the rare overflow case.
namespace {
std::once_flag g_range_once;
unsigned* g_range_host = nullptr;
void range_alloc() {
    unsigned* p = nullptr;
    if (hipHostMalloc((void**)&p, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && p) {
        *p = 0u;
        g_range_host = p;
    }
    (void)hipGetLastError();
}
}  // namespace

// device address of the flag word for the CURRENT device (nullptr when pinned memory cannot be had: the kernels then skip the note)
unsigned* um_range_flag_dev() {
    std::call_once(g_range_once, range_alloc);
    if (!g_range_host) return nullptr;
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, g_range_host, 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return (unsigned*)d;
}

extern "C" int um_range_flags(unsigned* flags_out, int reset) {
    if (!flags_out) {
        um_set_error("um_range_flags: null output");
        return UM_ERR_BAD_ARG;
    }
    std::call_once(g_range_once, range_alloc);
    if (!g_range_host) {
        *flags_out = 0u;
        return 0;
    }
    // read-and-reset is ONE atomic exchange: a kernel's system-scope OR that lands between a separate load and store would be lost
    // (round-5 ADVICE; UniMatch.forward calls this with reset = 1 while the previous forward's kernels may still be running).
    // The word is process-wide (one per library instance, all devices and models): an overflow is reported by whichever forward
    // reads the flags next.
    *flags_out = reset ? __atomic_exchange_n(g_range_host, 0u, __ATOMIC_ACQ_REL) : __atomic_load_n(g_range_host, __ATOMIC_ACQUIRE);
    return 0;
}

// runtime.hip — error strings, device selection and per-kernel HIP-event timing of libmmiss.
#include "common.h"
#include <atomic>
#include <map>

// ------------------------------------------------------------------ errors
static thread_local char g_err[1024] = "";

void mmiss_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mmiss_last_error(void) { return g_err; }
extern "C" int mmiss_abi_version(void) { return MMISS_ABI_VERSION; }

extern "C" int mmiss_device_count(int* count) {
    if (!count) MM_FAIL(MMISS_ERR_ARG, "mmiss_device_count: null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return MMISS_OK;
}

bool mmiss_is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // clear sticky "invalid value" of a pageable host pointer
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged
#if defined(hipMemoryTypeArray)
           || a.type == hipMemoryTypeArray
#endif
        ;
}

int mmiss_use_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        MM_FAIL(MMISS_ERR_HIP, "no HIP device visible (%s); libmmiss has no CPU fallback",
                e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    }
    if (device < 0 || device >= n) MM_FAIL(MMISS_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
    MM_HIP(hipSetDevice(device));
    static std::mutex mu;
    static std::map<int, bool> checked;
    std::lock_guard<std::mutex> lk(mu);
    if (!checked.count(device)) {
        hipDeviceProp_t prop;
        MM_HIP(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            MM_FAIL(MMISS_ERR_UNSUPPORTED, "device %d is %s; libmmiss is built for gfx950 (MI355X) only", device,
                    prop.gcnArchName);
        checked[device] = true;
    }
    return MMISS_OK;
}

// ------------------------------------------------------------------ tuning options (tests / experiments)
static std::mutex g_opt_mu;
static std::map<std::string, int> g_opts;
int mmiss_ensure_dyn_lds(const void* kernel, int lds) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, int> have;
    int dev = 0;
    MM_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    int& cur = have[std::make_pair(kernel, dev)];
    if (lds > cur) {
        MM_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        cur = lds;
    }
    return MMISS_OK;
}

// Options exist for tests and A/B experiments; a production process never sets one. Until the first
// mmiss_dbg_set_option call every lookup is one relaxed atomic load (no mutex, no std::string): the launch paths ask for
// ~25 options per encode on a single-request path that is already dispatch-bound.
static std::atomic<bool> g_opts_any{false};
int mmiss_option(const char* key, int dflt) {
    if (!g_opts_any.load(std::memory_order_acquire)) return dflt;
    std::lock_guard<std::mutex> lk(g_opt_mu);
    auto it = g_opts.find(key);
    return it == g_opts.end() ? dflt : it->second;
}
extern "C" int mmiss_dbg_set_option(const char* key, int value) {
    if (!key) MM_FAIL(MMISS_ERR_ARG, "null option key");
    std::lock_guard<std::mutex> lk(g_opt_mu);
    g_opts[key] = value;
    g_opts_any.store(true, std::memory_order_release);
    return MMISS_OK;
}

// ------------------------------------------------------------------ kernel timing
namespace {
struct ProfRec {
    std::string name;
    hipEvent_t e0, e1;
    double flops, bytes;
};
struct ProfAcc {
    long launches = 0;
    double ms = 0, flops = 0, bytes = 0;
};
std::mutex g_prof_mu;
bool g_prof_on = false;
std::string g_prof_filter;   // non-empty: only this kernel class is bracketed ...
int g_prof_stride = 1;       // ... and only every stride-th launch of it
long g_prof_seen = 0;
std::vector<ProfRec> g_prof_pending;
std::vector<hipEvent_t> g_prof_pool;
std::map<std::string, ProfAcc> g_prof_acc;
std::vector<std::string> g_prof_order;

hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) {
        hipEvent_t e = g_prof_pool.back();
        g_prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void prof_drain_locked(bool wait) {
    size_t keep = 0;
    for (size_t i = 0; i < g_prof_pending.size(); ++i) {
        ProfRec& r = g_prof_pending[i];
        bool done = true;
        if (wait)
            (void)hipEventSynchronize(r.e1);
        else
            done = (hipEventQuery(r.e1) == hipSuccess);
        if (!done) {
            if (keep != i) g_prof_pending[keep] = r;
            ++keep;
            continue;
        }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            if (!g_prof_acc.count(r.name)) g_prof_order.push_back(r.name);
            ProfAcc& a = g_prof_acc[r.name];
            a.launches += 1;
            a.ms += ms;
            a.flops += r.flops;
            a.bytes += r.bytes;
        }
        g_prof_pool.push_back(r.e0);
        g_prof_pool.push_back(r.e1);
    }
    (void)hipGetLastError();
    g_prof_pending.resize(keep);
}
}  // namespace

ProfScope::ProfScope(const char* name, hipStream_t s, double flops, double bytes) : slot(-1), stream(s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_on || g_prof_pending.size() > 200000) return;
    if (!g_prof_filter.empty()) {
        if (g_prof_filter != name) return;
        if ((g_prof_seen++ % g_prof_stride) != 0) return;
    }
    ProfRec r;
    r.name = name;
    r.e0 = prof_event();
    r.e1 = prof_event();
    r.flops = flops;
    r.bytes = bytes;
    if (!r.e0 || !r.e1) return;
    (void)hipEventRecord(r.e0, s);
    g_prof_pending.push_back(r);
    slot = (int)g_prof_pending.size() - 1;
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    // slots stay valid until mmiss_prof_read / mmiss_prof_reset drain the list (never while scopes are open)
    if (slot < (int)g_prof_pending.size()) (void)hipEventRecord(g_prof_pending[slot].e1, stream);
}

extern "C" int mmiss_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return MMISS_OK;
}

extern "C" int mmiss_prof_filter(const char* kernel, int stride) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_filter = kernel ? kernel : "";
    g_prof_stride = stride > 0 ? stride : 1;
    g_prof_seen = 0;
    return MMISS_OK;
}

extern "C" int mmiss_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_drain_locked(true);
    g_prof_acc.clear();
    g_prof_order.clear();
    return MMISS_OK;
}

extern "C" int mmiss_prof_read(char* buf, size_t cap) {
    if (!buf || cap < 3) MM_FAIL(MMISS_ERR_ARG, "mmiss_prof_read: buffer too small");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_drain_locked(true);
    std::string s = "[";
    bool first = true;
    for (const std::string& name : g_prof_order) {
        const ProfAcc& a = g_prof_acc[name];
        char line[512];
        snprintf(line, sizeof(line), "%s{\"kernel\": \"%s\", \"launches\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
                 first ? "" : ", ", name.c_str(), a.launches, a.ms, a.flops, a.bytes);
        s += line;
        first = false;
    }
    s += "]";
    if (s.size() + 1 > cap) MM_FAIL(MMISS_ERR_ARG, "mmiss_prof_read: need %zu bytes, have %zu", s.size() + 1, cap);
    memcpy(buf, s.c_str(), s.size() + 1);
    return MMISS_OK;
}

// capi.hip -- error plumbing, version and the optional per-kernel event timing of libngp_hip.
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "ngp_common.hpp"

namespace ngp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return NGP_ELAUNCH;
    }
    return NGP_OK;
}

// ---- profiling: events around selected launches, only when enabled -------------
struct ProfEntry {
    std::string name;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    double total_ms = 0;
    uint64_t launches = 0;
    double units = 0;
};
static bool g_prof_on = false;
static std::mutex g_prof_mu;
static std::vector<ProfEntry> g_prof;

static int prof_slot(const char* name) {
    for (size_t i = 0; i < g_prof.size(); i++)
        if (g_prof[i].name == name) return (int)i;
    g_prof.emplace_back();
    g_prof.back().name = name;
    return (int)g_prof.size() - 1;
}

ProfScope::ProfScope(const char* name, hipStream_t s, double units) : slot(-1), stream(s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    slot = prof_slot(name);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, s);
    g_prof[slot].events.emplace_back(a, b);
    g_prof[slot].units += units;
}
ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_prof[slot].events.back().second, stream);
}

bool prof_enabled() { return g_prof_on; }
void prof_add_units(const char* name, double units) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof[prof_slot(name)].units += units;
}

static void prof_collect(ProfEntry& e) {
    for (auto& p : e.events) {
        (void)hipEventSynchronize(p.second);
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
            e.total_ms += ms;
            e.launches++;
        }
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    e.events.clear();
}

}  // namespace ngp

using namespace ngp;

extern "C" {

const char* ngp_last_error(void) { return g_err; }
int ngp_version(void) { return 100; }

int ngp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ngp_prof_enable(int on) {
    g_prof_on = on != 0;
    return NGP_OK;
}

int ngp_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& e : g_prof) {
        prof_collect(e);
        e.total_ms = 0;
        e.launches = 0;
        e.units = 0;
    }
    return NGP_OK;
}

int ngp_prof_read(const char* name, double* total_ms, uint64_t* launches, double* units) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& e : g_prof) {
        if (e.name == name) {
            prof_collect(e);
            if (total_ms) *total_ms = e.total_ms;
            if (launches) *launches = e.launches;
            if (units) *units = e.units;
            return NGP_OK;
        }
    }
    set_error("prof_read: no kernel named '%s' was recorded", name);
    return NGP_EINVAL;
}

}  // extern "C"

// Optional per-kernel timing inside the library: when enabled (pn2_prof_enable(1)) every kernel launch is
// bracketed by two HIP events recorded on the launch stream, together with the launch's ALGORITHMIC bytes and
// flops.  bench.py uses it to report the roofline of the dominant kernel from live measurements; it
// is off by default and then costs one predictable branch per launch.  The only mutable global state of the
// library lives here, behind a mutex.
#include <mutex>
#include <string>
#include <vector>

#include "pn2_common.h"

namespace pn2 {
namespace prof {

bool g_enabled = false;

namespace {
struct Rec {
    const char* name;
    double bytes, flops;
    hipEvent_t e0, e1;
};
std::mutex g_mu;
std::vector<Rec> g_recs;
}  // namespace

Scope::Scope(const char* name, hipStream_t s, double bytes, double flops) : active_(g_enabled), s_(s) {
    if (!active_) return;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        active_ = false;
        return;
    }
    (void)hipEventRecord(e0, s);
    std::lock_guard<std::mutex> lk(g_mu);
    idx_ = g_recs.size();
    g_recs.push_back(Rec{name, bytes, flops, e0, e1});
}

Scope::~Scope() {
    if (!active_) return;
    hipEvent_t e1;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        e1 = g_recs[idx_].e1;
    }
    (void)hipEventRecord(e1, s_);
}

}  // namespace prof
}  // namespace pn2

extern "C" void pn2_prof_enable(int on) { pn2::prof::g_enabled = on != 0; }

// Synchronises on every recorded event, aggregates by (name, bytes, flops) and clears the records.
// names: NUL-separated names written into names_buf; returns the number of groups (<= max_groups).
extern "C" int pn2_prof_collect(char* names_buf, size_t names_cap, double* total_ms, long long* calls, double* bytes,
                                double* flops, int max_groups) {
    using namespace pn2::prof;
    std::lock_guard<std::mutex> lk(g_mu);
    struct Agg {
        const char* name;
        double bytes, flops, ms;
        long long calls;
    };
    std::vector<Agg> aggs;
    for (auto& r : g_recs) {
        float ms = 0.f;
        (void)hipEventSynchronize(r.e1);
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
        bool found = false;
        for (auto& a : aggs)
            if (a.name == r.name && a.bytes == r.bytes && a.flops == r.flops) {
                a.ms += ms;
                a.calls += 1;
                found = true;
                break;
            }
        if (!found) aggs.push_back(Agg{r.name, r.bytes, r.flops, ms, 1});
    }
    g_recs.clear();
    int n = 0;
    size_t pos = 0;
    for (auto& a : aggs) {
        if (n >= max_groups) break;
        const size_t len = std::char_traits<char>::length(a.name) + 1;
        if (pos + len > names_cap) break;
        std::char_traits<char>::copy(names_buf + pos, a.name, len);
        pos += len;
        total_ms[n] = a.ms;
        calls[n] = a.calls;
        bytes[n] = a.bytes;
        flops[n] = a.flops;
        ++n;
    }
    return n;
}

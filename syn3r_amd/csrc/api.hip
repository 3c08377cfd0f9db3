// Error string, version and arch queries of the C-ABI (include/syn3r_hip.h).
#include "common.h"
#include <string.h>

namespace syn3r {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace syn3r

extern "C" const char* syn3r_last_error(void) { return syn3r::g_err; }
extern "C" int syn3r_version(void) { return 100; }
extern "C" const char* syn3r_arch(void) { return "gfx950"; }

// ---------------------------------------------------------------- kernel tracer
// State is PER CALLING THREAD (thread_local): a thread that enables tracing times ITS OWN launches and reads its own
// report; other threads' launches are untouched (no global mutable state, no lock on the launch path).
#include <map>
#include <string>
#include <vector>

namespace syn3r {
namespace {
struct Span { std::string name; hipEvent_t a, b; };
thread_local bool g_on = false;
thread_local bool g_detail = false;
thread_local std::string g_filter;   // comma-separated substrings; empty = every kernel
thread_local std::vector<Span> g_spans;
thread_local std::vector<hipEvent_t> g_pool;
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
}  // namespace

bool trace_on() { return g_on; }
bool trace_detail() { return g_detail; }
static bool wanted(const char* name) {
    if (g_filter.empty()) return true;
    size_t pos = 0;
    while (pos <= g_filter.size()) {
        size_t end = g_filter.find(',', pos);
        if (end == std::string::npos) end = g_filter.size();
        if (end > pos && strstr(name, g_filter.substr(pos, end - pos).c_str())) return true;
        pos = end + 1;
    }
    return false;
}
bool trace_open(const char* name, hipEvent_t* start, hipEvent_t* stop) {
    if (!wanted(name)) return false;
    Span s{name, get_event(), get_event()};
    *start = s.a;
    *stop = s.b;
    g_spans.push_back(s);
    return true;
}
}  // namespace syn3r

extern "C" int syn3r_trace_enable(int on) {
    syn3r::g_on = on != 0;
    syn3r::g_detail = on == 2;
    return SYN3R_OK;
}

extern "C" int syn3r_trace_filter(const char* substrings) {
    syn3r::g_filter = substrings ? substrings : "";
    return SYN3R_OK;
}

// Synchronises the recorded events, aggregates per kernel name and writes
// "name calls total_ms\n" lines into buf (truncated to cap); clears the trace.
extern "C" int syn3r_trace_report(char* buf, size_t cap) {
    std::map<std::string, std::pair<long long, double>> agg;
    for (auto& s : syn3r::g_spans) {
        float ms = 0.f;
        if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            auto& e = agg[s.name];
            e.first += 1;
            e.second += ms;
        }
        syn3r::g_pool.push_back(s.a);
        syn3r::g_pool.push_back(s.b);
    }
    syn3r::g_spans.clear();
    size_t off = 0;
    if (buf && cap) buf[0] = 0;
    for (auto& kv : agg) {
        char line[256];
        int n = snprintf(line, sizeof(line), "%s %lld %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
        if (buf && off + n + 1 < cap) { memcpy(buf + off, line, n + 1); off += n; }
    }
    return SYN3R_OK;
}

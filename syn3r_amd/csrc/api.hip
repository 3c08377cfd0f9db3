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
// A trace SESSION (switch, filter, recorded spans) is an object owned by the thread that enabled tracing; which session a
// thread's launches record into is a thread-local pointer.  There is no process-global switch: a thread that never
// enabled or attached to a session is untouched by another thread's tracing.  A second thread joins a
// session explicitly - syn3r_trace_attach(session) - which is how the launches PyTorch's autograd thread makes for the
// rasteriser's backward reach the report of the thread that asked for the trace.
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace syn3r {
namespace {
struct Span { std::string name; hipEvent_t a, b; };
struct Session {
    std::mutex mu;                 // spans / pool: the owner and attached threads may launch concurrently
    bool on = false, detail = false;
    std::string filter;            // comma-separated substrings; empty = every kernel
    std::vector<Span> spans;
    std::vector<hipEvent_t> pool;
    hipEvent_t get_event() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    bool wanted(const char* name) const {
        if (filter.empty()) return true;
        size_t pos = 0;
        while (pos <= filter.size()) {
            size_t end = filter.find(',', pos);
            if (end == std::string::npos) end = filter.size();
            if (end > pos && strstr(name, filter.substr(pos, end - pos).c_str())) return true;
            pos = end + 1;
        }
        return false;
    }
};
// Sessions are handed to OTHER threads as raw pointers (syn3r_trace_session / syn3r_trace_attach: PyTorch's autograd thread joins
// the session of the thread that opened the trace), so a session must outlive the thread that created it: sessions live in a
// process-wide registry and are never freed (a few hundred bytes and a handful of pooled events per thread that ever traced).
thread_local Session* tl_own = nullptr;           // the session this thread created
thread_local Session* tl_cur = nullptr;           // the session this thread's launches record into
std::mutex g_sessions_mu;
std::vector<Session*> g_sessions;                 // keeps every session reachable for the life of the process
Session* own_session() {
    if (!tl_own) {
        tl_own = new Session();
        std::lock_guard<std::mutex> lk(g_sessions_mu);
        g_sessions.push_back(tl_own);
    }
    return tl_own;
}
}  // namespace

bool trace_on() { return tl_cur && tl_cur->on; }
bool trace_detail() { return tl_cur && tl_cur->detail; }
bool trace_open(const char* name, hipEvent_t* start, hipEvent_t* stop) {
    Session* s = tl_cur;
    if (!s) return false;
    std::lock_guard<std::mutex> lk(s->mu);
    if (!s->on || !s->wanted(name)) return false;
    Span sp{name, s->get_event(), s->get_event()};
    *start = sp.a;
    *stop = sp.b;
    s->spans.push_back(sp);
    return true;
}
}  // namespace syn3r

extern "C" int syn3r_trace_enable(int on) {
    syn3r::Session* s = syn3r::own_session();
    std::lock_guard<std::mutex> lk(s->mu);
    s->on = on != 0;
    s->detail = on == 2;
    syn3r::tl_cur = s;
    return SYN3R_OK;
}

extern "C" int syn3r_trace_filter(const char* substrings) {
    syn3r::Session* s = syn3r::own_session();
    std::lock_guard<std::mutex> lk(s->mu);
    s->filter = substrings ? substrings : "";
    return SYN3R_OK;
}

extern "C" void* syn3r_trace_session(void) { return syn3r::tl_cur; }

extern "C" int syn3r_trace_attach(void* session) {
    syn3r::tl_cur = (syn3r::Session*)session;       // null detaches
    return SYN3R_OK;
}

// Synchronises the recorded events of the calling thread's OWN session, aggregates per kernel name and writes
// "name calls total_ms\n" lines into buf (truncated to cap); clears the trace.
extern "C" int syn3r_trace_report(char* buf, size_t cap) {
    syn3r::Session* s = syn3r::own_session();
    std::lock_guard<std::mutex> lk(s->mu);
    std::map<std::string, std::pair<long long, double>> agg;
    for (auto& sp : s->spans) {
        float ms = 0.f;
        if (hipEventSynchronize(sp.b) == hipSuccess && hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            auto& e = agg[sp.name];
            e.first += 1;
            e.second += ms;
        }
        s->pool.push_back(sp.a);
        s->pool.push_back(sp.b);
    }
    s->spans.clear();
    size_t off = 0;
    if (buf && cap) buf[0] = 0;
    for (auto& kv : agg) {
        char line[256];
        int n = snprintf(line, sizeof(line), "%s %lld %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
        if (buf && off + n + 1 < cap) { memcpy(buf + off, line, n + 1); off += n; }
    }
    return SYN3R_OK;
}

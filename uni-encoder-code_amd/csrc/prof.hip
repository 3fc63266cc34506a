// Opt-in per-launch timing of the GEMM kernels with HIP events on the launch stream, so that
// bench.py can report the dominant kernel's achieved FLOP/s over the timed region without an
// external profiler.  The only process-global state in the library; off by default (then the
// hooks are a single predictable branch).  Not thread-safe: one launching thread per process.
#include <hip/hip_runtime.h>
#include <vector>

#include "common.h"
#include "prof.h"

namespace {
struct Rec { int kind; double flops, bytes; hipEvent_t e0, e1; };
bool g_on = false;
double g_next_bytes = 0.0;      // algorithmic bytes announced for the next recorded launch (uenc_prof_next_bytes)
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_used = 0;

hipEvent_t get_event() {
    if (g_pool_used == g_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_pool.push_back(e);
    }
    return g_pool[g_pool_used++];
}
}  // namespace

bool uenc_prof_on() { return g_on; }

void uenc_prof_begin(int kind, double flops, hipStream_t stream, double bytes) {
    if (bytes == 0.0) bytes = g_next_bytes;
    g_next_bytes = 0.0;
    Rec r{kind, flops, bytes, get_event(), get_event()};
    if (r.e0 == nullptr || r.e1 == nullptr) return;
    (void)hipEventRecord(r.e0, stream);
    g_recs.push_back(r);
}

void uenc_prof_end(hipStream_t stream) {
    if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().e1, stream);
}

// Grouped launches take their operands from a device-side table: the host that built the table announces the algorithmic
// bytes of the next launch here (profiling only; ignored while the timers are off).
extern "C" int uenc_prof_next_bytes(double bytes) {
    if (g_on) g_next_bytes = bytes;
    return 0;
}

extern "C" int uenc_prof_enable(int on) {
    g_on = (on != 0);
    g_recs.clear();
    g_pool_used = 0;
    return 0;
}

// Sums over all recorded launches of `kind` (call after synchronising the device).
extern "C" int uenc_prof_collect(int kind, double* ms_total, double* flops_total, long* launches) {
    double ms = 0.0, fl = 0.0;
    long n = 0;
    for (const Rec& r : g_recs) {
        if (r.kind != kind) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) continue;
        ms += t; fl += r.flops; ++n;
    }
    if (ms_total) *ms_total = ms;
    if (flops_total) *flops_total = fl;
    if (launches) *launches = n;
    return 0;
}

// Algorithmic bytes (operands read once + results written once) summed over the recorded launches of `kind`.
extern "C" int uenc_prof_collect_bytes(int kind, double* bytes_total) {
    double b = 0.0;
    for (const Rec& r : g_recs)
        if (r.kind == kind) b += r.bytes;
    if (bytes_total) *bytes_total = b;
    return 0;
}

#include "profile.h"

#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sat_hip.h"

namespace sat {
namespace {
struct Rec { std::string name; hipEvent_t e0, e1; double flops, bytes; };
std::mutex g_mu;
bool g_on = false;
std::string g_only;          // non-empty: only scopes of this name (or, ending in '*', of this prefix) are recorded
std::vector<Rec> g_recs;
}  // namespace

bool profile_enabled() { return g_on; }

ProfScope::ProfScope(const char* name, double flops, double bytes, hipStream_t stream) : slot(-1), st(stream) {
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_only.empty()) {         // "family" or "prefix*"
        const bool star = g_only.back() == '*';
        if (star ? strncmp(name, g_only.c_str(), g_only.size() - 1) != 0 : g_only != name) return;
    }
    Rec r; r.name = name; r.flops = flops; r.bytes = bytes;
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    hipEventRecord(r.e0, st);
    g_recs.push_back(r);
    slot = (int)g_recs.size() - 1;
}
ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    hipEventRecord(g_recs[slot].e1, st);
}
}  // namespace sat

using namespace sat;

extern "C" {
int sat_profile_start_only(const char* name) {
    if (!name || !*name) return sat_profile_start();
    sat_profile_start();
    std::lock_guard<std::mutex> lk(g_mu);
    g_only = name;
    return 0;
}
int sat_profile_start(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_only.clear();
    for (auto& r : g_recs) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    g_recs.clear();
    g_on = true;
    return 0;
}
int sat_profile_pause(int32_t paused) {        // keep the records and the family filter, stop / resume recording
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = !paused;
    return 0;
}
int sat_profile_stop(sat_profile_entry* out, int32_t max_entries, int32_t* n_out) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = false;
    std::map<std::string, sat_profile_entry> agg;
    for (auto& r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess) hipEventElapsedTime(&ms, r.e0, r.e1);
        auto& e = agg[r.name];
        if (e.launches == 0) { memset(&e, 0, sizeof e); snprintf(e.name, sizeof e.name, "%s", r.name.c_str()); }
        e.launches += 1; e.total_ms += ms; e.flops += r.flops; e.bytes += r.bytes;
        hipEventDestroy(r.e0); hipEventDestroy(r.e1);
    }
    g_recs.clear();
    int n = 0;
    for (auto& kv : agg) { if (out && n < max_entries) out[n] = kv.second; ++n; }
    if (n_out) *n_out = n < max_entries ? n : max_entries;
    return 0;
}
}

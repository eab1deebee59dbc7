// Optional in-library kernel timing with HIP events on the launch stream (used by bench.py for the roofline
// object).  Off by default: when off a scope costs one predictable branch.
#pragma once
#include "common.h"

namespace sat {
bool profile_enabled();
// records (name, flops, bytes) with an event pair around the launches issued while the scope is alive
struct ProfScope {
    int slot;
    hipStream_t st;
    ProfScope(const char* name, double flops, double bytes, hipStream_t stream);
    ~ProfScope();
};
}  // namespace sat

// Internal: per-kernel HIP-event timing hooks (see um_timing_enable / um_timing_collect in the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/unimatch_hip.h"

bool um_timing_on();
void* um_timing_begin(int kernel_id, hipStream_t stream);   // returns a token or nullptr when timing is off
void um_timing_end(void* token, hipStream_t stream);

struct ScopedKernelTimer {
    void* tok;
    hipStream_t s;
    ScopedKernelTimer(int kid, hipStream_t stream) : tok(um_timing_on() ? um_timing_begin(kid, stream) : nullptr), s(stream) {}
    ~ScopedKernelTimer() { um_timing_end(tok, s); }
};

void um_census_hit(int variant);   // UM_V_*: counts the launch when um_census_enable(1) is in effect

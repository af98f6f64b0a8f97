// In-library per-kernel-class timing with HIP events on the launch stream (bench.py's live
// roofline measurement: the events bracket exactly one kernel launch on the stream it runs on).
#pragma once
#include <hip/hip_runtime.h>

namespace vdsprof {
extern unsigned g_mask;  // bit per class; 0 = profiling off (the normal state: zero overhead)
void begin(int cls, hipStream_t s, double flops, double bytes);
void end(hipStream_t s);
struct Scope {
  bool on;
  hipStream_t s;
  Scope(int cls, hipStream_t st, double flops, double bytes) : on((g_mask >> cls) & 1u), s(st) {
    if (on) begin(cls, st, flops, bytes);
  }
  ~Scope() {
    if (on) end(s);
  }
};
}  // namespace vdsprof

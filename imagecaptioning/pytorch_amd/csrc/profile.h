// In-library launch instrumentation (the backend's counterpart of the reference's wall-clock
// `time/batch` prints, train.py:198-208): when enabled, every launch of an instrumented kernel
// class is bracketed by a pair of HIP events ON THE STREAM THE KERNEL RUNS ON, and the algorithmic
// bytes / flops of the launch are accumulated on the host.  bench.py reads the per-class totals to
// report roofline fractions measured inside the timed region.  Disabled = zero overhead.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum CapmiProfClass {
    CAPMI_PROF_GEMM_DECODE = 0,   // skinny NT GEMMs of the decode step (weight streaming, M <= 64)
    CAPMI_PROF_GEMM_BPTT = 1,     // skinny NN GEMMs of BPTT (dX = dG W)
    CAPMI_PROF_GEMM_FAT = 2,      // everything else (prefill, time-batched weight gradients)
    CAPMI_PROF_ATTENTION_FWD = 3,
    CAPMI_PROF_ATTENTION_BWD = 4,
    CAPMI_PROF_SELECT = 5,
    CAPMI_PROF_LSTM_CELL = 6,
    CAPMI_PROF_CIDERD = 7,
    CAPMI_PROF_ADAM = 8,
    CAPMI_PROF_GEMM_DECODE_STREAM = 9,   // the decode-step GEMMs that stream >= 24 MB of weights (LSTM gates, logit): ONE kernel
                                         // instance, gemm_lc_kernel<true,2>, the dominant row of the rocprofv3 table
    CAPMI_PROF_NCLASS = 10
};

namespace capmi_prof {
// Preferred form for the roofline kernels: the launch itself carries the event pair
// (hipExtLaunchKernelGGL(start, stop)): the timestamps come from the dispatch packet, no extra barrier
// packets are queued, so the timed region of bench.py is not perturbed.  Returns false when the class
// is not being profiled (then launch normally).
bool take_events(int cls, hipEvent_t *start, hipEvent_t *stop, double bytes, double flops);
bool enabled();
void begin(int cls, hipStream_t st, double bytes, double flops);
void end(int cls, hipStream_t st);
struct Scope {
    int cls;
    hipStream_t st;
    bool on;
    Scope(int c, hipStream_t s, double bytes, double flops) : cls(c), st(s), on(enabled()) {
        if (on) begin(cls, st, bytes, flops);
    }
    ~Scope() {
        if (on) end(cls, st);
    }
};
}  // namespace capmi_prof

// Side stream for work that is off a call's critical path (weight/bias gradients beside the dX chain of the temporal backward).
//
// OFF by default (HYB_SIDE_STREAM=1 enables it): measured on MI355X / ROCm 7.2 the parallel branch LOSES 4.7 % of the step (1.559 vs
// 1.487 ms, same box, alternating runs): the ~50 us of weight-gradient kernels do leave the critical path, but every fork/join edge of
// the replayed graph costs more than the 5 us kernels it hides.  (Issued eagerly the same idea lost 2 % in round 1 to host-side event
// calls; config 4 with its heavier d = 768, T = 64 gradient GEMMs: -0.7 %.)  Kept as a switch for future runtimes.
//
// When enabled it is used ONLY while the caller's stream is being captured into a hipGraph (fork/join = graph edges, no host cost); an
// eager call stays on the caller's stream.  The stream and its three events belong to the library (one set per device, created on the
// first NON-capturing call that could use them -- creating a stream is not a capturable operation -- and never destroyed).
#include <mutex>
#include <stdlib.h>
#include "hyb_common.h"

namespace {
constexpr int MAXDEV = 64;
HybSide g_side[MAXDEV];
bool g_ready[MAXDEV];
std::mutex g_mu;
bool side_enabled() {
    static const int on = getenv("HYB_SIDE_STREAM") ? atoi(getenv("HYB_SIDE_STREAM")) : 0;
    return on != 0;
}
}  // namespace

HybSide* hyb_side_for(hipStream_t main) {
    if (!side_enabled()) return nullptr;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &cs) != hipSuccess) return nullptr;
    if (cs != hipStreamCaptureStatusActive) {
        if (!g_ready[dev]) {                                   // not capturing: a safe moment to create this device's objects
            std::lock_guard<std::mutex> lock(g_mu);
            if (!g_ready[dev]) {
                HybSide s{};
                bool ok = hipStreamCreateWithFlags(&s.s, hipStreamNonBlocking) == hipSuccess;
                ok = ok && hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) == hipSuccess;
                ok = ok && hipEventCreateWithFlags(&s.done[0], hipEventDisableTiming) == hipSuccess;
                ok = ok && hipEventCreateWithFlags(&s.done[1], hipEventDisableTiming) == hipSuccess;
                if (ok) { g_side[dev] = s; g_ready[dev] = true; }
            }
        }
        return nullptr;
    }
    return g_ready[dev] ? &g_side[dev] : nullptr;
}

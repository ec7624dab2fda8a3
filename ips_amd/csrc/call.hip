// call.hip - one IPSNet.ips call whose selection loop is resident, enqueued by ONE library call (round 5).
//
// Reference: IPSNet.ips, architecture/ips_net.py:169-262, for ONE image (B_seq = 1, config/mnist_config.yml:4-5) or for
// feature slides (config/camelyon_config.yml).  The pieces are the library's own entry points - fill, the resident loop on
// the side stream (ipsx_scan_persistent_ws), the gate, the producer (ipsx_trunk_stream | ipsx_projector_stream), the
// conditional recovery launch (ipsx_scan_range_if_ws), the end of the call (ipsx_ips_finish) - and the two cross-stream
// hand-overs between them.  Enqueued from Python they cost ~12 ctypes calls, two torch event objects and ~170 us of host
// time per call, of which ~100 us sit IN FRONT of the producer's launch: a synchronised call of one image was 0.92 ms
// around a 0.77 ms kernel.  Enqueued here the host's share is one call; and nothing of the interpreter - the
// garbage collector, the allocator, another thread holding the GIL - can stall the host between the launch of the loop
// and the launch of the producer it waits for (tools/soak.py: what a loop's rare timeouts were made of; what still can is
// the OS descheduling the thread, e.g. a container's CPU quota - see ipsx_dbg_call_host_gap below).
//
// Timing: slots of HIP event pairs owned by the library bracket the PRODUCER's launch on request (bench.py's roofline
// figure: torch's own event objects cannot be recorded from here).

#include "ipsx_common.h"

#include <atomic>
#include <mutex>
#include <time.h>

namespace {

// Diagnostic (ipsx_dbg_call_host_gap; tools/soak.py): host time between the moment the loop's launch was handed to the
// runtime and the moment its producer's launch returned - the window in which a descheduled host thread (a container's
// CPU quota, a collector in another thread's interpreter) leaves a resident loop waiting for rows.  [0] longest (ns),
// [1] calls whose window was longer than 10 ms, [2] calls
std::atomic<unsigned long long> g_gap_max{0}, g_gap_long{0}, g_gap_calls{0};

inline unsigned long long host_ns() {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (unsigned long long)t.tv_sec * 1000000000ull + (unsigned long long)t.tv_nsec;
}

constexpr int kSlots = 64, kDevices = 16;

struct DeviceEvents {
    hipEvent_t fork = nullptr, join = nullptr;          // main -> side (the loop follows the fill), side -> main (the end follows the loop)
    hipEvent_t t0[kSlots] = {}, t1[kSlots] = {};
    bool ok = false;
    std::mutex enqueue;          // one call's launches and hand-overs go out as a block: the events are shared by the device's calls
};

DeviceEvents* device_events() {
    static DeviceEvents ev[kDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kDevices) return nullptr;
    DeviceEvents& e = ev[dev];
    std::lock_guard<std::mutex> lock(e.enqueue);
    if (!e.ok) {
        if (hipEventCreateWithFlags(&e.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&e.join, hipEventDisableTiming) != hipSuccess) return nullptr;
        e.ok = true;
    }
    return &e;
}

}  // namespace

using namespace ipsx;

IPSX_API int ipsx_ips_call_run(const ipsx_ips_call* c) {
    IPSX_REQUIRE(c && c->logits && c->mem_idx && c->words && c->x && c->emb && c->v_packed && c->src && c->mem_patch && c->mem_idx_out,
                 "ips_call: null pointer");
    IPSX_REQUIRE(c->b > 0 && c->n > c->m && c->m > 0 && c->i > 0 && c->h > 0 && c->n_token > 0, "ips_call: bad sizes");
    IPSX_REQUIRE((c->trunk != nullptr) != (c->lin != nullptr), "ips_call: exactly one producer (trunk for one image, lin for feature slides)");
    IPSX_REQUIRE(!c->trunk || c->b == 1, "ips_call: the trunk stream takes ONE image");
    IPSX_REQUIRE(c->side_stream && c->side_stream != c->stream, "ips_call: the loop needs a stream of its own");
    IPSX_REQUIRE(c->words_total >= 2 * (int64_t)c->b + 1, "ips_call: control words");
    DeviceEvents* ev = device_events();
    if (!ev) return fail(IPSX_EHIP, "ips_call: no events on this device");
    std::lock_guard<std::mutex> lock(ev->enqueue);
    hipStream_t main = as_stream(c->stream), side = as_stream(c->side_stream);
    int32_t* const tie = c->words;
    int32_t* const ready = c->words + c->b;
    int32_t* const status = c->words + 2 * c->b;
    int32_t* const ctl = status + 1;
    const int64_t n_iter = cdiv(c->n - c->m, c->i);
    auto hip_ok = [](hipError_t e, const char* what) { return e == hipSuccess ? IPSX_OK : fail(IPSX_EHIP, "ips_call: %s: %s", what, hipGetErrorString(e)); };

    // (the projector stream's control words end in 786 KB of hand-over accumulators that need no zeroing)
    const int64_t zero_words = c->lin ? std::min<int64_t>(c->words_total, 2 * (int64_t)c->b + 1 + (int64_t)ipsx_projector_stream_ctl_zero_words((int64_t)c->b * c->n))
                                      : c->words_total;
    IPSX_TRY(hip_ok(hipMemsetAsync(c->words, 0, (size_t)zero_words * sizeof(int32_t), main), "fill"));
    IPSX_TRY(hip_ok(hipEventRecord(ev->fork, main), "event"));
    IPSX_TRY(hip_ok(hipStreamWaitEvent(side, ev->fork, 0), "wait"));
    const unsigned long long h0 = host_ns();
    IPSX_TRY(ipsx_scan_persistent_ws(c->logits, c->b, c->n, c->m, c->i, c->h, c->n_token, c->mem_idx, nullptr, tie, ready,
                                     c->b > 1 ? 1 : 0, status, c->loops, c->scan_workspace, c->scan_workspace_bytes, side));
    // From here on the resident loop is in flight on the side stream.  A step that fails must not leave it behind
    // un-joined (advisor, round 5): the loop would spin to its bound, the caller's next call would zero `words` on the
    // main stream while that old loop still reads them, and nothing would order the two.  Every failure below therefore
    // goes through `bail`: the loop is told to give up (ready < 0: it leaves at once and sets bit 0 of the status word),
    // `join` is still recorded on the side stream and waited for on the main stream, and the error is returned.
    auto bail = [&](int rc) {
        (void)hipMemsetAsync(ready, 0xFF, (size_t)c->b * sizeof(int32_t), main);     // every progress word negative: give up
        (void)hipEventRecord(ev->join, side);
        (void)hipStreamWaitEvent(main, ev->join, 0);
        return rc;
    };
#define IPSX_TRY_JOINED(expr)                 \
    do {                                      \
        const int rc_ = (expr);               \
        if (rc_ != IPSX_OK) return bail(rc_); \
    } while (0)
    // producers must not take the compute units before a loop has its own (see ipsx_scan_gate)
    IPSX_TRY_JOINED(ipsx_scan_gate(status, main));
    const int slot = c->timing_slot;
    if (slot >= 0 && slot < kSlots) {
        if (!ev->t0[slot]) {                      // both events, or neither: a half-made pair would record a null event later
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
                if (e0) (void)hipEventDestroy(e0);
                return bail(fail(IPSX_EHIP, "ips_call: timing event"));
            }
            ev->t0[slot] = e0;
            ev->t1[slot] = e1;
        }
        IPSX_TRY_JOINED(hip_ok(hipEventRecord(ev->t0[slot], main), "timing event"));
    }
    if (c->trunk)
        IPSX_TRY_JOINED(ipsx_trunk_stream(c->trunk, static_cast<const float*>(c->x), c->n, c->emb, c->pos, c->v_packed, c->r, c->logits, ctl,
                                          ready, c->workgroups, c->quad_pulls, main));
    else
        IPSX_TRY_JOINED(ipsx_projector_stream(c->lin, static_cast<const float*>(c->x), (int64_t)c->b * c->n, c->n, c->ln_eps, c->emb, c->v_packed,
                                              c->r, c->logits, ctl, ready, c->workgroups, c->short_first, main));
    {
        const unsigned long long gap = host_ns() - h0;
        unsigned long long seen = g_gap_max.load(std::memory_order_relaxed);
        while (gap > seen && !g_gap_max.compare_exchange_weak(seen, gap, std::memory_order_relaxed)) {}
        if (gap > 10000000ull) g_gap_long.fetch_add(1, std::memory_order_relaxed);
        g_gap_calls.fetch_add(1, std::memory_order_relaxed);
    }
    if (slot >= 0 && slot < kSlots) IPSX_TRY_JOINED(hip_ok(hipEventRecord(ev->t1[slot], main), "timing event"));
    IPSX_TRY_JOINED(hip_ok(hipEventRecord(ev->join, side), "event"));
    IPSX_TRY(hip_ok(hipStreamWaitEvent(main, ev->join, 0), "wait"));
#undef IPSX_TRY_JOINED
    // a loop that gave up waiting is redone here, in the same call (a no-op otherwise)
    IPSX_TRY(ipsx_scan_range_if_ws(c->logits, c->b, c->n, c->m, c->i, c->h, c->n_token, 0, n_iter, c->mem_idx, nullptr, tie, status, 1,
                                   c->scan_workspace, c->scan_workspace_bytes, main));
    return ipsx_ips_finish(c->src, c->src_row_bytes, c->src_bstride_rows, c->n, c->pos_table, c->pos_row_bytes, c->pos_bstride_rows,
                           c->mem_idx, c->b, c->m, c->mem_patch, c->mem_pos, c->mem_idx_out, status, c->status_host, main);
}

IPSX_API int ipsx_ips_call_elapsed(int slot, float* ms) {
    IPSX_REQUIRE(ms && slot >= 0 && slot < kSlots, "ips_call_elapsed: bad arguments");
    DeviceEvents* ev = device_events();
    if (ev) ev->enqueue.lock();
    const bool recorded = ev && ev->t0[slot];
    if (ev) ev->enqueue.unlock();
    if (!recorded) return fail(IPSX_EINVAL, "ips_call_elapsed: slot %d was never recorded on this device", slot);
    const hipError_t e = hipEventElapsedTime(ms, ev->t0[slot], ev->t1[slot]);
    return e == hipSuccess ? IPSX_OK : fail(IPSX_EHIP, "ips_call_elapsed: %s", hipGetErrorString(e));
}

// Diagnostic (not part of include/ipsx.h): out3 = {longest host window loop launch -> producer launched (ns), windows > 10 ms,
// calls} since the last look; clears them
extern "C" __attribute__((visibility("default"))) void ipsx_dbg_call_host_gap(unsigned long long* out3) {
    out3[0] = g_gap_max.exchange(0);
    out3[1] = g_gap_long.exchange(0);
    out3[2] = g_gap_calls.exchange(0);
}

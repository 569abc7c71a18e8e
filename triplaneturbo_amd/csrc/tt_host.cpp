// tt_host.cpp -- status strings, launch checking, device queries, the work-queue counter scratch
#include "tt_host.h"

#include <stdio.h>
#include <stdlib.h>

#include <mutex>

extern "C" const char* tt_strerror(int status) {
    switch (status) {
        case TT_OK: return "ok";
        case TT_ERR_BAD_ARG: return "bad argument (null pointer, non-positive size or inconsistent shape)";
        case TT_ERR_UNSUPPORTED: return "unsupported configuration (planes must be square, C=32, hidden=64)";
        case TT_ERR_LAUNCH: return "HIP kernel launch failed";
        case TT_ERR_DEVICE: return "HIP device query failed / not a gfx950 device";
        default: return "unknown tt_status";
    }
}

extern "C" int tt_abi_version(void) { return TT_ABI_VERSION; }

// sha256 of the sources + build flags this binary was made from (_lib.source_hash; "unknown" for a hand-made build).
// The marker prefix lets _lib.needs_build read it from the file without loading the library.
#ifndef TT_SOURCE_HASH_STR
#define TT_SOURCE_HASH_STR "unknown"
#endif
static const char k_source_hash[] = "TT_SOURCE_HASH=" TT_SOURCE_HASH_STR;
extern "C" const char* tt_source_hash(void) { return k_source_hash + 15; }

int tt_check_launch() { return hipGetLastError() == hipSuccess ? TT_OK : TT_ERR_LAUNCH; }

int tt_num_cus() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return cus;
}

namespace {
// Work-queue counter slots: TT_SLOT_INTS ints each (8 per-XCD heads + a steal counter, then the per-launch magnitude
// bounds of the backward kernels' fp16 outer products, tt_host.h).
//   * eager launches: a RING of kRing slots per (device, stream).  Launches on a stream are ordered and every launch
//     first enqueues a one-wave kernel that zeroes ITS slot on that stream, so a slot is clean whatever happened to the
//     previous kernel that used it (fault, kill), two launches that may run concurrently (different streams) never
//     share counters, and two host threads that enqueue on the SAME stream at the same time (ctypes releases the GIL)
//     get different slots: the interleaving zero(A) zero(B) K1(A) K2(B) is harmless, where a single slot per stream
//     would let K2 start on counters K1 has already advanced.
//   * launches recorded during a stream capture: slots of a second pool handed out round-robin; the zeroing kernel is
//     captured as a node in front of the kernel node, so a slot may be handed out again after kGraphSlots captured
//     launches (a process that re-captures its graphs per shape / per epoch never runs out).  Not supported: more than
//     kGraphSlots captured launches alive in graphs that replay CONCURRENTLY, and more than kStreams streams with
//     launches in flight at once (the oldest stream entry is recycled, without a device-wide synchronisation -- which
//     would be illegal while another stream is capturing).
constexpr int kMaxDevices = 64, kStreams = 256, kRing = 4, kGraphSlots = 4096, kSlotInts = TT_SLOT_INTS;
struct DeviceScratch {
    int* base = nullptr;             // (kStreams * kRing + kGraphSlots) * kSlotInts ints
    hipStream_t streams[kStreams];   // stream owning eager ring i
    unsigned next[kStreams];         // next ring position of stream i
    int n_streams = 0;
    int evict = 0;                   // next stream entry to recycle once the table is full
    unsigned n_graph = 0;
    bool evict_warned = false;
};
DeviceScratch g_scratch[kMaxDevices];
std::mutex g_scratch_mu;
}  // namespace

__global__ void k_zero_slot(int* slot) { slot[threadIdx.x] = 0; }

int* tt_queue_counters(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) return nullptr;
    int* slot = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        DeviceScratch& d = g_scratch[dev];
        if (!d.base) {
            if (cap != hipStreamCaptureStatusNone) return nullptr;  // first use must not be inside a capture
            void* ptr = nullptr;
            if (hipMalloc(&ptr, (size_t)(kStreams * kRing + kGraphSlots) * kSlotInts * sizeof(int)) != hipSuccess)
                return nullptr;
            d.base = static_cast<int*>(ptr);
        }
        if (cap != hipStreamCaptureStatusNone) {
            if (d.n_graph == kGraphSlots)  // said ONCE: from here on captured launches reuse slots (limits: INTEGRATION.md 3)
                fprintf(stderr, "libtt_hip: %d launches captured on device %d; work-queue slots of captured launches are "
                                "reused from now on -- graphs holding more than %d launches in total must not replay "
                                "concurrently\n", kGraphSlots, dev, kGraphSlots);
            slot = d.base + (size_t)(kStreams * kRing + (d.n_graph++ % kGraphSlots)) * kSlotInts;
        } else {
            int k = 0;
            while (k < d.n_streams && d.streams[k] != stream) ++k;
            if (k == d.n_streams) {
                if (d.n_streams == kStreams) {  // stream handles come and go: recycle the oldest entry
                    if (!d.evict_warned) {
                        d.evict_warned = true;
                        fprintf(stderr, "libtt_hip: more than %d streams have launched on device %d; the oldest stream's "
                                        "work-queue ring is recycled (safe unless more than %d streams have launches in "
                                        "flight at once)\n", kStreams, dev, kStreams);
                    }
                    k = d.evict;
                    d.evict = (d.evict + 1) % kStreams;
                } else {
                    ++d.n_streams;
                }
                d.streams[k] = stream;
                d.next[k] = 0;
            }
            slot = d.base + (size_t)(k * kRing + (d.next[k]++ % kRing)) * kSlotInts;
        }
    }
    // zeroed by a one-wave KERNEL, not hipMemsetAsync: as a graph node the 64-byte memset was not reliably re-executed
    // in front of its kernel on replays >= 2 once few other nodes separated the launches (ROCm 7.2; tests/test_gpu_graph.py
    // caught it: a replay popped no work), a kernel node is
    hipLaunchKernelGGL(k_zero_slot, dim3(1), dim3(kSlotInts), 0, stream, slot);
    if (hipGetLastError() != hipSuccess) return nullptr;
    return slot;
}

// Test hook (tests/test_gpu_graph.py): fills every work-queue slot of `stream`'s ring (so also the one the next eager
// launch on it will use) with garbage, as a faulted or killed kernel would leave it.  The launch must still pop every
// item, because tt_queue_counters zeroes the slot on the stream in front of every launch.
extern "C" int tt_debug_poison_queue(void* stream) {
    for (int r = 0; r < kRing; ++r) {
        int* slot = tt_queue_counters((hipStream_t)stream);
        if (!slot) return TT_ERR_DEVICE;
        if (hipMemsetAsync(slot, 0x7f, kSlotInts * sizeof(int), (hipStream_t)stream) != hipSuccess) return TT_ERR_DEVICE;
    }
    return TT_OK;
}

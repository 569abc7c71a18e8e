// tt_host.cpp -- status strings, launch checking, device queries, the work-queue counter scratch
#include "tt_host.h"

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

int tt_check_launch() { return hipGetLastError() == hipSuccess ? TT_OK : TT_ERR_LAUNCH; }

int tt_num_cus() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return cus;
}

namespace {
// Work-queue counter slots: 64 B each (8 per-XCD heads + padding: one cache line).
//   * eager launches: ONE slot per (device, stream).  Launches on a stream are ordered, and every launch first
//     enqueues a one-wave kernel that zeroes its slot on that stream, so a slot is clean whatever happened to the
//     previous kernel that used it (fault, kill) and two launches that may run concurrently (different streams)
//     never share counters.
//   * launches recorded during a stream capture: a slot of their own from a second pool, never handed out again (a
//     graph may be replayed on any stream, concurrently with eager launches on the capture stream); the zeroing
//     kernel is captured as a node in front of the kernel node.
constexpr int kMaxDevices = 64, kStreamSlots = 64, kGraphSlots = 4096, kSlotInts = 16;
struct DeviceScratch {
    int* base = nullptr;                 // (kStreamSlots + kGraphSlots) * kSlotInts ints
    hipStream_t streams[kStreamSlots];   // stream owning eager slot i
    int n_streams = 0;
    int n_graph = 0;
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
            if (hipMalloc(&ptr, (size_t)(kStreamSlots + kGraphSlots) * kSlotInts * sizeof(int)) != hipSuccess)
                return nullptr;
            d.base = static_cast<int*>(ptr);
        }
        if (cap != hipStreamCaptureStatusNone) {
            if (d.n_graph >= kGraphSlots) return nullptr;
            slot = d.base + (size_t)(kStreamSlots + d.n_graph++) * kSlotInts;
        } else {
            int k = 0;
            while (k < d.n_streams && d.streams[k] != stream) ++k;
            if (k == d.n_streams) {
                if (d.n_streams == kStreamSlots) {
                    // more distinct streams than slots (stream handles come and go): once the device is idle no
                    // slot is in use, so the table starts over
                    if (hipDeviceSynchronize() != hipSuccess) return nullptr;
                    d.n_streams = 0;
                    k = 0;
                }
                ++d.n_streams;
                d.streams[k] = stream;
            }
            slot = d.base + (size_t)k * kSlotInts;
        }
    }
    // zeroed by a one-wave KERNEL, not hipMemsetAsync: as a graph node the 64-byte memset was not reliably re-executed
    // in front of its kernel on replays >= 2 once few other nodes separated the launches (ROCm 7.2; tests/test_gpu_graph.py
    // caught it: a replay popped no work), a kernel node is
    hipLaunchKernelGGL(k_zero_slot, dim3(1), dim3(kSlotInts), 0, stream, slot);
    if (hipGetLastError() != hipSuccess) return nullptr;
    return slot;
}

// Test hook (tests/test_gpu_graph.py): fills the work-queue slot the next eager launch on `stream` will use with
// garbage, as a faulted or killed kernel would leave it.  The launch must still pop every item, because
// tt_queue_counters zeroes the slot on the stream in front of every launch.
extern "C" int tt_debug_poison_queue(void* stream) {
    int* slot = tt_queue_counters((hipStream_t)stream);
    if (!slot) return TT_ERR_DEVICE;
    return hipMemsetAsync(slot, 0x7f, kSlotInts * sizeof(int), (hipStream_t)stream) == hipSuccess ? TT_OK
                                                                                                 : TT_ERR_DEVICE;
}

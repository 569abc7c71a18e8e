// tt_host.cpp -- status strings, launch checking, device queries, the work-queue counter scratch
#include "tt_host.h"

#include <stdlib.h>

#include <atomic>
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
constexpr int kMaxDevices = 64, kSlots = 256, kSlotInts = 16;  // 64 B per slot: one cache line
int* g_scratch[kMaxDevices] = {};
std::mutex g_scratch_mu;
std::atomic<unsigned> g_next_slot{0};
}  // namespace

int* tt_queue_counters(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    int* base;
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        if (!g_scratch[dev]) {
            void* ptr = nullptr;
            if (hipMalloc(&ptr, (size_t)kSlots * kSlotInts * sizeof(int)) != hipSuccess) return nullptr;
            if (hipMemset(ptr, 0, (size_t)kSlots * kSlotInts * sizeof(int)) != hipSuccess) return nullptr;
            g_scratch[dev] = static_cast<int*>(ptr);
        }
        base = g_scratch[dev];
    }
    (void)stream;  // the kernels leave their slot zeroed (item_pop): no memset to enqueue
    return base + (size_t)(g_next_slot.fetch_add(1) % kSlots) * kSlotInts;
}

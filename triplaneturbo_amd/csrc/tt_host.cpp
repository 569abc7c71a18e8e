// tt_host.cpp -- status strings, launch checking, device queries (no global mutable state)
#include "tt_host.h"

#include <stdlib.h>

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

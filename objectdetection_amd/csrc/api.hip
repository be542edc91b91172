// Version and error-text entry points of libodamd.so.
#include "common.h"
#include "odamd.h"

extern "C" {

int odamd_abi_version(void) { return ODAMD_ABI_VERSION; }

const char* odamd_error_string(int code) {
    switch (code) {
        case ODAMD_OK: return "ok";
        case ODAMD_E_ARG: return "invalid argument (null pointer, non-positive size or unsupported value)";
        case ODAMD_E_ALIGN: return "pointer or channel count violates the required alignment";
        case ODAMD_E_WORKSPACE: return "workspace too small";
        case ODAMD_E_SHAPE: return "inconsistent shapes";
        case ODAMD_E_LIMIT: return "size exceeds the fixed capacity of this allocation-free entry point";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown odamd error";
    }
}

}  // extern "C"

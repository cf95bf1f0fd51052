// C-ABI housekeeping entry points (version, error strings).
#include "common.h"

extern "C" int32_t nsos_abi_version(void) { return NSOS_ABI_VERSION; }

extern "C" const char* nsos_error_string(int32_t code) {
    switch (code) {
        case NSOS_OK: return "ok";
        case NSOS_ERR_NULL_POINTER: return "a required pointer argument is NULL";
        case NSOS_ERR_BAD_SHAPE: return "negative, zero or inconsistent sizes";
        case NSOS_ERR_UNSUPPORTED: return "shape outside what the gfx950 kernels are specialised for";
        case NSOS_ERR_BUFFER_TOO_SMALL: return "output buffer too small";
        case NSOS_ERR_MISALIGNED: return "pointer must be 16-byte aligned";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown nsos error";
    }
}

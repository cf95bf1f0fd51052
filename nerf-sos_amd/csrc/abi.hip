// C-ABI housekeeping entry points (version, error strings).
#include "common.h"

extern "C" int32_t nsos_abi_version(void) { return NSOS_ABI_VERSION; }

#ifndef NSOS_SOURCE_HASH
#define NSOS_SOURCE_HASH "unstamped"
#endif
// sha256 (first 16 hex digits) over the sources, headers and compiler flags this library was built from
// (__graft_entry__.source_hash): a stale .so or object is detectable without a GPU (tests/test_abi.py).
extern "C" const char* nsos_source_hash(void) { return NSOS_SOURCE_HASH; }

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

// capi.hip -- ABI bookkeeping entry points of libmulactseg_hip.so
#include "common.h"

extern "C" int mas_abi_version(void) { return MAS_ABI_VERSION; }

extern "C" const char* mas_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case MAS_ERR_NULL: return "null pointer argument";
        case MAS_ERR_SHAPE: return "bad shape (non-positive size, or image larger than the fixed-point accumulators allow)";
        case MAS_ERR_CLASSES: return "class count out of range (2..MAS_MAX_CLASSES)";
        case MAS_ERR_DTYPE: return "unknown superpixel-id dtype code";
        case MAS_ERR_ALIGN: return "pointer not aligned as required";
        case MAS_ERR_RANGE: return "argument out of range";
        case MAS_ERR_WORKSPACE: return "workspace too small";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown mulactseg error";
    }
}

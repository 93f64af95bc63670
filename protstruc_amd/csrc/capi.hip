// Library-level entry points of libprotstruc_hip.so (see include/protstruc_hip.h).
#include <hip/hip_runtime.h>
#include <string.h>

#include "../../include/protstruc_hip.h"

extern "C" int ps_k1_set_tuning(const char* key, int value);
extern "C" int ps_k1_get_tuning(const char* key, int* value);

extern "C" int ps_abi_version(void) { return 1; }

extern "C" const char* ps_error_string(int code) { return hipGetErrorString(static_cast<hipError_t>(code)); }

extern "C" int ps_set_tuning(const char* key, int value) {
    if (!key) return (int)hipErrorInvalidValue;
    if (!strncmp(key, "k1_", 3)) return ps_k1_set_tuning(key, value);
    return (int)hipErrorInvalidValue;
}

extern "C" int ps_get_tuning(const char* key, int* value) {
    if (!key || !value) return (int)hipErrorInvalidValue;
    if (!strncmp(key, "k1_", 3)) return ps_k1_get_tuning(key, value);
    return (int)hipErrorInvalidValue;
}

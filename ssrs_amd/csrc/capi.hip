// Library-level entry points of libssrs_hip.so: version, error text, device info.
#include <cstring>

#include "common.h"

namespace ssrs {

char *error_buffer()
{
    static thread_local char buf[512] = "";
    return buf;
}

int set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace ssrs

extern "C" int ssrs_version(void) { return SSRS_VERSION; }

extern "C" const char *ssrs_last_error(void) { return ssrs::error_buffer(); }

extern "C" int ssrs_device_info(int device, char *name, size_t name_len, int *compute_units,
                                size_t *hbm_bytes)
{
    hipDeviceProp_t prop;
    SSRS_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (name && name_len) {
        strncpy(name, prop.name, name_len - 1);
        name[name_len - 1] = '\0';
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return SSRS_OK;
}

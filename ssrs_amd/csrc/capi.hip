// Library-level entry points of libssrs_hip.so: version, error text, device info.
#include <dlfcn.h>

#include <cstring>

#include <rccl/rccl.h>          // types and enums only: the functions are resolved at run time

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

// ------------------------------------------------------------------ histogram reduce
// The one exchange step of the track-sharded run: sum of the uint32 presence
// histograms over the ranks of an RCCL communicator the CALLER created (one process per GPU).
// RCCL is not linked: its two entry points are looked up in the process first (the caller's own
// RCCL -- a process must not mix two copies of the library with one communicator), then in
// librccl.so.1.
namespace {
using ReduceFn = ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t);
using AllReduceFn = ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);

void *rccl_symbol(const char *name)
{
    if (void *p = dlsym(RTLD_DEFAULT, name)) return p;
    static void *handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    return handle ? dlsym(handle, name) : nullptr;
}
}  // namespace

extern "C" int ssrs_hist_reduce(uint32_t *hist, size_t n, int root, void *nccl_comm, void *stream)
{
    SSRS_REQUIRE(hist != nullptr && nccl_comm != nullptr, "ssrs_hist_reduce: hist / communicator is NULL");
    if (n == 0) return SSRS_OK;
    static ReduceFn reduce = reinterpret_cast<ReduceFn>(rccl_symbol("ncclReduce"));
    static AllReduceFn all_reduce = reinterpret_cast<AllReduceFn>(rccl_symbol("ncclAllReduce"));
    SSRS_REQUIRE(reduce && all_reduce, "ssrs_hist_reduce: RCCL (ncclReduce / ncclAllReduce) is not available in this process");
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
    const ncclResult_t rc = root < 0
        ? all_reduce(hist, hist, n, ncclUint32, ncclSum, comm, ssrs::as_stream(stream))
        : reduce(hist, hist, n, ncclUint32, ncclSum, root, comm, ssrs::as_stream(stream));
    if (rc != ncclSuccess) return ssrs::set_error(SSRS_ERR_HIP, "ssrs_hist_reduce: RCCL returned %d", static_cast<int>(rc));
    return SSRS_OK;
}

// Internal helpers shared by the translation units of libssrs_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/ssrs_hip.h"

namespace ssrs {

// thread-local message behind ssrs_last_error()
char *error_buffer();
int set_error(int code, const char *fmt, ...);

#define SSRS_HIP_CHECK(expr)                                                      \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess)                                                     \
            return ssrs::set_error(SSRS_ERR_HIP, "%s failed: %s (%s:%d)", #expr,  \
                                   hipGetErrorString(e_), __FILE__, __LINE__);    \
    } while (0)

#define SSRS_REQUIRE(cond, ...)                                     \
    do {                                                            \
        if (!(cond)) return ssrs::set_error(SSRS_ERR_INVALID, __VA_ARGS__); \
    } while (0)

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// MI355X: 256 CUs; memory-bound grids are capped at 8 blocks of 256 per CU and
// grid-stride the rest (cdna_hip_programming.md, Guideline 11).
constexpr int kBlock = 256;
constexpr int kMaxStreamBlocks = 256 * 8;

}  // namespace ssrs

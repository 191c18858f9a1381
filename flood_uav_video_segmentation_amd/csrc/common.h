// Shared host-side helpers for the floodseg HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>

namespace fs {

// Last error text, per thread; read through fs_last_error().
std::string& last_error();
int fail(const char* fmt, ...);

#define FS_HIP(expr)                                                                  \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess)                                                         \
            return ::fs::fail("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
    } while (0)

#define FS_REQUIRE(cond, ...)                                                         \
    do {                                                                              \
        if (!(cond)) return ::fs::fail(__VA_ARGS__);                                  \
    } while (0)

#define FS_TRY(expr)                                                                  \
    do {                                                                              \
        int _r = (expr);                                                              \
        if (_r != 0) return _r;                                                       \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace fs

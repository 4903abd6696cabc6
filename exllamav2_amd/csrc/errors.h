// errors.h -- error convention of the C ABI: every entry point returns 0 on success or a negative EXL2_E_* code and
// leaves a message retrievable with exl2_last_error().  (The reference throws c10::Error through TORCH_CHECK,
// exllamav2_ext/cpp/util.h:33-38; the Python host layer turns these codes back into RuntimeError.)
#pragma once
#include <stdio.h>
#include <stdarg.h>

#define EXL2_OK            0
#define EXL2_E_INVALID    -1     // bad argument / shape / dtype contract
#define EXL2_E_OOM        -2     // device allocation failed ("HIP out of memory", model.py:637-639 string-matches this)
#define EXL2_E_HIP        -3     // HIP runtime error
#define EXL2_E_UNSUPPORTED -4

void exl2_set_error(const char* fmt, ...);

#define EXL2_FAIL(code, ...) do { exl2_set_error(__VA_ARGS__); return (code); } while (0)
#define EXL2_REQUIRE(cond, ...) do { if (!(cond)) { exl2_set_error(__VA_ARGS__); return EXL2_E_INVALID; } } while (0)
#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { \
    exl2_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    return _e == hipErrorOutOfMemory ? EXL2_E_OOM : EXL2_E_HIP; } } while (0)

// Host-side device scoping (the reference wraps every binding in an at::cuda::OptionalCUDAGuard, e.g. ext_qmatrix.cpp:41):
// entry points that own a device index switch to it and put the caller's device back on the way out; one-time per-device
// set-up (function attributes, CU count, scratch) is keyed on the device that is current at the call.
struct DeviceGuard
{
    int prev = -1; bool switched = false;
    explicit DeviceGuard(int dev)
    {
        if (dev < 0 || hipGetDevice(&prev) != hipSuccess || prev == dev) return;
        switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define EXL2_MAX_DEVICES 32
static inline int exl2_current_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= EXL2_MAX_DEVICES) d = 0; return d; }
// true exactly once per device for the given flag array
static inline bool exl2_first_on_device(bool* flags) { const int d = exl2_current_device(); if (flags[d]) return false; flags[d] = true; return true; }

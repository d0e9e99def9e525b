// host_util.h -- error plumbing of the C ABI: no exception crosses an extern "C" entry point
// (same convention as the reference's libsdod.cpp:102-108: exception -> status code + message).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <atomic>
#include <stdexcept>
#include <string>

namespace sdod {

enum Status { // mirrors enum libsdod_status_code, csrc/libsdod/api/libsdod.h:11-18
    OK = 0,
    INVALID_CONTEXT = 1,
    INVALID_ARGUMENT = 2,
    FAILED_ALLOCATION = 3,
    RUNTIME_ERROR = 4,
    INTERNAL_ERROR = 5,
};

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string& msg);
const char* get_last_error();

// Per-kernel device timing without a profiler: while a LaunchTimer is installed on the calling thread (Graph::profile does
// it around every launch-list entry) the kernels are launched through hipExtLaunchKernelGGL with a start / stop event pair,
// which the command processor stamps at the kernel's own begin and end -- the duration rocprofv3 reports for the dispatch,
// without the queue / dispatch latency that an event pair AROUND an eager launch includes (2-4 us per launch).
struct LaunchTimer {
    hipEvent_t* start;
    hipEvent_t* stop;
    int cap;
    int used;
};
extern thread_local LaunchTimer* g_launch_timer;

// Kernel attributes (dynamic LDS size) are per device: true the first time the calling thread's CURRENT device meets the
// call site owning `mask` (one bit per device ordinal; two threads racing both set the attribute, which is idempotent).
inline bool first_use_on_device(std::atomic<unsigned long long>& mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    const unsigned long long bit = 1ull << dev;
    return (mask.fetch_or(bit, std::memory_order_relaxed) & bit) == 0;
}

} // namespace sdod

#define SDOD_LAUNCH(kernel, grid, block, smem, stream, ...)                                                          \
    do {                                                                                                            \
        sdod::LaunchTimer* lt_ = sdod::g_launch_timer;                                                              \
        if (lt_ != nullptr && lt_->used < lt_->cap) {                                                               \
            hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(smem), stream, lt_->start[lt_->used],        \
                                  lt_->stop[lt_->used], 0, __VA_ARGS__);                                            \
            ++lt_->used;                                                                                            \
        } else {                                                                                                    \
            hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);                                     \
        }                                                                                                           \
    } while (0)

#define SDOD_STR2(x) #x
#define SDOD_STR(x) SDOD_STR2(x)

#define SDOD_REQUIRE(cond, msg)                                                                                   \
    do {                                                                                                          \
        if (!(cond))                                                                                              \
            throw sdod::Error(sdod::INVALID_ARGUMENT,                                                             \
                              std::string(__func__) + ": " + (msg) + " [" __FILE__ ":" SDOD_STR(__LINE__) "]");    \
    } while (0)

#define SDOD_HIP_CHECK(expr)                                                                                      \
    do {                                                                                                          \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess)                                                                                     \
            throw sdod::Error(sdod::RUNTIME_ERROR, std::string(__func__) + ": HIP error " + hipGetErrorString(_e) + \
                                                       " [" __FILE__ ":" SDOD_STR(__LINE__) "]");                  \
    } while (0)

#define SDOD_TRY try {
#define SDOD_CATCH                                         \
    }                                                      \
    catch (const sdod::Error& e) {                         \
        sdod::set_last_error(e.what());                    \
        return e.code;                                     \
    }                                                      \
    catch (const std::bad_alloc& e) {                      \
        sdod::set_last_error(e.what());                    \
        return sdod::FAILED_ALLOCATION;                    \
    }                                                      \
    catch (const std::exception& e) {                      \
        sdod::set_last_error(e.what());                    \
        return sdod::INTERNAL_ERROR;                       \
    }                                                      \
    catch (...) {                                          \
        sdod::set_last_error("unspecified error");         \
        return sdod::INTERNAL_ERROR;                       \
    }

// misc.hip -- last-error slot and device query of the kernel-level C ABI.
#include "host_util.h"
#include "sdod_hip.h"

#include <cstring>

namespace sdod {
static thread_local std::string g_last_error;
thread_local LaunchTimer* g_launch_timer = nullptr;
void set_last_error(const std::string& msg) { g_last_error = msg; }
const char* get_last_error() { return g_last_error.c_str(); }
} // namespace sdod

extern "C" const char* sdod_hip_last_error(void) { return sdod::get_last_error(); }

extern "C" int sdod_hip_device_info(int* cu_count, size_t* hbm_bytes, char* arch, int arch_len) {
    SDOD_TRY
    int dev = 0;
    SDOD_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    SDOD_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    if (arch && arch_len > 0) {
        std::strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return 0;
    SDOD_CATCH
}

// Library-level entry points of liborv_mi355: version, thread-local error text, device check.
#include "common.hpp"
#include <string.h>

static thread_local char g_err[512] = "";

void orv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int orv_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orv_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return ORV_ELAUNCH;
    }
    return ORV_OK;
}

extern "C" int orv_version(void) { return 1; }
extern "C" const char* orv_last_error(void) { return g_err; }

extern "C" int orv_device_check(int device) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        orv_set_error("orv_device_check: hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
        return ORV_EDEVICE;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        orv_set_error("orv_device_check: device %d is %s, this library is built for gfx950 (MI355X) only", device,
                      prop.gcnArchName);
        return ORV_EDEVICE;
    }
    return ORV_OK;
}

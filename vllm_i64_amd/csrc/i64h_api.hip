// Library-wide pieces of libi64hip.so: error string, version, device probe.
#include "i64h_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void i64h_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static long long g_grid_wait_budget = 0;
long long i64h_grid_wait_budget() { return g_grid_wait_budget > 0 ? g_grid_wait_budget : 600000000ll; }

extern "C" {

// Cycles a grid barrier of a resident launch may wait before it gives up, raises the sticky error word and goes on
// (0 restores the default, about a quarter of a second).  A knob for tests of the error path.
int i64h_set_grid_wait_budget(int64_t cycles) {
    I64H_CHECK_ARG(cycles >= 0, "set_grid_wait_budget: %lld", (long long)cycles);
    g_grid_wait_budget = cycles;
    return 0;
}

const char* i64h_last_error(void) { return g_err; }

int i64h_version(void) { return 100; }

// 0 when the current device is a gfx950 part the code objects in this library were built for.
int i64h_device_check(void) {
    int dev = 0;
    I64H_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    I64H_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    I64H_CHECK_ARG(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "device %d is %s; libi64hip.so is built for gfx950 only", dev, prop.gcnArchName);
    return 0;
}

}  // extern "C"

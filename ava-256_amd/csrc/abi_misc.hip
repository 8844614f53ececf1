// abi_misc.hip -- version / error-string / device-query entry points of the C ABI.
#include <string.h>

#include "mvp_host.h"

extern "C" int mvp_abi_version(void) { return MVP_ABI_VERSION; }

extern "C" const char *mvp_error_string(int code) {
    switch (code) {
        case MVP_OK: return "ok";
        case MVP_ERR_BADARG: return "bad argument (null / misaligned pointer, bad size or non-finite scalar)";
        case MVP_ERR_UNSUPPORTED: return "shape not supported by this build";
        case MVP_ERR_NODEVICE: return "no usable HIP device";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}

extern "C" int mvp_device_arch(int device, char *buf, int buflen) {
    if (!buf || buflen <= 0) return MVP_ERR_BADARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
        (void)hipGetLastError();
        return MVP_ERR_NODEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return MVP_ERR_NODEVICE;
    strncpy(buf, prop.gcnArchName, (size_t)buflen - 1);
    buf[buflen - 1] = 0;
    return MVP_OK;
}

// api.hip - library-level entry points of libipsx (version, errors, device probe).
#include <string.h>

#include "ipsx_common.h"

namespace ipsx {

char* err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace ipsx

IPSX_API int ipsx_version(void) { return IPSX_VERSION; }
IPSX_API const char* ipsx_last_error(void) { return ipsx::err_buf(); }

IPSX_API int ipsx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

IPSX_API int ipsx_device_is_gfx950(int dev) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

// Error reporting and identification for libpasta_hip.so.
#include "common.h"
#include <hip/hip_version.h>

namespace pasta {

char* error_buffer() {
    static thread_local char buf[1024] = {0};
    return buf;
}

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 1024, fmt, ap);
    va_end(ap);
    return 1;
}

}  // namespace pasta

extern "C" const char* pasta_last_error(void) { return pasta::error_buffer(); }
extern "C" int pasta_abi_version(void) { return 21; }
extern "C" const char* pasta_build_info(void) {
#define PASTA_STR2(x) #x
#define PASTA_STR(x) PASTA_STR2(x)
    return "libpasta_hip gfx950 " __DATE__ " hip " PASTA_STR(HIP_VERSION_MAJOR) "." PASTA_STR(HIP_VERSION_MINOR);
}

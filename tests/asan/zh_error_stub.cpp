// The two error entry points of libzebra_hip.so (zh_api.hip) for the stand-alone sanitizer build of zh_refformat.cpp.
#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/zebra_hip.h"

static thread_local std::string g_err;

int zh_set_error(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
extern "C" ZH_API const char *zh_last_error(void) { return g_err.c_str(); }

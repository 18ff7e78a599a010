// error.cpp -- thread-local last-error string + ABI version.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/slnlp.h"

namespace slnlp {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace slnlp

extern "C" const char* slnlp_last_error(void) { return slnlp::g_err; }
extern "C" int slnlp_abi_version(void) { return SLNLP_ABI_VERSION; }

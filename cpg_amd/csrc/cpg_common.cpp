#include "cpg_common.h"
#include <stdarg.h>

namespace cpg {

static thread_local char g_err[512] = "";

char *err_buf() { return g_err; }

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static thread_local int g_shared_chip = 0;
int shared_chip_hint() { return g_shared_chip; }

}  // namespace cpg

extern "C" int cpg_set_shared_chip_hint(int32_t shared) {
    cpg::g_shared_chip = shared ? 1 : 0;
    return CPG_OK;
}
extern "C" int cpg_version(void) { return CPG_ABI_VERSION; }
extern "C" const char *cpg_last_error(void) { return cpg::err_buf(); }

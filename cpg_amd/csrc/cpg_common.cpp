#include "cpg_common.h"
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>

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

static thread_local PackCtx g_pack = {};
PackCtx &pack_ctx() { return g_pack; }

int pack_site(const PackJob &job, const float **pre, const char *what) {
    PackCtx &c = g_pack;
    if (c.mode == PACK_QUERY) {
        c.job = job;
        c.hit = true;
        return 1;
    }
    if (c.mode == PACK_USE && !c.hit) {
        if (c.use == nullptr || c.use_bytes != job.bytes || (((uintptr_t)c.use) & 15) != 0)
            return fail(CPG_E_INVALID, "%s: the packed operand handed in (%zu bytes) is not the one this launch streams (%zu bytes, 16-byte aligned)",
                        what, c.use_bytes, job.bytes);
        *pre = c.use;
        c.hit = true;
        return 2;
    }
    return 0;
}

// Process-wide: the weight-gradient planners that read it run inside autograd's backward, i.e. on the engine's per-device worker
// thread, not on the thread that called cpg_set_shared_chip_hint (round 3 kept it thread-local and the planners never saw it).
static std::atomic<int> g_shared_chip{0};
int shared_chip_hint() { return g_shared_chip.load(std::memory_order_relaxed); }

namespace {
enum Kind { BOOL, INT, WINO_KERNEL };
struct OptDef {
    const char *name;
    Kind kind;
};
// same order as enum Opt
const OptDef kDefs[OPT_COUNT] = {
    {"CPG_NO_WINO", BOOL},          {"CPG_NO_WINO_WGRAD", BOOL},   {"CPG_NO_WINO_ODD", BOOL},      {"CPG_NO_STEM", BOOL},
    {"CPG_NO_STEM_FUSE", BOOL},     {"CPG_NO_DEAD_SKIP", BOOL},    {"CPG_NO_S2", BOOL},            {"CPG_NO_V14", BOOL},
    {"CPG_DISABLE_CONV3X3", BOOL},  {"CPG_DISABLE_CONV1X1", BOOL}, {"CPG_DISABLE_CONV1X1_WGRAD", BOOL}, {"CPG_DISABLE_PW_GEMM", BOOL},
    {"CPG_C3_FORCE", INT},          {"CPG_C3W_BPC", INT},          {"CPG_W3_PICK", INT},           {"CPG_PWW_BPC", INT},
    {"CPG_STEM_BLOCKS", INT},       {"CPG_WINO_KERNEL", WINO_KERNEL}, {"CPG_WINO_NW", INT},        {"CPG_WINO_PERSIST", INT},
    {"CPG_WINO_GRIDS", INT},        {"CPG_WW_UNITS", INT},         {"CPG_WW_XCD", INT},            {"CPG_PW_TILE", INT},           {"CPG_WG3_SHARE", INT},
    {"CPG_WW_SHARE", INT},          {"CPG_WINO_TAIL", INT},         {"CPG_FC_SMALL", INT},
};

struct Table {
    std::atomic<int> v[OPT_COUNT];
    Table() {                                    // runs once, when the shared library is loaded
        for (int i = 0; i < OPT_COUNT; ++i) {
            int val = OPT_UNSET;
            if (const char *f = getenv(kDefs[i].name)) {
                switch (kDefs[i].kind) {
                    case BOOL: val = (f[0] == 0 || (f[0] == '0' && f[1] == 0)) ? 0 : 1; break;
                    case INT: val = atoi(f); break;
                    case WINO_KERNEL: val = f[0] == 'b' ? 0 : f[0] == 'p' ? 2 : f[0] == '6' ? 3 : 1; break;
                }
            }
            v[i].store(val, std::memory_order_relaxed);
        }
    }
};
Table g_opts;

int find_opt(const char *name) {
    if (name == nullptr) return -1;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (strcmp(name, kDefs[i].name) == 0) return i;
    return -1;
}
}  // namespace

int opt(Opt o) { return g_opts.v[o].load(std::memory_order_relaxed); }

}  // namespace cpg

extern "C" int cpg_set_shared_chip_hint(int32_t shared) {
    cpg::g_shared_chip.store(shared ? 1 : 0, std::memory_order_relaxed);
    return CPG_OK;
}
extern "C" int32_t cpg_get_shared_chip_hint(void) { return cpg::shared_chip_hint(); }
extern "C" int cpg_set_option(const char *name, int32_t value) {
    const int i = cpg::find_opt(name);
    if (i < 0) return cpg::fail(CPG_E_INVALID, "cpg_set_option: unknown option '%s'", name ? name : "(null)");
    cpg::g_opts.v[i].store(value, std::memory_order_relaxed);
    return CPG_OK;
}
extern "C" int cpg_get_option(const char *name, int32_t *value) {
    const int i = cpg::find_opt(name);
    if (i < 0 || value == nullptr) return cpg::fail(CPG_E_INVALID, "cpg_get_option: unknown option '%s'", name ? name : "(null)");
    *value = cpg::g_opts.v[i].load(std::memory_order_relaxed);
    return CPG_OK;
}
extern "C" int cpg_version(void) { return CPG_ABI_VERSION; }
extern "C" const char *cpg_last_error(void) { return cpg::err_buf(); }

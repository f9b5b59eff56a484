/* mg_common.h -- helpers shared by the stamped host-layer sources (C11). */
#ifndef MG_COMMON_H
#define MG_COMMON_H
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mg_multigrid.h"

#define MG_CAT_(a, b) a##b
#define MG_CAT(a, b) MG_CAT_(a, b)
#define MG_CAT3(a, b, c) MG_CAT(MG_CAT(a, b), c)

#define MG_PI (3.141592653589793) /* N3/inclusion.h:9 */

static inline int mg_fail(int status, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    mgx_set_last_error(buf);
    return status;
}

/* MG_CA_TRACE=1 in the environment: the slab driver prints the launches and exchanges of its communication-avoiding
 * schedule (level, pass, plane ranges) to stderr -- a debugging aid, read once */
static inline int mg_ca_trace(void) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("MG_CA_TRACE"); on = e && *e && *e != '0'; }
    return on;
}

#define MG_TRY(expr)            \
    do {                        \
        int st_ = (expr);       \
        if (st_) return st_;    \
    } while (0)

#define MG_REQUIRE(cond, status, ...)                      \
    do {                                                   \
        if (!(cond)) return mg_fail(status, __VA_ARGS__);  \
    } while (0)

#endif

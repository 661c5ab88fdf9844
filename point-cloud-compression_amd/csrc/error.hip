// error.hip -- error plumbing of the C ABI (include/pccx.h: pccx_last_error, pccx_version).  Pure host code, in a file of its own so that
// the host-only sanitizer build of the packers (oracle/Makefile: `make -C oracle asan`, g++ -fsanitize=address,undefined over
// error.hip + pack.hip + pack_h2.hip) links without any device code.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void pccx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
extern "C" const char *pccx_last_error(void) { return g_err; }
extern "C" int pccx_version(void) { return 100; }

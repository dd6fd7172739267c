// capi_common.cpp -- error string storage and version entry points of the C ABI.
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "sn_common.h"

long long sn_emd_workspace_floats(int b, int n, int m);
long long sn_emd_sweep2d_floats(int b, int n, int m);

static thread_local char g_err[512] = "";

int sn_set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int sn_abi_version(void) { return SN_ABI_VERSION; }
extern "C" const char *sn_last_error_string(void) { return g_err; }
extern "C" long long sn_workspace_bytes(const char *op, int B, int N, int M, int K)
{
    (void)K;
    if (!op) return -1;
    // approxmatch: the reference op allocates (b,(n+m)*2) floats (tf_approxmatch.cpp:167-168); this
    // implementation keeps the ratio vectors of all 10 levels so that `match` is written once.
    if (!strcmp(op, "approxmatch")) return sn_emd_workspace_floats(B, N, M) * 4;
    if (!strcmp(op, "matchcost")) return (long long)B * ((N + 255) / 256) * 4;  // per-workgroup partial sums
    if (!strcmp(op, "emd_loss"))  // level workspace + cost partials + the one-sweep form's tile partials
        return sn_emd_workspace_floats(B, N, M) * 4 + (long long)B * ((N + 255) / 256) * 4 + sn_emd_sweep2d_floats(B, N, M) * 4;
    return 0;
}

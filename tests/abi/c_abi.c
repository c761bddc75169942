/* Both public headers must be usable from plain C (the drop-in boundary is a C ABI). */
#include "gauss_hip.h"
#include "gauss_host.h"

int gauss_c_abi_check(void)
{
    gauss_window_desc w;
    gauss_ctx* ctx = 0;
    gauss_table* t = 0;
    (void)t;
    w.kind = GAUSS_WIN_IMPUTE;
    w.geno_format = GAUSS_GENO_U8;
    return gauss_hip_init(0, &ctx) + (int)sizeof(w);
}

// Error reporting and version of the C ABI.
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

void gt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gt_last_error_string(void) { return g_err; }
extern "C" int gt_version(void) { return 100; }

// sizeof of every struct that crosses the ABI, so a binding can verify its layout at load time
extern "C" int64_t gt_abi_sizeof(int32_t which) {
    switch (which) {
        case 0: return sizeof(GtOperand);
        case 1: return sizeof(GtGemm);
        case 2: return sizeof(GtSlabs);
        case 3: return sizeof(GtAttnDims);
        case 4: return sizeof(GtDecoderFwd);
        case 5: return sizeof(GtDecoderBwd);
        case 6: return sizeof(GtBiLstm);
        default: return -1;
    }
}

// Library identification for the C ABI (include/wsmgmap.h).
#include "wsmg_common.h"

// defined in wsmg_rnn.hip (hidden: not part of the ABI): 1 when that file was compiled without packed-fp32 instructions, which
// its gradient correctness rests on (Makefile: EXTRA_wsmg_rnn).  wsmgmap/_abi.py refuses a library whose build info lacks "rnn-nopk".
__attribute__((visibility("hidden"))) int wsmgi_rnn_no_pk_fp32();

extern "C" int wsmg_abi_version(void) { return 1; }
extern "C" const char* wsmg_build_info(void) {
    return wsmgi_rnn_no_pk_fp32() ? "libwsmgmap gfx950 (CDNA4) hipcc f32-mfma rnn-nopk abi1" : "libwsmgmap gfx950 (CDNA4) hipcc f32-mfma abi1";
}

// Library identification for the C ABI (include/wsmgmap.h).
#include "wsmg_common.h"

extern "C" int wsmg_abi_version(void) { return 1; }
extern "C" const char* wsmg_build_info(void) { return "libwsmgmap gfx950 (CDNA4) hipcc f32-mfma abi1"; }

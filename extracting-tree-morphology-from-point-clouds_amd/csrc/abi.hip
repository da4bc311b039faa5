// Library identification entry points of the C ABI (include/pn2_hip.h).
#include "pn2_common.h"

extern "C" int pn2_version(void) { return PN2_ABI_VERSION; }
extern "C" const char* pn2_arch(void) { return "gfx950"; }

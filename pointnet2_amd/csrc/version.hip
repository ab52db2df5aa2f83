// version.hip -- library identification for the C ABI (include/pn2ops.h).
#include "pn2_device.h"

extern "C" const char *pn2_version(void) { return "pn2ops 0.2.0 gfx950"; }

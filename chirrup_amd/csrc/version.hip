// ABI version and build target of libchirrup_amd.so.
#include "../../include/chirrup_amd.h"

extern "C" int chirrup_abi_version(void) { return 1; }
extern "C" const char *chirrup_target_arch(void) { return "gfx950"; }

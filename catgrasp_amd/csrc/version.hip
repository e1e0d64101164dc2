#include "../../include/catgrasp_amd.h"
extern "C" const char* cg_version(void) { return "catgrasp_amd 0.1 (gfx950)"; }

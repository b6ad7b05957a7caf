// Instantiates the fiber scheduler of the CPU emulation shim (TEST INFRASTRUCTURE ONLY).
#define CSN_EMU_IMPL
#include "hip_cpu_shim.h"

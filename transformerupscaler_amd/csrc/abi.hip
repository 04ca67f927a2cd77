#include <hip/hip_runtime.h>
extern "C" int tup_abi_version(void) { return 14; }

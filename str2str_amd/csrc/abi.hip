// ABI version of libstr2str_hip.so (include/str2str_hip.h).
#include "str2str_hip.h"
extern "C" int s2s_abi_version(void) { return 15; }

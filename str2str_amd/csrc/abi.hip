// ABI version of libstr2str_hip.so (include/str2str_hip.h) and the library-wide range flag (range_flag.h).
#include "range_flag.h"
#include "str2str_hip.h"

namespace s2s {
int* g_range_flag = nullptr;
}

extern "C" int s2s_abi_version(void) { return 32; }

extern "C" int s2s_set_range_flag(int* device_words) {   // kRangeWords ints (range_flag.h)
    s2s::g_range_flag = device_words;
    return 0;
}

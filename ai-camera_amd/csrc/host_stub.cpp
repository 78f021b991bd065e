// host_stub.cpp -- the two library-level symbols the HIP-free sanitizer build of the host association logic needs
// (tools/asan_host.sh: lsap.cpp + assoc_host.cpp + this file under g++ -fsanitize=address,undefined).  NOT part of libaicam.so.
#include "assoc_host.hpp"

namespace aic {
static thread_local std::string g_last_error;
void set_last_error(const std::string& m) { g_last_error = m; }
}  // namespace aic

extern "C" const char* aic_last_error(void) { return aic::g_last_error.c_str(); }
extern "C" int aic_abi_version(void) { return AIC_ABI_VERSION; }

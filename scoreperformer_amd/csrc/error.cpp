// Thread-local error string for the spn C-ABI (no C++ exceptions cross the boundary).
#include <string.h>
static thread_local char g_err[512] = "";
extern "C" void spn_set_error(const char* msg) { strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1); g_err[sizeof(g_err) - 1] = 0; }
extern "C" const char* spn_last_error(void) { return g_err; }
extern "C" int spn_abi_version(void) { return 1; }

#include "common.h"
namespace mi355 { thread_local char g_err[512] = ""; }
extern "C" {
const char* mi355det_last_error(void) { return mi355::g_err; }
int mi355det_version(void) { return 1; }
}

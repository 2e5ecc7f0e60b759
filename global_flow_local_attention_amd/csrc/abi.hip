// Library-level entry points of libgfla_hip.so: version, status strings, tuning knobs.
#include "gfla_common.h"

namespace gfla {
// per thread: a host thread that drives its own device (DataParallel-style workers, tests) tunes only its own launches
constexpr int kTuningKeys = 24;
static thread_local int g_tuning[kTuningKeys] = {0};
int tuning(int key) { return (key >= 0 && key < kTuningKeys) ? g_tuning[key] : 0; }
}  // namespace gfla

extern "C" {
int gfla_abi_version(void) { return 1; }

const char *gfla_status_string(int status) {
  switch (status) {
    case GFLA_OK: return "ok";
    case GFLA_ERR_NULL_POINTER: return "a required buffer is NULL";
    case GFLA_ERR_BAD_SHAPE: return "bad shape or kernel_size";
    case GFLA_ERR_UNSUPPORTED: return "shape outside the supported index range";
    case GFLA_ERR_LAUNCH: return "kernel launch failed (hipGetLastError)";
    default: return "unknown status";
  }
}

int gfla_set_tuning(int key, int value) {
  if (key < 0 || key >= gfla::kTuningKeys) return 0;
  int old = gfla::g_tuning[key];
  gfla::g_tuning[key] = value;
  return old;
}
}

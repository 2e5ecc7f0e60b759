// Library-level entry points of libgfla_hip.so: version, status strings, tuning knobs.
#include "gfla_common.h"

#include <atomic>

namespace gfla {
// PROCESS-global: torch runs the backward of a GPU autograd Function on the engine's per-device worker thread and
// nn.DataParallel runs replicas on worker threads, so a per-thread table would silently drop the caller's choice there.
constexpr int kTuningKeys = 64;
static std::atomic<int> g_tuning[kTuningKeys];
int tuning(int key) { return (key >= 0 && key < kTuningKeys) ? g_tuning[key].load(std::memory_order_relaxed) : 0; }

// Dispatch trace: how often each kernel path was enqueued (tests assert which path a call took, whatever thread made it).
static std::atomic<int64_t> g_path[GFLA_PATH_COUNT];
void note_path(int id) {
  if (id >= 0 && id < GFLA_PATH_COUNT) g_path[id].fetch_add(1, std::memory_order_relaxed);
}
}  // namespace gfla

extern "C" {
int gfla_abi_version(void) { return GFLA_ABI_VERSION; }

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
  return gfla::g_tuning[key].exchange(value, std::memory_order_relaxed);
}

int64_t gfla_path_count(int path) {
  return (path >= 0 && path < GFLA_PATH_COUNT) ? gfla::g_path[path].load(std::memory_order_relaxed) : -1;
}
}

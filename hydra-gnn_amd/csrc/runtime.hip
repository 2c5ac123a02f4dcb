// Library plumbing: error string, device probe, hipGraph capture, HIP-event timers.
#include "common.h"

namespace hmp {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace hmp

struct hmp_graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
};

struct hmp_timer {
  hipEvent_t a, b;
};

extern "C" {

int hmp_abi_version(void) { return HMP_ABI_VERSION; }

const char* hmp_last_error(void) { return hmp::err_buf(); }

/* struct sizes, so a binding can verify that its mirror of include/hydra_mp.h has the same layout */
size_t hmp_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(hmp_plan);
    case 1: return sizeof(hmp_gat_args);
    case 2: return sizeof(hmp_conv_spec);
    case 3: return sizeof(hmp_layer_spec);
    case 4: return sizeof(hmp_net_spec);
    case 5: return sizeof(hmp_batch);
    case 6: return sizeof(hmp_train_args);
    default: return 0;
  }
}

int hmp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  int ok = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
  }
  return ok;
}

int hmp_graph_begin(void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(stream != nullptr, "hmp_graph_begin: capture needs a non-null stream");
  HMP_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return HMP_OK;
}

int hmp_graph_end(void* stream, hmp_graph** out) {
  using namespace hmp;
  HMP_CHECK_ARG(stream != nullptr && out != nullptr, "hmp_graph_end: null argument");
  hipGraph_t g = nullptr;
  HMP_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
  hipGraphExec_t e = nullptr;
  hipError_t err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  if (err != hipSuccess) {
    hipGraphDestroy(g);
    HMP_FAIL(HMP_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(err));
  }
  hmp_graph* h = new hmp_graph{g, e};
  *out = h;
  return HMP_OK;
}

int hmp_graph_launch(hmp_graph* g, void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(g != nullptr, "hmp_graph_launch: null graph");
  HMP_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
  return HMP_OK;
}

void hmp_graph_destroy(hmp_graph* g) {
  if (!g) return;
  hipGraphExecDestroy(g->exec);
  hipGraphDestroy(g->graph);
  delete g;
}

int hmp_timer_create(hmp_timer** out) {
  using namespace hmp;
  HMP_CHECK_ARG(out != nullptr, "hmp_timer_create: null");
  hmp_timer* t = new hmp_timer;
  hipError_t e1 = hipEventCreate(&t->a), e2 = hipEventCreate(&t->b);
  if (e1 != hipSuccess || e2 != hipSuccess) {
    delete t;
    HMP_FAIL(HMP_E_HIP, "hipEventCreate failed");
  }
  *out = t;
  return HMP_OK;
}
int hmp_timer_start(hmp_timer* t, void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(t, "null timer");
  HMP_HIP(hipEventRecord(t->a, (hipStream_t)stream));
  return HMP_OK;
}
int hmp_timer_stop(hmp_timer* t, void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(t, "null timer");
  HMP_HIP(hipEventRecord(t->b, (hipStream_t)stream));
  return HMP_OK;
}
int hmp_timer_elapsed_ms(hmp_timer* t, float* ms) {
  using namespace hmp;
  HMP_CHECK_ARG(t && ms, "null timer");
  HMP_HIP(hipEventSynchronize(t->b));
  HMP_HIP(hipEventElapsedTime(ms, t->a, t->b));
  return HMP_OK;
}
void hmp_timer_destroy(hmp_timer* t) {
  if (!t) return;
  hipEventDestroy(t->a);
  hipEventDestroy(t->b);
  delete t;
}

}  // extern "C"

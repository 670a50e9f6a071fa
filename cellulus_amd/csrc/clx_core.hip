// Error reporting, version and device probing for libclx.
#include "clx_common.h"

static thread_local char g_err[512] = "";

void clx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* clx_last_error(void) { return g_err; }
extern "C" int clx_abi_version(void) { return 13; }
extern "C" int clx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// ---------------------------------------------------------------------------
// Optional in-library kernel timing (bench.py roofline): when enabled, the MFMA kernel
// launchers attach a HIP event pair to their launch (start / end of the kernel on its stream) and record the
// FLOPs the launch executes.  Off by default; no cost when off.
// ---------------------------------------------------------------------------
#include <vector>

namespace {
struct ProfRec { int kind; double flops; hipEvent_t e0, e1; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
}  // namespace

bool clx_prof_enabled() { return g_prof_on; }

// a launch's event pair: handed to hipExtLaunchKernelGGL, which stamps them with the kernel's own start and end
// (no separate event packets in the stream: bracketing hipEventRecord calls cost the step ~5 us per launch)
void clx_prof_events(int kind, double flops, hipEvent_t* e0, hipEvent_t* e1) {
  *e0 = *e1 = nullptr;
  ProfRec r;
  r.kind = kind; r.flops = flops;
  if (hipEventCreate(&r.e0) != hipSuccess) return;
  if (hipEventCreate(&r.e1) != hipSuccess) { (void)hipEventDestroy(r.e0); return; }
  g_prof.push_back(r);
  *e0 = r.e0; *e1 = r.e1;
}

extern "C" int clx_profile_enable(int on) {
  g_prof_on = on != 0;
  if (!g_prof_on || on == 2) {   // 2 = enable and clear
    for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof.clear();
  }
  return CLX_OK;
}

extern "C" int clx_profile_read(int kind, double* launches, double* total_ms, double* total_flops) {
  CLX_REQUIRE(launches && total_ms && total_flops, "clx_profile_read: null pointer");
  *launches = *total_ms = *total_flops = 0.0;
  for (auto& r : g_prof) {
    if (r.kind != kind) continue;
    if (hipEventSynchronize(r.e1) != hipSuccess) { clx_set_error("clx_profile_read: event sync failed"); return CLX_ERR_LAUNCH; }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
    *launches += 1.0; *total_ms += ms; *total_flops += r.flops;
  }
  return CLX_OK;
}

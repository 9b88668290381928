// Error reporting, build info and the event-based profiling hooks of libcocodr_hip.so.
#include <stdarg.h>

#include <vector>

#include "common.h"
#include "prof.h"

static thread_local char g_err[512] = "";

void cocodr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* cocodr_last_error(void) { return g_err; }
extern "C" const char* cocodr_build_info(void) { return "cocodr_hip gfx950 (MI355X/CDNA4) " __DATE__ " " __TIME__; }

int g_prof_kind = PROF_OFF;
int g_prof_paused = 0;
namespace {
struct ProfState {
  std::vector<hipEvent_t> begin, end;
  size_t used = 0;
  double flops = 0.0;
} g_ps;
}  // namespace

void prof_record(hipStream_t st, double flops, bool begin) {
  if (begin) {
    if (g_ps.used == g_ps.begin.size()) {
      hipEvent_t b, e;
      hipEventCreate(&b);
      hipEventCreate(&e);
      g_ps.begin.push_back(b);
      g_ps.end.push_back(e);
    }
    g_ps.flops += flops;
    hipEventRecord(g_ps.begin[g_ps.used], st);
  } else {
    hipEventRecord(g_ps.end[g_ps.used], st);
    g_ps.used++;
  }
}

extern "C" int cocodr_prof_begin(int kind) {
  CK_ARG(kind >= PROF_OFF && kind <= PROF_SCORE, "prof_begin: bad kind %d", kind);
  g_ps.used = 0;
  g_ps.flops = 0.0;
  g_prof_kind = kind;
  g_prof_paused = 0;
  return COCODR_OK;
}

extern "C" int cocodr_prof_pause(int paused) {
  g_prof_paused = paused != 0;
  return COCODR_OK;
}

extern "C" int cocodr_prof_end(int* launches, double* total_ms, double* total_flops) {
  g_prof_kind = PROF_OFF;
  if (hipDeviceSynchronize() != hipSuccess) {
    cocodr_set_error("prof_end: device synchronize failed");
    return COCODR_ERR_LAUNCH;
  }
  double ms = 0.0;
  for (size_t i = 0; i < g_ps.used; ++i) {
    float t = 0.f;
    hipEventElapsedTime(&t, g_ps.begin[i], g_ps.end[i]);
    ms += t;
  }
  if (launches) *launches = (int)g_ps.used;
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = g_ps.flops;
  g_ps.used = 0;
  return COCODR_OK;
}

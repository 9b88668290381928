// Optional per-kernel-class timing with HIP events on the launch stream (bench.py roofline leg).
#pragma once
#include <hip/hip_runtime.h>

enum { PROF_OFF = 0, PROF_GEMM = 1, PROF_ATTN = 2, PROF_SCORE = 3 };

extern int g_prof_kind;
extern int g_prof_paused;
void prof_record(hipStream_t st, double flops, bool begin);

struct ProfScope {
  bool on;
  hipStream_t st;
  ProfScope(int kind, hipStream_t s, double flops) : on(kind == g_prof_kind && !g_prof_paused), st(s) {
    if (on) prof_record(st, flops, true);
  }
  ~ProfScope() {
    if (on) prof_record(st, 0.0, false);
  }
};

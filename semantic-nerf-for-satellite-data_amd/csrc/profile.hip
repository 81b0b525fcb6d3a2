// Optional per-launch timing (snerf_profile_begin / _end): HIP events on the launch stream around every GEMM launch, summed
// per kernel family.  bench.py's roofline phase reads it; nothing here runs unless profiling was switched on.
#include "common.h"
#include "gemm.h"

#include <vector>

namespace snerf {

struct ProfRec { hipEvent_t a, b; double flops; int variant; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_prof_pool;
static size_t g_prof_used = 0;

static hipEvent_t prof_event() {
  if (g_prof_used == g_prof_pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    g_prof_pool.push_back(e);
  }
  return g_prof_pool[g_prof_used++];
}

// bracket of one launch: token >= 0 while profiling is on
int prof_hook_begin(double flops, int variant, hipStream_t st) {
  if (!g_prof_on || variant < 0 || variant >= SNERF_PROFILE_VARIANTS) return -1;
  ProfRec rec{prof_event(), prof_event(), flops, variant};
  if (!rec.a || !rec.b) return -1;
  (void)hipEventRecord(rec.a, st);
  g_prof.push_back(rec);
  return (int)g_prof.size() - 1;
}
void prof_hook_end(int token, hipStream_t st) {
  if (token >= 0) (void)hipEventRecord(g_prof[token].b, st);
}
namespace bsp {
int prof_hook_begin(double flops, int variant, hipStream_t st) { return snerf::prof_hook_begin(flops, variant, st); }
void prof_hook_end(int token, hipStream_t st) { snerf::prof_hook_end(token, st); }
}  // namespace bsp

int profile_begin() {
  g_prof.clear();
  g_prof_used = 0;
  g_prof_on = true;
  return SNERF_OK;
}

int profile_end(SnerfProfile* out) {
  g_prof_on = false;
  if (!out) { set_error("snerf_profile_end: null output"); return SNERF_ERR_NULL; }
  for (int v = 0; v < SNERF_PROFILE_VARIANTS; ++v) { out->ms[v] = 0.0; out->flops[v] = 0.0; out->launches[v] = 0; }
  for (const ProfRec& r : g_prof) {
    SNERF_HIP_CHECK(hipEventSynchronize(r.b));
    float ms = 0.f;
    SNERF_HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
    out->ms[r.variant] += ms;
    out->flops[r.variant] += r.flops;
    out->launches[r.variant] += 1;
  }
  g_prof.clear();
  g_prof_used = 0;
  return SNERF_OK;
}

}  // namespace snerf

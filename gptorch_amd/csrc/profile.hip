// Optional per-launch timing of the MFMA contraction kernel with HIP events
// recorded on the launch stream (used by bench.py's roofline leg; off by default).
#include <vector>
#include "gpn_common.h"

#include <mutex>

namespace gpn {

// One record per profiled launch.  `cls` (gpn_common.h PROF_*) says what the launch was, so that
// bench.py can price each kernel family against its own roofline; `work` = executed flops for the
// contraction classes, algorithmic bytes for the HBM-bound ones.
struct Rec { hipEvent_t a, b; double work; int cls; };
static bool g_on = false;
static std::mutex g_mutex;               // launches may come from several host threads
static std::vector<Rec> g_recs;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

bool profile_on() { return g_on; }
// returns the record index for profile_end (records of other threads may interleave)
int profile_begin(hipStream_t s, double work, int cls) {
  std::lock_guard<std::mutex> lock(g_mutex);
  Rec r{get_event(), get_event(), work, cls};
  (void)hipEventRecord(r.a, s);
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}
void profile_end(hipStream_t s, int idx) {
  std::lock_guard<std::mutex> lock(g_mutex);
  if (idx >= 0 && idx < (int)g_recs.size()) (void)hipEventRecord(g_recs[idx].b, s);
}

static int collect(double* out, int nclasses) {
  std::lock_guard<std::mutex> lock(g_mutex);
  for (int i = 0; i < 3 * nclasses; ++i) out[i] = 0.0;
  for (auto& r : g_recs) {
    float t = 0.f;
    GPN_HIP_CHECK(hipEventSynchronize(r.b));
    GPN_HIP_CHECK(hipEventElapsedTime(&t, r.a, r.b));
    const int c = (r.cls >= 0 && r.cls < nclasses) ? r.cls : -1;
    if (c >= 0) { out[3 * c] += 1.0; out[3 * c + 1] += t; out[3 * c + 2] += r.work; }
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  return GPN_OK;
}

}  // namespace gpn

extern "C" int gpn_profile_enable(int on) {
  gpn::g_on = on != 0;
  return GPN_OK;
}

// Synchronises every recorded event; out[3c + 0] = launches, out[3c + 1] = total ms,
// out[3c + 2] = total work of class c (PROF_* in gpn_common.h: executed flops for the
// contraction classes -- 2*M*N*K per launch, lower-tile launches count the tiles on/below the
// diagonal only -- algorithmic bytes for K assembly and the gradient sweep).  Clears the list.
extern "C" int gpn_profile_collect_classes(double* out_host, int nclasses) {
  if (!out_host) return -1;
  if (nclasses < 1 || nclasses > 64) return -2;
  return gpn::collect(out_host, nclasses);
}

// all contraction launches together: out[0] = launches, out[1] = ms, out[2] = executed flops
extern "C" int gpn_profile_collect(double* out3_host) {
  double c[3 * gpn::PROF_NCLASSES];
  const int rc = gpn::collect(c, gpn::PROF_NCLASSES);
  if (rc != GPN_OK) return rc;
  out3_host[0] = out3_host[1] = out3_host[2] = 0.0;
  for (int k = gpn::PROF_GEMM; k <= gpn::PROF_GEMM_TRI; ++k)
    for (int j = 0; j < 3; ++j) out3_host[j] += c[3 * k + j];
  return GPN_OK;
}

// Experimental (tools' build): a stream restricted to a subset of the 256 CUs (hipExtStreamCreateWithCUMask).
// mask_words = 8 x uint32 (bit i = CU i enabled).  Returns the hipStream_t through *out.
GPN_DEBUG_ONLY(
extern "C" int gpn_debug_masked_stream(const uint32_t* mask_words, int nwords, void** out) {
  hipStream_t s;
  GPN_HIP_CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask_words));
  *out = s;
  return GPN_OK;
})

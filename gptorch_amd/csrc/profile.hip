// Optional per-launch timing of the MFMA contraction kernel with HIP events
// recorded on the launch stream (used by bench.py's roofline leg; off by default).
#include <vector>
#include "gpn_common.h"

namespace gpn {

struct Rec { hipEvent_t a, b; double flops; };
static bool g_on = false;
static std::vector<Rec> g_recs;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

bool profile_on() { return g_on; }
void profile_begin(hipStream_t s, double flops) {
  Rec r{get_event(), get_event(), flops};
  (void)hipEventRecord(r.a, s);
  g_recs.push_back(r);
}
void profile_end(hipStream_t s) { (void)hipEventRecord(g_recs.back().b, s); }

}  // namespace gpn

extern "C" int gpn_profile_enable(int on) {
  gpn::g_on = on != 0;
  return GPN_OK;
}

// Synchronises every recorded event; out[0] = launches, out[1] = total ms,
// out[2] = total executed flops (2*M*N*K per launch; lower-tile launches count
// the tiles on/below the diagonal only).  Clears the record list.
extern "C" int gpn_profile_collect(double* out3_host) {
  using namespace gpn;
  double ms = 0.0, fl = 0.0;
  for (auto& r : g_recs) {
    float t = 0.f;
    GPN_HIP_CHECK(hipEventSynchronize(r.b));
    GPN_HIP_CHECK(hipEventElapsedTime(&t, r.a, r.b));
    ms += t;
    fl += r.flops;
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  out3_host[0] = (double)g_recs.size();
  out3_host[1] = ms;
  out3_host[2] = fl;
  g_recs.clear();
  return GPN_OK;
}

// Experimental: a stream restricted to a subset of the 256 CUs (hipExtStreamCreateWithCUMask).
// mask_words = 8 x uint32 (bit i = CU i enabled).  Returns the hipStream_t through *out.
extern "C" int gpn_debug_masked_stream(const uint32_t* mask_words, int nwords, void** out) {
  hipStream_t s;
  GPN_HIP_CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)nwords, mask_words));
  *out = s;
  return GPN_OK;
}

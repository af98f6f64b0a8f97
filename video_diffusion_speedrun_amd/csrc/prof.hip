// Event-pair recorder behind vds_prof_enable / vds_prof_collect (include/vds.h).
#include "prof.h"
#include "../../include/vds.h"
#include <vector>

namespace vdsprof {
unsigned g_mask = 0;
namespace {
struct Rec { int cls; hipEvent_t a, b; double flops, bytes; };
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace
void begin(int cls, hipStream_t s, double flops, double bytes) {
  Rec r{cls, get_event(), get_event(), flops, bytes};
  (void)hipEventRecord(r.a, s);
  g_recs.push_back(r);
}
void end(hipStream_t s) { (void)hipEventRecord(g_recs.back().b, s); }
}  // namespace vdsprof

static const char* kNames[VDS_PROF_NCLASS] = {"gemm_nt", "gemm_nn", "gemm_tn", "attn_fwd", "attn_bwd_delta",
                                              "attn_bwd_dkv", "attn_bwd_dq", "rmsnorm_mod_fwd", "rmsnorm_mod_bwd",
                                              "adamw", "qkv_rope_fwd", "qkv_rope_bwd", "gate_bwd",
                                              "attn_fwd_plain", "attn_bwd_dkv_plain", "attn_bwd_dq_plain", "gemm_fp8",
                                              "attn_fp8_fwd", "attn_fp8_dkv", "attn_fp8_dq", "fp8_quant"};

extern "C" const char* vds_prof_class_name(int cls) { return (cls >= 0 && cls < VDS_PROF_NCLASS) ? kNames[cls] : ""; }

extern "C" int vds_prof_enable(uint32_t class_mask) {
  using namespace vdsprof;
  for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
  g_recs.clear();
  g_mask = class_mask;
  return VDS_OK;
}

extern "C" int vds_prof_collect(vds_prof_stat* out) {
  using namespace vdsprof;
  if (!out) return VDS_ERR_ARG;
  for (int i = 0; i < VDS_PROF_NCLASS; ++i) out[i] = vds_prof_stat{0, 0.0, 0.0, 0.0};
  for (auto& r : g_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) return VDS_ERR_LAUNCH;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return VDS_ERR_LAUNCH;
    out[r.cls].launches += 1;
    out[r.cls].ms += ms;
    out[r.cls].flops += r.flops;
    out[r.cls].bytes += r.bytes;
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  return VDS_OK;
}

// The knob table of config.h: names, defaults, and the ONE place where the library reads its environment.
#include "config.h"
#include "common.h"
#include "../../include/vds.h"
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace vdscfg {
namespace {
struct Def { const char* name; const char* env; double def; };
const Def kDefs[N_KNOBS] = {
    {"gemm_tile", "VDS_GEMM_TILE", 0},
    {"gemm_narrow", "VDS_GEMM_NARROW", 1},
    {"gemm_group_m", "VDS_GEMM_GROUP_M", 4},
    {"gemm_group_m_tn", "VDS_GEMM_GROUP_M_TN", 0},
    {"gemm_tn_joint", "VDS_GEMM_TN_JOINT", 1},
    {"gemm_mid_tn", "VDS_GEMM_MID_TN", 0.75},
    {"gemm_mid_factor", "VDS_GEMM_MID_FACTOR", 1.3},
    {"attn_mfma16", "VDS_ATTN_MFMA16", 7},
    {"attn_delta_fold", "VDS_ATTN_DELTA_FOLD", 1},
    {"attn_tail_last", "VDS_ATTN_TAIL_LAST", 1},
    {"attn_fwd_wide", "VDS_ATTN_FWD_WIDE", 2},
    {"attn_wide_stores", "VDS_ATTN_WIDE_STORES", 1},
    {"attn_qsplit", "VDS_ATTN_QSPLIT", 0},
    {"cross_dkv16", "VDS_CROSS_DKV16", 1},
    {"attn8_dq_waves", "VDS_ATTN8_DQ_WAVES", 4},
    {"ew_min_rows", "VDS_EW_MIN_ROWS", 8},
    {"ew_wgs", "VDS_EW_WGS", 768},
    {"rmsnorm_q4", "VDS_RMSNORM_Q4", 1},
    {"rope_tok", "VDS_ROPE_TOK", 1},
    {"rope_tile", "VDS_ROPE_TILE", 4},
    {"adaln_mfma", "VDS_ADALN_MFMA", 1},
    {"comm_allpairs", nullptr, 0},
    {"deterministic", nullptr, 0},
};
const char* g_rccl_path = nullptr;
const char* env(const char* name) { return name ? getenv(name) : nullptr; }
struct Init {
  Init() {
    for (int i = 0; i < N_KNOBS; ++i) {
      const char* e = env(kDefs[i].env);
      g_val[i] = (e && *e) ? atof(e) : kDefs[i].def;
    }
    const char* s = env("VDS_COMM_SCHEDULE");
    if (s && strcmp(s, "allpairs") == 0) g_val[COMM_ALLPAIRS] = 1;
    g_rccl_path = env("VDS_RCCL_PATH");
  }
};
}  // namespace
double g_val[N_KNOBS];
static Init g_init;  // runs when the library is loaded
const char* rccl_path() { return g_rccl_path; }
}  // namespace vdscfg

namespace vdsdet {
static float* g_ws = nullptr;
static size_t g_ws_bytes = 0;
float* workspace(size_t floats, const char* who) {
  if (g_ws && floats * 4 <= g_ws_bytes) return g_ws;
  vdserr::set("%s: deterministic mode needs %zu bytes of workspace, vds_set_deterministic was given %zu", who, floats * 4,
              g_ws_bytes);
  return nullptr;
}
size_t workspace_bytes() { return g_ws ? g_ws_bytes : 0; }
}  // namespace vdsdet

extern "C" int vds_set_deterministic(int32_t on, void* workspace, size_t workspace_bytes) {
  const int prev = vdscfg::geti(vdscfg::DETERMINISTIC);
  if (on != 0 && on != 1) return VDS_ERR_ARG;
  if (on && (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < 4096)) return VDS_ERR_ARG;
  vdscfg::g_val[vdscfg::DETERMINISTIC] = on;
  vdsdet::g_ws = on ? (float*)workspace : nullptr;
  vdsdet::g_ws_bytes = on ? workspace_bytes : 0;
  return prev;
}

extern "C" int vds_knob_set(const char* name, double value) {
  if (!name) return VDS_ERR_ARG;
  for (int i = 0; i < vdscfg::N_KNOBS; ++i)
    if (strcmp(name, vdscfg::kDefs[i].name) == 0) {
      if (i == vdscfg::DETERMINISTIC) return VDS_ERR_ARG;  // (needs its workspace: vds_set_deterministic)
      vdscfg::g_val[i] = value;
      return VDS_OK;
    }
  return VDS_ERR_ARG;
}

extern "C" double vds_knob_get(const char* name) {
  if (name)
    for (int i = 0; i < vdscfg::N_KNOBS; ++i)
      if (strcmp(name, vdscfg::kDefs[i].name) == 0) return vdscfg::g_val[i];
  return NAN;
}

// Every tuning knob of the library in ONE table, filled once from the environment when the library is loaded (a static
// initialiser: no `getenv` on any launch path) and changeable afterwards only through the C ABI
// (vds_knob_set / vds_knob_get, include/vds.h) -- explicit, process-wide, documented state instead of hidden reads.
// Defaults are the measured-best settings; every knob exists for same-process A/B measurements and tests.
#pragma once
#include <cstddef>

namespace vdscfg {
enum Knob {
  // ---- GEMM (csrc/gemm.hip)
  GEMM_TILE,         // VDS_GEMM_TILE: 0 = cost model (default) | 128 | 256 | 2 (= 256 x 128)
  GEMM_NARROW,       // VDS_GEMM_NARROW: a last tile column of <= 128 columns runs the 256 x 128 body (default 1)
  GEMM_GROUP_M,      // VDS_GEMM_GROUP_M: row tiles per group of the tile order (default 4)
  GEMM_GROUP_M_TN,   // VDS_GEMM_GROUP_M_TN: the same for weight gradients; 0 = sized per launch (default)
  GEMM_TN_JOINT,     // VDS_GEMM_TN_JOINT: XCD-aware order over the joint (split, tile) list (default 1)
  GEMM_MID_TN,       // VDS_GEMM_MID_TN: time per unit of work of the 256 x 128 kernel on TN problems (default 0.75)
  GEMM_MID_FACTOR,   // VDS_GEMM_MID_FACTOR: price of a round of 256 x 128 tiles in rounds of 256^2 tiles (default 1.3)
  // ---- attention (csrc/attention.hip, csrc/attention_fp8.hip)
  ATTN_MFMA16,       // VDS_ATTN_MFMA16: bit mask, which head-dim-72 kernels run on v_mfma_f32_16x16x32_bf16 (default 7)
  ATTN_DELTA_FOLD,   // VDS_ATTN_DELTA_FOLD: the head-dim-72 dQ kernel computes -delta / lse2 itself and runs before dK/dV (default 1)
  ATTN_TAIL_LAST,    // VDS_ATTN_TAIL_LAST: the ragged tile of every head is scheduled after all full tiles (default 1)
  ATTN_FWD_WIDE,     // VDS_ATTN_FWD_WIDE: 64 queries per wave in the forward: 0 never | 1 always | 2 from 2048 queries (default)
  ATTN_WIDE_STORES,  // VDS_ATTN_WIDE_STORES: 16-byte stores of the 16x16 epilogues (default 1)
  ATTN_QSPLIT,       // VDS_ATTN_QSPLIT: 0 = rule (default) | 1 = off | N = force N query splits of the plain dK/dV kernel
  CROSS_DKV16,       // VDS_CROSS_DKV16: cross-attention dK/dV on the 16x16x32 kernel: 0 never | 1 when the grid fills the chip
                     // (default) | 2 always
  ATTN8_DQ_WAVES,    // VDS_ATTN8_DQ_WAVES: waves per workgroup of the fp8 dQ kernel: 4 (default) | 6
  // ---- row kernels (csrc/elementwise.hip)
  EW_MIN_ROWS,       // VDS_EW_MIN_ROWS: floor of the rows per workgroup of the row kernels (default 8: 8, and 32 from 8192 rows)
  EW_WGS,            // VDS_EW_WGS: target workgroup count of the row kernels (default 768)
  RMSNORM_Q4,        // VDS_RMSNORM_Q4: four rows per wave in rmsnorm_mod_fwd at D = 384 / 768 / 1152 (default 1)
  ROPE_TOK,          // VDS_ROPE_TOK: token-tile form of qkv_rope_bwd (default 1)
  ROPE_TILE,         // VDS_ROPE_TILE: tokens per workgroup of the token-tile qkv_rope_fwd: 0 = element-wise kernel | 2 | 4 (default) | 8
  ADALN_MFMA,        // VDS_ADALN_MFMA: MFMA form of the small (adaLN) linear forward (default 1)
  // ---- communication (csrc/comm.hip)
  COMM_ALLPAIRS,     // VDS_COMM_SCHEDULE=allpairs -> 1: all-pairs reduce-scatter instead of RCCL's (default 0)
  // ---- run-time only (no environment variable)
  DETERMINISTIC,     // vds_set_deterministic: fixed-order reductions everywhere (default 0)
  N_KNOBS
};
extern double g_val[N_KNOBS];
inline double get(Knob k) { return g_val[k]; }
inline int geti(Knob k) { return (int)g_val[k]; }
// VDS_RCCL_PATH (a string: not in the table), or nullptr
const char* rccl_path();
}  // namespace vdscfg

// deterministic mode (vds_set_deterministic): the caller's workspace for partial results
namespace vdsdet {
inline bool on() { return vdscfg::g_val[vdscfg::DETERMINISTIC] != 0; }
// the workspace if it holds `floats` floats, else nullptr (and the error message is set)
float* workspace(size_t floats, const char* who);
size_t workspace_bytes();
}  // namespace vdsdet

"""ctypes binding of libvds_hip.so (the C ABI declared in include/vds.h).

There is deliberately NO fallback: if the HIP library is missing or a kernel returns an
error this raises, it never routes to a CPU / eager-torch path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VDS_LIB_PATH") or os.path.join(_HERE, "libvds_hip.so")  # override: A/B of kernel builds

VDS_NT, VDS_NN, VDS_TN = 0, 1, 2
EPI_STORE, EPI_BIAS_GELU, EPI_GATE_RES, EPI_DGELU, EPI_F32 = 0, 1, 2, 3, 4

c_i32, c_i64, c_f32, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [("layout", c_i32), ("epilogue", c_i32), ("M", c_i32), ("N", c_i32), ("K", c_i32),
                ("A", c_vp), ("lda", c_i64), ("B", c_vp), ("ldb", c_i64), ("C", c_vp), ("ldc", c_i64),
                ("C2", c_vp), ("ldc2", c_i64), ("bias", c_vp), ("aux", c_vp), ("ldaux", c_i64),
                ("gate", c_vp), ("ldgate", c_i64), ("rows_per_batch", c_i32), ("split_k", c_i32), ("colsum", c_vp)]


class AttnArgs(C.Structure):
    _fields_ = [("B", c_i32), ("H", c_i32), ("Lq", c_i32), ("Lk", c_i32), ("head_dim", c_i32),
                ("q", c_vp), ("q_sb", c_i64), ("q_sh", c_i64), ("q_sl", c_i64),
                ("k", c_vp), ("k_sb", c_i64), ("k_sh", c_i64), ("k_sl", c_i64),
                ("v", c_vp), ("v_sb", c_i64), ("v_sh", c_i64), ("v_sl", c_i64),
                ("o", c_vp), ("o_sb", c_i64), ("o_sh", c_i64), ("o_sl", c_i64),
                ("lse", c_vp),
                ("d_o", c_vp), ("do_sb", c_i64), ("do_sh", c_i64), ("do_sl", c_i64),
                ("dq", c_vp), ("dq_sb", c_i64), ("dq_sh", c_i64), ("dq_sl", c_i64),
                ("dk", c_vp), ("dk_sb", c_i64), ("dk_sh", c_i64), ("dk_sl", c_i64),
                ("dv", c_vp), ("dv_sb", c_i64), ("dv_sh", c_i64), ("dv_sl", c_i64),
                ("delta", c_vp), ("kv_pad_ones", c_i32), ("ws_floats", c_i64)]


class Attn8Args(C.Structure):  # vds_attn_fp8_args
    _fields_ = [("B", c_i32), ("H", c_i32), ("Lq", c_i32), ("Lk", c_i32), ("head_dim", c_i32),
                ("q", c_vp), ("k", c_vp), ("v", c_vp),
                ("o", c_vp), ("o_sb", c_i64), ("o_sh", c_i64), ("o_sl", c_i64),
                ("lse", c_vp), ("d_o", c_vp),
                ("dq", c_vp), ("dq_sb", c_i64), ("dq_sh", c_i64), ("dq_sl", c_i64),
                ("dk", c_vp), ("dk_sb", c_i64), ("dk_sh", c_i64), ("dk_sl", c_i64),
                ("dv", c_vp), ("dv_sb", c_i64), ("dv_sh", c_i64), ("dv_sl", c_i64),
                ("stats", c_vp), ("deq", c_vp),
                ("o_q", c_vp), ("o_q_ld", c_i64), ("dq_q", c_vp), ("dq_q_ld", c_i64),
                ("e_amax_prev", c_vp), ("e_amax_cur", c_vp), ("e_dq_out", c_vp)]


class Fp8Out(C.Structure):  # vds_fp8_out
    _fields_ = [("q", C.c_void_p), ("ldq", C.c_int64), ("qt", C.c_void_p), ("ldqt", C.c_int64),
                ("amax_in", C.c_void_p), ("amax_out", C.c_void_p), ("dq_out", C.c_void_p), ("fmt", C.c_int32),
                ("colsum", C.c_void_p)]


class ProfStat(C.Structure):
    _fields_ = [("launches", c_i64), ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


PROF_NCLASS = 21


class AdamWTensor(C.Structure):
    _fields_ = [("p", c_vp), ("g", c_vp), ("m", c_vp), ("v", c_vp), ("p_bf16", c_vp),
                ("numel", c_i64), ("lr", c_f32), ("wd", c_f32)]


# name -> argtypes (restype is int unless noted).  Must list every symbol of include/vds.h.
SIGNATURES = {
    "vds_version": [],
    "vds_last_error": [],
    "vds_gemm_bf16": [C.POINTER(GemmArgs), c_vp],
    "vds_gemm_force_tile": [c_i32],
    "vds_knob_set": [C.c_char_p, C.c_double],
    "vds_knob_get": [C.c_char_p],
    "vds_set_deterministic": [c_i32, c_vp, C.c_size_t],
    "vds_attn_fwd": [C.POINTER(AttnArgs), c_vp],
    "vds_attn_bwd": [C.POINTER(AttnArgs), c_vp],
    "vds_attn_set_variant": [c_i32],
    "vds_kv_pad_ones": [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "vds_dv0_reduce": [c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "vds_attn_bwd_workspace_bytes": [C.POINTER(AttnArgs)],
    "vds_rmsnorm_mod_fwd": [c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_i32, c_i32, c_i32,
                            c_f32, c_vp],
    "vds_rmsnorm_mod_bwd": [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp,
                            c_i64, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp],
    "vds_gate_bwd": [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_i64, c_vp, c_vp, c_i32, c_i32, c_i32,
                     c_vp],
    "vds_colsum_bf16": [c_vp, c_i64, c_vp, c_i32, c_i32, c_vp],
    "vds_colsum_bf16_rows": [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp],
    "vds_qkv_rope_fwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "vds_qkv_rope_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32,
                         c_i32, c_i32, c_i32, c_i32, c_vp],
    "vds_rope_apply": [c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32,
                       c_i32, c_vp],
    "vds_rope_rows": [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp,
                      c_vp, c_vp],
    "vds_rope_rows_dev": [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp],
    "vds_small_linear_fwd": [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp],
    "vds_small_linear_bwd": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp],
    "vds_small_linear_fwd_batched": [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "vds_small_linear_bwd_batched": [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "vds_timestep_embedding": [c_vp, c_vp, c_i32, c_i32, c_vp],
    "vds_patchify": [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "vds_unpatchify": [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "vds_unpatchify_bwd": [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "vds_patchify_rows": [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "vds_unpatchify_rows": [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "vds_fill_registers": [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp],
    "vds_registers_bwd": [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_vp],
    "vds_noise_latents": [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_vp],
    "vds_flow_loss_workspace_floats": [c_i32, c_i64],
    "vds_flow_loss": [c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_i64, c_vp, c_vp],
    "vds_flow_loss_bwd": [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_vp],
    "vds_cfg_euler_step": [c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_i64, c_vp],
    "vds_adamw_multi": [c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_f32, c_f32, c_i32, c_f32, c_f32, c_vp],
    "vds_adamw_multi_dev": [c_vp, c_vp, c_vp, c_i32, c_i32, c_f32, c_f32, c_f32, c_vp, c_f32, c_vp],
    "vds_comm_available": [],
    "vds_comm_unique_id": [c_vp, C.c_size_t],
    "vds_comm_init": [c_i32, c_i32, c_vp, C.c_size_t],
    "vds_comm_info": [c_vp, c_vp, c_vp, c_vp],
    "vds_comm_destroy": [],
    "vds_all_gather_bf16": [c_vp, c_vp, c_i64, c_vp],
    "vds_all_gather_f32": [c_vp, c_vp, c_i64, c_vp],
    "vds_reduce_scatter_workspace_bytes": [c_i64],
    "vds_reduce_scatter_f32_avg": [c_vp, c_vp, c_i64, c_vp, C.c_size_t, c_vp],
    "vds_average_chunks_f32": [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp],
    "vds_all_reduce_f32_avg": [c_vp, c_i64, c_vp],
    "vds_cast_f32_bf16": [c_vp, c_vp, c_i64, c_vp],
    "vds_cast_bf16_f32": [c_vp, c_vp, c_i64, c_vp],
    "vds_gemm_fp8": [C.POINTER(GemmArgs), c_vp, c_vp, c_i32, c_i32, c_vp, c_vp],
    "vds_absmax": [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp],
    "vds_quant_fp8": [c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp],
    "vds_transpose_fp8": [c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp],
    "vds_rmsnorm_mod_fwd_fp8": [c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp,
                                c_vp, c_i32, c_i32, c_i32, c_f32, c_vp],
    "vds_gate_bwd_fp8": [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp,
                         c_vp, c_i32, c_i32, c_i32, c_vp],
    "vds_qkv_rope_bwd_fp8": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp,
                             c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "vds_attn_fp8_supported": [c_i32],
    "vds_attn_fp8_fwd": [C.POINTER(Attn8Args), c_vp],
    "vds_attn_fp8_bwd_workspace_bytes": [C.POINTER(Attn8Args)],
    "vds_attn_fp8_delta": [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32,
                           c_i32, c_i32, c_vp],
    "vds_attn_fp8_bwd": [C.POINTER(Attn8Args), c_vp],
    "vds_qkv_rope_fwd_fp8": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32,
                             c_i32, c_i32, c_i32, c_i32, c_vp],
    "vds_cross_qkv_fp8": [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp],
    "vds_selftest_lanemaps": [c_vp, c_vp],
    "vds_prof_enable": [C.c_uint32],
    "vds_prof_collect": [c_vp],
    "vds_prof_class_name": [c_i32],
}

_lib = None


class VdsError(RuntimeError):
    pass


def load():
    """Load the library (once).  Raises if it was not built: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VdsError(
            f"{LIB_PATH} not found: build the gfx950 kernels first (python -c 'import __graft_entry__ as g; "
            "g.build()' or make -C video_diffusion_speedrun_amd/csrc).  This package has no CPU/eager fallback.")
    import torch  # noqa: F401  -- first: libvds_hip.so must bind to the HIP runtime torch has loaded
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = (C.c_char_p if name in ("vds_last_error", "vds_prof_class_name") else
                      C.c_double if name == "vds_knob_get" else
                      C.c_size_t if name.endswith("_workspace_bytes") else
                      C.c_int64 if name.endswith("_workspace_floats") else C.c_int)
    _lib = lib
    return lib


_ERR = {-1: "VDS_ERR_ARG (bad argument / alignment)", -2: "VDS_ERR_UNSUPPORTED (shape not supported)",
        -3: "VDS_ERR_LAUNCH (kernel launch failed)"}


def check(rc: int, what: str):
    if rc != 0:
        msg = load().vds_last_error()
        raise VdsError(f"{what} failed: {_ERR.get(rc, rc)} {msg.decode() if msg else ''}")

"""ctypes binding of libslnlp.so (C ABI in include/slnlp.h).

The library is the product path: there is NO CPU fallback.  Loading works
without a GPU (symbols / layout queries only); any compute entry point needs a
MI355X and raises ``RuntimeError`` with ``slnlp_last_error()`` on failure.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SLNLP_PROBE_LIB=k: load lib/libslnlp_probe<k>.so instead (tools/probes only: `make PROBE=k` builds it with agent-scope fences
# around every kernel, csrc/common.hpp); the product never sets it
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib",
                        f"libslnlp_probe{os.environ['SLNLP_PROBE_LIB']}.so" if os.environ.get("SLNLP_PROBE_LIB") else "libslnlp.so")

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [("A", vp), ("lda", i64), ("a_kmajor", i32),
                ("B", vp), ("ldb", i64), ("b_kmajor", i32),
                ("C", vp), ("ldc", i64),
                ("M", i32), ("N", i32), ("K", i32),
                ("bias", vp), ("relu", i32),
                ("gate", vp), ("ldg", i64), ("gate_scale", f32),
                ("drop_p", f32), ("drop_site", i32), ("rng", vp),
                ("resid", vp), ("ldr", i64),
                ("rowsum_a", vp), ("precision", i32), ("gate_mode", i32),
                ("A_hi", vp), ("A_lo", vp), ("lda_p", i64), ("B_hi", vp), ("B_lo", vp), ("ldb_p", i64),
                ("C_hi", vp), ("C_lo", vp), ("ldc_p", i64), ("drop_head_dim", i32),
                ("col_scale", vp), ("C_q8", vp),
                ("batch", i32), ("batch_stride_a", i64), ("batch_stride_b", i64), ("batch_stride_c", i64)]


class TimedLaunch(C.Structure):
    _fields_ = [("blocks", i32), ("njobs", i32), ("geometry", i32), ("us", f32)]


class TfConfig(C.Structure):
    _fields_ = [("E", i32), ("H", i32), ("N", i32), ("F", i32), ("Vs", i32), ("Vt", i32),
                ("B", i32), ("S", i32), ("pad_src", i32), ("pad_tgt", i32),
                ("dropout", f32), ("precision", i32)]


class TfBuffers(C.Structure):
    _fields_ = [("params", vp), ("grads", vp), ("momentum", vp), ("pe", vp), ("workspace", vp),
                ("rng", vp), ("lr", vp), ("scalars", vp)]


class RnnConfig(C.Structure):
    _fields_ = [("lstm", i32), ("E", i32), ("Hd", i32), ("N", i32), ("Vs", i32), ("Vt", i32), ("B", i32), ("S", i32),
                ("pad_src", i32), ("pad_tgt", i32), ("bos_idx", i32), ("dropout", f32), ("precision", i32)]


class RnnCellDir(C.Structure):
    _fields_ = [("xproj", vp), ("hproj", vp), ("h", vp), ("c", vp), ("hprev_save", vp), ("cprev_save", vp),
                ("acts", vp), ("hn_save", vp), ("out", vp), ("t", i32), ("out_row0", i32), ("out_col0", i32)]


class RnnStepDir(C.Structure):
    _fields_ = [("h_in", vp), ("h_out", vp), ("w_hh", vp), ("b_hh", vp), ("xproj", vp), ("c", vp), ("cprev_save", vp),
                ("acts", vp), ("hn_save", vp), ("out", vp), ("t", i32), ("out_row0", i32), ("out_col0", i32)]


class RnnLayerDir(C.Structure):
    _fields_ = [("hprev", vp), ("h_final", vp), ("w_hh", vp), ("b_hh", vp), ("xproj", vp), ("c", vp), ("cprev", vp),
                ("acts", vp), ("hn", vp), ("out", vp), ("out_col0", i32), ("reverse", i32)]


class RnnCellBwdDir(C.Structure):
    _fields_ = [("dh_state", vp), ("dc_state", vp), ("dout", vp), ("acts", vp), ("cprev_save", vp),
                ("hprev_save", vp), ("hn_save", vp), ("dgx", vp), ("dgh", vp), ("carry", vp),
                ("t", i32), ("out_row0", i32), ("out_col0", i32),
                ("dh_extra", vp), ("extra_stride", i64), ("n_extra", i32)]


class RnnStepBwdDir(C.Structure):
    _fields_ = [("cell", RnnCellBwdDir), ("dgh_next", vp), ("w_hh", vp)]


class LnReduceEntry(C.Structure):
    _fields_ = [("partial", vp), ("dgamma", vp), ("dbeta", vp), ("nblk", i32), ("E", i32)]


# name -> (restype, argtypes).  Every symbol include/slnlp.h declares is listed;
# tests/test_abi.py checks the header and this table agree.
SIGNATURES = {
    "slnlp_last_error": (C.c_char_p, []),
    "slnlp_abi_version": (i32, []),
    "slnlp_gemm": (i32, [C.POINTER(GemmArgs), vp]),
    "slnlp_gemm_group_scratch_bytes": (i64, [C.POINTER(GemmArgs), C.POINTER(i32), i32]),
    "slnlp_set_rnn_step_tile": (i32, [i32]),
    "slnlp_gemm_group": (i32, [C.POINTER(GemmArgs), C.POINTER(i32), i32, vp, i64, vp]),
    "slnlp_gemm_wd": (i32, [C.POINTER(GemmArgs), C.POINTER(GemmArgs), vp, i64, vp]),
    "slnlp_gemm_wd_plan": (i32, [C.POINTER(GemmArgs), C.POINTER(GemmArgs), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
    "slnlp_gemm_rows": (i32, [C.POINTER(GemmArgs), vp]),
    "slnlp_gemm_rows_bwd": (i32, [C.POINTER(GemmArgs), C.POINTER(GemmArgs), vp]),
    "slnlp_set_rows_tile": (i32, [i32]),
    "slnlp_quant_rows_fp8": (i32, [vp, i64, i32, i32, vp, i64, vp, vp]),
    "slnlp_split_planes": (i32, [vp, i64, i32, i32, vp, vp, i64, vp]),
    "slnlp_embed_fwd": (i32, [vp, i64, i32, i32, i32, i32, vp, vp, vp, f32, f32, i32, vp, i64, vp]),
    "slnlp_embed_bwd_scratch_bytes": (i64, [i32, i32, i32]),
    "slnlp_embed_bwd": (i32, [vp, i64, i32, i32, i32, i32, vp, vp, f32, i64, f32, i32, vp, vp, vp]),
    "slnlp_attn_self_fwd": (i32, [vp, vp, i64, i64, i32, i32, i32, i32, i32, vp, vp, f32, i32, vp, vp]),
    "slnlp_attn_self_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, f32, i32, vp, vp]),
    "slnlp_attn_long_scratch_bytes": (i64, [i32, i32, i32]),
    "slnlp_attn_self_bwd_long": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp, f32, i32, vp, vp]),
    "slnlp_attn_cross_fwd": (i32, [vp, vp, i64, i32, i32, i32, i32, vp, vp, f32, i32, vp, vp]),
    "slnlp_attn_cross_bwd": (i32, [vp, vp, i64, vp, vp, i32, i32, i32, i32, vp, vp, i64, f32, i32, vp, vp]),
    "slnlp_layernorm_fwd": (i32, [vp, vp, vp, i32, i32, f32, vp, vp, vp]),
    "slnlp_layernorm_bwd": (i32, [vp, vp, vp, vp, i32, i32, vp, vp, vp, f32, i32, vp, vp, C.POINTER(i32), vp]),
    "slnlp_ln_param_reduce": (i32, [vp, i32, i32, vp]),
    "slnlp_lsm_nll": (i32, [vp, i64, vp, i32, i32, i64, vp, vp, vp, i64, vp, vp]),
    "slnlp_lsm_bwd": (i32, [vp, vp, i32, i32, vp, i64, vp]),
    "slnlp_clip_sgd_step": (i32, [vp, vp, vp, i64, vp, f32, f32, vp, vp, vp, vp]),
    "slnlp_clip_adam_step": (i32, [vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, f32, vp, vp, vp, vp]),
    "slnlp_dropout_mask": (i32, [vp, i32, i32, f32, i32, vp, vp]),
    "slnlp_rnn_cell_fwd": (i32, [i32, C.POINTER(RnnCellDir), i32, i32, i32, vp, f32, i64, f32, i32, vp, vp]),
    "slnlp_rnn_layer_fwd": (i32, [i32, C.POINTER(RnnLayerDir), i32, i32, i32, i32, vp, f32, i64, f32, i32, vp, i32, vp,
                                  C.POINTER(i32), vp]),
    "slnlp_rnn_step_fwd": (i32, [i32, C.POINTER(RnnStepDir), i32, i32, i32, vp, f32, i64, f32, i32, vp, i32, vp]),
    "slnlp_rnn_cell_bwd": (i32, [i32, C.POINTER(RnnCellBwdDir), i32, i32, i32, vp, i64, f32, i32, vp, vp]),
    "slnlp_rnn_step_bwd": (i32, [i32, C.POINTER(RnnStepBwdDir), i32, i32, i32, vp, i64, f32, i32, vp, i32, vp]),
    "slnlp_bahdanau_fwd": (i32, [vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, vp, vp, vp]),
    "slnlp_bahdanau_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "slnlp_rnn_num_params": (i32, [C.POINTER(RnnConfig)]),
    "slnlp_rnn_param_info": (i32, [C.POINTER(RnnConfig), i32, C.c_char_p, C.POINTER(i64 * 2), C.POINTER(i32),
                                   C.POINTER(i64)]),
    "slnlp_rnn_arena_floats": (i64, [C.POINTER(RnnConfig)]),
    "slnlp_rnn_workspace_bytes": (i64, [C.POINTER(RnnConfig)]),
    "slnlp_rnn_create": (i32, [C.POINTER(RnnConfig), C.POINTER(TfBuffers), C.POINTER(vp)]),
    "slnlp_rnn_destroy": (None, [vp]),
    "slnlp_rnn_forward": (i32, [vp, vp, vp, vp, i32, i32, vp, vp]),
    "slnlp_rnn_seed_dlogp": (i32, [vp, vp, vp]),
    "slnlp_rnn_backward": (i32, [vp, vp]),
    "slnlp_rnn_optim": (i32, [vp, f32, f32, vp]),
    "slnlp_rnn_optim_adam": (i32, [vp, vp, f32, f32, f32, f32, f32, vp]),
    "slnlp_rnn_set_destroy_sync": (i32, [vp, i32]),
    "slnlp_rnn_train_step": (i32, [vp, vp, vp, vp, i32, f32, f32, vp, vp]),
    "slnlp_rnn_graph_capture_train": (i32, [vp, vp, vp, vp, i32, f32, f32, vp, vp]),
    "slnlp_rnn_graph_launch": (i32, [vp, i32, vp]),
    "slnlp_rnn_set_persistent": (i32, [vp, i32]),
    "slnlp_rnn_set_fused_backward": (i32, [vp, i32]),
    "slnlp_rnn_health": (i32, [vp, C.POINTER(i32)]),
    "slnlp_rnn_tap": (i32, [vp, C.c_char_p, vp, i64, C.POINTER(i64), vp]),
    "slnlp_tf_num_params": (i32, [C.POINTER(TfConfig)]),
    "slnlp_tf_param_info": (i32, [C.POINTER(TfConfig), i32, C.c_char_p, C.POINTER(i64 * 2), C.POINTER(i32),
                                  C.POINTER(i64)]),
    "slnlp_tf_arena_floats": (i64, [C.POINTER(TfConfig)]),
    "slnlp_tf_workspace_bytes": (i64, [C.POINTER(TfConfig)]),
    "slnlp_tf_create": (i32, [C.POINTER(TfConfig), C.POINTER(TfBuffers), C.POINTER(vp)]),
    "slnlp_tf_destroy": (None, [vp]),
    "slnlp_tf_forward": (i32, [vp, vp, vp, i32, i32, vp, vp]),
    "slnlp_tf_seed_dlogp": (i32, [vp, vp, vp]),
    "slnlp_tf_backward": (i32, [vp, vp]),
    "slnlp_tf_optim": (i32, [vp, f32, f32, vp]),
    "slnlp_tf_train_step": (i32, [vp, vp, vp, i32, f32, f32, vp, vp]),
    "slnlp_tf_graph_capture_train": (i32, [vp, vp, vp, i32, f32, f32, vp, vp]),
    "slnlp_tf_graph_launch": (i32, [vp, i32, vp]),
    "slnlp_tf_tap": (i32, [vp, C.c_char_p, vp, i64, C.POINTER(i64), vp]),
    "slnlp_tf_params_changed": (i32, [vp]),
    "slnlp_tf_optim_adam": (i32, [vp, vp, f32, f32, f32, f32, f32, vp]),
    "slnlp_tf_debug_layout": (i32, [vp, C.c_char_p, i64]),
    "slnlp_tf_set_destroy_sync": (i32, [vp, i32]),
    "slnlp_set_stream_policy": (i32, [i32]),
    "slnlp_set_thread_stream_policy": (i32, [i32]),
    "slnlp_set_backward_passes": (i32, [i32, i32]),
    "slnlp_get_backward_passes": (i32, [C.POINTER(i32), C.POINTER(i32)]),
    "slnlp_launch_timer_start": (i32, [i32]),
    "slnlp_launch_timer_stop": (i32, [C.POINTER(TimedLaunch), i32]),
    "slnlp_set_plane_tile": (i32, [i32]),
    "slnlp_set_gemm_ks": (i32, [i32]),
    "slnlp_set_fp8_tile": (i32, [i32]),
    "slnlp_tf_lockstep_workspace_bytes": (i64, [vp, i32]),
    "slnlp_tf_lockstep_create": (i32, [vp, i32, vp, i64, vp, vp]),
    "slnlp_tf_lockstep_destroy": (None, [vp]),
    "slnlp_tf_lockstep_set_data": (i32, [vp, i32, vp, vp, i64, vp, vp, vp]),
    "slnlp_tf_lockstep_step": (i32, [vp, i32, i64, i32, i32, i32, C.c_float, C.c_float, vp]),
    "slnlp_tf_lockstep_epoch": (i32, [vp, i32, i32, i32, C.c_float, C.c_float, vp]),
    "slnlp_tf_lockstep_num_launches": (i32, [vp, i32, i32, i32]),
    "slnlp_tf_lockstep_set_adam": (i32, [vp, vp, f32, f32, f32, f32]),
    "slnlp_tf_lockstep_set_destroy_sync": (i32, [vp, i32]),
    "slnlp_rnn_lockstep_workspace_bytes": (i64, [vp, i32]),
    "slnlp_rnn_lockstep_create": (i32, [vp, i32, vp, i64, vp, vp]),
    "slnlp_rnn_lockstep_destroy": (None, [vp]),
    "slnlp_rnn_lockstep_set_data": (i32, [vp, i32, vp, vp, vp, i64, vp, vp, vp]),
    "slnlp_rnn_lockstep_step": (i32, [vp, i32, i64, i32, i32, i32, C.c_float, C.c_float, vp]),
    "slnlp_rnn_lockstep_epoch": (i32, [vp, i32, i32, i32, C.c_float, C.c_float, vp]),
    "slnlp_rnn_lockstep_num_launches": (i32, [vp, i32, i32, i32]),
    "slnlp_rnn_lockstep_set_adam": (i32, [vp, vp, f32, f32, f32, f32]),
    "slnlp_rnn_lockstep_set_destroy_sync": (i32, [vp, i32]),
}

_lib = None


def load():
    """Load libslnlp.so (once).  Raises if the in-tree build is missing --
    run ``python -c 'import __graft_entry__ as g; g.build()'`` or ``make`` in
    sign-language-nlp_amd/."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"libslnlp.so not built at {LIB_PATH}: the HIP extension is the "
                               "only compute path (no CPU fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = load().slnlp_last_error().decode(errors="replace")
        raise RuntimeError(f"libslnlp {what} failed (code {rc}): {msg}")


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("slnlp: no MI355X visible -- the HIP path is the only compute path "
                           "(there is deliberately no CPU fallback)")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream

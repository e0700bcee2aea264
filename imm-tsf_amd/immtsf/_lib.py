"""ctypes binding of libimmtsf_hip.so (the C ABI declared in include/immtsf.h).

There is deliberately no CPU fallback: if the shared library is missing or a call fails, the product modules
raise.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libimmtsf_hip.so")

ABI_VERSION = 6
FORM_NO_PROJ = 16        # immtsf_fusion_cfg.form bit (IMMTSF_FORM_NO_PROJ)
FORM_HALF_OUT = 32       # ... IMMTSF_FORM_HALF_OUT: the fp32 output is not written, its bf16 image is all the consumer reads
FORM_LOWRANK_OUT = 64    # ... IMMTSF_FORM_LOWRANK_OUT: dZ = dP Wc is not written, the producer's backward takes (dP, Wc)
BWD_PHASE_A, BWD_PHASE_B, BWD_PHASE_C = 1, 2, 4      # immtsf_fusion_cfg.bwd_phase bits: data paths (IMMTSF_BWD_PHASE_*)
BWD_WGRAD_A, BWD_WGRAD_B, BWD_WGRAD_C = 16, 32, 64   # ... and parameter gradients (IMMTSF_BWD_WGRAD_*)


class ImmtsfError(RuntimeError):
    pass


_ERR = {-1: "IMMTSF_EINVAL (bad dimension / null pointer)", -2: "IMMTSF_EWORKSPACE (workspace too small)",
        -3: "IMMTSF_EUNSUPPORTED (shape outside kernel limits)"}

c_f32p = C.c_void_p   # device pointers travel as integers
c_u8p = C.c_void_p
c_i32p = C.c_void_p
c_stream = C.c_void_p


class FusionCfg(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("T", C.c_int32), ("C", C.c_int32), ("d_m", C.c_int32),
                ("d", C.c_int32), ("H", C.c_int32), ("precision", C.c_int32), ("training", C.c_int32),
                ("p_drop", C.c_float), ("kappa", C.c_float), ("seed", C.c_uint64), ("seed_step_dev", C.c_void_p), ("grads_prezeroed", C.c_int32),
                ("form", C.c_int32), ("in_h", C.c_void_p), ("aux_h", C.c_void_p), ("out_h", C.c_void_p), ("sched_flag", C.c_void_p),
                ("bwd_phase", C.c_int32), ("reserved0", C.c_int32), ("note_index", C.c_void_p), ("lr_grad", C.c_void_p)]


class LowRankGrad(C.Structure):
    """immtsf_lowrank_grad: an upstream gradient as coef (rows, rank; pitch ld) x basis (rank, d)"""
    _fields_ = [("coef", C.c_void_p), ("basis", C.c_void_p), ("rank", C.c_int32), ("ld", C.c_int32)]


def _ptr_struct(name, fields):
    return type(name, (C.Structure,), {"_fields_": [(f, C.c_void_p) for f in fields], "FIELDS": tuple(fields)})


T2VParams = _ptr_struct("T2VParams", [
    "Q_param", "input_proj_w", "input_proj_b", "t2v_lin_w", "t2v_lin_b", "t2v_per_w", "t2v_per_b", "kv_w", "kv_b",
    "attn_in_w", "attn_in_b", "attn_out_w", "attn_out_b", "ln_w", "ln_b", "proj_out_w", "proj_out_b"])
RecAvgParams = _ptr_struct("RecAvgParams", [
    "log_recency_sigma", "input_proj_w", "input_proj_b", "ln_w", "ln_b", "proj_w", "proj_b"])
XAddParams = _ptr_struct("XAddParams", [
    "proj_q_w", "proj_k_w", "proj_v_w", "attn_in_w", "attn_in_b", "attn_out_w", "attn_out_b", "res_w", "res_b",
    "ln_w", "ln_b"])
TTCNParams = _ptr_struct("TTCNParams", [
    "te_scale_w", "te_scale_b", "te_per_w", "te_per_b", "W1", "b1", "W2", "b2", "W3", "b3", "T_bias"])
class EncoderLayerCfg(C.Structure):
    _fields_ = [("Bs", C.c_int32), ("S", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("F", C.c_int32), ("precision", C.c_int32),
                ("training", C.c_int32), ("p_attn", C.c_float), ("p_drop", C.c_float), ("eps", C.c_float), ("seed", C.c_uint64),
                ("seed_step_dev", C.c_void_p), ("site_base", C.c_uint64), ("grads_prezeroed", C.c_int32)]


EncoderLayerParams = _ptr_struct("EncoderLayerParams", [
    "in_w", "in_b", "out_w", "out_b", "ln1_w", "ln1_b", "w1", "b1", "w2", "b2", "ln2_w", "ln2_b"])
class FFNBlockCfg(C.Structure):
    _fields_ = [("R", C.c_int32), ("D", C.c_int32), ("F", C.c_int32), ("act", C.c_int32), ("precision", C.c_int32), ("training", C.c_int32),
                ("p_drop", C.c_float), ("eps", C.c_float), ("seed", C.c_uint64), ("seed_step_dev", C.c_void_p), ("site_base", C.c_uint64),
                ("grads_prezeroed", C.c_int32)]


FFNBlockParams = _ptr_struct("FFNBlockParams", ["w1", "b1", "w2", "b2", "ln_w", "ln_b"])
GCNParams = _ptr_struct("GCNParams", [
    "nodevec1", "nodevec2", "gate1_w", "gate1_b", "gate2_w", "gate2_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b",
    "mlp_w", "mlp_b"])
DecoderParams = _ptr_struct("DecoderParams", ["W1", "b1", "W2", "b2", "W3", "b3"])
Time2VecParams = _ptr_struct("Time2VecParams", ["w0", "b0", "w", "b"])
NoteIndex = _ptr_struct("NoteIndex", ["mask", "mtxt", "lengths", "offsets", "rowmap", "seg"])
GRParams = _ptr_struct("GRParams", [
    "w_ih", "w_hh", "b_ih", "b_hh", "res_w", "res_b", "gate_w", "gate_b", "ln_w", "ln_b"])

class Store(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("tt", "vals", "mask", "row_off", "hist_len", "note_tau", "note_src", "note_off",
                                           "emb")] + [("C", C.c_int32), ("d_m", C.c_int32)]


# name -> (restype, argtypes).  Must list EVERY function include/immtsf.h declares (tests/test_abi.py checks).
_P = C.POINTER
_PROTOS = {
    "immtsf_abi_version": (C.c_int, []),
    "immtsf_abi_sizes": (C.c_int, [C.c_void_p, C.c_int32]),
    "immtsf_ragged_index": (C.c_int, [c_f32p, C.c_int32, C.c_int32, C.c_int32, c_u8p, c_i32p, c_i32p, c_i32p, c_i32p,
                                      c_u8p, c_i32p, c_stream]),
    "immtsf_note_index_build": (C.c_int, [c_i32p, C.c_int32, C.c_int32, _P(NoteIndex), c_stream]),
    "immtsf_ttf_t2v_xattn_workspace_bytes": (C.c_size_t, [_P(FusionCfg)]),
    "immtsf_ttf_t2v_xattn_scratch_bytes": (C.c_size_t, [_P(FusionCfg)]),
    "immtsf_instance_norm": (C.c_int, [c_f32p, C.c_int32, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, c_stream]),
    "immtsf_notes_stage": (C.c_int, [c_f32p, C.c_int32, c_i32p, c_i32p, C.c_int32, C.c_void_p, C.c_int32, c_f32p, c_i32p, C.c_int32, c_f32p, c_f32p,
                                     c_f32p, c_f32p, c_stream]),
    "immtsf_ttf_t2v_xattn_folded": (C.c_int, [_P(FusionCfg)]),
    "immtsf_ttf_t2v_xattn_accepts_lowrank": (C.c_int, [_P(FusionCfg), C.c_int32]),
    "immtsf_mmf_xrank_lowrank_basis": (C.c_int, [_P(FusionCfg), C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]),
    "immtsf_ttf_t2v_xattn_forward": (C.c_int, [_P(FusionCfg), _P(T2VParams), c_f32p, c_f32p, c_f32p, c_u8p, C.c_void_p,
                                               C.c_size_t, c_i32p, c_stream]),
    "immtsf_ttf_t2v_xattn_backward": (C.c_int, [_P(FusionCfg), _P(T2VParams), c_f32p, c_f32p, c_f32p, C.c_void_p,
                                                C.c_size_t, C.c_void_p, C.c_size_t, _P(T2VParams), c_stream]),
    "immtsf_ttf_t2v_xattn_forward_packed": (C.c_int, [_P(FusionCfg), _P(T2VParams), c_f32p, c_i32p, c_i32p, c_f32p, c_f32p,
                                                      c_u8p, C.c_void_p, C.c_size_t, c_stream]),
    "immtsf_ttf_t2v_xattn_backward_packed": (C.c_int, [_P(FusionCfg), _P(T2VParams), c_f32p, c_i32p, c_f32p, c_f32p,
                                                       C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _P(T2VParams),
                                                       c_stream]),
    "immtsf_ttf_recavg_workspace_bytes": (C.c_size_t, [_P(FusionCfg)]),
    "immtsf_ttf_recavg_scratch_bytes": (C.c_size_t, [_P(FusionCfg)]),
    "immtsf_ttf_recavg_forward": (C.c_int, [_P(FusionCfg), _P(RecAvgParams), c_f32p, c_f32p, c_f32p, c_f32p, c_u8p,
                                            C.c_void_p, C.c_size_t, c_i32p, c_stream]),
    "immtsf_ttf_recavg_backward": (C.c_int, [_P(FusionCfg), _P(RecAvgParams), c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p,
                                             C.c_size_t, C.c_void_p, C.c_size_t, _P(RecAvgParams), c_stream]),
    "immtsf_mmf_xattn_add_workspace_bytes": (C.c_size_t, [_P(FusionCfg)]),
    "immtsf_mmf_xattn_add_scratch_bytes": (C.c_size_t, [_P(FusionCfg)]),
    "immtsf_mmf_xattn_add_forward": (C.c_int, [_P(FusionCfg), _P(XAddParams), c_f32p, c_f32p, c_u8p, c_f32p, C.c_void_p,
                                               C.c_size_t, c_stream]),
    "immtsf_mmf_xattn_add_backward": (C.c_int, [_P(FusionCfg), _P(XAddParams), c_f32p, c_f32p, c_u8p, c_f32p, c_f32p,
                                                c_f32p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _P(XAddParams),
                                                c_stream]),
    "immtsf_mmf_xattn_kv_workspace_bytes": (C.c_size_t, [_P(FusionCfg)]),
    "immtsf_mmf_xattn_kv_scratch_bytes": (C.c_size_t, [_P(FusionCfg)]),
    "immtsf_mmf_xattn_q_workspace_bytes": (C.c_size_t, [_P(FusionCfg)]),
    "immtsf_mmf_xattn_q_scratch_bytes": (C.c_size_t, [_P(FusionCfg)]),
    "immtsf_mmf_xattn_kv_forward": (C.c_int, [_P(FusionCfg), _P(XAddParams), c_f32p, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "immtsf_mmf_xattn_kv_backward": (C.c_int, [_P(FusionCfg), _P(XAddParams), c_f32p, c_f32p, c_f32p, C.c_void_p,
                                               C.c_size_t, C.c_void_p, C.c_size_t, _P(XAddParams), c_stream]),
    "immtsf_mmf_xrank_pw": (C.c_int32, [_P(FusionCfg)]),
    "immtsf_mmf_xrank_p_workspace_bytes": (C.c_size_t, [_P(FusionCfg)]),
    "immtsf_mmf_xrank_p_scratch_bytes": (C.c_size_t, [_P(FusionCfg)]),
    "immtsf_mmf_xrank_q_workspace_bytes": (C.c_size_t, [_P(FusionCfg)]),
    "immtsf_mmf_xrank_fold": (C.c_int, [_P(FusionCfg), _P(XAddParams), c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "immtsf_mmf_xrank_p_forward": (C.c_int, [_P(FusionCfg), _P(XAddParams), c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_int32,
                                           c_stream]),
    "immtsf_mmf_xrank_p_backward": (C.c_int, [_P(FusionCfg), _P(XAddParams), c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t,
                                            C.c_void_p, C.c_size_t, _P(XAddParams), c_stream]),
    "immtsf_mmf_xrank_p_backward_data": (C.c_int, [_P(FusionCfg), _P(XAddParams), c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t,
                                                 C.c_void_p, C.c_size_t, c_stream]),
    "immtsf_mmf_xrank_fold_z": (C.c_int, [_P(FusionCfg), _P(XAddParams), c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "immtsf_mmf_xrank_p_forward_z": (C.c_int, [_P(FusionCfg), _P(XAddParams), c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t,
                                             C.c_int32, c_stream]),
    "immtsf_mmf_xrank_p_backward_data_z": (C.c_int, [_P(FusionCfg), _P(XAddParams), c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p,
                                                   C.c_size_t, C.c_void_p, C.c_size_t, c_stream]),
    "immtsf_mmf_xrank_p_backward_pre_z": (C.c_int, [_P(FusionCfg), c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, c_f32p,
                                                  c_f32p, c_stream]),
    "immtsf_mmf_xrank_p_backward_params": (C.c_int, [_P(FusionCfg), _P(XAddParams), c_f32p, C.c_void_p, C.c_size_t, C.c_void_p,
                                                   C.c_size_t, _P(XAddParams), C.c_int32, C.c_int32, c_stream]),
    "immtsf_mmf_xrank_q_forward": (C.c_int, [_P(FusionCfg), c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_u8p, c_f32p, C.c_void_p,
                                           C.c_size_t, c_stream]),
    "immtsf_mmf_xrank_q_backward": (C.c_int, [_P(FusionCfg), c_f32p, c_f32p, c_f32p, c_u8p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                            c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "immtsf_mmf_xrank_q_train_scratch_bytes": (C.c_size_t, [_P(FusionCfg)]),
    "immtsf_mmf_xrank_q_train": (C.c_int, [_P(FusionCfg), c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_u8p, c_f32p, c_f32p, c_f32p, C.c_float,
                                         c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p,
                                         C.c_void_p, c_stream]),
    "immtsf_attn_mid_supported": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "immtsf_attn_mid_forward": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                        C.c_int32, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p, c_f32p, c_f32p, c_stream]),
    "immtsf_attn_mid_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_float, C.c_float, C.c_uint64, C.c_uint64, C.c_void_p, c_f32p, c_f32p, c_f32p, c_stream]),
    "immtsf_mmf_xattn_q_fold_floats": (C.c_size_t, [_P(FusionCfg)]),
    "immtsf_mmf_xattn_q_fold": (C.c_int, [_P(FusionCfg), _P(XAddParams), c_f32p, c_stream]),
    "immtsf_mmf_xattn_q_forward": (C.c_int, [_P(FusionCfg), _P(XAddParams), c_f32p, c_f32p, c_u8p, c_f32p, c_f32p, C.c_void_p,
                                             C.c_size_t, c_stream]),
    "immtsf_mmf_xattn_q_backward": (C.c_int, [_P(FusionCfg), _P(XAddParams), c_f32p, c_f32p, c_u8p, c_f32p, c_f32p, c_f32p,
                                              c_f32p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, _P(XAddParams), C.c_int32,
                                              c_stream]),
    "immtsf_mmf_xattn_q_backward_params": (C.c_int, [_P(FusionCfg), _P(XAddParams), c_f32p, c_u8p, c_f32p, C.c_void_p,
                                                     C.c_size_t, C.c_void_p, C.c_size_t, _P(XAddParams), c_stream]),
    "immtsf_mmf_gr_add_workspace_bytes": (C.c_size_t, [_P(FusionCfg), C.c_int32]),
    "immtsf_mmf_gr_add_scratch_bytes": (C.c_size_t, [_P(FusionCfg), C.c_int32]),
    "immtsf_mmf_gr_add_forward": (C.c_int, [_P(FusionCfg), C.c_int32, _P(GRParams), c_f32p, c_f32p, c_u8p, c_f32p,
                                            C.c_void_p, C.c_size_t, c_stream]),
    "immtsf_mmf_gr_add_backward": (C.c_int, [_P(FusionCfg), C.c_int32, _P(GRParams), c_f32p, c_f32p, c_u8p, c_f32p,
                                             c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                             _P(GRParams), c_stream]),
    "immtsf_mmf_gr_pw": (C.c_int32, [_P(FusionCfg), C.c_int32]),
    "immtsf_mmf_gr_p_workspace_bytes": (C.c_size_t, [_P(FusionCfg), C.c_int32]),
    "immtsf_mmf_gr_p_scratch_bytes": (C.c_size_t, [_P(FusionCfg), C.c_int32]),
    "immtsf_mmf_gr_p_forward": (C.c_int, [_P(FusionCfg), C.c_int32, _P(GRParams), c_f32p, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "immtsf_mmf_gr_p_backward": (C.c_int, [_P(FusionCfg), C.c_int32, _P(GRParams), c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t, C.c_void_p,
                                           C.c_size_t, _P(GRParams), c_stream]),
    "immtsf_mmf_gr_q_train": (C.c_int, [_P(FusionCfg), C.c_int32, _P(GRParams), c_f32p, c_f32p, c_u8p, c_f32p, c_f32p, c_f32p, C.c_float, c_f32p,
                                        c_f32p, c_f32p, c_f32p, _P(GRParams), c_f32p, C.c_void_p, C.c_void_p, c_stream]),
    "immtsf_ttcn_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "immtsf_ttcn_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "immtsf_ttcn_forward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p,
                                      _P(TTCNParams), c_f32p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, c_stream]),
    "immtsf_ttcn_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p,
                                       _P(TTCNParams), c_f32p, c_f32p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p,
                                       C.c_size_t, _P(TTCNParams), C.c_int32, c_stream]),
    "immtsf_masked_mse_sums": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, c_stream]),
    "immtsf_masked_mse": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_float,
                                    c_stream]),
    "immtsf_masked_mse_counted": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, C.c_float, c_stream]),
    "immtsf_masked_mse_finish": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p,
                                           C.c_float, c_stream]),
    "immtsf_tpatchgnn_gcn_lds_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "immtsf_encoder_layer_workspace_bytes": (C.c_size_t, [_P(EncoderLayerCfg)]),
    "immtsf_encoder_layer_scratch_bytes": (C.c_size_t, [_P(EncoderLayerCfg)]),
    "immtsf_encoder_layer_forward": (C.c_int, [_P(EncoderLayerCfg), _P(EncoderLayerParams), c_f32p, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "immtsf_encoder_layer_backward": (C.c_int, [_P(EncoderLayerCfg), _P(EncoderLayerParams), c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t,
                                                C.c_void_p, C.c_size_t, _P(EncoderLayerParams), c_stream]),
    "immtsf_residual_layernorm_forward": (C.c_int, [c_f32p, c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p, C.c_float, C.c_int32, C.c_float,
                                                    C.c_uint64, C.c_uint64, C.c_void_p, c_f32p, c_f32p, c_f32p, c_stream]),
    "immtsf_residual_layernorm_backward": (C.c_int, [c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_float, C.c_uint64,
                                                     C.c_uint64, C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "immtsf_ffn_block_workspace_bytes": (C.c_size_t, [_P(FFNBlockCfg)]),
    "immtsf_ffn_block_scratch_bytes": (C.c_size_t, [_P(FFNBlockCfg)]),
    "immtsf_ffn_block_forward": (C.c_int, [_P(FFNBlockCfg), _P(FFNBlockParams), c_f32p, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "immtsf_ffn_block_backward": (C.c_int, [_P(FFNBlockCfg), _P(FFNBlockParams), c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_size_t,
                                            C.c_void_p, C.c_size_t, _P(FFNBlockParams), c_stream]),
    "immtsf_inception_merge": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), c_f32p, c_f32p, c_stream]),
    "immtsf_inception_unmerge": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, c_f32p, c_f32p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                           c_stream]),
    "immtsf_conv2d_same_cl_forward": (C.c_int, [C.c_int32, c_f32p] + [C.c_int32] * 5 + [c_f32p, c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p,
                                                                                     c_f32p, c_stream]),
    "immtsf_conv2d_same_cl_scratch_floats": (C.c_size_t, [C.c_int32] * 6),
    "immtsf_conv2d_same_cl_backward": (C.c_int, [C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p] + [C.c_int32] * 5 + [c_f32p, C.c_int32, C.c_int32,
                                                                                                               c_f32p, c_f32p, c_f32p, c_f32p,
                                                                                                               c_stream]),
    "immtsf_period_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, c_i32p, c_i32p, c_stream]),
    "immtsf_conv2d_period_forward": (C.c_int, [C.c_int32, c_f32p, C.c_int32, C.c_int32, c_i32p, c_i32p, C.c_int32, C.c_int32, c_f32p, c_f32p,
                                               C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_int32, c_stream]),
    "immtsf_conv2d_periods_forward": (C.c_int, [C.c_int32, c_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, c_i32p, c_i32p, C.c_int32, C.c_int32,
                                                c_f32p, c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_int32, c_stream]),
    "immtsf_conv2d_periods_scratch_floats": (C.c_size_t, [C.c_int32] * 6),
    "immtsf_conv2d_periods_backward": (C.c_int, [C.c_int32, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, c_i32p, c_i32p, C.c_int32,
                                                 C.c_int32, c_f32p, C.c_int32, C.c_int32, c_f32p, C.c_int32, c_f32p, c_f32p, c_f32p, C.c_void_p,
                                                 c_stream]),
    "immtsf_conv2d_periods_implicit_ok": (C.c_int, [C.c_int32] * 5),
    "immtsf_conv2d_periods_backward_x_scratch_floats": (C.c_size_t, [C.c_int32] * 6),
    "immtsf_conv2d_periods_backward_x": (C.c_int, [C.c_int32, c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, c_i32p, c_i32p,
                                                   C.c_int32, C.c_int32, c_f32p, C.c_int32, C.c_int32, c_f32p, C.c_int32, c_f32p, c_f32p, c_f32p,
                                                   C.c_int32, c_stream]),
    "immtsf_period_aggregate_forward": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_f32p, c_stream]),
    "immtsf_period_aggregate_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_f32p, c_f32p,
                                                   c_stream]),
    "immtsf_conv2d_period_scratch_floats": (C.c_size_t, [C.c_int32] * 5),
    "immtsf_conv2d_period_backward": (C.c_int, [C.c_int32, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, c_i32p, c_i32p, C.c_int32, C.c_int32,
                                                c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, c_stream]),
    "immtsf_tpatchgnn_gcn_forward": (C.c_int, [C.c_int32] * 6 + [c_f32p, _P(GCNParams), c_f32p, c_stream]),
    "immtsf_tpatchgnn_gcn_saved_floats": (C.c_size_t, [C.c_int32] * 6),
    "immtsf_tpatchgnn_gcn_forward_saved": (C.c_int, [C.c_int32] * 6 + [c_f32p, _P(GCNParams), c_f32p, c_f32p, c_stream]),
    "immtsf_tpatchgnn_gcn_backward_saved": (C.c_int, [C.c_int32] * 6 + [c_f32p, _P(GCNParams), c_f32p, c_f32p, _P(GCNParams), c_stream]),
    "immtsf_tpatchgnn_gcn_backward": (C.c_int, [C.c_int32] * 6 + [c_f32p, _P(GCNParams), c_f32p, c_f32p, _P(GCNParams),
                                                c_stream]),
    "immtsf_tpatchgnn_decoder_lds_bytes": (C.c_size_t, [C.c_int32] * 5),
    "immtsf_tpatchgnn_decoder_forward": (C.c_int, [C.c_int32] * 6 + [c_f32p, c_f32p, _P(DecoderParams), c_f32p, c_stream]),
    "immtsf_tpatchgnn_decoder_backward": (C.c_int, [C.c_int32] * 6 + [c_f32p, c_f32p, _P(DecoderParams), c_f32p, c_f32p, c_f32p,
                                                    _P(DecoderParams), c_stream]),
    "immtsf_tpatchgnn_decoder_forward_p": (C.c_int, [C.c_int32] * 7 + [c_f32p, c_f32p, _P(DecoderParams), c_f32p, c_stream]),
    "immtsf_tpatchgnn_decoder_backward_p": (C.c_int, [C.c_int32] * 7 + [c_f32p, c_f32p, _P(DecoderParams), c_f32p, c_f32p, c_f32p,
                                                   _P(DecoderParams), c_stream]),
    "immtsf_tpatchgnn_decoder_forward_te": (C.c_int, [C.c_int32] * 7 + [c_f32p, c_f32p, _P(Time2VecParams), _P(DecoderParams), c_f32p,
                                                      c_stream]),
    "immtsf_tpatchgnn_decoder_backward_te": (C.c_int, [C.c_int32] * 7 + [c_f32p, c_f32p, _P(Time2VecParams), _P(DecoderParams), c_f32p,
                                                       c_f32p, _P(DecoderParams), _P(Time2VecParams), c_stream]),
    "immtsf_collate_series": (C.c_int, [_P(Store), c_i32p, C.c_int32, C.c_int32, C.c_int32, C.c_float] + [c_f32p] * 6 + [c_stream]),
    "immtsf_collate_patches": (C.c_int, [_P(Store), c_i32p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_int32,
                                         C.c_float, c_f32p, c_f32p, c_f32p, c_stream]),
    "immtsf_collate_notes": (C.c_int, [_P(Store), c_i32p, C.c_int32, C.c_int32, c_f32p, c_f32p, c_i32p, c_i32p, C.c_void_p,
                                       c_stream]),
    "immtsf_gemm": (C.c_int, [C.c_int32, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32, c_f32p,
                              C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32, c_stream]),
    "immtsf_linear_backward": (C.c_int, [C.c_int32, c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, c_f32p,
                                         c_f32p, c_f32p, c_f32p, C.c_int32, c_stream]),
    "immtsf_linear_bf16_forward": (C.c_int, [C.c_int32, c_f32p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_int32, c_stream]),
    "immtsf_linear_bf16_backward": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_f32p, C.c_void_p,
                                              C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_stream]),
    "immtsf_time2vec_forward": (C.c_int, [c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "immtsf_time2vec_backward": (C.c_int, [c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                           c_f32p, c_f32p, C.c_int32, c_stream]),
    "immtsf_bf16_twin_register": (C.c_int, [c_f32p, C.c_void_p, C.c_size_t]),
    "immtsf_bf16_twin_unregister": (C.c_int, [c_f32p]),
    "immtsf_bf16_twin_enable": (C.c_int, [C.c_int32]),
    "immtsf_patch_flatten3": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, c_stream]),
    "immtsf_f32_to_bf16": (C.c_int, [c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "immtsf_bf16_to_f32": (C.c_int, [C.c_void_p, c_f32p, C.c_size_t, c_stream]),
    "immtsf_gemm_batched": (C.c_int, [C.c_int32, C.c_int32, c_f32p, C.c_int32, C.c_int64, C.c_int64, c_f32p, C.c_int32,
                                      C.c_int64, C.c_int64, c_f32p, C.c_int32, C.c_int64, C.c_int64, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, c_stream]),
    "immtsf_softmax_rows_forward": (C.c_int, [c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_u8p,
                                              C.c_float, C.c_uint64, C.c_uint64, C.c_int32, C.c_void_p, c_stream]),
    "immtsf_softmax_rows_backward": (C.c_int, [c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                               C.c_uint64, C.c_uint64, C.c_void_p, c_stream]),
    "immtsf_attention_short_forward": (C.c_int, [c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_float,
                                                 C.c_uint64, C.c_uint64, C.c_void_p, c_f32p, c_stream]),
    "immtsf_attention_short_backward": (C.c_int, [c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32,
                                                  C.c_float, C.c_uint64, C.c_uint64, C.c_void_p, c_f32p, c_stream]),
    "immtsf_layernorm_forward": (C.c_int, [c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p, C.c_float, c_f32p, c_f32p,
                                           c_f32p, C.c_float, C.c_uint64, C.c_uint64, c_stream]),
    "immtsf_layernorm_backward": (C.c_int, [c_f32p, C.c_int32, C.c_int32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                            c_f32p, C.c_float, C.c_uint64, C.c_uint64, c_stream]),
    "immtsf_dropout_mask": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, c_u8p, c_stream]),
    "immtsf_set_side_stream": (C.c_int, [C.c_int32]),
    "immtsf_side_stream_enabled": (C.c_int, []),
    "immtsf_flag_set": (C.c_int, [C.c_void_p, c_stream]),
    "immtsf_flag_wait": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, c_stream]),
    "immtsf_flags_clear": (C.c_int, [C.c_void_p, C.c_int32, c_stream]),
    "immtsf_flags_clear_set": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, c_stream]),
    "immtsf_flag_bump": (C.c_int, [C.c_void_p, c_stream]),
    "immtsf_flag_wait_ge": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, c_stream]),
    "immtsf_flag_trace": (C.c_int, [C.c_int32]),
    "immtsf_flag_trace_read": (C.c_int, [C.c_void_p, C.c_int32]),
    "immtsf_timing_enable": (C.c_int, [C.c_int32]),
    "immtsf_debug_gemm_config": (C.c_int, [C.c_int32, C.c_int32]),
    "immtsf_timing_collect": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "immtsf_gemm_bf16_group_tn": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, c_stream]),
    "immtsf_embed_forward": (C.c_int, [C.c_int32, c_f32p] + [C.c_int32] * 7 + [c_f32p, c_f32p, c_f32p, C.c_float, C.c_uint64, C.c_uint64,
                                       C.c_void_p, c_stream]),
    "immtsf_embed_backward": (C.c_int, [C.c_int32, c_f32p] + [C.c_int32] * 7 + [c_f32p, c_f32p, c_f32p, C.c_int32, c_f32p, C.c_float,
                                        C.c_uint64, C.c_uint64, C.c_void_p, c_stream]),
    "immtsf_gemm_bf16": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, c_f32p, C.c_int32, C.c_void_p,
                                   C.c_int32, c_f32p, c_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32,
                                   c_i32p, C.c_int32, c_i32p, c_stream]),
    "immtsf_debug_gemm2_config": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "immtsf_gemm3_bf16": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, c_f32p, C.c_int32, C.c_void_p,
                                    C.c_int32, c_f32p, c_f32p, c_i32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                    C.c_int32, c_i32p, c_stream]),
    "immtsf_debug_gemm3_config": (C.c_int, [C.c_int32, C.c_int32]),
    "immtsf_gemm3_tn_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "immtsf_gemm3_tn_bf16": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, c_f32p, C.c_int32, c_f32p, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_float, C.c_int32, c_i32p, C.c_void_p, C.c_size_t, c_stream]),
    "immtsf_adam_step_dev": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_uint64, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_float, C.c_void_p, C.c_float, c_f32p, C.c_void_p, c_stream]),
    "immtsf_adam_step_dev_zero": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_uint64, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_float, C.c_void_p, C.c_float, c_f32p, C.c_void_p, c_stream]),
    "immtsf_adam_step_guarded": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_uint64, C.c_float, C.c_float, C.c_float,
                                           C.c_float, C.c_float, C.c_void_p, C.c_float, c_f32p, C.c_void_p, C.c_void_p, C.c_int32, c_stream]),
    "immtsf_adam_sqnorm": (C.c_int, [c_f32p, C.c_uint64, c_f32p, C.c_void_p, C.c_void_p, c_stream]),
    "immtsf_adam_apply": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_uint64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_int32, C.c_void_p, C.c_float, c_f32p, C.c_void_p, c_stream]),
    "immtsf_adam_step": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_uint64, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_float, C.c_int32, C.c_float, c_f32p, c_stream]),
    "immtsf_adam_prepare": (C.c_int, [c_f32p, C.c_void_p, C.c_uint64, c_f32p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      c_f32p, C.c_void_p, c_stream]),
    "immtsf_flag_wait_ge_multi": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, c_stream]),
    "immtsf_adam_range": (C.c_int, [c_f32p, c_f32p, C.c_void_p, c_f32p, c_f32p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.c_float,
                                    C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_float, c_f32p, C.c_int32, C.c_void_p, c_stream]),
    "immtsf_guard_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, c_stream]),
}

# the structs of the ABI in immtsf_abi_sizes' order (tests/test_abi.py compares ctypes.sizeof with the library's sizeof)
def abi_structs():
    return [FusionCfg, T2VParams, RecAvgParams, XAddParams, GRParams, TTCNParams, GCNParams, DecoderParams, Time2VecParams,
            EncoderLayerCfg, EncoderLayerParams, FFNBlockCfg, FFNBlockParams, Store, NoteIndex, LowRankGrad]


_lib = None


def load():
    """dlopen the library once and attach prototypes.  Raises ImmtsfError when it is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    global LIB_PATH
    LIB_PATH = os.environ.get("IMMTSF_LIB", LIB_PATH)       # measurement builds (tools/); the default is the in-tree library
    if not os.path.exists(LIB_PATH):
        raise ImmtsfError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C imm-tsf_amd/csrc`.  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImmtsfError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if lib.immtsf_abi_version() != ABI_VERSION:
        raise ImmtsfError("libimmtsf_hip.so ABI version mismatch; rebuild it")
    structs = abi_structs()
    sizes = (C.c_int32 * len(structs))()
    if lib.immtsf_abi_sizes(sizes, len(structs)) != len(structs) or any(int(sizes[i]) != C.sizeof(t) for i, t in enumerate(structs)):
        raise ImmtsfError("libimmtsf_hip.so struct layout differs from this binding (immtsf_abi_sizes); rebuild it")
    _lib = lib
    return lib


def exported_names():
    return sorted(_PROTOS)


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc < 0:
        raise ImmtsfError(f"{what}: {_ERR.get(rc, rc)}")
    raise ImmtsfError(f"{what}: hipError_t {rc}")


def ptr(t):
    """device pointer of a tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)

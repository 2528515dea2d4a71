"""ctypes binding of libdsvg_hip.so (the C ABI declared in include/dsvg.h).

The product path has no CPU fallback: if the shared library is missing or a tensor is not on a HIP
device, the ops raise.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"``
(or ``deepsvg_amd/csrc/build.sh``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libdsvg_hip.so")

DSVG_F32 = 0
DSVG_BF16 = 1
# == DSVG_ABI_VERSION of include/dsvg.h at the time SIGNATURES below was written: load() refuses a library built from another
# header (a stale .so with the old argument lists would otherwise be called with a stream where a size is expected)
ABI_VERSION = 9

c_i32, c_i64, c_u32, c_f32 = C.c_int32, C.c_int64, C.c_uint32, C.c_float
vp = C.c_void_p


class GemmDesc(C.Structure):
    """mirror of ``dsvg_gemm_desc`` (include/dsvg.h)"""
    _fields_ = [
        ("dtype", c_i32), ("M", c_i32), ("N", c_i32), ("K", c_i32),
        ("A", vp), ("lda", c_i64), ("a_kc", c_i32),
        ("B", vp), ("ldb", c_i64), ("b_kc", c_i32),
        ("C", vp), ("ldc", c_i64), ("c_f32", c_i32),
        ("bias", vp),
        ("res", vp), ("ldres", c_i64), ("res_pre", c_i32),
        ("act", c_i32),
        ("gate", vp), ("ldgate", c_i64), ("gate_scale", c_f32),
        ("drop_p", c_f32), ("drop_site", c_u32),
        ("a_drop_p", c_f32), ("a_drop_site", c_u32), ("a_drop_ld", c_i64),
        ("seed", vp),
        ("accumulate", c_i32),
        ("split_k", c_i32), ("workspace", vp), ("workspace_bytes", c_i64),
        ("rowsum", vp),
        ("impl", c_i32),
    ]


class GsFwdLayer(C.Structure):
    """mirror of ``dsvg_gs_fwd_layer`` (include/dsvg.h)"""
    _fields_ = [(n, vp) for n in ("packed_fwd_layer", "in_bias", "out_bias", "b1", "b2", "gamma1", "beta1", "gamma2", "beta2",
                                  "seq_add", "x2", "mean1", "rstd1", "xn1", "qkv", "ao", "x1", "mean2", "rstd2", "xn2", "h")] + [
        ("site0", c_u32), ("reserved_", c_u32)]


class GsBwdLayer(C.Structure):
    """mirror of ``dsvg_gs_bwd_layer`` (include/dsvg.h)"""
    _fields_ = [(n, vp) for n in ("packed_bwd_layer", "x", "mean1", "rstd1", "qkv", "x1", "mean2", "rstd2", "h", "gamma1", "gamma2",
                                  "dx", "dx1", "dym", "dpre", "dx1m", "dqkv", "dg", "dgamma2", "dbeta2", "dgamma1", "dbeta1",
                                  "workspace")] + [("site0", c_u32), ("reserved_", c_u32)]


# name -> (restype, argtypes); every symbol declared in include/dsvg.h
SIGNATURES = {
    "dsvg_last_error": (C.c_char_p, []),
    "dsvg_version": (c_i32, []),
    "dsvg_gemm": (c_i32, [C.POINTER(GemmDesc), vp]),
    "dsvg_gemm_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "dsvg_reduce_partials": (c_i32, [vp, c_i64, c_i64, vp, c_i32, vp]),
    "dsvg_defer_scope": (c_i32, [c_i32, vp]),
    "dsvg_flush_deferred": (c_i32, [vp]),
    "dsvg_defer_zero": (c_i32, [vp, c_i64, vp]),
    "dsvg_gemm_group_scope": (c_i32, [c_i32, vp]),
    "dsvg_colsum": (c_i32, [c_i32, vp, c_i64, c_i64, c_i32, vp, c_i32, c_f32, c_u32, vp, vp, c_i64, vp]),
    "dsvg_colsum_workspace_bytes": (c_i64, [c_i64, c_i32]),
    "dsvg_layernorm_fwd": (c_i32, [c_i32, vp, vp, vp, vp, vp, vp, c_i64, c_i32, c_f32, vp]),
    "dsvg_layernorm_bwd": (c_i32, [c_i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, c_i32, c_i64, c_i32, vp, c_i64, vp]),
    "dsvg_layernorm_bwd_workspace_bytes": (c_i64, [c_i64, c_i32]),
    "dsvg_layernorm_bwd_masked": (c_i32, [c_i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, c_i32, c_i64, c_i32, vp, c_i64, vp, c_f32,
                                          c_u32, vp, vp]),
    "dsvg_attention_fwd": (c_i32, [c_i32, vp, vp, vp, c_i64, vp, vp, c_i64, c_i32, c_i32, c_f32, c_f32, c_u32, vp, vp]),
    "dsvg_attention_bwd": (c_i32, [c_i32, vp, vp, vp, c_i64, vp, vp, vp, c_i64, c_i32, c_i32, c_f32, c_f32, c_u32, vp, vp]),
    "dsvg_attention_tiles": (c_i32, [vp, c_i64, c_i32, vp, vp, vp]),
    "dsvg_pack_tokens": (c_i32, [vp, vp, vp, c_i64, c_i32, c_i32, vp, vp, vp, vp, vp]),
    "dsvg_visible_first": (c_i32, [vp, c_i64, vp, vp, vp, vp]),
    "dsvg_head_pack_elems": (c_i64, [c_i32]),
    "dsvg_head_pack": (c_i32, [vp, c_i32, vp, vp]),
    "dsvg_head_argmax": (c_i32, [vp, vp, vp, c_i64, c_i32, c_i32, vp, vp]),
    "dsvg_head_sample": (c_i32, [vp, vp, vp, c_i64, c_i32, c_i32, c_f32, vp, c_u32, vp, vp]),
    "dsvg_head_lse_workspace_bytes": (c_i64, [c_i64]),
    "dsvg_head_lse": (c_i32, [vp, vp, vp, c_i64, c_i32, c_i32, vp, vp, vp, vp, vp, vp, c_i64, vp]),
    "dsvg_head_dlogits": (c_i32, [vp, vp, vp, c_i64, c_i32, c_i32, vp, vp, vp, vp, vp, vp, c_f32, vp, c_i64, vp]),
    "dsvg_attention_bwd_outproj": (c_i32, [vp, vp, vp, c_i64, vp, vp, vp, vp, c_i64, c_i32, c_f32, c_f32, c_u32, vp, vp]),
    "dsvg_attn_pack_bwd_elems": (c_i64, [c_i32]),
    "dsvg_attn_pack_bwd": (c_i32, [vp, vp, c_i32, vp, vp]),
    "dsvg_ffn_wgrad_finish_many": (c_i32, [vp, c_i32, vp]),
    "dsvg_ffn_debug_clock": (c_i32, [vp]),
    "dsvg_gs_debug_clock": (c_i32, [vp]),
    "dsvg_gather_groups": (c_i32, [c_i32, vp, vp, vp, c_i64, c_i32, c_i32, c_i64, vp]),
    "dsvg_build_masks": (c_i32, [vp, c_i64, c_i32, c_i32, c_i32, vp, vp, vp, vp]),
    "dsvg_embed_gather": (c_i32, [c_i32, vp, vp, vp, vp, vp, vp, vp, vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, vp]),
    "dsvg_embed_scatter": (c_i32, [c_i32, vp, vp, vp, vp, vp, vp, vp, vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32,
                                   c_i32, vp, c_i64, vp]),
    "dsvg_embed_scatter_workspace_bytes": (c_i64, [c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32]),
    "dsvg_group_index": (c_i32, [vp, c_i64, c_i32, c_i32, vp, vp]),
    "dsvg_add_pos_fwd": (c_i32, [c_i32, vp, vp, vp, c_i64, c_i32, c_i32, c_f32, c_u32, vp, vp]),
    "dsvg_add_pos_bwd": (c_i32, [c_i32, vp, vp, vp, c_i32, c_i64, c_i32, c_i32, c_f32, c_u32, vp, vp, c_i64, vp]),
    "dsvg_add_pos_bwd_workspace_bytes": (c_i64, [c_i64, c_i32, c_i32]),
    "dsvg_masked_mean_fwd": (c_i32, [c_i32, vp, vp, vp, vp, c_i64, c_i32, c_i32, vp]),
    "dsvg_masked_mean_bwd": (c_i32, [c_i32, vp, vp, vp, c_i64, vp, c_i64, c_i32, c_i32, vp]),
    "dsvg_bcast_add_fwd": (c_i32, [c_i32, vp, vp, c_i64, c_i32, c_i32, c_f32, c_u32, vp, vp]),
    "dsvg_bcast_add_bwd": (c_i32, [c_i32, vp, vp, c_i64, c_i64, c_i32, c_i32, c_f32, c_u32, vp, c_i64, vp]),
    "dsvg_bcast_add_bwd_masked": (c_i32, [vp, vp, vp, c_i64, c_i64, c_i32, c_i32, c_i64, c_f32, c_u32, c_u32, vp, c_i64, vp]),
    "dsvg_loss_targets": (c_i32, [vp, vp, vp, c_i64, c_i32, c_i32, c_i32, c_i32, vp, vp, vp, vp, vp, vp, vp]),
    "dsvg_masked_ce_fwd": (c_i32, [c_i32, vp, c_i64, c_i32, vp, vp, c_i64, c_i32, vp, vp, vp, c_i64, vp, vp]),
    "dsvg_masked_ce_workspace_bytes": (c_i64, [c_i64]),
    "dsvg_masked_ce_bwd": (c_i32, [c_i32, vp, c_i64, c_i32, vp, vp, vp, vp, vp, c_f32, vp, c_i64, c_i64, c_i32, vp, c_i32, vp]),
    "dsvg_live_rows": (c_i32, [vp, c_i64, c_i32, vp, vp, vp, c_i64, vp]),
    "dsvg_live_rows_workspace_bytes": (c_i64, [c_i64]),
    "dsvg_scatter_rows": (c_i32, [c_i32, vp, vp, vp, c_i64, c_i32, c_i32, vp]),
    "dsvg_sumsq": (c_i32, [vp, c_i64, vp, vp, c_i64, vp]),
    "dsvg_sumsq_workspace_bytes": (c_i64, [c_i64]),
    "dsvg_adamw_step": (c_i32, [vp, vp, vp, vp, c_i64, vp, c_f32, c_f32, c_f32, c_f32, vp, vp, c_f32, c_f32, vp]),
    "dsvg_cast_weights": (c_i32, [c_i32, vp, vp, vp, c_i64, c_i64, vp]),
    "dsvg_advance_step": (c_i32, [vp, vp, vp]),
    "dsvg_pack_images": (c_i32, [vp, vp, c_i64, vp, c_i32, vp, vp, vp, vp, vp, c_i32, vp, vp, vp, c_i32, vp, vp, vp, vp, vp]),
    "dsvg_copy_many": (c_i32, [vp, vp, vp, c_i32, vp]),
    "dsvg_loss_combine_fwd": (c_i32, [vp, vp, c_i32, vp, vp]),
    "dsvg_loss_combine_bwd": (c_i32, [vp, vp, vp, c_i32, vp, vp]),
    "dsvg_gate_mul": (c_i32, [c_i32, vp, vp, vp, c_i64, c_f32, vp]),
    "dsvg_add": (c_i32, [c_i32, vp, vp, vp, c_i64, vp]),
    "dsvg_drop_apply": (c_i32, [c_i32, vp, vp, c_i64, c_f32, c_u32, vp, vp]),
    "dsvg_assemble_batch": (c_i32, [vp, c_i64, vp, c_i64, vp, c_i64, c_i32, c_i32, c_i32, c_f32, c_i32, vp, vp, vp,
                                    vp]),
    "dsvg_attention_causal_fwd": (c_i32, [c_i32, vp, vp, vp, c_i64, c_i32, c_i32, c_f32, c_f32, c_u32, vp, vp]),
    "dsvg_attention_causal_bwd": (c_i32, [c_i32, vp, vp, vp, vp, c_i64, c_i32, c_i32, c_f32, c_f32, c_u32, vp, vp]),
    "dsvg_seq_lens": (c_i32, [vp, c_i64, c_i32, c_i32, vp, vp]),
    "dsvg_attention_long_fwd": (c_i32, [c_i32, vp, vp, vp, c_i64, c_i32, c_i32, c_f32, c_i32, c_i32, c_f32, c_u32, vp,
                                        vp]),
    "dsvg_attention_long_bwd": (c_i32, [c_i32, vp, vp, vp, vp, c_i64, c_i32, c_i32, c_f32, c_i32, c_f32, c_u32, vp, vp]),
    "dsvg_prefix_mean_fwd": (c_i32, [c_i32, vp, vp, vp, c_i64, c_i32, c_i32, vp]),
    "dsvg_prefix_mean_bwd": (c_i32, [c_i32, vp, vp, vp, c_i64, c_i32, c_i32, vp]),
    "dsvg_match_costs": (c_i32, [c_i32, vp, c_i64, vp, c_i64, vp, c_i64, vp, vp, vp, c_i64, c_i32, c_i32, c_i32, c_i32,
                                 c_i32, c_i32, c_i32, c_f32, c_f32, c_f32, vp, vp, vp]),
    "dsvg_match_assign": (c_i32, [vp, vp, c_i64, c_i32, c_i32, vp, vp, vp, vp]),
    "dsvg_argmax_rows": (c_i32, [c_i32, vp, c_i64, c_i32, c_i64, c_i32, vp, vp]),
    "dsvg_sample_rows": (c_i32, [c_i32, vp, c_i64, c_i32, c_i64, c_i32, c_f32, vp, c_u32, vp, vp]),
    "dsvg_ffn_pack_bytes": (c_i64, [c_i32, c_i32]),
    "dsvg_ffn_pack": (c_i32, [vp, vp, c_i32, c_i32, c_i32, vp, vp, vp, vp, vp]),
    "dsvg_ffn_fwd": (c_i32, [vp, vp, vp, vp, vp, vp, vp, vp, c_i64, c_f32, c_f32, c_u32, c_u32, vp, c_i32, vp]),
    "dsvg_ffn_bwd": (c_i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, c_i64, c_f32, c_f32, c_u32, c_u32, vp, vp]),
    "dsvg_attn_bwd_dx_workspace_bytes": (c_i64, [c_i64]),
    "dsvg_attn_bwd_dx_debug_clock": (c_i32, [vp]),
    "dsvg_attn_bwd_dx": (c_i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, c_i32, c_i64, vp, c_i64, vp, c_f32, c_u32, vp, vp]),
    "dsvg_ffn_bwd_dx": (c_i32, [vp, vp, vp, vp, vp, c_i64, c_f32, vp, c_f32, c_u32, vp, vp]),
    "dsvg_ffn_wgrad_finish": (c_i32, [vp] * 12),
    "dsvg_probe_trread": (c_i32, [vp, vp, vp]),
    "dsvg_attn_pack_bytes": (c_i64, [c_i32]),
    "dsvg_attn_pack": (c_i32, [vp, vp, c_i32, c_i32, c_i32, vp, vp]),
    "dsvg_attn_block_fwd": (c_i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, c_i64, c_i32, c_i64, vp, vp, vp, vp, vp, vp,
                                    c_f32, c_f32, c_f32, c_u32, c_u32, vp, vp, c_i64, c_u32, vp]),
    "dsvg_gs_pack_bytes": (c_i64, [c_i32]),
    "dsvg_gs_pack": (c_i32, [vp, vp, c_i32, c_i32, c_i32, c_i32, vp, vp, vp]),
    "dsvg_gs_layer_fwd": (c_i32, [vp] * 12 + [c_i64, c_i64, c_i32] + [vp] * 11 + [c_f32, c_f32, c_f32, c_u32, vp, c_i64, c_i32, vp]),
    "dsvg_gs_bwd_workspace_bytes": (c_i64, [c_i64, c_i32]),
    "dsvg_latent_chain_fwd": (c_i32, [vp, vp, vp, c_i32, vp, vp, vp, c_i64, vp]),
    "dsvg_latent_chain_bwd": (c_i32, [vp, vp, vp, c_i32, vp, vp, c_i64, vp]),
    "dsvg_gs_layer_bwd": (c_i32, [vp] * 13 + [c_i64, c_i32] + [vp] * 10 + [c_f32, c_f32, c_u32, vp, vp, c_i64, vp, vp]),
    # (layers: pointer to an array of GsFwdLayer / GsBwdLayer below)
    "dsvg_gs_stack_fwd": (c_i32, [vp, vp, c_i32, vp, c_i64, c_i64, c_i32, c_f32, c_f32, c_f32, vp, vp]),
    "dsvg_gs_stack_bwd": (c_i32, [vp, vp, c_i32, vp, c_i64, c_i32, c_f32, c_f32, vp, c_i64, c_i64, vp]),
}

_lib = None


class DsvgError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and attach the signatures.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DsvgError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c \"import __graft_entry__ as g; "
            f"g.build()\"` (needs hipcc); there is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    built = lib.dsvg_version()
    if built != ABI_VERSION:
        raise DsvgError(f"{LIB_PATH} was built with DSVG_ABI_VERSION {built}, this binding expects {ABI_VERSION}: rebuild "
                        f"the library (deepsvg_amd/csrc/build.sh)")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().dsvg_last_error()
        raise DsvgError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")

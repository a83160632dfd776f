"""ctypes binding of libogmm_hip.so (include/ogmm_hip.h).  No fallback: if the library is missing or a
symbol does not resolve, importing the ops fails loudly -- the product path has no CPU route."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libogmm_hip.so")

ABI_VERSION = 28

ACT_NONE, ACT_RELU, ACT_LEAKY02, ACT_SIGMOID = 0, 1, 2, 3
PREC_F32, PREC_F16X3, PREC_F16X3_FRAG, PREC_F16_FRAG = 0, 1, 2, 3
STATUS_EDGECONV_PROTOCOL, STATUS_EM_EXIT_PROTOCOL = 2, 4          # bits of the device status word (include/ogmm_hip.h)


class GemmDesc(Structure):
    """Mirror of `struct ogmm_gemm`."""
    _fields_ = [
        ("A", c_void_p), ("lda", c_int64), ("K1", c_int32),
        ("A2", c_void_p), ("lda2", c_int64), ("K2", c_int32),
        ("B", c_void_p), ("ldb", c_int64),
        ("C", c_void_p), ("ldc", c_int64),
        ("Res", c_void_p), ("ldr", c_int64),
        ("M", c_int32), ("N", c_int32),
        ("batch_outer", c_int32), ("batch_inner", c_int32),
        ("sA_o", c_int64), ("sA_i", c_int64), ("sA2_o", c_int64), ("sA2_i", c_int64),
        ("sB_o", c_int64), ("sB_i", c_int64), ("sC_o", c_int64), ("sC_i", c_int64),
        ("sR_o", c_int64), ("sR_i", c_int64),
        ("scale", c_void_p), ("shift", c_void_p), ("row_affine", c_int32),
        ("alpha", c_float),
        ("act", c_int32),
        ("pool_k", c_int32), ("pool_out", c_void_p), ("ldp", c_int64), ("store_c", c_int32),
        ("precision", c_int32), ("B_hi", c_void_p), ("B_lo", c_void_p), ("ldb_h", c_int64), ("overflow", c_void_p),
        ("col_stats", c_void_p), ("a_scale", c_void_p), ("a_shift", c_void_p), ("a_relu", c_int32), ("group_rows", c_int32),
        ("ovl_orow", c_void_p), ("ovl_ocol", c_void_p), ("ovl_ld", c_int64), ("ovl_rowpart", c_void_p), ("ovl_colpart", c_void_p), ("row_rscale", c_void_p),
        ("rd_w", c_void_p), ("rd_b", c_void_p), ("rd_act", c_int32), ("rd_out", c_void_p), ("rd_ld", c_int64),
        ("a_gather_ids", c_void_p), ("a_gather_map", c_void_p), ("a_gather_S", c_int32), ("a_gather_N", c_int32), ("a_gather_rows", c_int64),
        ("col_stats_slot_mask", c_int32), ("col_stats_slot_stride", c_int64),
        ("terms", c_int32),
        ("nb_mean", c_void_p), ("nb_rstd", c_void_p), ("nb_scale", c_void_p), ("nb_shift", c_void_p), ("nb_act", c_int32),
        ("a_trans", c_int32), ("a_colsum", c_void_p),
    ]


# name -> argtypes (restype is int for all but the two noted); the list doubles as the export check in tests
PROTOTYPES = {
    "ogmm_abi_version": [],
    "ogmm_last_error": [],
    "ogmm_topk_rows": [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "ogmm_knn": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "ogmm_pack_clouds": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p],
    "ogmm_knn_pos_head_supported": [c_int, c_int],
    "ogmm_knn_pos_head_workspace_bytes": [c_int, c_int],
    "ogmm_knn_pos_head": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogmm_fps": [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "ogmm_gather_rows": [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p],
    "ogmm_gemm_nt": [POINTER(GemmDesc), c_void_p],
    "ogmm_gemm_overlap_fusable": [c_int, c_int, c_int],
    "ogmm_gemm_rowdot_fusable": [c_int, c_int, c_int, c_int],
    "ogmm_gemm_normbwd_fusable": [c_int, c_int, c_int, c_int],
    "ogmm_gemm_atrans_supported": [c_int, c_int, c_int, c_int64, c_int],
    "ogmm_gemm_gather_fusable": [c_int, c_int, c_int, c_int64],
    "ogmm_overlap_finalize": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p],
    "ogmm_row_rnorm": [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p],
    "ogmm_edgeconv_first": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p],
    "ogmm_edgeconv_fused": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p] +
                           [c_void_p, c_void_p, c_void_p, c_void_p, c_float] * 3 + [c_void_p, c_int64, c_void_p],
    "ogmm_edgeconv_pc": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p] +
                        [c_void_p, c_void_p, c_void_p, c_void_p, c_float] * 3 + [c_void_p, c_int64, c_void_p, c_void_p],
    "ogmm_pos_hidden": [c_void_p, c_void_p, c_int, c_int, c_int, c_int] + [c_void_p] * 6 + [c_void_p, c_void_p, c_void_p],
    "ogmm_attention_workspace_bytes": [c_int, c_int, c_int, c_int],
    "ogmm_attention": [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int64, c_void_p, c_void_p],
    "ogmm_attention_terms": [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int64, c_int, c_void_p, c_void_p],
    "ogmm_add_n": [c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p],
    "ogmm_attention_bwd_supported": [c_int, c_int],
    "ogmm_attention_bwd": [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_float,
                           c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p],
    "ogmm_attention_bwd_f16x3": [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_float,
                                 c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p],
    "ogmm_softmax_rows": [c_void_p, c_int64, c_int, c_int64, c_void_p],
    "ogmm_instnorm_relu": [c_void_p, c_int64, c_int, c_int, c_int, c_float, c_void_p],
    "ogmm_instnorm_finalize": [c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p],
    "ogmm_l2norm_pack_frag": [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p],
    "ogmm_l2norm_pack_frag_rnorm": [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p],
    "ogmm_pack_frag": [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p],
    "ogmm_l2norm_rows": [c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p],
    "ogmm_rowdot": [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p],
    "ogmm_overlap_cross": [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p],
    "ogmm_overlap_cross_workspace_bytes": [c_int, c_int],
    "ogmm_overlap_cross_ws": [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p],
    "ogmm_gmm_em_exit_workspace_bytes": [c_int, c_int, c_int, c_int, c_int],
    "ogmm_gmm_em_chip_max_group": [c_int, c_int],
    "ogmm_gmm_em_chip_cached": [c_int, c_int],
    "ogmm_gmm_em": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_double, c_int,
                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogmm_gmm_em_workspace_bytes": [c_int, c_int, c_int],
    "ogmm_gmm_em_multi": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_double, c_int,
                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogmm_gmm_feat_mean": [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    "ogmm_match_kabsch": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogmm_kabsch": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "ogmm_clu_infonce": [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p],
    "ogmm_icp_point_to_point": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_int, c_double, c_double,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogmm_min_sqdist": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "ogmm_rotation_from_cov": [c_void_p, c_int, c_void_p, c_void_p],
    "ogmm_icp_workspace_bytes": [c_int, c_int],
    "ogmm_icp_point_to_point_ws": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_int, c_double, c_double,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    # training mode
    "ogmm_norm_finalize": [c_void_p, c_int64, c_int, c_int64, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogmm_norm_param_grads": [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "ogmm_bn_update_running": [c_void_p, c_void_p, c_int, c_int, c_int64, c_float, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogmm_colstats": [c_void_p, c_int64, c_int64, c_int, c_int64, c_void_p, c_void_p],
    "ogmm_affine_act": [c_void_p, c_int64, c_int64, c_int, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p],
    "ogmm_norm_bwd_reduce": [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int64, c_int, c_int64,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p],
    "ogmm_norm_bwd_apply": [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int64, c_int, c_int64,
                            c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p],
    "ogmm_affine_act_pool": [c_void_p, c_int64, c_int64, c_int, c_int, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64,
                             c_void_p, c_void_p],
    "ogmm_maxpool_k": [c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p],
    "ogmm_maxpool_k_bwd": [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p, c_int64, c_void_p],
    "ogmm_overlap_cross_train": [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p],
    "ogmm_overlap_cross_bwd": [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                               c_void_p, c_void_p, c_void_p, c_int64, c_void_p],
    "ogmm_weight_grad_thin_streams": [c_int, c_int],
    "ogmm_weight_grad_thin": [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p],
    "ogmm_weight_grad_thin_f16x3": [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "ogmm_kabsch_bwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogmm_pow2_scale": [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "ogmm_nearest_point": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "ogmm_edge_features": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "ogmm_pos_features": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "ogmm_small_bmm_nn": [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_int64, c_int64, c_void_p],
    "ogmm_scatter_add_rows": [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p],
    "ogmm_small_bmm_nt": [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int64, c_int64, c_void_p],
    "ogmm_l2norm_rows_bwd": [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p],
    "ogmm_debug_edgeconv_probe": [c_void_p],
    "ogmm_debug_edgeconv_pc_probe": [c_void_p],
    "ogmm_transpose_pad": [c_void_p, c_int64, c_int64, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "ogmm_split_weight": [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p],
    "ogmm_pack_frag_t": [c_void_p, c_int64, c_int64, c_int, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64,
                         c_void_p],
}

_lib = None


class OgmmError(RuntimeError):
    pass


def load():
    """Loads the in-tree shared library once and checks ABI version and symbols."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise OgmmError("libogmm_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(or `make -C ogmm_amd/csrc`); there is no CPU fallback for the product path")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError here = missing export
        fn.argtypes = argtypes
        fn.restype = c_char_p if name == "ogmm_last_error" else (c_int64 if name.endswith(("_bytes", "_streams")) else c_int)
    if lib.ogmm_abi_version() != ABI_VERSION:
        raise OgmmError("libogmm_hip.so ABI %d != binding ABI %d" % (lib.ogmm_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise OgmmError("%s failed: %s" % (name, lib.ogmm_last_error().decode(errors="replace")))

"""ctypes loader for libjxgpu.so (the HIP/C-ABI product library).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C janusx_amd/csrc``.  There is no CPU
fallback: if the shared object is missing or a call fails, a RuntimeError is raised (the reference raises
RuntimeError from its PyO3 layer too, e.g. src/stats/grm.rs:3079-3086).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libjxgpu.so")
_LIB = None

c_p = C.c_void_p
c_i = C.c_int
c_l = C.c_int64
c_d = C.c_double
c_f = C.c_float

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/jxgpu.h one to one
SIGNATURES = {
    "jx_last_error": [],
    "jx_version": [],
    "jx_set_progress": [c_p, c_p, c_l],
    "jxg_device_count": [],
    "jxg_set_device": [c_i],
    "jxg_device_info": [c_p],
    "jxg_num_tiles": [c_i],
    "jxg_repack_p32": [c_p, c_l, c_i, c_l, c_p, c_i, c_p, c_l, c_p, c_p],
    "jxg_row_counts_p32": [c_p, c_l, c_i, c_p, c_p],
    "jxg_row_counts_raw_masked": [c_p, c_l, c_l, c_p, c_p, c_p],
    "jxg_grm_accumulate": [c_p, c_l, c_i, c_p, c_p, c_l, c_p, c_i, c_i, c_p],
    "jxg_grm_accumulate_rows": [c_p, c_l, c_i, c_p, c_p, c_l, c_p, c_i, c_i, c_i, c_i, c_p],
    "jxg_grm_finalize": [c_p, c_i, c_d, c_p, c_i, c_p],
    "jxg_spgrm_work_bytes": [c_i],
    "jxg_spgrm_count": [c_p, c_i, c_d, c_d, c_i, c_p, c_p, c_p],
    "jxg_spgrm_fill": [c_p, c_i, c_d, c_d, c_i, c_p, c_p, c_p, c_p, c_p],
    "jxg_spgrm_count_bands": [c_p, c_i, c_d, c_d, c_i, c_i, c_i, c_p, c_p, c_p],
    "jxg_spgrm_fill_bands": [c_p, c_i, c_d, c_d, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p],
    "jxg_spgrm_densify": [c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_p],
    "jxg_eigh_f64": [c_p, c_i, c_d, c_p, c_p],
    "jxg_eigh_grid_check": [c_i],
    "jxg_eigh_dist_staging_doubles": [c_i],
    "jxg_eigh_set_dist": [c_i, c_i, c_p, c_p, c_p, c_l, c_i],
    "jxg_eigh_set_local": [c_i],
    "jxg_eigh_set_gather": [c_p, c_p],
    "jxg_eigh_set_agree": [c_p, c_p],
    "jxg_eigh_last_dist_agree": [],
    "jxg_eigh_set_band_dist": [c_i, c_i, c_p, c_p, c_p, c_l, c_i, c_i],
    "jxg_eigh_band_staging_doubles": [c_i],
    "jxg_eigh_last_band_sharded": [],
    "jxg_eigh_last_dc_windowed": [],
    "jxg_dgemm_f64": [c_i, c_i, c_i, c_i, c_i, c_d, c_p, c_l, c_p, c_l, c_d, c_p, c_l, c_i, c_p],
    "jxg_oz_dgemm_f64": [c_i, c_i, c_i, c_i, c_i, c_d, c_p, c_l, c_p, c_l, c_d, c_p, c_l, c_p, c_p],
    "jxg_oz_planes": [],
    "jxg_oz_set_planes": [c_i],
    "jxg_dsymm_lower_f64": [c_i, c_i, c_d, c_p, c_l, c_p, c_l, c_d, c_p, c_l, c_p],
    "jxg_dsyr2k_lower_nt_f64": [c_i, c_i, c_d, c_p, c_l, c_p, c_l, c_d, c_p, c_l, c_p],
    "jxg_sy2st_f64": [c_p, c_i, c_p, c_p, c_p, c_p, c_p],
    "jxg_tri_tiles_doubles": [c_i],
    "jxg_tri_tiles_pack_f64": [c_p, c_i, c_p, c_i, c_p],
    "jxg_symmetrize_f64": [c_p, c_i, c_p],
    "jxg_transpose_f64": [c_p, c_p, c_i, c_p],
    "jxg_gather_sub_f64": [c_p, c_i, c_i, c_p, c_i, c_p, c_p],
    "jxg_cast_f64_to_f32": [c_p, c_p, c_l, c_p],
    "jxg_rotate_xy": [c_p, c_i, c_p, c_i, c_p, c_p],
    "jxg_lmm_reml_null": [c_p, c_p, c_p, c_i, c_i, c_d, c_d, c_i, c_d, c_p, c_p],
    "jxg_ut_split": [c_p, c_i, c_p, c_p, c_i, c_p],
    "jxg_rotate_packed": [c_p, c_l, c_i, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p],
    "jxg_lut_split": [c_p, c_l, c_p, c_p],
    "jxg_rotate_packed16": [c_p, c_l, c_i, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p],
    "jxg_lut_split_rows": [c_p, c_l, c_i, c_p, c_p, c_l, c_p, c_p, c_p],
    "jxg_ut_rowsum": [c_p, c_i, c_p, c_p],
    "jxg_rotate_packed16x": [c_p, c_l, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p],
    "jxg_rotate_packed16x_ld": [c_p, c_l, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_l, c_p],
    "jxg_ut_quant3": [c_p, c_i, c_p, c_p, c_p],
    "jxg_rotate_packed16x_q": [c_p, c_l, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_p],
    "jxg_rotate_dense_f32": [c_p, c_i, c_i, c_p, c_p, c_p],
    "jxg_lmm_scan": [c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_d, c_d, c_d, c_i, c_i, c_d, c_i, c_d, c_p, c_p, c_p],
    "jxg_lmm_tables_bytes": [c_i, c_i, c_d, c_d],
    "jxg_lmm_tables_build": [c_p, c_p, c_p, c_i, c_i, c_d, c_d, c_p, c_p],
    "jxg_lmm_scan_tab": [c_p, c_i, c_i, c_p, c_p, c_i, c_d, c_d, c_p, c_d, c_i, c_i, c_d, c_i, c_d, c_p, c_p, c_p],
    "jxg_lmm_scan_exact": [c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_d, c_d, c_d, c_i, c_i, c_d, c_i, c_d, c_p, c_p, c_p],
    "jxg_fvlmm_prepare": [c_p, c_p, c_p, c_i, c_i, c_d, c_p, c_p, c_p, c_p, c_p],
    "jxg_fvlmm_scan": [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_d, c_i, c_i, c_d, c_d, c_p, c_p],
    "jxg_lmm2_scan": [c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_d, c_d, c_d, c_i, c_i, c_d, c_d, c_p, c_p],
    "jxg_lmm2_scan_exact": [c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_d, c_d, c_d, c_i, c_i, c_d, c_d, c_p, c_p],
    "jxg_lmm2_null_ml": [c_p, c_p, c_p, c_i, c_i, c_d, c_d, c_i, c_d, c_i, c_d, c_p, c_p],
    "jx_lmm2_chunk": [c_p, c_p, c_p, c_i, c_i, c_d, c_d, c_p, c_l, c_p, c_d, c_i, c_d, c_p],
    "jx_lmm2_null_ml": [c_p, c_p, c_p, c_i, c_i, c_d, c_d, c_i, c_d, c_i, c_d, c_p],
    "jxg_lmm_loglike_null": [c_p, c_p, c_p, c_i, c_i, c_d, c_p, c_p],
    "jx_ml_loglike_null": [c_p, c_p, c_p, c_i, c_i, c_d, c_p],
    "jxg_gblup_fit": [c_p, c_i, c_d, c_p, c_d, c_d, c_d, c_i, c_p, c_p, c_p],
    "jxg_packed_tdot": [c_p, c_l, c_i, c_p, c_i, c_p, c_p, c_p, c_p],
    "jxg_packed_tdot_f32": [c_p, c_l, c_i, c_p, c_i, c_p, c_p, c_p, c_p],
    "jxg_t32_bytes": [c_i, c_i],
    "jxg_p32_transpose": [c_p, c_l, c_i, c_p, c_i, c_p, c_p],
    "jxg_packed_dot_t32": [c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p],
    "jxg_packed_dot": [c_p, c_l, c_i, c_p, c_i, c_p, c_p, c_p, c_p],
    "jxg_cross_dot": [c_p, c_i, c_l, c_p, c_i, c_p, c_i, c_p, c_d, c_p, c_p],
    "jxg_splmm_exact_scan_dev": [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_d, c_i, c_p, c_p],
    "jxg_fvlmm_scan_dev": [c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_d, c_i, c_i, c_d, c_d, c_p, c_p],
    "jxg_last_kernel_ms": [c_i],
    "jx_assoc_tsv_write": [C.c_char_p, c_p, c_p, c_l, c_p, c_p, c_p, c_i],
    "jx_assoc_tsv_append": [C.c_char_p, c_p, c_p, c_l, c_p, c_p, c_p, c_i, c_i],
    "jx_row_counts": [c_p, c_l, c_i, c_p, c_i, c_p],
    "jx_grm_packed": [c_p, c_l, c_i, c_p, c_p, c_p, c_i, c_i, c_p, c_i, c_p, c_p],
    "jx_grm_stream_payload_f32": [c_p, c_l, c_i, c_i, c_f, c_f, c_f, c_p, c_p, c_p],
    "jx_eigh_f64": [c_p, c_i, c_d, c_p, c_p],
    "jx_lmm_rotate_x_y_with_ut_f64": [c_p, c_i, c_p, c_i, c_p, c_p, c_p],
    "jx_lmm_reml_null": [c_p, c_p, c_p, c_i, c_i, c_d, c_d, c_i, c_d, c_p],
    "jx_lmm_reml_chunk": [c_p, c_p, c_p, c_i, c_i, c_d, c_d, c_p, c_l, c_p, c_i, c_d, c_i, c_d, c_p],
    "jx_fvlmm_assoc_chunk": [c_p, c_p, c_p, c_i, c_i, c_d, c_p, c_l, c_p, c_i, c_d, c_p],
    "jx_gblup_reml_grm": [c_p, c_i, c_l, c_p, c_i, c_p, c_p, c_i, c_d, c_d, c_d, c_i, c_d, c_i, c_p, c_p, c_p],
    "jx_spgrm_packed_to_jxgrm": [c_p, c_l, c_i, c_p, c_p, c_p, c_i, c_i, c_d, c_i, c_i, C.c_char_p, c_p, c_p],
    "jx_spgrm_set_part": [c_i, c_i],
    "jx_spgrm_merge_parts": [C.c_char_p, c_i, c_i, c_p],
    "jx_he_traces_packed": [c_p, c_l, c_i, c_p, c_l, c_p, c_p, c_i, c_p, c_p, c_i, c_i, C.c_uint64, c_i, c_d, c_p],
    "jx_pcg_set_dist": [c_i, c_i, c_p, c_p, c_p, c_l],
    "jx_pcg_dist_count": [],
    "jx_pcg_image_scope": [c_i],
    "jxg_scratch_trim": [],
    "jx_rrblup_pcg_packed": [c_p, c_l, c_i, c_p, c_l, c_p, c_p, c_i, c_p, c_p, c_i, c_d, c_d, c_i, c_p, c_p, c_p, c_p],
    "jx_rrblup_exact_snp_packed": [c_p, c_l, c_i, c_p, c_l, c_p, c_p, c_i, c_p, c_p, c_i, c_d, c_d, c_d, c_i, c_p, c_p,
                                   c_p, c_p],
    "jx_assoc_packed": [c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_d, c_d, c_i, c_d, c_i,
                        c_d, c_i, c_d, c_p],
    "jxg_sps_ldr": [c_i],
    "jxg_sps_work_doubles": [c_i, c_i],
    "jxg_sps_rows_to_cols_f64": [c_p, c_i, c_i, c_l, c_p, c_i, c_p],
    "jxg_sps_solve_multi": [c_i, c_p, c_p, c_p, c_d, c_p, c_p, c_i, c_i, c_d, c_i, c_p, c_p, c_p, c_p],
    "jxg_sps_scan_sums": [c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_p],
    "jx_assoc_packed_chain": [c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_d, c_d, c_i, c_d, c_i,
                              c_d, c_i, c_d, c_p, c_i, c_p, c_l],
    "jxg_lmm_scan_chain_tab": [c_p, c_i, c_i, c_p, c_p, c_i, c_d, c_d, c_p, c_d, c_i, c_p, c_i, c_p, c_i, c_d, c_p, c_p, c_p],
    "jxg_lmm_scan_chain": [c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_d, c_d, c_d, c_i, c_p, c_i, c_p, c_i, c_d, c_p, c_p, c_p],
    "jxg_lmm_scan_exact_chain": [c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_d, c_d, c_d, c_i, c_p, c_i, c_p, c_i, c_d, c_p, c_p, c_p],
    "jxg_lmm_series_doubles": [c_i, c_d, c_d],
    "jxg_lmm_series_coef_tab": [c_p, c_i, c_i, c_p, c_i, c_d, c_d, c_p, c_p, c_p, c_p],
    "jxg_lmm_series_brent_tab": [c_i, c_i, c_p, c_p, c_i, c_d, c_d, c_p, c_d, c_i, c_i, c_d, c_p, c_p, c_p, c_i, c_p, c_i, c_d,
                                 c_p, c_p, c_p],
    "jx_assoc_packed_gm": [c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_d, c_d, c_i, c_d, c_i,
                           c_d, c_i, c_d, c_p, c_i],
    "jxg_rotate_packed16x_fused": [c_p, c_l, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_i,
                                   c_p],
    "jxg_fvlmm_finish_dev": [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_d, c_i, c_i, c_d, c_d, c_i, c_p, c_p],
    "jxg_lm_scan_p32": [c_p, c_l, c_i, c_p, c_i, c_p, c_p, c_i, c_p, c_d, c_p, c_p, c_p],
    "jxg_splmm_grammar_scan_p32": [c_p, c_l, c_i, c_p, c_i, c_p, c_p, c_i, c_p, c_d, c_d, c_d, c_p, c_p, c_p],
    "jxg_splmm_gamma_sums": [c_p, c_i, c_i, c_l, c_i, c_p, c_p, c_p, c_p, c_p],
    "jx_lm_residualize": [c_p, c_p, c_p, c_i, c_i, c_p, c_p],
    "jx_lm_assoc_packed": [c_p, c_p, c_p, c_i, c_p, c_l, c_i, c_p, c_p, c_p, c_i, c_p],
    "jx_lm_assoc_dense": [c_p, c_p, c_p, c_i, c_p, c_l, c_i, c_p],
    "jxg_lm_scan_dense": [c_p, c_i, c_i, c_l, c_p, c_i, c_p, c_d, c_p, c_p, c_p],
    "jxg_decode_rows_p32": [c_p, c_l, c_i, c_p, c_i, c_p, c_p, c_l, c_p],
    "jxg_rot_miss_max": [c_i, c_d],
    "jxg_rotate_missing_dense": [c_p, c_l, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_l, c_p],
    "jxg_lut_split_rows_m": [c_p, c_l, c_i, c_p, c_p, c_l, c_p, c_p, c_p, c_i, c_p],
    "jxg_transpose_f32": [c_p, c_i, c_p, c_p],
    "jxg_rotate_missing_correct": [c_p, c_l, c_i, c_p, c_i, c_p, c_p, c_p, c_l, c_p],
}
_RESTYPES = {"jx_last_error": C.c_char_p, "jxg_last_kernel_ms": C.c_float, "jxg_lmm_tables_bytes": C.c_int64,
             "jxg_t32_bytes": C.c_int64, "jxg_eigh_dist_staging_doubles": C.c_int64, "jxg_eigh_band_staging_doubles": C.c_int64,
             "jxg_spgrm_work_bytes": C.c_int64, "jxg_tri_tiles_doubles": C.c_int64, "jx_assoc_tsv_write": C.c_int64,
             "jx_assoc_tsv_append": C.c_int64, "jx_pcg_dist_count": C.c_int64, "jxg_scratch_trim": C.c_int64,
             "jxg_lmm_series_doubles": C.c_int64, "jxg_sps_work_doubles": C.c_int64}


def lib():
    """Load libjxgpu.so (once). Raises RuntimeError when it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C janusx_amd/csrc` (there is no CPU fallback)")
        # PyTorch-ROCm wheels bundle their own libamdhip64/rocBLAS/rocSOLVER under the same SONAMEs as /opt/rocm.
        # If libjxgpu pulled the /opt/rocm copies in first and torch arrived later, the process would hold two
        # HIP runtimes (observed on the GPU box: "no ROCm-capable device is detected" at the first launch).
        # Importing torch first, when it is installed, keeps one runtime per process; without torch the
        # /opt/rocm runtime is used and everything works standalone.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        h = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(h, name)
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, C.c_int)
        _LIB = h
    return _LIB


def check(status: int) -> None:
    if status != 0:
        msg = lib().jx_last_error()
        raise RuntimeError(msg.decode("utf-8", "replace") if msg else f"libjxgpu call failed ({status})")

"""Every `JXGPU_*` environment switch of the library, classified (tests/test_abi_and_host.py keeps this table complete: a name
that appears in the sources and not here fails the CPU suite).

The SUPPORTED product state is the one with none of them set; JX_LMM_UNIFIED_NO_WARM_START (the reference's own variable) is the
only environment setting that is part of the product's documented behaviour.  Classes:
  knob      product configuration (resources, routes by size, multi-rank modes): results agree within the parity bars on every value
  numerics  selects an arithmetic form whose RESULT differs beyond summation order; the value named under `default` is the product,
            every other value is a diagnostic form, covered by the test named (or marked untested = not a product state)
  form      an alternative kernel / launch shape computing the same numbers (same bits or summation order only)
  trace     prints timings
  test      hooks for the test-suite and bench.py
"""

SWITCHES = {
    # ---- numerics: the result changes with the value ------------------------------------------------------------------------
    "JXGPU_OZ_PLANES": ("numerics", "6 (5 inside pipeline.eigh_from_grm(f32_consumer=True))",
                        "digit planes of the sliced int8 products of the eigensolver (4..6); 4 is NOT a product state",
                        "test_sliced_int8_gemm[5|6], test_full_size_c2/c3/c4_properties[f32_consumer], test_end_to_end_two_stage"),
    "JXGPU_EIGH_F32_PLANES": ("numerics", "5", "0 keeps 6 planes for the f32-consuming pipeline",
                              "test_full_size_c3_properties[0.0-False] (6) / [0.0-True] (5)"),
    "JXGPU_STEDC_OZ_PLANES": ("numerics", "unset (follows JXGPU_OZ_PLANES)", "planes of the divide-and-conquer merges alone", "untested"),
    "JXGPU_GRM_MISS_DIGITS": ("numerics", "3", "2 = the two-digit dense missing-call GRM of rounds 4-5 (beta off by 1e-5..1e-4 end to end)",
                              "test_end_to_end_two_stage (default); 2 is NOT a product state"),
    "JXGPU_GRM_MISS": ("numerics", "1", "0: rows with missing calls on the fp16 split kernel (f32 accumulation, like the reference)",
                       "test_grm_missing_calls_sparse_correction"),
    "JXGPU_GRM_MISS_MAX": ("numerics", "0.012", "largest missing share for the sparse correction", "test_grm_missing_calls_sparse_correction"),
    "JXGPU_GRM_MISS_DENSE_MIN": ("numerics", "0.0015", "missing share from which the dense digit form replaces the sparse correction",
                                 "test_grm_missing_calls_sparse_correction, scripts/diag_e2e_two_stage.py"),
    "JXGPU_GRM_MISS_DENSE_ROWS": ("numerics", "16384", "fewest SNPs for the dense digit form", "test_grm_missing_calls_sparse_correction"),
    "JXGPU_GRM_FP4": ("form", "0", "1: the count Gram of the 256-tile form on the fp4 matrix pipes from a nibble image (exact like the int8 kernel)",
                      "test_grm_count_gram_on_the_fp4_pipes"),
    "JXGPU_GRM_I8": ("numerics", "1", "0: exact-integer SNPs on the fp16 single-product kernel (f32 sums instead of exact i32)", "untested"),
    "JXGPU_GRM_EXACT": ("numerics", "1", "0: every SNP on the fp16 hi/lo split kernel", "untested"),
    "JXGPU_ROT_I8": ("numerics", "1", "0: design rows on the fp16 hi/lo rotation instead of the exact int8 planes",
                     "test_full_size_rotation_kernels_with_mixed_rows"),
    "JXGPU_ROT_EXACT": ("numerics", "1", "0: no integer-LUT fast path in the fp16 rotation", "untested"),
    "JXGPU_ROT_MISS_DENSE": ("numerics", "auto", "form of the missing-call term of the int8 rotation (both exact)",
                             "test_rotation_rows_with_a_few_missing_calls_take_the_exact_path"),
    "JXGPU_ROT_MISS_MAX": ("numerics", "auto", "missing calls per row up to which the gather form is taken",
                           "test_rotation_rows_with_a_few_missing_calls_take_the_exact_path"),
    "JXGPU_PCG_I8": ("numerics", "1", "0: the PCG / HE operator halves on the f32 bit-plane table kernels (f32 partial sums per tile) instead of "
                     "exact int8 plane sums of a four-digit image of the vector", "test_pcg_operator_forms_agree"),
    "JXGPU_FVLMM_FUSED": ("numerics", "1", "0 / 2: unfused / always-fused fixed-lambda scan (f64 sums in a different order + f32 G~ round trip)",
                          "test_fixed_lambda_scan_fused_into_the_rotation_epilogue"),
    "JXGPU_SCAN_EXACT": ("numerics", "unset", "reference-formulation scan kernel instead of the tabulated one (1e-9 apart)",
                         "test_fast_scan_matches_exact_scan (through the entry points)"),
    "JXGPU_SCAN_SERIES": ("numerics", "1", "0: no per-SNP Chebyshev series (3e-16 apart)", "test_fast_scan_matches_exact_scan"),
    "JXGPU_SCAN_INTERP": ("numerics", "1", "0: Brent evaluates the tabulated objective directly instead of its per-SNP Chebyshev interpolant "
                          "(3.7e-15 apart; dim <= 4, series form only)",
                          "test_warm_start_chain_device_pipeline_matches_trajectories, test_fast_scan_matches_exact_scan, full-size legs"),
    "JXGPU_LMM2_EXACT": ("numerics", "unset", "LMM2 on the reference-formulation kernel", "test_lmm2_routes"),
    "JXGPU_EIGH": ("numerics", "auto", "onestage | twostage | rocsolver: which eigensolver (eigenvectors differ within 1e-10)",
                   "test_eigh_two_stage_path_and_its_fallback"),
    "JXGPU_EIGH_TWOSTAGE_MIN": ("numerics", "1500", "size from which the two-stage reduction runs", "test_eigh_two_stage_path_and_its_fallback"),
    "JXGPU_STEDC": ("numerics", "own", "rocsolver: vendor divide and conquer (diagnostic)", "test_eigh_own_divide_and_conquer"),
    "JXGPU_ORMTR": ("numerics", "own", "rocsolver back-transformation (diagnostic)", "untested"),
    "JXGPU_ORMTR_OZ": ("numerics", "1", "0: Q1 on the f64 MFMA GEMM instead of the sliced int8 one", "untested"),
    "JXGPU_STEDC_OZ": ("numerics", "1", "0: merges on the f64 MFMA GEMM", "untested"),
    "JXGPU_OZ_MIN_N": ("numerics", "3000 / 750", "size from which Q1 / the merges take the sliced products", "untested"),
    # ---- knobs ---------------------------------------------------------------------------------------------------------------
    "JXGPU_DIST_BACKEND": ("knob", "nccl", "collective backend of `jx` multi-rank runs", "test_cli_two_ranks_share_one_gpu"),
    "JXGPU_DIST_EIGH": ("knob", "1", "0: replicated eigendecomposition on several ranks", "test_bench_two_ranks_share_one_gpu"),
    "JXGPU_DIST_EIGH_MIN_N": ("knob", "16384", "size from which the eigensolver's stages are dealt over the ranks", "test_bench_two_ranks_share_one_gpu"),
    "JXGPU_DIST_EIGH_ONESTAGE": ("knob", "unset", "sharded one-stage reduction", "untested"),
    "JXGPU_DIST_BAND": ("knob", "1", "0: band reduction replicated", "untested"),
    "JXGPU_DIST_BAND_MIN_N": ("knob", "16384", "size from which the band reduction is sharded", "test_sharded_band_reduction_rccl_single_rank"),
    "JXGPU_DIST_BAND_BLOCK": ("test", "auto", "block rows of the sharded band reduction", "test_sharded_band_reduction_rccl_single_rank"),
    "JXGPU_DIST_DC_WINDOW": ("knob", "1", "0: top-level merge replicated", "untested"),
    "JXGPU_SYR2K_PIPE_WGS": ("knob", "7/8 of the CUs", "workgroups of the persistent rank-2k update (the rest of the CUs serve the panel chain)",
                             "test_rank_2k_update_forms_give_the_same_bits"),
    "JXGPU_SPLMM_ROUTE": ("knob", "auto", "dense | block | factor: form of K + lambda I of the sparse-GRM routes",
                          "test_splmm_block_route_matches_the_dense_route, test_splmm_giant_component_both_sides_of_the_limit"),
    "JXGPU_SPLMM_BLOCK": ("knob", "4096", "samples per diagonal block of the block route", "test_splmm_block_route_matches_the_dense_route"),
    "JXGPU_SPLMM_BLOCK_MIN_N": ("knob", "16384", "size from which the block route is taken", "untested"),
    "JXGPU_SPGRM_ACC_GB": ("knob", "auto", "HBM for the sparse-GRM builder's accumulator", "untested"),
    "JXGPU_SPECTRAL_CACHE": ("knob", "1", "0: no reuse of the last sparse GRM's decomposition", "untested"),
    # ---- forms (same numbers) -----------------------------------------------------------------------------------------------
    **{k: ("form", "default", "kernel / launch shape of the same computation", t) for k, t in {
        "JXGPU_BC_OWNED": "test_bulge_chasing_position_owned_equals_sweep_owned", "JXGPU_SB2ST_WGS": "untested", "JXGPU_GRM_I8_TILE": "test_sparse_grm_row_panels_write_the_same_file",
        "JXGPU_GRM_TILE": "untested", "JXGPU_GRM_EXACT_BK": "untested", "JXGPU_ROT256": "untested", "JXGPU_SCAN_NOLDS": "untested",
        "JXGPU_SCAN_NOTILE": "untested", "JXGPU_SBBACK_BAL": "untested", "JXGPU_SBBACK_BAL5": "test_eigh_balanced_q2_forms",
        "JXGPU_SBBACK_BAL_MIN": "untested", "JXGPU_SBBACK_BAL_PER": "untested", "JXGPU_SBBACK_GROUPS": "untested", "JXGPU_SBBACK_NW": "untested",
        "JXGPU_SBBACK_PAIR": "untested", "JXGPU_SBBACK_SOLO": "untested", "JXGPU_STEDC_LEAF": "test_eigh_own_divide_and_conquer",
        "JXGPU_STEDC_OWNLEAF": "untested", "JXGPU_STEDC_PAR": "untested", "JXGPU_STEDC_PARMIN": "untested", "JXGPU_SY2SB_LOOKAHEAD": "untested",
        "JXGPU_SYTRD_KT": "untested", "JXGPU_SYTRD_SYR2K": "untested", "JXGPU_SYTRD_TAIL": "untested", "JXGPU_SYTRD_TARGET": "untested",
        "JXGPU_DSYMM_SLOTS": "untested", "JXGPU_DSYMM_SPLIT": "untested", "JXGPU_ORMTR_NB": "untested",
        "JXGPU_ROT_I8_DMA": "test_int8_rotation_forms_give_the_same_bits",
        "JXGPU_SCAN_CHAIN_SPLIT": "test_warm_start_chain_kernels_on_rotated_rows_with_invalid_rows",
        "JXGPU_REPACK_WINDOW": "test_repack_of_a_sample_subset_window_form",
        "JXGPU_SYR2K_PIPE": "test_rank_2k_update_forms_give_the_same_bits"}.items()},
    # ---- ablation masks (wrong results by design: timing experiments only) --------------------------------------------------------
    "JXGPU_QB_SKIP": ("test", "unset", "skips parts of the Q2 kernel (timing ablation; results are WRONG)", "test_q2_staggered_units_equal_lockstep (the lockstep value 64 only)"),
    "JXGPU_BC_SKIP": ("test", "unset", "bulge-chasing ablation (results WRONG)", "untested"),
    "JXGPU_BO_SLEEP": ("test", "unset", "hand-off delay of the bulge chasing (timing)", "untested"),
    # ---- traces, test hooks -------------------------------------------------------------------------------------------------
    "JXGPU_EIGH_TRACE": ("trace", "unset", "stage times of a decomposition to stderr", "-"),
    "JXGPU_PCG_TRACE": ("trace", "unset", "set-up marks of the PCG / HE / staging paths", "-"),
    "JXGPU_BENCH_BACKEND": ("test", "nccl", "bench.py: gloo for ranks sharing a GPU", "test_bench_two_ranks_share_one_gpu"),
    "JXGPU_BENCH_STEP_SLEEP_MS": ("test", "unset", "bench.py diagnostic: idle time in front of every odd step", "-"),
    "JXGPU_BENCH_CHILD": ("test", "unset", "bench.py: marks a child process", "-"),
    "JXGPU_BENCH_FORCE_DIST": ("test", "unset", "bench.py: one rank on the multi-rank code path", "test_distributed_eigh_rccl_callback_single_rank"),
    "JXGPU_DIST_EIGH_FORCE": ("test", "unset", "sharded eigensolver forms with one rank", "test_sharded_band_reduction_rccl_single_rank"),
    "JXGPU_DIST_EIGH_TEST_DISAGREE": ("test", "unset", "fault injection of the replica agreement check", "test_distributed_eigh_two_ranks_share_one_gpu"),
    "JXGPU_DIST_EIGH_TEST_DISAGREE2": ("test", "unset", "fault injection of the replica agreement check", "test_distributed_eigh_two_ranks_share_one_gpu"),
    "JXGPU_PCG_TEST_FAIL": ("test", "unset", "fault injection of the marker-sharded PCG", "test_marker_sharded_pcg_two_ranks_share_one_gpu"),
    "JXGPU_SCAN_CHAIN_FORCE_DIRECT": ("test", "unset", "row of a chain scan forced onto direct evaluations (its chain takes the one-kernel form)",
                                      "test_warm_start_chain_kernels_on_rotated_rows_with_invalid_rows"),
    "JXGPU_SPLMM_COMPONENT_MAX": ("test", "auto", "component limit of the spectral sparse routes", "test_splmm_giant_component_both_sides_of_the_limit"),
    "JXGPU_SPGRM_PANEL_ROWS": ("test", "auto", "row panels of the sparse-GRM builder", "test_sparse_grm_row_panels_write_the_same_file"),
}


def markdown_table(classes=("numerics",)):
    """The table of DESIGN.md Appendix S, generated from SWITCHES."""
    rows = ["| switch | product value | what another value does | covered by |", "|---|---|---|---|"]
    for name, (cls, default, what, test) in sorted(SWITCHES.items()):
        if cls in classes:
            cell = lambda t: str(t).replace(" | ", " / ").replace("|", "/")      # noqa: E731
            rows.append(f"| `{name}` | {cell(default)} | {cell(what)} | {cell(test)} |")
    return "\n".join(rows)

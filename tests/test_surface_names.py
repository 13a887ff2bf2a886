"""CPU: every function / type of the hot path's flat index (SURVEY.md section 8 a') has a same-named
counterpart in the package module that mirrors its reference file."""

import os

import pytest

from conftest import pkg

FLAT_INDEX = {
    "utils.rlgc": """filter_update _fft_cache_3d clear_rlgc_caches next_gpu_fft_size _axis_linear_fft_padding
        pad_for_linear_fft remove_padding_zyx _symmetric_padded_axis_indices enforce_symmetric_boundary pad_psf
        fft_conv _observed_region_mask kl_div _child_log_prefix _resolve_tiled_axis_geometry _axis_retained_bounds
        rlgc _is_gpu_memory_error _chunked_rlgc_once chunked_rlgc""",
    "utils.multiview_registration": """_clear_cupy_memory _max_z_projection_gpu _overlap_slices_after_translation
        register_pair_to_fixed warp_array_to_reference_gpu warp_array_to_reference_with_affine_and_sofima_flow_gpu
        transform_points_to_reference""",
    "utils.sofima_registration": """SofimaRegistrationConfig _resolve_patch_and_step _stabilize_axial_flow_component
        _normalized_mean_squared_error _compose_flow_fields_same_grid _refine_flow_vectors_subpixel
        _median_initial_flow_field _relax_flow_field _estimate_sofima_flow_field_xyz_px_impl
        estimate_sofima_flow_field_xyz_px""",
    "utils.decode_warping": """load_bit_round_transform_zyx_um compose_decode_warp_transform_zyx_um
        compose_optional_warp_transform_zyx_um warp_image_to_reference_frame warp_bit_image_to_reference
        warp_image_with_sofima_metadata""",
    "PixelDecoder": """DEFAULT_DECODE_LOWPASS_SIGMA DEFAULT_DECODE_MAGNITUDE_THRESHOLD DEFAULT_2D_MINIMUM_PIXELS
        DEFAULT_3D_MINIMUM_PIXELS _start_gpu_worker_process _join_gpu_workers decode_tiles_worker
        _optimize_norm_worker""",
    "DataRegistration": """_registration_diag _restrict_worker_to_assigned_gpu _resolve_psf _release_worker_gpu_memory
        _run_chunked_rlgc_remembering_crop _load_deconvolve_fiducial_round _process_fiducial_rounds_on_gpu
        _process_sofima_rounds_on_gpu _apply_bits_on_gpu time_stamp""",
}
METHODS = {
    ("PixelDecoder", "PixelDecoder"): """_load_codebook _resolve_excluded_gene_ids _codebook_fingerprint
        _iterative_normalization_metadata _normalize_codebook _suppress_excluded_codeword_assignments
        _normalization_pixels _load_global_normalization_vectors _global_normalization_vectors
        _load_iterative_normalization_vectors _iterative_normalization_vectors _load_bit_data _lowpass_image _lp_filter
        _effective_lowpass_sigma _default_minimum_pixels _scale_pixel_traces _clip_pixel_traces
        _normalize_pixel_traces _calculate_distances _decode_pixels _warp_pixel _decoded_z_to_source_z
        _extract_barcodes _save_barcodes decoded_barcodes decoded_image save_decoded_barcodes
        _prepare_normalization_state _cleanup decode_one_tile optimize_normalization_by_decoding decode_all_tiles""",
    ("DataRegistration", "DataRegistration"): """_entity_root _has_valid_deconvolved_fiducial_image
        _has_valid_deconvolved_readout_image _has_valid_feature_predictor_outputs _is_tile_complete register_all_tiles
        register_one_tile apply_registration_to_one_tile _generate_registrations _apply_registration_to_bits""",
}


@pytest.mark.parametrize("module", sorted(FLAT_INDEX))
def test_module_level_names_exist(module):
    mod = pkg(module)
    missing = [n for n in FLAT_INDEX[module].split() if not hasattr(mod, n)]
    assert not missing, f"{module}: {missing}"


@pytest.mark.parametrize("module,cls", sorted(METHODS))
def test_operator_methods_exist(module, cls):
    klass = getattr(pkg(module), cls)
    missing = [n for n in METHODS[(module, cls)].split() if not hasattr(klass, n)]
    assert not missing, f"{module}.{cls}: {missing}"


def test_small_control_helpers(monkeypatch, tmp_path):
    DR = pkg("DataRegistration")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2,3")
    assert DR._restrict_worker_to_assigned_gpu(3) == 0 and os.environ["HIP_VISIBLE_DEVICES"] == "3"

    class Store:
        _fiducial_root_path = tmp_path / "fiducial"
        _readouts_root_path = tmp_path / "readouts"

    dr = DR.DataRegistration.__new__(DR.DataRegistration)
    dr._datastore = Store()
    assert dr._entity_root("tile0003", round_id="round002") == tmp_path / "fiducial" / "tile0003" / "round002"
    assert dr._entity_root("tile0003", bit_id="bit007") == tmp_path / "readouts" / "tile0003" / "bit007"
    with pytest.raises(ValueError):
        dr._entity_root("tile0003")
    with pytest.raises(ValueError):
        dr._entity_root("tile0003", round_id="round001", bit_id="bit001")
    rl = pkg("utils.rlgc")
    left, right = rl._symmetric_padded_axis_indices(10, 3, 2)
    assert list(left) == [5, 4, 3] and list(right) == [7, 6]

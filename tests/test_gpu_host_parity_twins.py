"""GPU-box twins of the host-side reference-run parity tests: post-decode filtering,
duplicate removal, the LR filter, the F1 scorer and the iterative-normalisation vectors are host code by nature, but
the record of the GPU run should show them green on that box too - and fed by a real decode where that makes sense
(the decode_all_tiles -> blank-fraction filter -> duplicate removal chain is in test_gpu_extract.py, the iterative
normalisation loop against a reference run in test_gpu_optimize_normalization.py)."""

import pytest

import test_pixeldecoder_host as host
import test_postdecode as post

pytestmark = pytest.mark.gpu


def test_blank_fraction_filter_reference_run_on_the_gpu_box(golden):
    post.test_blank_fraction_filter_matches_reference_run(golden)


def test_duplicate_removal_reference_run_on_the_gpu_box(golden):
    post.test_duplicate_removal_matches_reference_run(golden)


def test_lr_filter_reference_run_on_the_gpu_box(golden):
    post.test_lr_filter_matches_reference_run(golden)


def test_f1_scorer_reference_run_on_the_gpu_box(golden):
    post.test_f1_scorer_matches_reference_run(golden)


def test_iterative_normalization_reference_run_on_the_gpu_box(golden):
    host.test_iterative_normalization_matches_reference_run(golden)


def test_filter_driver_methods_on_the_gpu_box(golden):
    post.test_pixeldecoder_filter_driver_methods(golden)

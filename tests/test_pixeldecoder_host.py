"""CPU: host-side logic of the PixelDecoder mirror (no kernel launches)."""


import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN, pkg
from helpers import codebook_frame, store_from_stack


def _decoder(golden, name="decode_pixels_22", **kw):
    PD = pkg("PixelDecoder")
    g = golden(name)
    cb = codebook_frame(g["codebook_bits"], list(g["gene_ids"]))
    store = store_from_stack([np.zeros((g["codebook_bits"].shape[1], 2, 4, 4), np.float32)], cb)
    return PD.PixelDecoder(store, merfish_bits=g["codebook_bits"].shape[1], verbose=0, **kw), g, store


def test_constructor_codebook_and_thresholds(golden):
    dec, g, _ = _decoder(golden)
    assert dec._pixel_assignment_threshold == float(g["pixel_threshold"])
    assert dec._transcript_distance_threshold == float(g["transcript_threshold"])
    assert np.array_equal(dec._decoding_matrix, g["decoding_matrix"])
    assert dec._blank_count == int(g["blank_count"]) and dec._is_3D
    assert dec._default_minimum_pixels() == 16.0
    PD = pkg("PixelDecoder")
    assert PD.DEFAULT_DECODE_LOWPASS_SIGMA == (3.0, 1.0, 1.0)
    assert PD.DEFAULT_DECODE_MAGNITUDE_THRESHOLD == (1.5, 10.0)
    with pytest.raises(ValueError):
        _decoder(golden, decode_mode="4d")


def test_excluded_gene_ids_resolution(golden):
    dec, g, _ = _decoder(golden, excluded_gene_ids=["gene003", "gene010", "gene003", " "])
    assert dec._excluded_codeword_indices == tuple(int(v) for v in g["excluded_indices"])
    with pytest.raises(ValueError, match="not present"):
        _decoder(golden, excluded_gene_ids=["nope"])
    fp = dec._codebook_fingerprint()
    assert len(fp) == 64 and dec._iterative_normalization_metadata()["codebook_sha256"] == fp


def test_suppress_excluded_assignments_mirrors_reference_test():
    """tests/test_optimization_codeword_exclusions.py:114-120 of the reference."""
    PD = pkg("PixelDecoder").PixelDecoder
    decoded = np.array([0, 1, 2, -1, 1], dtype=np.int16)
    index = np.array([0, 1, 2, 1, 1], dtype=np.int64)
    PD._suppress_excluded_codeword_assignments(decoded, index, (1,))
    assert decoded.tolist() == [0, -1, 2, -1, -1]
    PD._suppress_excluded_codeword_assignments(decoded, index, ())
    assert decoded.tolist() == [0, -1, 2, -1, -1]


def test_warp_pixel_known_answer(golden):
    PD = pkg("PixelDecoder").PixelDecoder
    g = golden("warp_pixel")
    got = PD._warp_pixel(g["pixel"], g["spacing"], g["origin"], g["affine"], g["camera_to_stage"])
    np.testing.assert_allclose(got, g["expected"], rtol=0, atol=1e-5)


def test_iterative_normalization_matches_reference_run(golden):
    dec, _g, store = _decoder(golden, "decode_pixels_16")
    g = golden("iterative_normalization")
    with open(GOLDEN / "iterative_normalization_barcodes.json") as fh:
        df = pd.read_json(fh, orient="split")
    dec._df_barcodes_loaded = df
    dec._global_background_vector = g["global_background"]
    dec._global_normalization_vector = g["global_normalization"]
    dec._iterative_normalization_vectors()
    norm, bkd = store.load_decode_normalization_vectors(None, "iterative")
    assert np.array_equal(norm, g["normalization"]) and np.array_equal(bkd, g["background"])
    assert store.load_decode_normalization_metadata(None, "iterative")["codebook_sha256"] == dec._codebook_fingerprint()
    # cached vectors fitted with another codebook are rejected (PixelDecoder.py:1223-1230)
    store._norm_meta[(None, "iterative")]["codebook_sha256"] = "0" * 64
    with pytest.raises(ValueError, match="different active codebook"):
        dec._load_iterative_normalization_vectors()


def test_normalization_state_precedence(golden):
    dec, _g, store = _decoder(golden, "decode_pixels_16")
    with pytest.raises(ValueError):
        dec._prepare_normalization_state("bogus", True)
    dec._prepare_normalization_state("none", True)
    assert not dec._global_normalization_loaded and not dec._iterative_normalization_loaded
    store.save_decode_normalization_vectors(None, "global", np.ones(16, np.float32), np.zeros(16, np.float32))
    dec._prepare_normalization_state("global", True)
    assert dec._global_normalization_loaded and not dec._iterative_normalization_loaded


def _decoder_with_codebook():
    """The three-gene decoder of the reference's test module (tests/test_optimization_codeword_exclusions.py:17-30)."""
    PD = pkg("PixelDecoder").PixelDecoder
    dec = PD.__new__(PD)
    dec._gene_ids = ["GeneA", "GeneB", "GeneC"]
    dec._codebook_matrix = np.asarray([[1, 1, 0, 0], [1, 0, 1, 0], [1, 0, 0, 1]], dtype=np.int8)
    return dec


def test_resolve_exclusions_is_exact_deduplicated_and_index_stable():
    """Mirror of the reference test of the same name (:96-104)."""
    dec = _decoder_with_codebook()
    gene_ids, indices = dec._resolve_excluded_gene_ids([" GeneB ", "GeneB"])
    assert gene_ids == ("GeneB",) and indices == (1,)
    with pytest.raises(ValueError, match="case-sensitive"):
        dec._resolve_excluded_gene_ids(["geneb"])
    assert dec._resolve_excluded_gene_ids(None) == ((), ()) and dec._resolve_excluded_gene_ids([]) == ((), ())
    assert dec._resolve_excluded_gene_ids(["GeneC", "GeneA"]) == (("GeneA", "GeneC"), (2, 0))


def test_resolve_exclusions_rejects_removing_entire_codebook():
    """Mirror of the reference test of the same name (:107-111)."""
    dec = _decoder_with_codebook()
    with pytest.raises(ValueError, match="every codeword"):
        dec._resolve_excluded_gene_ids(["GeneA", "GeneB", "GeneC"])


def test_excluded_winner_becomes_background_without_index_fallback():
    """Mirror of the reference test of the same name (:114-120): torch tensors behave like the arrays."""
    PD = pkg("PixelDecoder").PixelDecoder
    decoded = np.asarray([0, 1, 2, 1, -1], dtype=np.int16)
    nearest = np.asarray([0, 1, 2, 1, 1], dtype=np.int16)
    PD._suppress_excluded_codeword_assignments(decoded, nearest, (1,))
    np.testing.assert_array_equal(decoded, np.asarray([0, -1, 2, -1, -1]))


def test_run_scoped_normalization_metadata_round_trips():
    """Mirror of the reference datastore test (:275-300) on the in-memory store double."""
    ds_mod = pkg("datastore")
    import pandas as pd

    store = ds_mod.InMemoryDataStore(num_tiles=1, num_rounds=1, num_bits=4,
                                     codebook=pd.DataFrame({"gene_id": ["a"], "bit01": [1]}))
    metadata = {"scope": "iterative_optimization", "excluded_gene_ids": ["GeneB"], "codebook_sha256": "abc123"}
    store.save_decode_normalization_vectors("run1", "iterative", np.ones(4, dtype=np.float32),
                                            np.zeros(4, dtype=np.float32), decode_mode="3d", metadata=metadata)
    assert store.load_decode_normalization_metadata("run1", "iterative") == metadata
    nrm, bkd = store.load_decode_normalization_vectors("run1", "iterative")
    np.testing.assert_array_equal(nrm, np.ones(4, dtype=np.float32))
    np.testing.assert_array_equal(bkd, np.zeros(4, dtype=np.float32))
    assert store.load_decode_normalization_vectors("other", "iterative") is None or \
        store.load_decode_normalization_vectors("other", "iterative") == (None, None)


def test_warp_pixels_equals_the_per_point_loop():
    """``_warp_pixels`` is vectorised (round 4); the reference applies ``_warp_pixel`` row by row (:3134-3143).  Same
    float64 bits on 200 000 random points, with and without the camera-to-stage affine."""
    PD = pkg("PixelDecoder")
    rng = np.random.default_rng(5)
    pts = rng.uniform(0, 2048, size=(200_000, 3)) * np.array([0.03, 1.0, 1.0])
    spacing = np.array([0.315, 0.098, 0.098], dtype=np.float32)
    origin = np.array([2761.3, 107.81, 12.5], dtype=np.float32)
    c2s = np.array([[1, 0, 0, 0], [0, -0.07, -1, 0], [0, -1, 0.07, 0], [0, 0, 0, 1]], dtype=np.float32)
    aff = np.array([[1, 0.001, 0, 0.5], [0, 1, 0.002, 1.25], [0.003, 0, 1, -2.0], [0, 0, 0, 1]], dtype=np.float32)
    for cam in (c2s, None):
        got = PD.PixelDecoder._warp_pixels(pts, spacing, origin, aff, cam)
        want = np.stack([PD.PixelDecoder._warp_pixel(p, spacing, origin, aff, cam) for p in pts[:20_000]])
        assert got.shape == (200_000, 3) and np.array_equal(got[:20_000], want)

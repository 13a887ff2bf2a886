"""Golden vectors from the real scikit-image (not importable by the test interpreter).

Run with an interpreter that has scikit-image - in the build container
``/opt/conda/bin/python3.9 tests/golden/make_golden_skimage.py`` (scikit-image 0.18.3,
NumPy 1.26) - to write ``tests/golden/skimage_regionprops.npz``.  It pins the third-party
arithmetic PixelDecoder._extract_barcodes delegates to (PixelDecoder.py:2946-3042):
``label(background=-1, connectivity=3)`` (cuCIM mirrors the scikit-image definition), the
per-plane 8-connected variant, ``remove_small_objects`` and ``regionprops_table`` with
``area``, ``centroid``, ``intensity_mean`` (0.18 name: ``mean_intensity``) on the
multichannel float16 scaled image, ``inertia_tensor_eigvals``, ``intensity_min`` of the
distance image and ``intensity_mean`` of the magnitude image.

Input: the decoded / scaled / magnitude / distance volumes of ``decode_pixels_16.npz``
(themselves written by the reference's own decode functions, make_golden.py).
"""

from pathlib import Path

import numpy as np
import skimage
from skimage.measure import label, regionprops, regionprops_table
from skimage.morphology import remove_small_objects

here = Path(__file__).resolve().parent


def main():
    g = np.load(here / "decode_pixels_16.npz")
    decoded = g["decoded"]
    scaled = g["scaled"]          # (B, Z, Y, X) float16
    magnitude = g["magnitude"]    # float16
    distance = g["distance"]      # float16
    labels3 = label(decoded, background=-1, connectivity=3)
    labels2 = np.zeros_like(labels3)
    off = 0
    for z in range(decoded.shape[0]):   # PixelDecoder.py:2956-2972
        lab, n = label(decoded[z], background=-1, connectivity=2, return_num=True)
        if n:
            labels2[z] = np.where(lab != 0, lab + off, 0)
            off += int(n)
    kept = remove_small_objects(labels3, min_size=3)   # == max_size 2 of the current API
    inten = np.moveaxis(scaled, 0, -1)
    t = regionprops_table(kept, intensity_image=inten,
                          properties=["label", "area", "centroid", "mean_intensity", "inertia_tensor_eigvals"])
    tm = regionprops_table(kept, intensity_image=magnitude, properties=["label", "mean_intensity"])
    # (regionprops_table of 0.18 casts min_intensity to int - a bug fixed in 0.19 - so take the property itself)
    dmin = np.array([r.min_intensity for r in regionprops(kept, intensity_image=distance.astype(np.float32))], dtype=np.float32)
    B = scaled.shape[0]
    out = {
        "skimage_version": np.array(skimage.__version__),
        "labels3": labels3.astype(np.int32),
        "labels2": labels2.astype(np.int32),
        "kept": kept.astype(np.int32),
        "label": np.asarray(t["label"]),
        "area": np.asarray(t["area"], dtype=np.float64),
        "centroid": np.stack([t[f"centroid-{k}"] for k in range(3)], axis=1),
        "intensity_mean": np.stack([t[f"mean_intensity-{k}"] for k in range(B)], axis=1),
        "inertia_tensor_eigvals": np.stack([t[f"inertia_tensor_eigvals-{k}"] for k in range(3)], axis=1),
        "magnitude_mean": np.asarray(tm["mean_intensity"]),
        "distance_min": dmin,
    }
    np.savez_compressed(here / "skimage_regionprops.npz", **out)
    print({k: (v.shape, v.dtype) for k, v in out.items()})


def make_phase_correlation():
    """scikit-image 0.18.3 ``phase_cross_correlation`` (no ``normalization`` / ``disambiguate`` yet: the
    un-normalised variant) on blob images with known sub-pixel shifts, 2-D and 3-D, float32 in."""
    from scipy import ndimage as ndi
    from skimage.registration import phase_cross_correlation

    rng = np.random.default_rng(5)
    out = {"skimage_version": np.array(skimage.__version__)}
    cases = [((64, 72), (3.3, -5.7)), ((48, 50), (-0.25, 0.35)), ((12, 40, 44), (1.2, -3.4, 2.6)),
             ((9, 33, 31), (0.0, 4.0, -2.5))]
    for i, (shape, shift) in enumerate(cases):
        img = np.zeros(shape, dtype=np.float32)
        idx = tuple(rng.integers(2, s - 2, size=40) for s in shape)
        img[idx] = rng.uniform(0.5, 1.0, size=40).astype(np.float32)
        img = ndi.gaussian_filter(img, 1.3).astype(np.float32)
        mov = ndi.shift(img, shift, order=3, mode="grid-wrap").astype(np.float32)
        mov += rng.normal(0, 1e-3, size=shape).astype(np.float32)
        for up in (1, 10):
            got = phase_cross_correlation(img, mov, upsample_factor=up, return_error=False)
            out[f"shift_{i}_up{up}"] = np.asarray(got, dtype=np.float64)
        out[f"ref_{i}"], out[f"mov_{i}"], out[f"true_{i}"] = img, mov, np.asarray(shift)
    np.savez_compressed(here / "skimage_phase_correlation.npz", **out)
    print({k: v.tolist() for k, v in out.items() if k.startswith("shift")})


if __name__ == "__main__":
    main()
    make_phase_correlation()

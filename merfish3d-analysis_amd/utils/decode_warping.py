"""Decode-time transform composition and warp dispatch (mirror of utils/decode_warping.py).

``T = inv(chromatic) @ round`` (:55-81); identity transform with no flow field passes the
image through as float32 (:153-157); a SOFIMA field is used unless its status starts with
``identity_fallback`` (:165).  Interpolation itself is the HIP warp of
``utils/multiview_registration.py``.  Arrays may be NumPy (host in/out like the reference)
or torch tensors in HBM (device in/out).
"""

from __future__ import annotations

from collections.abc import Mapping, Sequence
from typing import Any

import numpy as np


def load_bit_round_transform_zyx_um(datastore: Any, *, tile, bit_id: str):
    """(round id | None, 4x4) for one readout bit; reference-round bits are identity (:15-52)."""
    round_index = datastore.load_local_round_linker(tile=tile, bit=bit_id) - 1
    if round_index <= 0:
        return None, np.eye(4, dtype=np.float32)
    round_id = datastore.round_ids[round_index]
    transform = datastore.load_local_round_transform_zyx_um(tile=tile, round=round_id)
    if transform is None:
        raise RuntimeError(f"Missing local round transform for tile={tile} round={round_id}.")
    return round_id, np.asarray(transform, dtype=np.float32)


def compose_decode_warp_transform_zyx_um(*, round_transform_zyx_um, chromatic_transform_zyx_um):
    """inv(chromatic) @ round in float32 (:55-81)."""
    return np.linalg.inv(np.asarray(chromatic_transform_zyx_um, dtype=np.float32)) @ (
        np.asarray(round_transform_zyx_um, dtype=np.float32)
    )


def compose_optional_warp_transform_zyx_um(*, round_transform_zyx_um, chromatic_transform_zyx_um):
    """Same with ``None`` meaning identity (:84-120)."""
    rt = np.eye(4, dtype=np.float32) if round_transform_zyx_um is None else np.asarray(
        round_transform_zyx_um, dtype=np.float32)
    ct = np.eye(4, dtype=np.float32) if chromatic_transform_zyx_um is None else np.asarray(
        chromatic_transform_zyx_um, dtype=np.float32)
    return compose_decode_warp_transform_zyx_um(round_transform_zyx_um=rt, chromatic_transform_zyx_um=ct)


def _as_float32(image):
    try:
        import torch

        if isinstance(image, torch.Tensor):
            return image.to(torch.float32)
    except ImportError:  # pragma: no cover
        pass
    return np.asarray(image, dtype=np.float32)


def warp_image_to_reference_frame(image, *, transform_zyx_um, spacing_zyx_um: Sequence[float],
                                  loaded_flow_field=None, gpu_id: int = 0):
    """Affine or affine+flow warp into the reference frame, float32 out (:123-181)."""
    if loaded_flow_field is None and np.allclose(transform_zyx_um, np.eye(4, dtype=np.float32)):
        return _as_float32(image)

    from .multiview_registration import warp_array_to_reference_gpu

    if loaded_flow_field is not None:
        flow, flow_attrs = loaded_flow_field
        if not str(flow_attrs.get("sofima_status", "")).startswith("identity_fallback"):
            return _as_float32(warp_image_with_sofima_metadata(
                image, transform_zyx_um=transform_zyx_um, spacing_zyx_um=spacing_zyx_um,
                sofima_flow_field_xyz_px=flow, flow_attrs=flow_attrs, gpu_id=gpu_id))
    return _as_float32(warp_array_to_reference_gpu(
        image, transform_zyx_um=transform_zyx_um, spacing_zyx_um=spacing_zyx_um,
        reference_shape=tuple(image.shape), gpu_id=gpu_id))


def warp_bit_image_to_reference(image, *, datastore: Any, tile, bit_id: str,
                                emission_wavelength_um: float, gpu_id: int = 0):
    """Warp one native bit image into the round-1 local reference frame (:184-245)."""
    round_id, round_transform = load_bit_round_transform_zyx_um(datastore, tile=tile, bit_id=bit_id)
    chromatic = datastore.load_chromatic_affine_transform_zyx_um(wavelength_um=emission_wavelength_um)
    transform = compose_decode_warp_transform_zyx_um(
        round_transform_zyx_um=round_transform, chromatic_transform_zyx_um=chromatic)
    loaded_flow_field = None
    if round_id is not None:
        loaded_flow_field = datastore.load_local_sofima_flow_field(
            tile=tile, round=round_id, return_future=False)
    return warp_image_to_reference_frame(
        image, transform_zyx_um=transform, spacing_zyx_um=datastore.voxel_size_zyx_um,
        loaded_flow_field=loaded_flow_field, gpu_id=gpu_id)


def warp_image_with_sofima_metadata(image, *, transform_zyx_um, spacing_zyx_um: Sequence[float],
                                    sofima_flow_field_xyz_px, flow_attrs: Mapping[str, Any],
                                    reference_shape: Sequence[int] | None = None, gpu_id: int = 0):
    """Flow warp driven by the datastore's flow metadata (:248-304)."""
    from .multiview_registration import warp_array_to_reference_with_affine_and_sofima_flow_gpu

    if reference_shape is None:
        reference_shape = flow_attrs["reference_shape_zyx_px"]
    return _as_float32(warp_array_to_reference_with_affine_and_sofima_flow_gpu(
        image, transform_zyx_um=transform_zyx_um, spacing_zyx_um=spacing_zyx_um,
        reference_shape=tuple(int(v) for v in reference_shape),
        sofima_flow_field_xyz_px=sofima_flow_field_xyz_px,
        flow_field_stride_zyx_px=flow_attrs["map_stride_zyx_px"],
        flow_field_box_start_xyz_px=flow_attrs["map_box_start_xyz_px"], gpu_id=gpu_id))


__all__ = [
    "compose_decode_warp_transform_zyx_um",
    "compose_optional_warp_transform_zyx_um",
    "load_bit_round_transform_zyx_um",
    "warp_bit_image_to_reference",
    "warp_image_to_reference_frame",
    "warp_image_with_sofima_metadata",
]

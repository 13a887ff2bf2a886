"""Child ranks of tests/test_gpu_two_rank_pipeline.py (started by ``python -m torch.distributed.run --nproc-per-node 2``):
two processes share the one GPU of a test box over a ``gloo`` group (RCCL refuses two ranks on one device; the
collectives' payloads - int64[256] histograms, a count - then travel as CPU tensors) and ONE on-disk qi2lab datastore,
which is how the operators meet on a multi-GPU node: whole tiles per rank in ``DataRegistration.register_all_tiles``
(DataRegistration.py:2139-2156 shards inside a tile; the build shards tiles first) and in
``PixelDecoder.decode_all_tiles`` (PixelDecoder.py:4811-4820), pooled normalisation statistics by all-reduce, tables
through the store, rank 0 filters.  usage: dist_pipeline_child.py <datastore path>"""

import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def run(path: Path, distributed: bool) -> dict:
    import numpy as np
    import pandas as pd
    import torch

    from conftest import pkg

    rank = 0
    if distributed:
        import torch.distributed as dist

        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
        rank = dist.get_rank()
    DR, PD, QS = pkg("DataRegistration"), pkg("PixelDecoder"), pkg("qi2lab_store")
    store = QS.Qi2labDiskStore(path, device="cuda:0", num_threads=2, write_behind=True, write_threads=2)
    predictors = np.load(path / "test_predictors.npz")
    state = {}

    def predictor(image):
        prob = predictors[f"{state['key'][0]}__{state['key'][1]}"]
        flat = np.argsort(prob.ravel())[-20:]
        z, y, x = np.unravel_index(flat, prob.shape)
        return pd.DataFrame({"axis-0": z, "axis-1": y, "axis-2": x}), prob

    load = store.load_local_corrected_image

    def tracking_load(tile, round=None, bit=None, return_future=True):
        if bit is not None:
            state["key"] = (store._tile(tile), store._bit(bit))
        return load(tile, round=round, bit=bit, return_future=return_future)

    store.load_local_corrected_image = tracking_load
    reg = DR.DataRegistration(store, decon_fiducial=True, decon_readout=True, perform_deformable_registration=False,
                              verbose=0, feature_predictor=predictor)
    reg.register_all_tiles()
    store.load_local_corrected_image = load
    dec = PD.PixelDecoder(store, merfish_bits=len(store.bit_ids), verbose=0)
    dec._global_normalization_vectors(gpu_id=0)
    dec.decode_all_tiles(assign_to_cells=False, minimum_pixels=3, normalization_method="global",
                         target_gross_misid_rate=0.05)
    out = {"rank": rank, "suspect_all_ranks": reg.suspect_registrations_all_ranks,
           "resident_left": len(store._resident), "pending_left": len(store._pending_writes)}
    if distributed:
        import torch.distributed as dist

        dist.barrier()
    if rank == 0:
        final = store.load_global_filtered_decoded_spots()
        norm, bkd = store.load_decode_normalization_vectors(None, "global")
        out.update(transcripts=int(len(final)), normalization=np.asarray(norm, dtype=float).round(3).tolist(),
                   background=np.asarray(bkd, dtype=float).round(3).tolist(),
                   per_tile=[int((final["tile_idx"] == t).sum()) for t in range(len(store.tile_ids))],
                   table_digest=int(pd.util.hash_pandas_object(
                       final[["gene_id", "tile_idx", "tile_z", "tile_y", "tile_x"]].sort_values(
                           ["tile_idx", "tile_z", "tile_y", "tile_x", "gene_id"]).reset_index(drop=True),
                       index=False).sum() % (1 << 62)))
    if distributed:
        import torch.distributed as dist

        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    res = run(Path(sys.argv[1]), distributed="WORLD_SIZE" in os.environ)
    if res["rank"] == 0:
        print("RESULT " + json.dumps(res), flush=True)

#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of scripts/gpu_probe_rlgc.py
into per-kernel HBM traffic, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
prescribes: both counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it
is doubled; WRITE_SIZE is exact.

    python scripts/pmc_summary.py profiles/r01_pmc_fetch_size_rlgc_probe.csv \
        profiles/r01_pmc_write_size_rlgc_probe.csv 128 2187 2187 > profiles/r01_pmc_rlgc_traffic.json
"""
import collections
import csv
import json
import re
import sys

CONV = ("k_x_forward", "k_y_pass", "k_y_forward_mirror", "k_z_conv", "k_x_inverse")  # name prefixes of the passes


def rows(path):
    out = []
    for r in csv.DictReader(open(path)):
        m = re.search(r"(k_\w+(?:<[^>]*>)?)", r["Kernel_Name"])
        if m:
            out.append((int(r["Dispatch_Id"]), m.group(1), float(r["Counter_Value"]) * 1024.0))
    out.sort()
    return out


def load(path):
    d = collections.defaultdict(list)
    for _i, name, v in rows(path):
        d[name].append(v)
    return d


def per_call_totals(path, scale):
    """Bytes per m3d_rlgc_run call (k_psf_hash opens each call)."""
    totals = []
    for _i, name, v in rows(path):
        if name.startswith("k_psf_hash"):
            totals.append(0.0)
        if totals:
            totals[-1] += scale * v
    return totals


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    zp, yp, xp = (int(v) for v in sys.argv[3:6])
    vol = 4.0 * zp * yp * xp
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, [0.0]), write.get(k, [0.0])
        kernels[k] = {
            "dispatches": len(f),
            "fetch_size_bytes_raw_mean": sum(f) / len(f),
            "write_size_bytes_mean": sum(w) / len(w),
            "hbm_bytes_corrected_mean": 2.0 * sum(f) / len(f) + sum(w) / len(w),
            "in_padded_f32_volumes": round((2.0 * sum(f) / len(f) + sum(w) / len(w)) / vol, 3),
        }
    # one Z pass per convolution (the two ratio convolutions share their X inverse pass); the OTF's
    # forward-only Z pass of the first call is counted too
    # (round 2: the two ratio convolutions share one paired Z launch - the all-planes variant "..., 0, true, ...>" - which
    # therefore counts twice)
    n_conv = sum(len(v) * (2 if ", 0, true" in k else 1) for k, v in fetch.items() if k.startswith("k_z_conv"))
    total = 0.0
    for k, v in kernels.items():
        if k.startswith(CONV):
            total += v["hbm_bytes_corrected_mean"] * v["dispatches"]
    # the probe runs the solver for 1, n and 2n iterations: the difference of the last two calls is n
    # iterations' worth of traffic without the per-call setup (OTF, update_norm, speculative forward)
    calls = [f + w for f, w in zip(per_call_totals(sys.argv[1], 2.0), per_call_totals(sys.argv[2], 1.0))]
    n_iter = int(sys.argv[6]) if len(sys.argv) > 6 else 2
    per_iteration = (calls[-1] - calls[-2]) / n_iter if len(calls) >= 2 else None
    out = {
        "rlgc_call_hbm_bytes": calls,
        "rlgc_iteration_hbm_bytes": per_iteration,
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over scripts/gpu_probe_rlgc.py 64 2048 2048 2",
        "correction": "bytes = 2 * FETCH_SIZE[KiB] * 1024 + WRITE_SIZE[KiB] * 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)",
        "calibration": ("k_floor / k_crop / k_image_sum / k_otf_abs2 (streaming reads of known size) confirm the x2 to three digits, "
                        "and the L2's own request counters (TCC_EA0_RDREQ x 128 B, TCC_EA0_WRREQ x 64 B; round 3, "
                        "profiles/r03_l2_fabric.txt) give the same bytes: the corrected figures are what crosses the L2's "
                        "memory side.  Kernels that show more than their algorithmic bytes (the X inverse passes that read "
                        "half lines, some Z passes) really request those lines twice; the repeats are served by the Infinity "
                        "Cache, which sits behind these counters (round 1 had read the excess as a counter artefact)"),
        "padded_volume_bytes": vol,
        "fft_conv_launches": n_conv,
        "fft_conv_hbm_bytes_per_launch": total / max(n_conv, 1),
        "kernels": kernels,
    }
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()

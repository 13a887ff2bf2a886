#!/usr/bin/env python3
"""Reduce the SQ counter passes of scripts/gpu_pmc_sq.sh per kernel: mean counter values per dispatch and the
ratios that say what a kernel waits on (MI355X_MICROARCH.md, "rocprofv3 PMC slots": WAIT_ANY + WAIT_INST_ANY +
ACTIVE_INST_ANY ~ WAVE_CYCLES, all in quad-cycles summed over waves).

    python scripts/pmc_sq_summary.py gpurun_out/pmc_sq_wait.csv gpurun_out/pmc_sq_lds.csv gpurun_out/pmc_sq_mem.csv \
        > profiles/r02_pmc_sq_rlgc.json
"""
import collections
import csv
import json
import re
import sys

csv.field_size_limit(1 << 30)


def load(path, acc, meta):
    for r in csv.DictReader(open(path)):
        m = re.search(r"(k_\w+(?:<[^>]*>)?)", r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1)
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = {"workgroup_size": int(r["Workgroup_Size"]), "lds_block_size": int(r["LDS_Block_Size"]),
                   "vgpr": int(r["VGPR_Count"]), "accum_vgpr": int(r["Accum_VGPR_Count"]),
                   "sgpr": int(r["SGPR_Count"]), "grid_size": int(r["Grid_Size"])}


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for p in sys.argv[1:]:
        load(p, acc, meta)
    out = {}
    for k in sorted(acc):
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        d = dict(meta[k])
        d["dispatches"] = max(len(v) for v in acc[k].values())
        d["counters_mean"] = {n: round(v, 1) for n, v in sorted(c.items())}
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        r = {}
        if wc:
            r["wait_any_frac"] = round(c.get("SQ_WAIT_ANY", 0.0) / wc, 4)          # parked: s_waitcnt / barrier
            r["wait_inst_any_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4)  # issue stalls
            r["wait_inst_lds_frac"] = round(c.get("SQ_WAIT_INST_LDS", 0.0) / wc, 4)
            r["active_inst_any_frac"] = round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4)
        if c.get("SQ_BUSY_CYCLES"):
            # SQ_LEVEL_WAVES accumulates resident waves per cycle; per busy cycle (summed over SEs/XCDs alike)
            r["mean_waves_per_busy_cycle"] = round(c.get("SQ_LEVEL_WAVES", 0.0) / c["SQ_BUSY_CYCLES"], 3)
        if c.get("SQ_WAVES") and wc:
            r["quad_cycles_per_wave"] = round(wc / c["SQ_WAVES"], 1)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            r["lds_bank_conflict_frac_of_lds_cycles"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        if c.get("SQ_WAVES"):
            for n in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU", "SQ_INSTS_SMEM"):
                if n in c:
                    r[n.lower() + "_per_wave"] = round(c[n] / c["SQ_WAVES"], 1)
        d["ratios"] = r
        out[k] = d
    json.dump({"source": "rocprofv3 --pmc (scripts/gpu_pmc_sq.sh), mean per dispatch", "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()

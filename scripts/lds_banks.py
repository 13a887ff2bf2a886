#!/usr/bin/env python3
"""LDS-array cycles of the static X passes' access patterns (N = 2187 = 9 * 9 * 3 * 9, 256 threads) by the banking
rules of MI355X_MICROARCH.md, section LDS: an 8-byte access is served per lane group - 4 x 16 contiguous lanes over 32
four-byte banks for ds_read2_b64 / ds_write_b64 / ds_write2_b64 (what hipcc emits for most of these accesses), 2 x 32
lanes over 64 banks for ds_read_b64 - one cycle per group plus one per extra distinct address on a bank.
Usage: python3 scripts/lds_banks.py   (no GPU; the table in profiles/r03_lds_banks.txt is its output)"""
import numpy as np

N, S = 2187, 243


def cycles(elem, live, wide):
    groups = [range(0, 32), range(32, 64)] if wide else [range(16 * g, 16 * g + 16) for g in range(4)]
    banks = 64 if wide else 32
    total = 0
    for grp in groups:
        per = {}
        for lane in grp:
            if live[lane]:
                for d in (2 * elem[lane], 2 * elem[lane] + 1):
                    per.setdefault(d % banks, set()).add(d)
        if per:
            total += max(len(s) for s in per.values())
    return total


def run(name, addr, wide=False, threads=256):
    got = best = 0
    for w in range(threads // 64):
        a = [addr(64 * w + lane) for lane in range(64)]
        live = [x is not None for x in a]
        a = [0 if x is None else x for x in a]
        if any(live):
            got += cycles(a, live, wide)
            best += cycles(list(range(64)), live, wide)   # the same lanes on consecutive elements
    print(f"{name:58s} {'ds_read_b64 ' if wide else 'read2 / write'}  {got:4d} cycles, conflict free {best:4d}  x {got / best:.2f}")


def pos9(kp):   # digit-reversed position of 9-block k' (radix list 9, 9, 3, 9), in blocks
    return (kp % 9) * 27 + ((kp // 9) % 9) * 3 + kp // 81


if __name__ == "__main__":
    for k in range(3):
        t = lambda tid, k=k: min(tid + 256 * k, 728)
        run(f"stage 27, j along the lanes (round 2), items {256 * k}..", lambda tid: (t(tid) // 9) * 27 + t(tid) % 9)
        run(f"stage 27, block along the lanes,       items {256 * k}..", lambda tid: (t(tid) % 81) * 27 + t(tid) // 81)
    run("stage 243 (block = t / 27, j = t % 27)", lambda tid: None if tid >= 243 else (tid // 27) * 243 + tid % 27)
    run("edge stages (element t + 243 q)", lambda tid: None if tid >= 243 else tid)
    run("merge, block k' (lane = (pair, row))", lambda tid: None if tid >> 1 >= 122 else (tid & 1) * N + 9 * pos9(tid >> 1))
    run("merge, block S - k'", lambda tid: None if not 0 < tid >> 1 < 122 else (tid & 1) * N + 9 * pos9(S - (tid >> 1)))
    run("last forward stage, block k' (lane = pair)", lambda tid: None if tid >= 122 else 9 * pos9(tid))
    run("last forward stage, block S - k'", lambda tid: None if tid >= 122 else 9 * pos9((S - tid) % S))

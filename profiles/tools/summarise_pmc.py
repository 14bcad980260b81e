#!/usr/bin/env python
"""Turn the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_traffic.json.

FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md "HBM"): FETCH_SIZE reports exactly
half of the bytes of a wide coalesced streaming read (8-16 B/lane loads here), so it is doubled; WRITE_SIZE is exact.
    python profiles/tools/summarise_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <key> [out.json] [valu_counter_collection.csv]
The optional third pass (--pmc SQ_INSTS_VALU) adds "<key>/valu_insts": wave64 VALU instructions per launch per stage, from which
bench.py derives roofline.valu_issue_frac.
"""
import collections
import csv
import datetime
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bench import source_hash  # noqa: E402  (hash of the kernel sources the passes were run on)

NAMES = {"k_forward_spectra": "al_forward_spectra", "k_ir_spectra": "al_ir_spectra", "k_emitter_gains": "al_emitter_gains",
         "k_signal_spectra": "al_signal_spectra",
         "k_spectral_mac": "al_spectral_mac", "k_moving_fused": "al_spectral_mac", "k_block_synthesis": "al_block_synthesis", "k_event_levels": "al_event_levels",
         "k_mixdown": "al_mixdown"}


def per_kernel(path, counter):
    """Per C-ABI stage: the average dispatch of every DISTINCT kernel the stage launches, summed (al_spectral_mac launches up to
    three kernels per call -- capsule loop, tile kernel, sliding window; averaging over all of a stage's dispatches would halve
    the stage's traffic whenever two of them run)."""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            for k, v in NAMES.items():
                if k in row["Kernel_Name"]:
                    acc[v][row["Kernel_Name"]].append(float(row["Counter_Value"]))
                    break
    return {stage: sum(sum(v) / len(v) for v in kernels.values()) for stage, kernels in acc.items()}


def main():
    fetch, write, key = sys.argv[1:4]
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(__file__), "..", "pmc_traffic.json")
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    table = json.load(open(out)) if os.path.exists(out) else {}
    if table.get("source_hash") != source_hash():
        table = {}   # entries measured on other kernel sources are stale: bench.py refuses them anyway
    table["source_hash"] = source_hash()
    table["collected"] = datetime.date.today().isoformat()
    table[key] = {k: int(2 * f.get(k, 0) * 1024 + w.get(k, 0) * 1024) for k in NAMES.values()}
    table[key + "/detail"] = {k: {"fetch_KiB_raw": f.get(k, 0), "write_KiB": w.get(k, 0)} for k in NAMES.values()}
    if len(sys.argv) > 5:
        v = per_kernel(sys.argv[5], "SQ_INSTS_VALU")
        table[key + "/valu_insts"] = {k: int(v.get(k, 0)) for k in NAMES.values()}
    json.dump(table, open(out, "w"), indent=1)
    print(json.dumps(table[key], indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs into per-kernel HBM traffic.

Usage: python profiles/summarize_pmc.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <out.json>
Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE reports exactly half of the bytes of
a coalesced streaming read, so the read side is doubled; both counters are in units of 1 KiB... this
rocprofv3 build reports them in KB (1e3 B), checked against the known 4.29 GB streams of the layer kernels.
"""
import collections
import csv
import json
import os
import sys


def load(path):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "dva::" not in name:
            continue
        short = name.split("dva::")[1].split("(")[0]
        out[short].append(float(r["Counter_Value"]))
    return out


def main(src, dst):
    fetch = load(os.path.join(src, "pmc_FETCH_SIZE", "pmc_counter_collection.csv"))
    write = load(os.path.join(src, "pmc_WRITE_SIZE", "pmc_counter_collection.csv"))
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f = max(fetch.get(k, [0.0]))      # largest launch of the kernel = the V-sized one
        w = max(write.get(k, [0.0]))
        res[k] = {"fetch_bytes_corrected": 2 * f * 1e3, "write_bytes": w * 1e3,
                  "hbm_bytes": 2 * f * 1e3 + w * 1e3}
    json.dump(res, open(dst, "w"), indent=1)
    for k, v in res.items():
        print(f"{k:40s} {v['hbm_bytes'] / 1e9:8.2f} GB")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

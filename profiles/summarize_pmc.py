#!/usr/bin/env python
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs into per-kernel HBM traffic.

Usage: python profiles/summarize_pmc.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <out.json>
Units and correction (MI355X_MICROARCH.md, HBM section): both counters are in KiB (1024 B); on gfx950 FETCH_SIZE
reports exactly half of the bytes of a coalesced 16-byte-per-lane streaming read, so the read side is doubled.
Calibration instead of trust: the same run contains dva::copy_kernel (bench.py's copy ceiling: 4 GiB read + 4 GiB
written per launch, float4 grid-stride) -- the measured / known ratios of that kernel are printed and stored under
"_calibration"; they are NOT applied to the other kernels (access widths differ), only reported.
"""
import collections
import csv
import json
import os
import sys

KIB = 1024.0
COPY_BYTES = float(1 << 32)


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
        res[k] = {"fetch_bytes_corrected": 2 * f * KIB, "write_bytes": w * KIB,
                  "hbm_bytes": 2 * f * KIB + w * KIB}
    if "copy_kernel" in res:
        c = res["copy_kernel"]
        res["_calibration"] = {"kernel": "copy_kernel", "known_read_bytes": COPY_BYTES, "known_write_bytes": COPY_BYTES,
                               "read_measured_over_known": c["fetch_bytes_corrected"] / COPY_BYTES,
                               "write_measured_over_known": c["write_bytes"] / COPY_BYTES}
    # what build these counters belong to (bench.py prints traffic only when the sources it runs hash to the same value)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepviewagg_amd import _lib
    res["_stamp"] = {"csrc_sha256": _lib.source_sha256()}
    json.dump(res, open(dst, "w"), indent=1)
    for k, v in res.items():
        if k.startswith("_"):
            print(k, v)
        else:
            print(f"{k:48s} {v['hbm_bytes'] / 1e9:8.2f} GB")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

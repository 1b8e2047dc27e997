#!/usr/bin/env python3
"""Sum FETCH_SIZE / WRITE_SIZE (KiB, rocprofv3 --pmc, one counter per pass) over the conv_fwd_h2 3x3 dispatches of
tools/eval_once.py and print the per-launch HBM traffic that bench.py reports as roofline.traffic.
usage: traffic_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <batch>"""
import csv
import json
import sys


def total(path, counter):
    n, s = 0, 0.0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "conv_fwd_h2_kernel<3" in r["Kernel_Name"]:
            n += 1
            s += float(r["Counter_Value"])
    return n, s * 1024.0


nf, fetch = total(sys.argv[1], "FETCH_SIZE")
nw, write = total(sys.argv[2], "WRITE_SIZE")
assert nf == nw and nf > 0, (nf, nw)
out = {"kernel": "conv_fwd_h2_k3", "batch": int(sys.argv[4]), "launches": nf,
       "fetch_bytes_per_launch": fetch / nf, "write_bytes_per_launch": write / nw,
       "traffic_bytes_per_launch": (fetch + write) / nf,
       "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE, separate passes over tools/eval_once.py; WRITE_SIZE is exact "
               "for these stores; FETCH_SIZE is taken as reported (MI355X_MICROARCH.md halves only 16-B-per-lane "
               "streaming reads; this kernel's activation reads are dword-per-lane, and on the single-Cout-tile "
               "layers FETCH_SIZE comes out at 1.1x the input bytes, i.e. not halved)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))

#!/usr/bin/env python3
"""Per kernel family: HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB per dispatch)
over tools/eval_once.py -- what bench.py reports as roofline.traffic.

usage: traffic_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <workload> <batch>
                          [<launch table csv of bench.py --launch-table at the same workload and batch>]

Corrections (MI355X_MICROARCH.md, HBM section): WRITE_SIZE is exact for 16-B-per-lane streaming stores and float
atomics.  FETCH_SIZE reports HALF the bytes of wide coalesced streaming reads (16 B per lane): doubled for the
kernels whose reads are of that kind (GroupNorm: calibrated -- 2 x FETCH_SIZE = WRITE_SIZE = the tensor's bytes);
taken as reported for the conv / attention kernels, whose activation reads are dword-per-lane (calibrated on the
single-Cout-tile 3x3 layers: FETCH_SIZE = 1.1 x the input bytes, i.e. not halved).  `fetch_scale` records the factor.
The convs' RESIDUAL reads are 16-B-per-lane streaming reads (the quad epilogue), so FETCH_SIZE tallies them at half:
with the launch table (which says which launches add a residual and how big it is) the missing half is added back per
family (`residual_half_bytes_added_per_launch`)."""
import csv
import json
import sys

FAMILIES = {  # family -> (kernel-name substrings, FETCH_SIZE correction factor)
    "conv_fwd_h2_k3": (("conv_fwd_h2_kernel<3", "conv3x3_h2_m16_kernel"), 1.0),
    "conv_fwd_h2_k1": (("conv_fwd_h2_kernel<1", "conv1x1_h2_persist_kernel", "conv1x1_h2_m16_kernel",
                        "conv1x1_h2_ws_kernel"), 1.0),
    "attention_fwd_h2": (("attn_fwd_h2_kernel",), 1.0),
    "groupnorm_fwd": (("gn_fwd_reg_kernel", "gn_fwd_small_kernel", "gn_cat_fwd", "gn_fwd_cat", "groupnorm_fwd_kernel"), 2.0),
}


def totals(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for fam, (subs, _) in FAMILIES.items():
            if any(s in r["Kernel_Name"] for s in subs):
                n, s = out.get(fam, (0, 0.0))
                out[fam] = (n + 1, s + float(r["Counter_Value"]) * 1024.0)
    return out


fetch, write = totals(sys.argv[1], "FETCH_SIZE"), totals(sys.argv[2], "WRITE_SIZE")
# residual bytes per family from the launch table: shape "B=.. Cin=.. Cout=.. HxW k=.. .. res=1" -> B * Cout * H * W * 4
res_bytes, table_launches = {}, {}
if len(sys.argv) > 6:
    import re
    for r in csv.DictReader(open(sys.argv[6])):
        fam = r["kernel"]
        table_launches[fam] = table_launches.get(fam, 0) + 1
        m = re.search(r"B=(\d+) .*Cout=(\d+) (\d+)x(\d+) .*res=1", r["shape"])
        if m:
            b, co, hh, ww = map(int, m.groups())
            res_bytes[fam] = res_bytes.get(fam, 0.0) + 4.0 * b * co * hh * ww
kernels = {}
for fam, (_, scale) in FAMILIES.items():
    if fam not in fetch or fam not in write:
        continue
    (nf, f), (nw, w) = fetch[fam], write[fam]
    assert nf == nw and nf > 0, (fam, nf, nw)
    add = 0.0
    if fam in res_bytes and table_launches.get(fam) == nf:  # same dispatch list as the launch table
        add = 0.5 * res_bytes[fam]
    kernels[fam] = {"launches": nf, "fetch_bytes_per_launch": (f * scale + add) / nf, "write_bytes_per_launch": w / nw,
                    "traffic_bytes_per_launch": (f * scale + add + w) / nf, "fetch_scale": scale,
                    "residual_half_bytes_added_per_launch": add / nf}
out = {"workload": sys.argv[4], "batch": int(sys.argv[5]), "kernels": kernels,
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs over one eager evaluation "
               "(tools/eval_once.py) at this batch; corrections per MI355X_MICROARCH.md, see tools/traffic_summary.py"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))

#!/usr/bin/env python3
"""profiles/rNN_*_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_run.sh, for one kernel.

  python scripts/pmc_kernel_traffic.py <pmc dir> <out.json> <kernel regex> <particles> <bytes per scalar> '<config json>' ['<command>']

Mean over the dispatches whose name matches.  Units and the gfx950 correction follow MI355X_MICROARCH.md: both counters are in
KB; FETCH_SIZE counts 16-B/lane streamed reads at half their bytes, so half of the kernel's known streamed read (6 scalars x
particles) is added back; WRITE_SIZE is exact.  Also prints the SQ counters of the same kernel (per dispatch)."""
import csv
import glob
import json
import re
import sys

root, out, pat, particles, esz, config = sys.argv[1], sys.argv[2], sys.argv[3], int(float(sys.argv[4])), int(sys.argv[5]), json.loads(sys.argv[6])
command = sys.argv[7] if len(sys.argv) > 7 else "bench.py"


def rows(sub):
    path = glob.glob("%s/%s/*/*counter_collection.csv" % (root, sub))[0]
    return [r for r in csv.DictReader(open(path)) if re.search(pat, r["Kernel_Name"])]


def mean_kb(sub, counter):
    vals = [float(r["Counter_Value"]) for r in rows(sub) if r["Counter_Name"] == counter]
    return sum(vals) / len(vals), len(vals)


fetch, nf = mean_kb("fetch", "FETCH_SIZE")
write, nw = mean_kb("write", "WRITE_SIZE")
fetch_b, write_b = fetch * 1024, write * 1024
stream = 6.0 * esz * particles
fetch_corr = fetch_b + 0.5 * stream
sq = {}
for r in rows("sq"):
    sq.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
sq = {k: sum(v) / len(v) for k, v in sq.items()}
json.dump({
    "bytes_per_launch": fetch_corr + write_b, "fetch_size_raw_bytes": fetch_b, "write_size_raw_bytes": write_b,
    "fetch_size_corrected_bytes": fetch_corr, "dispatches_averaged": [nf, nw], "kernel": pat,
    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `%s` (scripts/pmc_run.sh), mean over the dispatches of the kernel; gfx950 "
              "correction per MI355X_MICROARCH.md: FETCH_SIZE counts 16-B/lane streamed reads at half their bytes, so half of the kernel's known streamed read "
              "(6 scalars x particles) is added back; WRITE_SIZE is exact" % command,
    "algorithmic_bytes_per_launch": 2 * stream, "sq_counters_per_dispatch": sq,
    "config": config,
}, open(out, "w"), indent=1)
print(open(out).read())

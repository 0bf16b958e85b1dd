#!/usr/bin/env python3
"""profiles/rNN_c3_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_c3.sh.

  python scripts/pmc_c3_traffic.py gpurun_out/pmc_c3 profiles/r02_c3_traffic.json [particles] [grid]

Mean over the in-place dispatches of push3_tiles_kernel<float>.  Units and the gfx950 correction follow
MI355X_MICROARCH.md: both counters are in KB; FETCH_SIZE counts 16-B/lane streamed reads at half their bytes, so
half of the kernel's known streamed read (24 B x particles) is added back; WRITE_SIZE is exact."""
import csv
import glob
import json
import re
import sys

root, out = sys.argv[1], sys.argv[2]
particles = int(float(sys.argv[3])) if len(sys.argv) > 3 else 500000000
grid = int(sys.argv[4]) if len(sys.argv) > 4 else 256
PAT = r"push3_tiles_kernel<float, false, false, false"


def mean_kb(sub, counter):
    path = glob.glob("%s/%s/*/*counter_collection.csv" % (root, sub))[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and re.search(PAT, r["Kernel_Name"])]
    return sum(vals) / len(vals), len(vals)


fetch, nf = mean_kb("fetch", "FETCH_SIZE")
write, nw = mean_kb("write", "WRITE_SIZE")
fetch_b, write_b = fetch * 1024, write * 1024
fetch_corr = fetch_b + 0.5 * 24 * particles
json.dump({
    "bytes_per_launch": fetch_corr + write_b, "fetch_size_raw_bytes": fetch_b, "write_size_raw_bytes": write_b,
    "fetch_size_corrected_bytes": fetch_corr, "dispatches_averaged": [nf, nw],
    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --only-c3 --steps 4 --warmup 1` (scripts/pmc_c3.sh), mean over "
              "the in-place push3_tiles_kernel<float> dispatches; gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE counts 16-B/lane streamed reads at "
              "half their bytes, so half of the kernel's known streamed read (24 B x particles) is added back; WRITE_SIZE is exact",
    "algorithmic_bytes_per_launch": 48.0 * particles,
    "config": {"workload": "c3", "particles": particles, "grid": grid, "dtype": "f32"},
}, open(out, "w"), indent=1)
print(open(out).read())

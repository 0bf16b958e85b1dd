#!/usr/bin/env python3
"""profiles/rNN_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_bench.sh.

  python scripts/pmc_traffic.py profiles/r01_pmc profiles/r01_traffic.json

Mean over the dispatches of the reference-RNG fused push (in-place and re-binning launches).  Units
and the gfx950 correction follow MI355X_MICROARCH.md: both counters are in KB; FETCH_SIZE counts
16-B/lane streamed reads at half their bytes, so half of the kernel's known streamed read
(41 B x particles) is added back; WRITE_SIZE is exact."""
import csv
import json
import re
import sys

root, out = sys.argv[1], sys.argv[2]
# the configuration the passes were taken on (scripts/pmc_bench.sh runs bench.py's defaults); bench.py attaches
# the figure only to a run of exactly this configuration
particles = int(sys.argv[3]) if len(sys.argv) > 3 else 100000000
grid = int(sys.argv[4]) if len(sys.argv) > 4 else 1024


def mean_kb(path, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter
            and re.search(r"push_tiles_kernel<float, true, (true|false), false, true>", r["Kernel_Name"])]
    return sum(vals) / len(vals), len(vals)


fetch, nf = mean_kb(root + "/fetch_counter_collection.csv", "FETCH_SIZE")
write, nw = mean_kb(root + "/write_counter_collection.csv", "WRITE_SIZE")
fetch_b, write_b = fetch * 1024, write * 1024
fetch_corr = fetch_b + 0.5 * 41 * particles
json.dump({
    "bytes_per_launch": fetch_corr + write_b, "fetch_size_raw_bytes": fetch_b, "write_size_raw_bytes": write_b,
    "fetch_size_corrected_bytes": fetch_corr, "dispatches_averaged": [nf, nw],
    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 5 --warmup 2` "
              "(" + root + "/*.csv), mean over the reference-RNG push_tiles_kernel<float,true,*,false,true> dispatches "
              "(in-place and re-binning launches); gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE counts 16-B/lane "
              "streamed reads at half their bytes, so half of the kernel's known streamed read (41 B x particles) is "
              "added back; WRITE_SIZE is exact",
    "algorithmic_bytes_per_launch": 48.0 * 2 * particles,
    "config": {"particles_per_gpu": particles, "grid": [grid, grid], "rng": "reference", "dtype": "f32"},
}, open(out, "w"), indent=1)
print(open(out).read())

#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — collected separately, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: TCC has 4 slots, FETCH_SIZE costs 3 and WRITE_SIZE 2)
into profiles/<tag>_traffic.json: HBM bytes per launch of every codec kernel.

usage: python tools/collect_traffic.py <fetch_dir> <write_dir> <out.json> [bench arguments the passes ran with]
The bench arguments are recorded ("command", "codec"): bench.py only quotes a traffic file whose command matches its own.

Counter units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB. The guide's gfx950 correction (FETCH_SIZE
reads exactly half of a wide 16-B-per-lane streaming read) is calibrated for that access shape only; the
codec kernels read 4-8 B per lane at scattered addresses, for which the guide calls the absolute value
uncalibrated. Both the raw value and the x2-corrected upper bound are recorded; bench.py reports the raw one.
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def per_kernel(directory, counter):
    files = glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True)
    acc = defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in acc.items() if k.startswith("k_")}, {k: len(v) for k, v in acc.items()}


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    fetch, nf = per_kernel(fetch_dir, "FETCH_SIZE")
    write, nw = per_kernel(write_dir, "WRITE_SIZE")
    codec = "rop"
    words = note.split()
    if "--codec" in words:
        codec = words[words.index("--codec") + 1]
    command = note if "--stage" in note else note + " --stage full"
    workload = words[words.index("--workload") + 1] if "--workload" in words else "enwik"
    nbytes = int(float(words[words.index("--bytes") + 1])) if "--bytes" in words else None
    import datetime
    res = {"note": note, "command": "bench.py " + command, "codec": codec, "unit": "bytes per launch (mean over dispatches)",
           "measured_on": datetime.date.today().isoformat(), "workload": workload, "bytes": nbytes, "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        res["kernels"][k] = {"fetch_raw": round(f), "write": round(w), "hbm_raw": round(f + w),
                             "hbm_fetch_x2_upper": round(2 * f + w), "dispatches": nf.get(k, 0)}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()

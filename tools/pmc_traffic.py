#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command) into
profiles/<tag>_hbm_traffic_pmc.json: HBM bytes per launch of every tb kernel.

    python tools/pmc_traffic.py fetch.csv write.csv out.json "<note about the command>" [images=1024,windows=170,pairs=512]

The optional last argument records the units ONE launch of the profiled command processed (images for the extractor
kernels, BA windows for k_ba_*, stereo pairs for the matcher / pose kernels): bench.py uses a figure as measured only
when its own launch has the same units, and flags it as scaled otherwise.

hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB and, on gfx950, FETCH_SIZE reports
half of wide coalesced reads (MI355X_MICROARCH.md, HBM / rocprofv3 section); the factor is uncalibrated for narrow
gathers, so kernels dominated by those are over-counted."""
import csv
import json
import re
import sys


def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = re.sub(r"[<(].*$", "", name)
        if not name.startswith("k_"):
            continue
        n, tot = acc.get(name, (0, 0.0))
        acc[name] = (n + 1, tot + float(r["Counter_Value"]))
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    units = {}
    if len(sys.argv) > 5:
        for kv in sys.argv[5].split(","):
            k, v = kv.split("=")
            units[k.strip()] = int(v)
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    kernels = {}
    for k in sorted(f):
        n, tot = f[k]
        nw, totw = w.get(k, (n, 0.0))
        fk, wk = tot / n, totw / max(nw, 1)
        kernels[k] = {"launches_sampled": n, "fetch_size_kb_raw": round(fk, 1), "write_size_kb": round(wk, 1),
                      "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
    doc = {"note": note, "kernels": kernels}
    if units:
        doc["units_per_launch"] = units
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in kernels.items():
        print("%-16s %4d launches  %10.1f MB / launch" % (k, v["launches_sampled"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()

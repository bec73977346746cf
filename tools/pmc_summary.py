#!/usr/bin/env python3
"""Per-kernel mean of every counter in rocprofv3 counter_collection CSVs: python tools/pmc_summary.py a.csv [b.csv ...] [--kernel substr]"""
import csv
import re
import sys
from collections import defaultdict

files = [a for a in sys.argv[1:] if not a.startswith("--")]
filt = None
if "--kernel" in sys.argv:
    filt = sys.argv[sys.argv.index("--kernel") + 1]
    files = [f for f in files if f != filt]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for path in files:
    for r in csv.DictReader(open(path)):
        name = re.sub(r"[(].*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))
        if filt and filt not in name:
            continue
        a = acc[name][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(acc):
    print(k)
    for c, (n, tot) in sorted(acc[k].items()):
        print("   %-28s %14.1f  (mean of %d dispatches)" % (c, tot / n, n))

#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv compactly: name, calls, average us, share."""
import csv
import signal
import sys

signal.signal(signal.SIGPIPE, signal.SIG_DFL)  # `| head` closes the pipe early

for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:44]:44s} {r['Calls']:>6s} avg {float(r['AverageNs']) / 1e3:9.1f} us  tot {float(r['TotalDurationNs']) / 1e6:9.3f} ms  {float(r['Percentage']):5.1f}%")

#!/usr/bin/env python3
"""Print a compact per-kernel summary of a rocprofv3 --kernel-trace --stats CSV directory."""
import csv, glob, sys
d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(f"{r['Name'][:72]:72s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:9.3f} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={float(r['Percentage']):6.2f}")

#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel from the rocpd database under a directory:
   python tools/pmc_summary.py <dir> [kernel-substring]"""
import os, sqlite3, sys
path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
hits = [os.path.join(r, f) for r, _, fs in os.walk(path) for f in fs if f.endswith(".db")]
db = sqlite3.connect(hits[0])
rows = db.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id), sum(duration) from counters_collection group by 1, 2").fetchall()
for k, c, v, n, d in sorted(rows):
    if pat in k:
        print(f"{k[:48]:48s} {c:28s} {v:14.5g}  ({n} dispatches)")

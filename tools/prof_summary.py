"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel table:
   python tools/prof_summary.py gpurun_out/prof/x_results.db [out.md [out.json]]
out.json: {kernel family (same key as tools/pmc_traffic.py): {calls, total_ms, avg_us}} -- read by bench.py's per-class HBM figures."""
import json
import os
import re
import sqlite3
import sys


def family(name):
    name = re.sub(r"^void\s+", "", name)
    return re.sub(r"\(.*$", "", name)


def short(name):
    return family(name)[:90]


def main():
    path = sys.argv[1]
    if os.path.isdir(path):                       # rocprofv3 -d <dir>: find the rocpd database below it
        hits = [os.path.join(r, f) for r, _, fs in os.walk(path) for f in fs if f.endswith(".db")]
        if not hits:
            sys.exit("no .db under " + path)
        path = hits[0]
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg, fam = {}, {}
    for n, s, e in rows:
        f = fam.setdefault(family(n), [0, 0.0])
        f[0] += 1
        f[1] += (e - s) / 1e6
        k = short(n)
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {k} | {a[0]} | {a[1] / 1e3:.3f} | {a[1] / a[0]:.1f} | {a[2]:.1f} | {a[3]:.1f} | {100 * a[1] / total:.1f} |")
    lines.append(f"\ntotal GPU kernel time: {total / 1e3:.3f} ms over {len(rows)} dispatches")
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    if len(sys.argv) > 3:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import provenance
        out = {k: {"calls": v[0], "total_ms": round(v[1], 4), "avg_us": round(1e3 * v[1] / v[0], 2)} for k, v in fam.items()}
        out["_provenance"] = provenance.stamp()              # (not a kernel: readers skip keys that start with "_")
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()

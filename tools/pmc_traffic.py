#!/usr/bin/env python3
"""HBM traffic per launch of every kernel family of a bench.py run, from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950: MI355X_MICROARCH.md "rocprofv3 PMC slots").

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <dirF> -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d <dirW> -- python bench.py ...
    python tools/pmc_traffic.py <dirF> <dirW> <out.json> [<out.md>]

Units / corrections (same guide, "HBM"): both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of
wide coalesced reads at 64 bytes, so it is DOUBLED; WRITE_SIZE is taken as is.
"""
import json
import os
import re
import sqlite3
import sys


def per_kernel(path, counter):
    hits = [os.path.join(r, f) for r, _, fs in os.walk(path) for f in fs if f.endswith(".db")]
    db = sqlite3.connect(hits[0])
    rows = db.execute("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection "
                      "where counter_name = ? group by 1", (counter,)).fetchall()
    out = {}
    for k, v, n in rows:
        fam = re.sub(r"^void\s+", "", k)
        fam = re.sub(r"\(.*$", "", fam)
        out[fam] = (out.get(fam, (0.0, 0))[0] + v, out.get(fam, (0.0, 0))[1] + n)
    return out


def main():
    dir_f, dir_w, out_json = sys.argv[1:4]
    out_md = sys.argv[4] if len(sys.argv) > 4 else None
    F, W = per_kernel(dir_f, "FETCH_SIZE"), per_kernel(dir_w, "WRITE_SIZE")
    res = {}
    for k in sorted(set(F) | set(W)):
        f, nf = F.get(k, (0.0, 0))
        w, nw = W.get(k, (0.0, 0))
        n = max(nf, nw)
        if n == 0:
            continue
        res[k] = {"launches": n, "fetch_bytes_per_launch": 2.0 * f * 1024 / max(nf, 1), "write_bytes_per_launch": w * 1024 / max(nw, 1)}
        res[k]["hbm_bytes_per_launch"] = res[k]["fetch_bytes_per_launch"] + res[k]["write_bytes_per_launch"]
    # the group bench.py reports as the dominant kernel: every pre-split / in-kernel-split bf16x3 forward + data-gradient GEMM
    grp = [v for k, v in res.items() if k.startswith(("conv_igemm_spx_kernel", "conv_igemm_sp_kernel", "conv_igemm_halo_kernel", "conv_igemm_rowhalo_stream_kernel"))]
    n = sum(v["launches"] for v in grp)
    summary = {"kernels": res}
    if n:
        summary["conv_igemm_sp"] = {
            "launches": n,
            "fetch_bytes_per_launch": sum(v["fetch_bytes_per_launch"] * v["launches"] for v in grp) / n,
            "write_bytes_per_launch": sum(v["write_bytes_per_launch"] * v["launches"] for v in grp) / n,
        }
        summary["conv_igemm_sp"]["hbm_bytes_per_launch"] = (summary["conv_igemm_sp"]["fetch_bytes_per_launch"] +
                                                            summary["conv_igemm_sp"]["write_bytes_per_launch"])
    # whole-run totals: bytes moved per optimizer step = sum over every kernel of (bytes per launch x launches) / Adam launches
    steps = res.get("adam_kernel", {}).get("launches", 0)
    summary["total_hbm_bytes"] = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in res.values())
    summary["optimizer_steps"] = steps
    if steps:
        summary["hbm_bytes_per_optimizer_step"] = summary["total_hbm_bytes"] / steps
    summary["method"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over the same bench.py command; KiB -> bytes; "
                         "FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 bytes)")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import provenance
    summary["provenance"] = provenance.stamp()
    json.dump(summary, open(out_json, "w"), indent=1)
    if out_md:
        with open(out_md, "w") as f:
            f.write("| kernel | launches | fetch MB / launch (x2 corrected) | write MB / launch | HBM MB / launch |\n|---|---|---|---|---|\n")
            for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]):
                f.write(f"| {k[:70]} | {v['launches']} | {v['fetch_bytes_per_launch'] / 1e6:.2f} | {v['write_bytes_per_launch'] / 1e6:.2f} | "
                        f"{v['hbm_bytes_per_launch'] / 1e6:.2f} |\n")
            if n:
                g = summary["conv_igemm_sp"]
                f.write(f"\nconv_igemm_sp group (bench.py roofline kernel): {g['launches']} launches, "
                        f"{g['hbm_bytes_per_launch'] / 1e6:.2f} MB HBM traffic per launch "
                        f"({g['fetch_bytes_per_launch'] / 1e6:.2f} read + {g['write_bytes_per_launch'] / 1e6:.2f} written)\n")
            f.write("\n" + summary["method"] + "\n")


if __name__ == "__main__":
    main()

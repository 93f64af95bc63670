#!/usr/bin/python3
"""Condense rocprofv3 output directories into the small files committed under profiles/.

    python3 tools/summarize_rocprof.py stats <dir> <out.csv>      # *_kernel_stats.csv of a --kernel-trace --stats run
    python3 tools/summarize_rocprof.py pmc <dir> <out.json> [skip] # *_counter_collection.csv of a --pmc run
    python3 tools/summarize_rocprof.py pmcseq <dir> <out.json> <kernel substring>   # the same, one entry per DISPATCH, in order
    python3 tools/summarize_rocprof.py ranges <dir> <out.json> <bench line file>    # kernel trace of a bench.py run, cut at the
                                                                                    # K1 dispatch ordinals the line reports

`pmc` sums every counter over the dispatches of each kernel (after dropping the first `skip` dispatches of each
kernel: warm-ups; default 2), and reports per-kernel per-dispatch means.  Counter units are left as rocprofv3 reports
them (SQ_* cycle counters count quad-cycles summed over all SIMDs / XCDs; see MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits


def stats(d, out):
    rows = []
    for f in find(d, "kernel_stats.csv"):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "AverageNs", "MinNs", "MaxNs", "StdDev", "TotalDurationNs", "Percentage"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"],
                        r["TotalDurationNs"], r["Percentage"]])


def pmc(d, out, skip=2):
    per = {}
    for f in find(d, "counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"]) + f" grid={r['Grid_Size']}"     # one entry per launch shape of a kernel
                e = per.setdefault(k, {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "lds": int(r["LDS_Block_Size"]),
                                       "grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"]), "disp": {}})
                dd = e["disp"].setdefault(int(r["Dispatch_Id"]), {"t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
                dd[r["Counter_Name"]] = dd.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    res = {}
    for k, e in per.items():
        ids = sorted(e["disp"])[skip:] or sorted(e["disp"])
        names = sorted({c for i in ids for c in e["disp"][i] if c != "t"})
        res[k] = {"dispatches_used": len(ids), "vgpr": e["vgpr"], "sgpr": e["sgpr"], "lds_bytes": e["lds"], "grid": e["grid"],
                  "workgroup": e["wg"], "mean_ns_under_pmc": sum(e["disp"][i]["t"] for i in ids) / len(ids),
                  "per_dispatch_mean": {c: sum(e["disp"][i].get(c, 0.0) for i in ids) / len(ids) for c in names}}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)


def pmcseq(d, out, kernel_substr):
    """Counters of every dispatch whose kernel name contains `kernel_substr`, in dispatch order (for workloads that launch one
    kernel shape in several modes, which `pmc` would merge)."""
    disp = {}
    for f in find(d, "counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if kernel_substr not in r["Kernel_Name"]:
                    continue
                dd = disp.setdefault(int(r["Dispatch_Id"]), {"kernel": short(r["Kernel_Name"]), "grid": int(r["Grid_Size"]),
                                                              "ns_under_pmc": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
                dd[r["Counter_Name"]] = dd.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    with open(out, "w") as fh:
        json.dump([dict(dispatch=i, **disp[i]) for i in sorted(disp)], fh, indent=1)


def trace(d, out, kernel_substr, last_k):
    """Per-dispatch durations of the kernels whose name contains `kernel_substr`, from *_kernel_trace.csv."""
    rows = []
    for f in find(d, "kernel_trace.csv"):
        with open(f) as fh:
            rows += [r for r in csv.DictReader(fh) if kernel_substr in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    tail = ms[-last_k:]
    res = {"kernel": short(rows[0]["Kernel_Name"]) if rows else None, "dispatches": len(ms),
           "mean_ms_all_dispatches_incl_autotune_and_warmup": sum(ms) / len(ms) if ms else None,
           f"mean_ms_last_{last_k}_dispatches_timed_region": sum(tail) / len(tail) if tail else None,
           "min_ms": min(ms) if ms else None, "max_ms": max(ms) if ms else None}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)


def ranges(d, out, bench_json, prefix="k1_"):
    """Cut the kernel trace of a bench.py process at the dispatch ordinals its result line reports
    (roofline.k1_dispatch_ordinals_timed_region and allocation_lottery.kept_pair_timed_like_the_headline.k1_dispatch_ordinals):
    the n-th dispatch whose kernel name starts with `prefix` is the n-th K1 launch of the process."""
    rows = []
    for f in find(d, "kernel_trace.csv"):
        with open(f) as fh:
            rows += [r for r in csv.DictReader(fh) if short(r["Kernel_Name"]).startswith(prefix)]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    with open(bench_json) as fh:
        line = [l for l in fh.read().splitlines() if l.startswith("{") and '"metric"' in l][-1]
    res_line = json.loads(line)
    roof = res_line["roofline"]
    cuts = {"timed_region": roof["k1_dispatch_ordinals_timed_region"]}
    kept = roof.get("allocation_lottery", {}).get("kept_pair_timed_like_the_headline", {})
    if "k1_dispatch_ordinals" in kept:
        cuts["kept_pair_timed_like_the_headline"] = kept["k1_dispatch_ordinals"]
    res = {"k1_dispatches_in_trace": len(rows), "nbytes_per_launch": roof["algorithmic_bytes_per_launch"]}
    for name, (a, b) in cuts.items():
        sel = rows[a:b]
        ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in sel]
        res[name] = {"dispatch_ordinals": [a, b], "kernels": sorted({short(r["Kernel_Name"]) for r in sel}),
                     "grids": sorted({"x".join(r[k] for k in sorted(r) if k.startswith("Grid_Size")) for r in sel}),
                     "mean_ms": sum(ms) / len(ms) if ms else None, "min_ms": min(ms) if ms else None, "max_ms": max(ms) if ms else None,
                     "frac_of_hbm_peak_at_the_trace_mean": (roof["algorithmic_bytes_per_launch"] / (sum(ms) / len(ms) * 1e-3) / 8e12) if ms else None}
    res["timed_region"]["bench_line_kernel_ms_hip_events"] = roof["kernel_ms"]
    res["timed_region"]["buffer_fill_GBps"] = roof.get("buffer_fill_GBps")
    if "kept_pair_timed_like_the_headline" in res:
        res["kept_pair_timed_like_the_headline"]["bench_line_kernel_ms_hip_events"] = kept.get("kernel_ms")
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "trace":
        trace(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]))
    elif sys.argv[1] == "ranges":
        ranges(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "pmcseq":
        pmcseq(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 2)

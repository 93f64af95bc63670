#!/usr/bin/env python3
"""profiles/k1_traffic.json (what bench.py reports as roofline.traffic) from the two counter passes of
tools/gpu_profile_r03.sh:  python3 tools/make_k1_traffic.py <bench_pmc_w.json> <bench_pmc_f.json> <out.json> <round>

WRITE_SIZE and FETCH_SIZE are reported in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports half of wide
coalesced reads).  The headline kernel is the `k1_pairdist_a15_pat<128, ...>` entry with the headline grid
(64 * 512 * 4 workgroups of 256 threads)."""
import json
import sys

w, f, out, rnd = sys.argv[1:5]
B, N, A = 64, 512, 15
GRID = B * N * (N // 128) * 256


def pick(path, counter):
    d = json.load(open(path))
    hits = {k: v for k, v in d.items() if k.startswith("k1_pairdist_a15_pat<128") and k.endswith(f"grid={GRID}")}
    if not hits:
        raise SystemExit(f"no headline K1 entry in {path}: {list(d)}")
    k, v = max(hits.items(), key=lambda kv: kv[1]["dispatches_used"])
    return k, v["per_dispatch_mean"][counter] * 1024.0, v["dispatches_used"]


kw, wbytes, nw = pick(w, "WRITE_SIZE")
kf, fbytes, nf = pick(f, "FETCH_SIZE")
alg = B * N * N * A * A * 5
res = {"B": B, "N_res": N, "N_atom": A, "kernel": kw.split(" grid=")[0],
       "hbm_bytes_per_launch": wbytes + 2 * fbytes, "write_bytes_per_launch": wbytes,
       "fetch_bytes_per_launch_corrected_x2": 2 * fbytes, "algorithmic_bytes_per_launch": alg,
       "ratio_to_algorithmic": (wbytes + 2 * fbytes) / alg, "dispatches_averaged": {"WRITE_SIZE": nw, "FETCH_SIZE": nf},
       "source": f"round {rnd}: rocprofv3 --output-format csv --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes) on "
                 "`python3 bench.py --no-cpu-baseline --steps 3` (tools/gpu_profile_r03.sh); counter unit KiB; FETCH_SIZE "
                 "doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); mean over the headline-kernel "
                 "dispatches of the process (autotune candidate, warm-up, timed)"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))

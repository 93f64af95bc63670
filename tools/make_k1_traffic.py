#!/usr/bin/env python3
"""profiles/k1_traffic.json (what bench.py reports as roofline.traffic) from the two counter passes of
tools/gpu_profile_rNN.sh:  python3 tools/make_k1_traffic.py <bench_pmc_w.json> <bench_pmc_f.json> <out.json> <round> [script]

WRITE_SIZE and FETCH_SIZE are reported in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports half of wide
coalesced reads).  A bench process launches the pattern kernel with every tile length the tuner tries
(`k1_pairdist_a15_pat<128>`, `<64>`, `<32>`; headline grid = 64 * 512 * 512/JT workgroups of 256 threads), so one pair of
passes gives the traffic of each; bench.py reports the entry of the kernel its timed launches took (roofline.kernel)."""
import json
import re
import sys

w, f, out, rnd = sys.argv[1:5]
script = sys.argv[5] if len(sys.argv) > 5 else f"tools/gpu_profile_r{int(rnd):02d}.sh"
B, N, A = 64, 512, 15
alg = B * N * N * A * A * 5


def entries(path, counter):
    res = {}
    for k, v in json.load(open(path)).items():
        m = re.match(r"(k1_pairdist_a15_pat<(\d+).*) grid=(\d+)$", k)
        if m and int(m.group(3)) == B * N * (N // int(m.group(2))) * 256:
            res[m.group(1)] = (v["per_dispatch_mean"][counter] * 1024.0, v["dispatches_used"])
    if not res:
        raise SystemExit(f"no headline K1 entry in {path}")
    return res


ws, fs = entries(w, "WRITE_SIZE"), entries(f, "FETCH_SIZE")
kernels = {}
for k in sorted(set(ws) & set(fs)):
    (wb, nw), (fb, nf) = ws[k], fs[k]
    kernels[k] = {"hbm_bytes_per_launch": wb + 2 * fb, "write_bytes_per_launch": wb, "fetch_bytes_per_launch_corrected_x2": 2 * fb,
                  "ratio_to_algorithmic": (wb + 2 * fb) / alg, "dispatches_averaged": {"WRITE_SIZE": nw, "FETCH_SIZE": nf}}
res = {"B": B, "N_res": N, "N_atom": A, "algorithmic_bytes_per_launch": alg, "kernels": kernels,
       "source": f"round {rnd}: rocprofv3 --output-format csv --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes) on "
                 f"`python3 bench.py --no-cpu-baseline --steps 3` ({script}); counter unit KiB; FETCH_SIZE "
                 "doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); mean over the dispatches of "
                 "each kernel at the headline grid in the process (tuner candidates, warm-up, timed)"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))

#!/usr/bin/env python3
"""profiles/k3_valu_bound.json (what bench.py's other_configs.config3 reports as valu_bound_us) from the SQ counter pass of
tools/gpu_profile_r05.sh over `tools/profile_workload.py k3`:

    python3 tools/make_k3_valu_bound.py <k3_pmc.json> <out.json> <round>

VALU-issue bound of a launch = SQ_ACTIVE_INST_VALU (quad-cycles in which a SIMD issued a VALU instruction, summed over the
chip) x 4 cycles / 1024 SIMDs / 2.4 GHz (MI355X peak engine clock, MI355X_MICROARCH.md): the time the launch's own VALU
instruction stream needs if every SIMD issues one every cycle it can, with no wait for memory, LDS, barriers or launch.
A second figure, issue_bound_us = SQ_INSTS_VALU / 1024 SIMDs x 2.08 ns: the rate at which a SIMD with four waves issues K3's
kind of instruction stream (32 packed fp32 : 1 transcendental) as measured by tools/microbench/valu_issue.hip
(profiles/r05_valu_issue.log: packed ops 4.9-5.0 cycles of 2.4 GHz, v_rsq / v_rcp / v_sqrt 8.6, plain VOP2 3.8) -- the paper
rate of one instruction per 4 cycles at 2.4 GHz is not reached by any packed instruction.
Keys are the kernel names ps_k3_plan_f32 / ps_featuriser_plan_f32 report."""
import json
import re
import sys

src, out, rnd = sys.argv[1:4]
SIMDS, CLOCK_GHZ = 1024, 2.4
ISSUE_NS = 2.08          # ns per wave-instruction per SIMD, mix of 32 packed : 1 v_rsq_f32 at four waves per SIMD (valu_issue.hip)
SWEEP = ("NP", "SRC", "NC", "VEC", "FAITHFUL")
FEAT = ("EXACT", "NC", "VEC", "M16", "WT", "FAITHFUL")


def plan_name(rocprof_name):
    m = re.match(r"(k3_sweep|k3_featurise)<([^>]*)>", rocprof_name)
    if not m:
        return None
    vals = [{"true": "1", "false": "0"}.get(v.strip(), v.strip()) for v in m.group(2).split(",")]
    names = SWEEP if m.group(1) == "k3_sweep" else FEAT
    vals += ["0"] * (len(names) - len(vals))      # a defaulted trailing FAITHFUL
    return m.group(1) + "<" + ",".join(f"{n}={v}" for n, v in zip(names, vals)) + ">"


kernels = {}
for k, v in json.load(open(src)).items():
    name = plan_name(k)
    if not name:
        continue
    c = v["per_dispatch_mean"]
    kernels[name] = {"valu_bound_us": c["SQ_ACTIVE_INST_VALU"] * 4 / SIMDS / (CLOCK_GHZ * 1e3),
                     "issue_bound_us": (c["SQ_INSTS_VALU"] / SIMDS * ISSUE_NS / 1e3) if c.get("SQ_INSTS_VALU") else None,
                     "SQ_ACTIVE_INST_VALU": c["SQ_ACTIVE_INST_VALU"], "SQ_INSTS_VALU": c.get("SQ_INSTS_VALU"),
                     "us_under_pmc": v["mean_ns_under_pmc"] / 1e3, "grid_threads": v["grid"], "vgpr": v["vgpr"],
                     "dispatches_averaged": v["dispatches_used"]}
res = {"B": 128, "N_res": 512, "simds": SIMDS, "clock_GHz": CLOCK_GHZ, "issue_ns_per_wave_instruction": ISSUE_NS, "kernels": kernels,
       "source": f"profiles/k3_valu_bound.json: round {rnd}, rocprofv3 --pmc SQ_ACTIVE_INST_VALU ... on `tools/profile_workload.py k3` "
                 f"(tools/gpu_profile_r{int(rnd):02d}.sh), committed; bound = SQ_ACTIVE_INST_VALU x 4 / {SIMDS} SIMDs / {CLOCK_GHZ} GHz; issue_bound = SQ_INSTS_VALU / {SIMDS} x {ISSUE_NS} ns (tools/microbench/valu_issue.hip); NOT re-measured inside the bench run"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))

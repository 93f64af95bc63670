# Write-request counters of the fused featuriser at aligned / unaligned lengths (round 4)
set -o pipefail
O=gpurun_out/${1:-r04featpmc}; shift
NS="${*:-512 500 496 511}"
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for pass in "q TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "h TCC_HIT_sum TCC_MISS_sum" "r TCC_EA0_RDREQ_sum" "a SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"; do
    set -- $pass; tag=$1; shift
    timeout -k 10 240 rocprofv3 --output-format csv --pmc "$@" -d $O/$tag -o f -- python3 tools/k3_featuriser_shapes.py 3 $NS > $O/$tag.log 2>&1; echo "$tag rc=$?"
    python3 tools/summarize_rocprof.py pmcseq $O/$tag $O/feat_$tag.json k3_featurise
    rm -rf $O/$tag
done
python3 - "$O" $NS <<'P'
import json, sys
O, NS = sys.argv[1], sys.argv[2:]
seqs = {t: json.load(open(f"{O}/feat_{t}.json")) for t in "qhra"}
per = 6
for k, N in enumerate(NS):
    row = {"N": int(N)}
    for t, s in seqs.items():
        grp = s[k * per:(k + 1) * per][3:]
        for key in grp[0]:
            if key in ("dispatch", "kernel", "grid"): continue
            row[key] = round(sum(g[key] for g in grp) / len(grp))
    print(row)
P

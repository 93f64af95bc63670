# Counters of K1's row-phase kernel at the unaligned shapes, per plane (tools/profile_workload.py k1u), round 4:
# write requests (all / 64-byte), read requests and fetched bytes (does a partial-line write make the L2 fetch the line?), SQ
set -o pipefail
O=gpurun_out/${1:-r04x}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for pass in "q TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "r TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "f FETCH_SIZE" "w WRITE_SIZE" "h TCC_HIT_sum TCC_MISS_sum" "a SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
    set -- $pass; tag=$1; shift
    timeout -k 10 240 rocprofv3 --output-format csv --pmc "$@" -d $O/$tag -o k1u -- python3 tools/profile_workload.py k1u 2 > $O/$tag.log 2>&1; echo "$tag rc=$?"
    python3 tools/summarize_rocprof.py pmcseq $O/$tag $O/k1u_$tag.json rowphase
    rm -rf $O/$tag
done
python3 - "$O" <<'P'
import json, sys
O = sys.argv[1]
seqs = {t: json.load(open(f"{O}/k1u_{t}.json")) for t in "qrfwha"}
n = len(seqs["q"])
for i in range(n):
    row = {}
    for t, s in seqs.items():
        if i < len(s):
            row.update({k: (round(v) if isinstance(v, float) else v) for k, v in s[i].items() if k not in ("dispatch", "ns_under_pmc")})
    print(i, row)
P

#!/bin/bash
# Per-kernel register / scratch / LDS usage of one HIP source, from the compiler's own remarks (no GPU needed):
#   tools/kernel_resources.sh protstruc_amd/csrc/pairwise_distance.hip [filter-regex]
# Prints: kernel name, VGPRs, spilled VGPRs, scratch bytes, occupancy (waves/SIMD), static LDS bytes.
src=${1:?source file}; filt=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -c "$src" -o /dev/null \
    -Rpass-analysis=kernel-resource-usage --cuda-device-only 2>&1 |
python3 -c '
import re, sys, subprocess
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+(VGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|TotalSGPRs): (\d+)", line)
    if m and cur: rows[cur][m.group(1)] = int(m.group(2))
names = list(rows)
dem = subprocess.run(["/usr/bin/c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
filt = re.compile(sys.argv[1])
for n, d in zip(names, dem):
    if not filt.search(d): continue
    r = rows[n]
    d = re.sub(r"\(anonymous namespace\)::", "", d); d = re.sub(r"\(.*", "", d)
    print("%-64s vgpr %3d spill %3d scratch %4d occ %d lds %6d" % (d[:64], r.get("VGPRs", -1), r.get("VGPRs Spill", -1), r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1), r.get("LDS Size [bytes/block]", -1)))
' "$filt"

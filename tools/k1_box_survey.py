"""Run the key K1 configurations on whatever box we got and print a one-line summary (box-to-box survey)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
p = torch.cuda.get_device_properties(0)
uid = getattr(p, "uuid", "?")
print("GPU", p.name, "uuid", uid, "CUs", p.multi_processor_count, "clock_kHz", getattr(p, "clock_rate", "?"), flush=True)
try:
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showperflevel", "--showcomputepartition", "--showmemorypartition"], capture_output=True, text=True, timeout=40).stdout
    print("\n".join(l for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "Power", "Performance", "artition")))[:600], flush=True)
except Exception as e:
    print("rocm-smi unavailable:", e)
print(subprocess.run([os.path.join(os.path.dirname(os.path.abspath(__file__)), "microbench", "xcc_census")], capture_output=True, text=True).stdout, flush=True)
os.environ["ROUNDS"] = "4"
sys.argv = ["k1_ab.py", "xcd_remap=0,rows_per_block=1", "xcd_remap=1,rows_per_block=1", "xcd_remap=1,rows_per_block=2",
            "xcd_remap=1,rows_per_block=4", "xcd_remap=0,rows_per_block=4"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "k1_ab.py")).read())

#!/usr/bin/env python3
"""Runs the multi-process GPU rehearsals of a ``pytest -m gpu`` session and leaves one verdict file per rehearsal.

Started by tests/conftest.py at session start, BEFORE the pytest process has touched the GPU (a process that has
initialised the GPU must not start other programs on this pool).  This orchestrator never touches the GPU itself.  It
keeps the number of processes on the card bounded by running in two phases:

  phase A (in parallel):  gloo_world2  tools/rehearse_rowshard.py --world 2 --backend gloo          (2 ranks)
                          rccl_world1  tools/rehearse_rowshard.py --world 1 --backend nccl --quick  (1 rank)
                          c_abi_demo   examples/c_abi_demo.c, compiled with gcc and run             (1 process)
  phase B (after A):      bench_gpus2  python3 bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --no-cpu-baseline
                                       -- the exact command form the driver uses for N > 1, self-launching (2 ranks)

For every name it writes ``<outdir>/<name>.log`` (stdout + stderr; bench_gpus2: stderr only, stdout goes to
``<name>.stdout``) and, when the rehearsal has ended, ``<outdir>/<name>.exit`` holding its exit code.
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--outdir", required=True)
    args = ap.parse_args()
    od = args.outdir
    os.makedirs(od, exist_ok=True)
    rehearse = os.path.join(ROOT, "tools", "rehearse_rowshard.py")
    demo = ("import sys, subprocess; sys.path.insert(0, %r); from protstruc_amd import build; "
            "build.build(verbose=False); exe = build.build_c_example(verbose=False); "
            "sys.exit(subprocess.run([exe]).returncode)" % ROOT)
    phase_a = {
        "gloo_world2": [sys.executable, rehearse, "--out", os.path.join(od, "gloo_world2.json"), "--world", "2",
                        "--backend", "gloo"],
        "rccl_world1": [sys.executable, rehearse, "--out", os.path.join(od, "rccl_world1.json"), "--world", "1",
                        "--backend", "nccl", "--quick"],
        "c_abi_demo": [sys.executable, "-c", demo],
    }
    phase_b = {
        "bench_gpus2": [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"],
    }

    def run_phase(cmds, timeout):
        procs = {}
        for name, cmd in cmds.items():
            log = open(os.path.join(od, name + ".log"), "w")
            out = open(os.path.join(od, name + ".stdout"), "w") if name.startswith("bench") else log
            procs[name] = subprocess.Popen(cmd, stdout=out, stderr=log, cwd=ROOT)
        for name, p in procs.items():
            try:
                code = p.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                p.kill()                  # exactly the child we started
                p.wait()
                code = -9
            with open(os.path.join(od, name + ".exit.tmp"), "w") as f:
                f.write(str(code))
            os.replace(os.path.join(od, name + ".exit.tmp"), os.path.join(od, name + ".exit"))

    run_phase(phase_a, 300)      # (normally ~40 s; the two phases together stay inside a 900 s test-tier limit)
    run_phase(phase_b, 420)


if __name__ == "__main__":
    main()

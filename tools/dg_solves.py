"""Whole DG time steps with the device solves (bench.py's `with_solves` leg of --variant dg alone, no CPU leg): iteration
counts, time per step and set-up time; KNPEMI_AMG_VERBOSE=1 prints the hierarchy.  Meant to be run under rocprofv3
--kernel-trace --stats to see which solver kernels carry the time."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--solve-steps", type=int, default=10)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    import bench
    import torch
    out = bench.run_dg(a, torch, cpu=True, rank=1)      # rank != 0: no CPU restatement leg
    print(json.dumps({"ms_per_step": out["ms_per_step"], "with_solves": out.get("with_solves")}))

#!/usr/bin/env python3
"""Row kernels by themselves: `reps` back-to-back launches of the EMI and KNP assemblies on one workload's mesh (fields at
the perturbed initial state), each bracketed by HIP events.  usage: tools/time_rows.py [workload] [reps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (sets the package paths)
from knpemi import _lib as L  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "config3"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    import argparse
    args = argparse.Namespace(scaling="weak", knp_twice=False, no_overlap=True, frozen_state=True)
    case, stepper, _ = bench.build_problem(workload, args, 0, 1)
    dp, lib = stepper.dp, stepper.dp.lib
    case.s.perturb()
    stepper.reset()
    out = {}
    for name, fn, flags in (("emi_rows_kernel", lib.knpemi_assemble_emi, stepper.flags_emi),
                            ("knp_rows_kernel", lib.knpemi_assemble_knp, stepper.flags_knp)):
        for _ in range(3):
            L.check(fn(dp.h, flags))
        dp.sync()
        kid = L.KERNEL_NAMES.index(name)
        L.check(lib.knpemi_profile(dp.h, 1 << kid))
        for _ in range(reps):
            L.check(fn(dp.h, flags))
        n, ms = C.c_int64(), C.c_double()
        L.check(lib.knpemi_profile_read(dp.h, kid, C.byref(n), C.byref(ms)))
        L.check(lib.knpemi_profile(dp.h, 0))
        out[name] = ms.value / max(1, n.value) * 1e3
    survey, design, sizes = bench.algorithmic_bytes(case, dp)
    print(workload, {k: (round(v, 2), round(survey[k] / (v * 1e-6) / 1e9 / bench.HBM_PEAK_GBS, 3)) for k, v in out.items()},
          "(us, fraction of 8 TB/s by the SURVEY accounting)")


if __name__ == "__main__":
    main()

"""Host wall time of the phases of whole time steps with the device solves (bench.py's `with_solves` state: the recorded
trajectory replayed to step 10, then real solves), each phase closed by a device synchronisation: where a step's time goes
without a profiler attached.  usage: python tools/step_breakdown.py [--workload config2] [--steps 20]"""
import argparse
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    import torch
    from knpemi import _lib as L
    args = argparse.Namespace(scaling="weak", knp_twice=False, no_overlap=False, frozen_state=False)
    with contextlib.redirect_stdout(io.StringIO()):
        case, stepper, halo = bench.build_problem(a.workload, args, 0, 1)
        phi_t, c_t, its, rhs = bench.record_trajectory(case, stepper, bench.WITH_SOLVES_START + a.steps + 4, halo)
    replay = bench.Replay(case, stepper, halo, phi_t, c_t, torch)
    replay.restart(bench.WITH_SOLVES_START)
    dp, lib = stepper.dp, stepper.dp.lib
    its = {"emi": [], "knp": []}
    solve_emi, solve_knp = bench.device_solvers(case, its)
    phases = {"ode + emi assembly": 0.0, "emi solve": 0.0, "knp assembly": 0.0, "knp solve + update": 0.0}
    t_emi = [0.0]
    t_knp = [0.0]

    def emi(d):
        d.sync()
        t0 = time.perf_counter()
        solve_emi(d)
        d.sync()
        t_emi[0] += time.perf_counter() - t0

    def knp(d):
        d.sync()
        t0 = time.perf_counter()
        solve_knp(d)
        d.sync()
        t_knp[0] += time.perf_counter() - t0
    stepper.solve_emi, stepper.solve_knp = emi, knp
    for _ in range(2):
        stepper.step(halo)
    dp.sync()
    t_emi[0] = t_knp[0] = 0.0
    its["emi"].clear(); its["knp"].clear()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        stepper.step(halo)
    dp.sync()
    total = time.perf_counter() - t0
    n = a.steps
    print(f"{a.workload}: {total / n * 1e3:.3f} ms per step with a synchronisation before and after each solve; "
          f"EMI solve {t_emi[0] / n * 1e6:.1f} us ({sum(its['emi']) / n:.2f} CG iterations), "
          f"KNP solve + update {t_knp[0] / n * 1e6:.1f} us ({sum(its['knp']) / n:.2f} BiCGStab iterations), "
          f"rest (sweep, assemblies) {(total - t_emi[0] - t_knp[0]) / n * 1e6:.1f} us")


if __name__ == "__main__":
    main()

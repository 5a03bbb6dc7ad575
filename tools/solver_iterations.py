"""Iteration counts / timings of the device Krylov solves (diagnostic)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import conftest  # noqa
from helpers import Setup
from knpemi.pdeSolver import create_solver_emi, create_solver_knp
from knpemi import _lib as L
import contextlib, io
cases = [("2d", 1), ("tet", 0), ("hex", 0), ("2d", 3), ("tet", 1)]
for kind, r in cases:
    with contextlib.redirect_stdout(io.StringIO()):
        s = Setup(kind, r); s.perturb(); s.phi[1].x.array[:] += -0.0744
        for t in s.subdomain_list:
            for k in range(2):
                s.c[t][k].x.array[:] = s.c_prev[t][k].x._a
        emi = create_solver_emi(s.a_emi, s.L_emi, s.phi, s.entity_maps, s.subdomain_list, None, direct=True, p=s.p_emi)
        A, b = emi.assemble()
    dp = emi.dp
    n = A.shape[0]
    print(kind, r, "n", n, flush=True)
    for pc in (L.PC_AMG, L.PC_JACOBI):
        dp.solver_setup(L.B_EMI, pc)
        for rtol in ((1e-5, 1e-5, 1e-10) if pc else (1e-5,)):
            for f, tag in ((s.phi[0], 0), (s.phi[1], 1)):
                f.x.array[:] = 0
                dp.push(L.F_PHI, dp.sub_index[tag], 0, f)
            try:
                dp.sync(); t0 = time.time(); res = dp.solve(L.B_EMI, rtol, 1e-40, 3000); dp.sync(); t1 = time.time()
                print("  emi pc", pc, "rtol", rtol, res, "%.2f ms" % ((t1 - t0) * 1e3), dp.solver_info(L.B_EMI), flush=True)
            except Exception as e:
                print("  emi pc", pc, rtol, e)
    dp.solver_setup(L.B_EMI, L.PC_AMG)
    with contextlib.redirect_stdout(io.StringIO()):
        knp = create_solver_knp(s.a_knp, s.L_knp, s.c, s.entity_maps, s.subdomain_list, None, direct=True, p=s.p_knp)
        Ak, bk = knp.assemble()
    c0 = {(t, k): s.c[t][k].x._a.copy() for t in s.subdomain_list for k in range(2)}
    for pc, rtol in ((0, 1e-7), (0, 1e-12), (1, 1e-7), (1, 1e-7), (1, 1e-12)):
        dp.solver_setup(L.B_KNP, pc) if rtol == 1e-7 else None
        for (t, k), v in c0.items():
            s.c[t][k].x.array[:] = v
        for t in s.subdomain_list:
            for k in range(2):
                dp.push(L.F_C, dp.sub_index[t], k, s.c[t][k])
        try:
            dp.sync(); t0 = time.time(); res = dp.solve(L.B_KNP, rtol, 1e-40, 3000); dp.sync(); t1 = time.time()
            print("  knp pc", pc, "rtol", rtol, res, "%.2f ms" % ((t1 - t0) * 1e3), flush=True)
        except Exception as e:
            print("  knp", rtol, e)

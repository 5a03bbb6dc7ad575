"""AMG strength-threshold sweep on the hexahedral boxes (debug aid)."""
import sys, os, time, contextlib, io
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import conftest  # noqa
from helpers import Setup
from knpemi.pdeSolver import create_solver_emi, create_solver_knp
from knpemi import _lib as L
for kind, r in [("hex", 0), ("hex", 1)]:
    with contextlib.redirect_stdout(io.StringIO()):
        s = Setup(kind, r); s.perturb(); s.phi[1].x.array[:] += -0.0744
        for t in s.subdomain_list:
            for k in range(2):
                s.c[t][k].x.array[:] = s.c_prev[t][k].x._a
        emi = create_solver_emi(s.a_emi, s.L_emi, s.phi, s.entity_maps, s.subdomain_list, None, direct=True, p=s.p_emi)
        A, b = emi.assemble()
    dp = emi.dp
    print(kind, r, "n", A.shape[0], flush=True)
    for theta in (0.02, 0.08, 0.15, 0.25, 0.4, 0.6):
        dp.solver_setup(L.B_EMI, L.PC_AMG, theta)
        for f, tag in ((s.phi[0], 0), (s.phi[1], 1)):
            f.x.array[:] = 1e-3 * theta
            dp.push(L.F_PHI, dp.sub_index[tag], 0, f)
        try:
            dp.sync(); t0 = time.time()
            res = dp.solve(L.B_EMI, 1e-8, 1e-40, 500); dp.sync(); t1 = time.time()
            print("  theta", theta, res, "%.2f ms" % ((t1 - t0) * 1e3), dp.solver_info(L.B_EMI), flush=True)
        except Exception as e:
            print("  theta", theta, e)

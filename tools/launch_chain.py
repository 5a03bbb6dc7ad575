"""What a dependent trivial kernel costs inside a bench-like process (torch initialised first, the handle's stream), through
knpemi_debug_launch_chain; `--no-torch` leaves torch out (the library then runs on /opt/rocm's HIP runtime)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "knp-emi-fenics-x_amd"), os.path.join(ROOT, "examples", "idealized_geometries")]
if "--no-torch" not in sys.argv:
    import torch
    torch.cuda.init()
    torch.zeros(8, device="cuda")
import contextlib, io
from knpemi import _lib as L
from setup_problem import Setup
with contextlib.redirect_stdout(io.StringIO()):
    s = Setup("tet", 0)
dp = s.a_emi.dp
for kind, name in ((0, "empty one-wave kernel"), (1, "y = x + 1 over 26 417 doubles"), (2, "start_kernel (one block)")):
    out = []
    for graph in (0, 1):
        us = C.c_double()
        L.check(dp.lib.knpemi_debug_launch_chain(dp.h, kind, 26417, 12, 400, graph, C.byref(us)))
        out.append(us.value)
    print(f"{name:36s} stream {out[0]:6.2f} us   graph {out[1]:6.2f} us per kernel")

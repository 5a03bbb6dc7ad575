"""Generates the golden fixtures under tests/golden/ from the oracle (run here, committed with
its output).  The reference cannot be executed offline, so these are the
oracle's own outputs on seeded inputs: they pin the oracle against regressions and give the GPU
tests fixed vectors; they are NOT outputs of the reference itself ("parity unpinned").

    python tests/golden/make_golden.py
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in ("knp-emi-fenics-x_amd", "oracle", "examples/idealized_geometries", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

import knpemi_oracle as o  # noqa: E402
from helpers import Setup  # noqa: E402

SEED = 12345


def probe_vector(n, seed):
    return np.random.default_rng(seed).uniform(-1.0, 1.0, n)


def assembly_fixture(kind, r, full):
    with contextlib.redirect_stdout(io.StringIO()):
        s = Setup(kind, r, build_forms=False)
    s.perturb(SEED)
    _, P, params, ions = s.oracle()
    c_all, phi, phiM, mm = s.oracle_fields()
    out = {}
    for split in (True, False):
        A, Pm, b = o.assemble_emi(P, params, ions, c_all, phiM, mm, splitting_scheme=split)
        Ak, bk = o.assemble_knp(P, params, ions, c_all, phi, phiM, mm, s.dt, splitting_scheme=split)
        tag = "split" if split else "nosplit"
        v, vk = probe_vector(A.shape[0], 1), probe_vector(Ak.shape[0], 2)
        out.update({f"{tag}_b_emi": b, f"{tag}_b_knp": bk,
                    f"{tag}_A_emi_v": A @ v, f"{tag}_P_emi_v": Pm @ v, f"{tag}_A_knp_v": Ak @ vk,
                    f"{tag}_A_emi_diag": A.diagonal(), f"{tag}_A_knp_diag": Ak.diagonal(),
                    f"{tag}_nnz": np.array([A.nnz, Pm.nnz, Ak.nnz])})
        if full and split:
            for name, M in (("A_emi", A), ("P_emi", Pm), ("A_knp", Ak)):
                M = M.tocsr()
                M.sort_indices()
                out[f"{name}_indptr"], out[f"{name}_indices"], out[f"{name}_data"] = M.indptr, M.indices, M.data
    np.savez_compressed(os.path.join(HERE, f"assembly_{kind}_r{r}.npz"), **out)
    print(kind, r, {k: v.shape for k, v in list(out.items())[:3]})


def ode_fixture():
    F, R, T = 96485.0, 8.314, 300.0
    out = {}
    cases = {
        "hh_si": dict(dt=1e-4, stim=10.0, Cm=0.02, psi=F / (R * T),
                      conc=dict(K_e=3.3236967382705265, K_i=124.15397583491901,
                                Na_e=100.71925900027354, Na_i=12.838513108648856)),
        "hh_mv": dict(dt=0.1, stim=1.0, Cm=1.0, psi=96500e3 / (8.315e3 * 307e3),
                      conc=dict(K_e=3.092970607490389, K_i=99.3100014897692, Na_e=144.66, Na_i=15.5)),
        "glial": dict(dt=0.1, stim=0.0, Cm=1.0, psi=96500e3 / (8.315e3 * 307e3),
                      conc=dict(K_e=3.092970607490389, K_i=99.3100014897692, Na_e=144.66, Na_i=15.5)),
    }
    for model, cs in cases.items():
        M = o.MODELS[model]
        ix = M["pidx"]
        for stim in sorted({0.0, cs["stim"]}):
            p = np.array(M["params"], float)
            p[ix["Cm"]], p[ix["psi"]] = cs["Cm"], cs["psi"]
            for k, v in cs["conc"].items():
                p[ix[k]] = v
            p[ix["Cl_e"]] = p[ix["Na_e"]] + p[ix["K_e"]]
            p[ix["Cl_i"]] = p[ix["Na_i"]] + p[ix["K_i"]]
            p[ix["z_Na"]], p[ix["z_K"]], p[ix["z_Cl"]] = 1.0, 1.0, -1.0
            p[ix["stim_amplitude"]] = stim
            y = np.array(M["states"], float)
            key = f"{model}_stim{stim:g}"
            out[f"{key}_p0"], out[f"{key}_y0"] = p.copy(), y.copy()
            out[f"{key}_rhs0"] = np.array(M["rhs"](y, 0.0, p.copy()))
            st, pa = y[None, :].copy(), p[None, :].copy()
            traj = []
            for k in range(10):
                o.ode_sweep(model, st, pa, k * cs["dt"], cs["dt"])
                traj.append(np.concatenate([st[0], pa[0, o._ich_slice(len(p))]]))
            out[f"{key}_traj"] = np.array(traj)   # rows: state (n_s) + side-effect currents (3)
            out[f"{key}_dt"] = np.array(cs["dt"])
    np.savez_compressed(os.path.join(HERE, "ode_models.npz"), **out)
    print("ode", len(out))


if __name__ == "__main__":
    assembly_fixture("2d", 1, full=True)
    assembly_fixture("tet", 0, full=False)
    assembly_fixture("hex", 0, full=False)
    ode_fixture()

"""Oracle-side adapters (TEST INFRASTRUCTURE): turn a driver set-up object (duck-typed: `mesh`, `ct`,
`ft`, `ion_list`, `c_prev`, `phi`, ... as built by examples/idealized_geometries/setup_problem.py) into
the plain arrays and floats the oracle works on.  Imported by tests/, smoke() and bench.py's
cpu_baseline leg only."""
import numpy as np

import knpemi_oracle as o


def oracle_problem(s, subdomains=None):
    mesh, ct, ft = s.mesh, s.ct, s.ft
    if subdomains is None:
        subdomains = {tag: list(sd.get("membrane_tags", [])) for tag, sd in s.subdomain_list.items()}
    P = o.OracleProblem(mesh.x, mesh.cells, mesh.cell_type, ct.dense(), mesh.facets[ft.indices],
                        ft.values, subdomains)
    pp = s.physical_parameters
    params = dict(dt=float(s.dt), F=float(pp['F']), psi=float(pp['psi']), C_M=float(pp['C_M']),
                  C_phi=float(pp['C_phi']))
    ions = [dict(name=i['name'], z=i['z'], D={t: float(i['D'][t]) for t in s.subdomain_list})
            for i in s.ion_list]
    return o, P, params, ions


def oracle_fields(s):
    tags = list(s.subdomain_list)
    c_all = {t: [f.x._a.copy() for f in s.c_prev[t]] + [s.ion_list[-1][f'c_{t}'].x._a.copy()] for t in tags}
    phi = {t: s.phi[t].x._a.copy() for t in tags}
    phiM = {t: s.phi_M_prev[t].x._a.copy() for t in tags if t > 0}
    mm = {t: [dict(tag=m['ode'].tag,
                   I_ch_k={n: (f.x._a.copy() if hasattr(f, 'x') else float(f)) for n, f in m['I_ch_k'].items()})
              for m in s.subdomain_list[t].get('mem_models', [])] for t in tags if t > 0}
    return c_all, phi, phiM, mm

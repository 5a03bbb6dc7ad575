"""Form descriptors: what `emi_system` / `knp_system` return instead of UFL forms."""
from __future__ import annotations

import numpy as np

from .fem.mesh import compute_interface_data


class FormDescriptor:
    """Opaque stand-in for a UFL form: names the integrals (`system`, `role`) and
    carries references to the coefficient Functions, exactly the objects the
    reference's UFL expressions hold on to."""

    def __init__(self, system, role, shared):
        self.system, self.role, self.shared = system, role, shared

    def __getattr__(self, name):
        try:
            return self.__dict__["shared"][name]
        except KeyError:
            raise AttributeError(name) from None

    def __repr__(self):
        return f"<knpemi {self.system}.{self.role} form descriptor>"


class Measure:
    """`dx(tag)`-style handle: (kind, subdomain id) -> integration entities."""

    def __init__(self, kind, mesh, data):
        self.kind, self.mesh, self.data = kind, mesh, data

    def __call__(self, tag):
        return (self.kind, tag, self.data.get(tag))


class Measures:
    def __init__(self, mesh, ct, ft):
        dx = {int(t): ct.find(t) for t in np.unique(ct.values)}
        ds = {int(t): ft.find(t) for t in np.unique(ft.values)}
        dS = {}
        ptr, _, _ = mesh.facet_cells()
        for t in np.unique(ft.values):
            facets = ft.find(t)
            interior = facets[(ptr[facets + 1] - ptr[facets]) == 2]
            dS[int(t)] = compute_interface_data(ct, interior).flatten()
        self.dx, self.dS, self.ds = Measure("dx", mesh, dx), Measure("dS", mesh, dS), Measure("ds", mesh, ds)

    def as_tuple(self):
        return self.dx, self.dS, self.ds


def bind_membrane_models(dp, subdomain_list, ion_list):
    """Attach every MembraneModel of every cell to its slot of the device problem."""
    names = [ion['name'] for ion in ion_list]
    for tag, subdomain in subdomain_list.items():
        if tag == 0:
            continue
        for j, mm in enumerate(subdomain.get('mem_models', [])):
            ode = mm['ode']
            if hasattr(ode, "_bind"):
                ode._bind(dp, dp.sub_index[tag], j, names)

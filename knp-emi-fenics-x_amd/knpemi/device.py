"""Device problem: the host-side owner of one `knpemi_handle` (include/knpemi_hip.h).

It flattens the mixed-dimensional mesh data the reference keeps in DOLFINx /
scifem objects (sub-meshes, membrane sub-mesh, entity maps, interface data:
`examples/idealized_geometries/run_3D.py:156-171`, `src/knpemi/emiWeakForm.py:28-51`)
into the plain arrays of `knpemi_problem_desc`, mirrors `Function.x.array`
contents to the GPU on demand (version-tracked, so unchanged arrays are not
re-uploaded) and exposes the assembled operators as scipy CSR matrices.
"""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np
import scipy.sparse as sp

from . import _lib as L

try:                                    # content fingerprint of a host array (DeviceProblem._stamp)
    import xxhash

    def _digest(a):
        return xxhash.xxh3_64_intdigest(memoryview(np.ascontiguousarray(a)).cast("B"))
except ImportError:                     # pragma: no cover -- zlib is always there
    import zlib

    def _digest(a):
        m = memoryview(np.ascontiguousarray(a)).cast("B")
        return (zlib.crc32(m) << 32) | zlib.adler32(m)
from .fem.function import as_float

_CELL_KIND = {"triangle": L.TRIANGLE, "tetrahedron": L.TETRAHEDRON, "hexahedron": L.HEXAHEDRON}
_MODEL_IDS = {"hh_si": L.MODEL_HH_SI, "hh_mv": L.MODEL_HH_MV, "glial": L.MODEL_GLIAL}


def _ptr_array(arrays, ctype):
    arr = (ctype * len(arrays))()
    for i, a in enumerate(arrays):
        arr[i] = a.ctypes.data_as(ctype) if a is not None else ctype()
    return arr


def flatten_problem(mesh, ft, subdomain_list):
    """Plain-array description of the problem (the contents of knpemi_problem_desc).

    Returns a dict of numpy arrays; `facet_tag[s]` holds the facet tag of every
    membrane facet of sub-domain s (mapped to membrane-model indices later).
    """
    tags = list(subdomain_list.keys())
    if tags[0] != 0:
        raise ValueError("the first sub-domain must be the ECS with tag 0 (run_3D.py:146)")
    subs = [subdomain_list[t] for t in tags]
    ecs = subs[0]["mesh_sub"]
    ft_dense = ft.dense()
    out = dict(tags=tags, x=[], cells=[], n_q=[0], facet_e=[None], facet_i=[None], facet_q=[None],
               facet_tag=[None], q_to_e=[None], q_to_i=[None], q_x=[None])
    for sd in subs:
        m = sd["mesh_sub"]
        if m.cell_type != mesh.cell_type:
            raise ValueError("sub-mesh cell type differs from the parent mesh")
        out["x"].append(np.ascontiguousarray(m.x, np.float64))
        out["cells"].append(np.ascontiguousarray(m.cells, np.int32))
    for sd in subs[1:]:
        mem = sd["mesh_mem"]
        ics = sd["mesh_sub"]
        pv = mem.parent_vertices[mem.cells]          # (nF, nf) parent vertex ids
        e = np.searchsorted(ecs.parent_vertices, pv)
        i = np.searchsorted(ics.parent_vertices, pv)
        ok = (e < ecs.num_vertices) & (i < ics.num_vertices)
        if not (ok.all() and np.array_equal(ecs.parent_vertices[e], pv)
                and np.array_equal(ics.parent_vertices[i], pv)):
            raise RuntimeError("a membrane facet is not shared by the ECS and the cell "
                               "(Facet is assumed to be an interior facet, utils.py:46)")
        out["n_q"].append(mem.num_vertices)
        out["facet_e"].append(np.ascontiguousarray(e, np.int32))
        out["facet_i"].append(np.ascontiguousarray(i, np.int32))
        out["facet_q"].append(np.ascontiguousarray(mem.cells, np.int32))
        out["facet_tag"].append(ft_dense[mem.parent_entities].astype(np.int32))
        out["q_to_e"].append(np.searchsorted(ecs.parent_vertices, mem.parent_vertices).astype(np.int32))
        out["q_to_i"].append(np.searchsorted(ics.parent_vertices, mem.parent_vertices).astype(np.int32))
        out["q_x"].append(mem.x)
    return out


class DeviceProblem:
    """One GPU-resident KNP-EMI problem (topology + fields + operators)."""

    _registry = weakref.WeakValueDictionary()

    def __init__(self, mesh, ct, ft, subdomain_list, ion_list, device=None):
        lib = L.load()
        if lib.knpemi_device_count() < 1:
            raise RuntimeError("no HIP device visible: the knpemi hot path runs on MI355X only "
                               "(there is no CPU fallback)")
        if not 2 <= len(ion_list) <= L.MAX_IONS:
            raise NotImplementedError(f"2 to {L.MAX_IONS} ionic species (the last one eliminated) are supported; the "
                                      "reference drivers use three")
        self.lib = lib
        self.mesh, self.ct, self.ft = mesh, ct, ft
        self.subdomain_list = subdomain_list
        self.tags = list(subdomain_list.keys())
        self.sub_index = {t: s for s, t in enumerate(self.tags)}
        self.K = len(ion_list)
        flat = flatten_problem(mesh, ft, subdomain_list)
        self.flat = flat
        S = len(self.tags)
        # membrane models: index in subdomain['mem_models'] by facet tag
        n_models = [0] * S
        facet_model = [None] * S
        self.models = {}
        for s in range(1, S):
            mms = subdomain_list[self.tags[s]].get("mem_models", [])
            n_models[s] = len(mms)
            fm = np.full(flat["facet_tag"][s].shape[0], -1, np.int32)
            for j in reversed(range(len(mms))):  # first model with a given tag wins
                fm[flat["facet_tag"][s] == int(mms[j]["ode"].tag)] = j
            facet_model[s] = fm
            for j, mm in enumerate(mms):
                self.models[(s, j)] = mm
        self.n_vert = np.array([a.shape[0] for a in flat["x"]], np.int32)
        self.n_cell = np.array([a.shape[0] for a in flat["cells"]], np.int32)
        self.n_q = np.array(flat["n_q"], np.int32)
        self.n_facet = np.array([0] + [a.shape[0] for a in flat["facet_e"][1:]], np.int32)
        self.n_models = np.array(n_models, np.int32)
        self.voff = np.concatenate([[0], np.cumsum(self.n_vert)]).astype(np.int64)
        self._keep = (flat, facet_model)
        desc = L.ProblemDesc()
        desc.gdim = mesh.gdim
        desc.cell_kind = _CELL_KIND[mesh.cell_type]
        desc.n_sub = S
        desc.n_ions = self.K
        desc.n_vert = L.iptr(self.n_vert)
        desc.n_cell = L.iptr(self.n_cell)
        desc.x = _ptr_array(flat["x"], L.c_dbl_p)
        desc.cells = _ptr_array(flat["cells"], L.c_int_p)
        desc.n_q = L.iptr(self.n_q)
        desc.n_facet = L.iptr(self.n_facet)
        desc.facet_e = _ptr_array(flat["facet_e"], L.c_int_p)
        desc.facet_i = _ptr_array(flat["facet_i"], L.c_int_p)
        desc.facet_q = _ptr_array(flat["facet_q"], L.c_int_p)
        desc.facet_model = _ptr_array(facet_model, L.c_int_p)
        desc.q_to_e = _ptr_array(flat["q_to_e"], L.c_int_p)
        desc.q_to_i = _ptr_array(flat["q_to_i"], L.c_int_p)
        desc.n_models = L.iptr(self.n_models)
        uc = getattr(mesh, "uniform_cell", None)      # edge vectors of every cell of a generated uniform box mesh
        if uc is not None and mesh.cell_type in ("hexahedron", "tetrahedron"):     # (tetrahedra: the grid they were split from)
            for i, v in enumerate(np.asarray(uc, np.float64).reshape(9)):
                desc.uniform_cell[i] = float(v)
        if device is None:
            import os
            device = int(os.environ.get("LOCAL_RANK", "0")) % lib.knpemi_device_count()
        h = C.c_void_p()
        L.check(lib.knpemi_create(C.byref(desc), int(device), C.byref(h)))
        self.h = h
        self.device = int(device)
        self._uploaded = {}       # (field, sub, idx) -> (id(vector), version)
        self._patterns = {}
        self._params_key = None
        self.ion_names = [ion["name"] for ion in ion_list]

    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            self.lib.knpemi_destroy(h)
            self.h = None

    # -- registry: emi_system / knp_system / MembraneModel share one handle ----
    @classmethod
    def get(cls, mesh, ct, ft, subdomain_list, ion_list):
        key = id(mesh)
        dp = cls._registry.get(key)
        if dp is None or dp.subdomain_list is not subdomain_list:
            dp = cls(mesh, ct, ft, subdomain_list, ion_list)
            cls._registry[key] = dp
            mesh._knpemi_device_problem = dp   # keeps the problem alive with the mesh
        return dp

    # -- parameters ---------------------------------------------------------------
    def set_params(self, physical_params, ion_list, dt):
        dt = as_float(dt)
        F, psi = as_float(physical_params["F"]), as_float(physical_params["psi"])
        C_M = as_float(physical_params["C_M"])
        # its own entry of the parameter dictionary in the reference (run_2D.py:187,208): the EMI forms read C_phi, the KNP
        # forms C_M / dt (emiWeakForm.py:164,231-236; knpWeakForm.py:181-182)
        C_phi = as_float(physical_params["C_phi"]) if "C_phi" in physical_params else C_M / dt
        if not C_phi > 0.0:
            raise ValueError("C_phi must be positive")
        rho = physical_params.get("rho", {})
        key = (dt, F, psi, C_M, C_phi, tuple(ion["z"] for ion in ion_list),
               tuple(as_float(ion["D"][t]) for ion in ion_list for t in self.tags),
               as_float(rho.get("z", 0.0)), tuple(as_float(rho.get(t, 0.0)) for t in self.tags))
        if key == self._params_key:
            return
        p = L.Params()
        p.dt, p.F, p.psi, p.C_M, p.C_phi = dt, F, psi, C_M, C_phi
        for k, ion in enumerate(ion_list):
            p.z[k] = float(ion["z"])
            for s, t in enumerate(self.tags):
                p.D[s][k] = as_float(ion["D"][t])
        p.rho_z = as_float(rho.get("z", 0.0))
        for s, t in enumerate(self.tags):
            p.rho[s] = as_float(rho.get(t, 0.0))
        L.check(self.lib.knpemi_set_params(self.h, C.byref(p)))
        self._params_key = key

    # -- Function I/O ----------------------------------------------------------------
    def push(self, field, sub, idx, fn):
        """Upload `fn.x` if it changed since the last upload of this field."""
        vec = fn.x
        key = (field, sub, idx)
        stamp = self._stamp(vec)
        if self._uploaded.get(key) == stamp:
            return
        a = vec._a
        L.check(self.lib.knpemi_set_field(self.h, field, sub, idx, L.dptr(a), a.shape[0]))
        self._uploaded[key] = stamp

    @staticmethod
    def _stamp(vec):
        """What "unchanged since the last upload" means: same vector object, no `.array` access since (`version`),
        and the same bytes -- a caller may keep a view (`a = f.x.array; ...; a[:] = v`, common in DOLFINx driver code)
        and write through it without touching `.array` again.  The content is fingerprinted with a 64-bit hash of the
        whole buffer (xxh3: ~10 GB/s, one pass over host memory, far cheaper than the upload it may save); a sum or a
        sampled dot product would miss permuted or compensating edits."""
        a = vec._a
        return (id(vec), vec.version, a.shape[0], _digest(a))

    def push_array(self, field, sub, idx, a):
        a = np.ascontiguousarray(a, np.float64)
        L.check(self.lib.knpemi_set_field(self.h, field, sub, idx, L.dptr(a), a.shape[0]))
        self._uploaded.pop((field, sub, idx), None)

    def pull(self, field, sub, idx, fn):
        """Download a device field into `fn.x` and mark it as in sync."""
        vec = fn.x
        a = vec._a
        L.check(self.lib.knpemi_get_field(self.h, field, sub, idx, L.dptr(a), a.shape[0]))
        vec.version += 1
        self._uploaded[(field, sub, idx)] = self._stamp(vec)

    def pull_array(self, field, sub, idx, n):
        a = np.empty(n, np.float64)
        L.check(self.lib.knpemi_get_field(self.h, field, sub, idx, L.dptr(a), n))
        return a

    def trace(self, sub, ue, ui):
        nq = int(self.n_q[sub])
        qe, qi = np.empty(nq), np.empty(nq)
        ue = np.ascontiguousarray(ue, np.float64)
        ui = np.ascontiguousarray(ui, np.float64)
        if ue.shape[0] != self.n_vert[0] or ui.shape[0] != self.n_vert[sub]:
            raise ValueError("trace: function size does not match the sub-mesh")
        L.check(self.lib.knpemi_trace(self.h, sub, L.dptr(ue), L.dptr(ui), L.dptr(qe), L.dptr(qi)))
        return qe, qi

    # -- operators ------------------------------------------------------------------
    def assemble_emi(self, want_p=True, splitting_scheme=True):
        flags = (L.WANT_P if want_p else 0) | (0 if splitting_scheme else L.NO_SPLITTING)
        L.check(self.lib.knpemi_assemble_emi(self.h, flags))

    def assemble_knp(self, splitting_scheme=True, early_membrane=False):
        """A_knp and b_knp.  `early_membrane`: the two-part form of the membrane integrals
        (knpemi_assemble_knp_membrane_early + KNPEMI_MEMBRANE_EARLY, here back to back on one stream; the stepper can
        run the first part beside the EMI solve); default: the one-part facet kernel."""
        flags = 0 if splitting_scheme else L.NO_SPLITTING
        if early_membrane:
            L.check(self.lib.knpemi_assemble_knp_membrane_early(self.h, flags))
            flags |= L.MEMBRANE_EARLY
        L.check(self.lib.knpemi_assemble_knp(self.h, flags))

    def _pattern(self, which):
        key = which if which != L.P_EMI else L.A_EMI
        if key not in self._patterns:
            n, nnz = C.c_int64(), C.c_int64()
            L.check(self.lib.knpemi_csr_dims(self.h, which, C.byref(n), C.byref(nnz)))
            rp = np.empty(n.value + 1, np.int32)
            ci = np.empty(max(nnz.value, 1), np.int32)
            L.check(self.lib.knpemi_get_csr_pattern(self.h, which, L.iptr(rp), L.iptr(ci)))
            self._patterns[key] = (n.value, nnz.value, rp, ci[:nnz.value])
        return self._patterns[key]

    def csr(self, which):
        n, nnz, rp, ci = self._pattern(which)
        vals = np.empty(max(nnz, 1), np.float64)
        L.check(self.lib.knpemi_get_csr_values(self.h, which, L.dptr(vals)))
        return sp.csr_matrix((vals[:nnz], ci, rp), shape=(n, n))

    def rhs(self, which):
        n = self._pattern(L.A_EMI if which == L.B_EMI else L.A_KNP)[0]
        b = np.empty(n, np.float64)
        L.check(self.lib.knpemi_get_rhs(self.h, which, L.dptr(b)))
        return b

    def set_csr_values(self, which, vals):
        """Caller-supplied operator values in the layout `csr(which).data` has (knpemi_set_csr_values)."""
        vals = np.ascontiguousarray(vals, np.float64)
        assert vals.size == self._pattern(which)[1]
        L.check(self.lib.knpemi_set_csr_values(self.h, which, L.dptr(vals)))

    def set_rhs(self, which, b):
        b = np.ascontiguousarray(b, np.float64)
        assert b.size == self._pattern(L.A_EMI if which == L.B_EMI else L.A_KNP)[0]
        L.check(self.lib.knpemi_set_rhs(self.h, which, L.dptr(b)))

    def set_solution(self, which, x):
        x = np.ascontiguousarray(x, np.float64)
        L.check(self.lib.knpemi_set_solution(self.h, which, x.ctypes.data_as(C.c_void_p), 0))

    def get_solution(self, which, n):
        x = np.empty(n, np.float64)
        L.check(self.lib.knpemi_get_solution(self.h, which, L.dptr(x)))
        return x

    def solve(self, which, rtol, atol, maxit=10000):
        """Device Krylov solve of the assembled system (`knpemi_solve_emi/knp`); returns (iterations, relres)."""
        it, rr = C.c_int(), C.c_double()
        fn = self.lib.knpemi_solve_emi if which == L.B_EMI else self.lib.knpemi_solve_knp
        L.check(fn(self.h, float(rtol), float(atol), int(maxit), C.byref(it), C.byref(rr)))
        return it.value, rr.value

    def solver_setup(self, which, precond, theta=0.0):
        L.check(self.lib.knpemi_solver_setup(self.h, which, precond, float(theta)))

    def solver_info(self, which):
        lev, b, oc = C.c_int(), C.c_int(), C.c_double()
        L.check(self.lib.knpemi_solver_info(self.h, which, C.byref(lev), C.byref(oc), C.byref(b)))
        return dict(levels=lev.value, op_complexity=oc.value, builds=b.value)

    def update_pde(self):
        L.check(self.lib.knpemi_update_pde(self.h))

    def sync(self):
        L.check(self.lib.knpemi_sync(self.h))

    def timer_start(self):
        L.check(self.lib.knpemi_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_double()
        L.check(self.lib.knpemi_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

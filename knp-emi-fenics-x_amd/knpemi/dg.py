"""Host side of the DG + symmetric-interior-penalty variant (include/knpemi_hip.h, `knpemi_dg_*`):
broken P1 on triangles and tetrahedra, broken Q1 on hexahedra (the cell type of the reference's own 3-D idealized mesh,
`examples/idealized_geometries/make_mesh_3D.py:100-102`; tensor-product vertex order, as `create_box` produces).

The reference has no such class -- its code is continuous Galerkin on sub-meshes; only `README.md:5-7` and the marker
convention of `examples/idealized_geometries/make_mesh_2D.py:88-90` ("interior facets tagged 0") refer to the DG method
of the legacy solver.  `DGProblem` therefore takes what those mesh scripts produce (one mesh, a cell marker, a facet
marker) and the `ion_list` / `physical_params` dictionaries of the drivers (`run_2D.py:204-251`), and exposes the
assembled systems as scipy CSR matrices the way `knpemi.device.DeviceProblem` does for the CG path.

Broken dof (cell c, local vertex j) = c * nv + j; fields are arrays of shape (n_cells, nv).  Membrane node
(facet f, vertex a) = f * nf + a; membrane fields are (n_mem_facets, nf).  All arithmetic happens in the HIP kernels
(csrc/kernels_dg.hip); without the extension or a GPU the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import scipy.sparse as sp

from . import _lib as L

_KIND = {"triangle": L.TRIANGLE, "tetrahedron": L.TETRAHEDRON, "hexahedron": L.HEXAHEDRON}


class DGProblem:
    def __init__(self, mesh, ct, ft, subdomain_tags, membrane_tags, n_ions=3, device=0):
        """`ct`, `ft`: cell and facet MeshTags of `mesh` (or dense arrays); `subdomain_tags`: cell tags in sub-domain
        order, ECS first (`run_2D.py:145-169`); `membrane_tags`: facet tags that mark membranes."""
        if mesh.cell_type not in _KIND:
            raise ValueError("the DG variant is built for triangles, tetrahedra and hexahedra")
        self.lib = L.load()
        self.mesh = mesh
        self.nv = mesh.cells.shape[1]
        self.nf = mesh.facets.shape[1]                      # vertices per facet
        self.K = int(n_ions)
        cell_tags = np.asarray(ct.dense() if hasattr(ct, "dense") else ct)
        tag_to_sub = {int(t): i for i, t in enumerate(subdomain_tags)}
        try:
            self.cell_sub = np.array([tag_to_sub[int(t)] for t in cell_tags], np.int32)
        except KeyError as e:
            raise ValueError(f"cell tag {e} is not in subdomain_tags") from None
        self.n_sub = len(subdomain_tags)
        if hasattr(ft, "indices"):
            sel = np.isin(ft.values, list(membrane_tags))
            fidx = np.sort(ft.indices[sel])
            order = np.argsort(ft.indices[sel], kind="stable")
            self.mem_tags = np.asarray(ft.values[sel])[order]
        else:
            dense = np.asarray(ft)
            fidx = np.flatnonzero(np.isin(dense, list(membrane_tags)))
            self.mem_tags = dense[fidx]
        self.mem_facets = np.ascontiguousarray(mesh.facets[fidx], np.int32).reshape(-1, self.nf)
        self.n_cells = mesh.num_cells
        self.n = self.n_cells * self.nv
        self.nmf = self.mem_facets.shape[0]
        x = np.ascontiguousarray(mesh.x, np.float64)
        cells = np.ascontiguousarray(mesh.cells, np.int32)
        desc = L.DGDesc(_KIND[mesh.cell_type], self.n_sub, self.K, self.n_cells, mesh.num_vertices, self.nmf,
                        L.dptr(x), L.iptr(cells), L.iptr(self.cell_sub), L.iptr(self.mem_facets))
        h = C.c_void_p()
        L.check(self.lib.knpemi_dg_create(C.byref(desc), int(device), C.byref(h)))
        self.h = h
        self.device = int(device)
        n, nnz, nq = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(self.lib.knpemi_dg_dims(self.h, C.byref(n), C.byref(nnz), C.byref(nq)))
        assert n.value == self.n and nq.value == self.nmf * self.nf
        self.nnz = nnz.value
        self.indptr = np.zeros(self.n + 1, np.int32)
        self.indices = np.zeros(self.nnz, np.int32)
        L.check(self.lib.knpemi_dg_get_pattern(self.h, L.iptr(self.indptr), L.iptr(self.indices)))
        self.X = x[cells]                                   # (n_cells, nv, d) coordinates of the broken dofs
        self.XM = x[self.mem_facets]                        # (nmf, nf, d)
        self.gamma = 10.0

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                self.lib.knpemi_dg_destroy(h)
            except Exception:
                pass

    # -- parameters and fields ---------------------------------------------------------------------------------
    def set_params(self, physical_params, ion_list, gamma=None, rho=None):
        """`physical_params`: dt, F, psi, C_M (floats or knpemi.fem.Constant); `ion_list`: dicts with 'z' and 'D'
        (sequence or {sub-domain index: value}); `rho`: (rho_z, per-sub-domain densities) or None."""
        val = lambda v: float(getattr(v, "value", v))
        p = L.DGParams()
        p.dt, p.F, p.psi, p.C_M = (val(physical_params[k]) for k in ("dt", "F", "psi", "C_M"))
        if gamma is not None:
            self.gamma = float(gamma)
        p.gamma = self.gamma
        if len(ion_list) != self.K:
            raise ValueError("ion_list does not match n_ions")
        for k, ion in enumerate(ion_list):
            p.z[k] = val(ion["z"])
            for s in range(self.n_sub):
                p.D[s][k] = val(ion["D"][s])
        if rho is not None:
            p.rho_z = val(rho[0])
            for s in range(self.n_sub):
                p.rho[s] = val(rho[1][s])
        L.check(self.lib.knpemi_dg_set_params(self.h, C.byref(p)))

    def _set(self, field, idx, a, n):
        a = np.ascontiguousarray(a, np.float64).reshape(-1)
        if a.size != n:
            raise ValueError("field has the wrong size")
        L.check(self.lib.knpemi_dg_set_field(self.h, field, idx, L.dptr(a), n))

    def _get(self, field, idx, n, shape):
        a = np.zeros(n)
        L.check(self.lib.knpemi_dg_get_field(self.h, field, idx, L.dptr(a), n))
        return a.reshape(shape)

    def set_concentration(self, k, c):
        self._set(L.DG_C, k, c, self.n)

    def get_concentration(self, k):
        return self._get(L.DG_C, k, self.n, (self.n_cells, self.nv))

    def set_potential(self, phi):
        self._set(L.DG_PHI, 0, phi, self.n)

    def get_potential(self):
        return self._get(L.DG_PHI, 0, self.n, (self.n_cells, self.nv))

    def set_membrane_potential(self, phi_M):
        self._set(L.DG_PHI_M, 0, phi_M, self.nmf * self.nf)

    def get_membrane_potential(self):
        return self._get(L.DG_PHI_M, 0, self.nmf * self.nf, (self.nmf, self.nf))

    def set_channel_current(self, k, I):
        self._set(L.DG_I_CH, k, I, self.nmf * self.nf)

    def get_channel_current(self, k):
        return self._get(L.DG_I_CH, k, self.nmf * self.nf, (self.nmf, self.nf))

    def set_source(self, k, f):
        self._set(L.DG_SOURCE, k, f, self.n)

    def membrane_dofs(self):
        e = np.zeros(self.nmf * self.nf, np.int32)
        i = np.zeros(self.nmf * self.nf, np.int32)
        L.check(self.lib.knpemi_dg_get_membrane_dofs(self.h, L.iptr(e), L.iptr(i)))
        return e.reshape(self.nmf, self.nf), i.reshape(self.nmf, self.nf)

    # -- the hot path ------------------------------------------------------------------------------------------
    def assemble_emi(self, splitting_scheme=True):
        L.check(self.lib.knpemi_dg_assemble_emi(self.h, 0 if splitting_scheme else L.NO_SPLITTING))

    def assemble_knp(self, splitting_scheme=True):
        L.check(self.lib.knpemi_dg_assemble_knp(self.h, 0 if splitting_scheme else L.NO_SPLITTING))

    def matrix(self, which):
        """which = 0: potential system; 1 + k: concentration system of solved ion k."""
        v = np.zeros(self.nnz)
        L.check(self.lib.knpemi_dg_get_values(self.h, which, L.dptr(v)))
        return sp.csr_matrix((v, self.indices, self.indptr), shape=(self.n, self.n))

    def rhs(self, which):
        b = np.zeros(self.n)
        L.check(self.lib.knpemi_dg_get_rhs(self.h, which, L.dptr(b)))
        return b

    def update(self, c_new):
        """End of step: c_prev <- c_new (K-1, n), eliminated ion, phi_M <- phi_i - phi_e."""
        a = np.ascontiguousarray(c_new, np.float64).reshape(-1)
        if a.size != (self.K - 1) * self.n:
            raise ValueError("c_new has the wrong size")
        L.check(self.lib.knpemi_dg_update(self.h, a.ctypes.data_as(C.c_void_p), 0))

    # -- device solves ------------------------------------------------------------------------------------------
    def solve_emi(self, rtol=1e-5, atol=1e-40, maxit=1000):
        """CG + auxiliary-space AMG on the assembled potential system, constants projected out (pdeSolver.py:24-35,
        74-78); the potential field takes the solution.  Returns (iterations, relative residual)."""
        it, rr = C.c_int(), C.c_double()
        L.check(self.lib.knpemi_dg_solve_emi(self.h, rtol, atol, maxit, C.byref(it), C.byref(rr)))
        return it.value, rr.value

    def solve_knp(self, rtol=1e-7, atol=1e-40, maxit=1000, update=False):
        """BiCGStab + AMG on the K - 1 assembled concentration systems (pdeSolver.py:99-110), starting from the previous
        concentrations; update=True runs the end-of-step update on the solution without leaving the device."""
        it, rr = C.c_int(), C.c_double()
        L.check(self.lib.knpemi_dg_solve_knp(self.h, rtol, atol, maxit, C.byref(it), C.byref(rr), int(bool(update))))
        return it.value, rr.value

    def set_extrapolation(self, on=True):
        """Both solves start from 2 x_n - x_(n-1) instead of x_n."""
        L.check(self.lib.knpemi_dg_set_extrapolation(self.h, int(bool(on))))

    def solution(self):
        """The concentrations of the last solve_knp, (K-1, n)."""
        c = np.zeros((self.K - 1, self.n))
        L.check(self.lib.knpemi_dg_get_solution(self.h, L.dptr(c)))
        return c

    # -- membrane ODEs -----------------------------------------------------------------------------------------
    def ode_bind(self, model_id, states_row, params_row, ion_param, v_index):
        """One state / parameter row per membrane node, all initialised to the given rows (MembraneModel.__init__,
        odeSolver.py:40-55); ion_param[3 k + (0, 1, 2)] = parameter columns of ion k's ECS trace, intracellular trace
        and channel current."""
        nq = self.nmf * self.nf
        full = lambda a: np.ascontiguousarray(a, np.float64) if np.ndim(a) == 2 else np.tile(np.asarray(a, np.float64), (nq, 1))
        st, pr = full(states_row), full(params_row)      # a full [n_mem_nodes][n] table (e.g. a stimulus written into
        if st.shape[0] != nq or pr.shape[0] != nq:        # the rows of some nodes) or one row for all nodes
            raise ValueError("state / parameter tables need one row per membrane node")
        ip = np.ascontiguousarray(ion_param, np.int32)
        self._ode_shape = (st.shape[1], pr.shape[1])
        L.check(self.lib.knpemi_dg_ode_bind(self.h, model_id, st.shape[1], pr.shape[1], L.dptr(st), L.dptr(pr),
                                            L.iptr(ip), int(v_index)))

    def ode_step(self, t0, dt, rtol=1e-8, atol=1e-10, set_v=True, set_traces=True):
        flags = (L.ODE_SET_V if set_v else 0) | (L.ODE_SET_TRACES if set_traces else 0)
        L.check(self.lib.knpemi_dg_ode_step(self.h, t0, dt, rtol, atol, flags))

    def ode_tables(self):
        nq = self.nmf * self.nf
        st = np.zeros((nq, self._ode_shape[0]))
        pr = np.zeros((nq, self._ode_shape[1]))
        L.check(self.lib.knpemi_dg_ode_get_tables(self.h, L.dptr(st), L.dptr(pr)))
        return st, pr

    def ode_stats(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(self.lib.knpemi_dg_ode_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def update_device(self, c_new_ptr):
        """As `update`, with c_new (K-1, n) already in device memory (pointer as int)."""
        L.check(self.lib.knpemi_dg_update(self.h, C.c_void_p(c_new_ptr), 1))

    def profile(self, on):
        L.check(self.lib.knpemi_dg_profile(self.h, int(on)))      # True / 1: every launch, n: every n-th, 0: off

    def profile_read(self, which):
        """(launches, average microseconds) of the bracketed launches of kernel `which` since the last read."""
        n, ms = C.c_int64(), C.c_double()
        L.check(self.lib.knpemi_dg_profile_read(self.h, which, C.byref(n), C.byref(ms)))
        return n.value, (ms.value / n.value * 1e3 if n.value else 0.0)

    def sync(self):
        L.check(self.lib.knpemi_dg_sync(self.h))

    def time_kernel(self, which, reps=20, splitting_scheme=True):
        ms = C.c_double()
        L.check(self.lib.knpemi_dg_time_kernel(self.h, which, 0 if splitting_scheme else L.NO_SPLITTING, reps, C.byref(ms)))
        return ms.value


class DGSlab:
    """One rank's part of the DG problem on `make_mesh_3D(r, cell, l)` (cell = "tetrahedron": broken P1 on the 6-tet split,
    "hexahedron": broken Q1) cut into x-slabs of whole hexahedron layers: its own layers plus ONE ghost layer on each interior side (the cells across the cut facets), generated
    locally with the global coordinates (`make_mesh_3D_slab`) -- nothing global is ever built.

    Rows of owned cells are complete: their facet neighbours are all local.  Once per step the ghost cells' dofs (K
    concentrations + potential) are refreshed from the owning neighbour rank (`exchange`: pack kernel -> point-to-point
    over `torch.distributed` -> unpack kernel, on the handle's stream with RCCL, host-staged with gloo).  Membrane nodes
    on ghost cells are integrated redundantly from identical inputs (the ODE sweep is deterministic), so phi_M and I_ch
    need no exchange -- the same arrangement as the CG path (knpemi/fem/distributed.py).
    """

    def __init__(self, r, l, rank, world, n_ions=3, device=0, cell="tetrahedron"):
        from .fem.idealized import make_mesh_3D_slab
        from .fem.distributed import Halo
        nx = l * 16 * 2 ** r
        cuts = [(nx * k) // world for k in range(world + 1)]
        self.a, self.b = cuts[rank], cuts[rank + 1]                 # owned hexahedron layers [a, b)
        if self.b - self.a < 1:
            raise ValueError("fewer hexahedron layers than ranks")
        lo, hi = max(self.a - 1, 0), min(self.b + 1, nx)
        mesh, ct, ft = make_mesh_3D_slab(r, cell, l, (lo, hi))
        self.dp = DGProblem(mesh, ct, ft, [0, 1], [1], n_ions=n_ions, device=device)
        dp = self.dp
        nxl = hi - lo
        self.cells_per_hex = 6 if cell == "tetrahedron" else 1
        self.layer = lo + (np.arange(dp.n_cells) // self.cells_per_hex) % nxl   # global layer of every local cell (x fastest)
        self.owned_cells = (self.layer >= self.a) & (self.layer < self.b)
        self.rank, self.world, self.nx = rank, world, nx

        def dofs_of_layer(g):
            c = np.flatnonzero(self.layer == g)                     # ascending local id = the same order on both ranks
            return (c[:, None] * dp.nv + np.arange(dp.nv)[None, :]).ravel().astype(np.int64)
        plan = {}
        if rank > 0:
            plan["low"] = dict(nb=rank - 1, send=dofs_of_layer(self.a), recv=dofs_of_layer(self.a - 1))
        if rank < world - 1:
            plan["high"] = dict(nb=rank + 1, send=dofs_of_layer(self.b - 1), recv=dofs_of_layer(self.b))
        self.plan = plan
        self._halo = Halo()          # reuses its packed-buffer plan and transport (mode agreed once, at attach)
        self._d = None

    def attach(self):
        import torch
        import torch.distributed as dist
        h, dp = self._halo, self.dp
        h.dp, h.L, h.torch, h.dist = dp, L, torch, dist
        dev = torch.device("cuda", dp.device)
        h._device = dev
        stream_ordered = os.environ.get("KNPEMI_HALO_SYNC") is None
        try:
            h._ext = torch.cuda.ExternalStream(dp.lib.knpemi_dg_stream(dp.h), device=dev)
        except (RuntimeError, TypeError):
            h._ext, stream_ordered = None, False
        self._native = False
        if dist.get_backend() == "gloo":
            h.mode = "gloo host-staged"
        else:
            # first choice: the library's own RCCL communicator on the DG handle's stream (knpemi_dg_comm_*), voted on
            native = os.environ.get("KNPEMI_HALO_TORCH") is None and os.environ.get("KNPEMI_HALO_SYNC") is None
            if native:
                buf = C.create_string_buffer(128)
                ok = dp.lib.knpemi_comm_unique_id(buf, 128) == 0 if dist.get_rank() == 0 else True
                box = [buf.raw if ok else None]
                dist.broadcast_object_list(box, src=0)
                native = box[0] is not None
                if native:
                    rc = dp.lib.knpemi_dg_comm_init(dp.h, dist.get_rank(), dist.get_world_size(), box[0], 128)
                    flag = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=dev)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    native = bool(int(flag.item()))
            if native:
                self._native = True
                h.mode = "library RCCL (knpemi_dg_comm_sendrecv)"
            else:
                flag = torch.tensor([1 if stream_ordered else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                h.mode = "stream-ordered RCCL" if int(flag.item()) else "host-synchronised RCCL"
        h._stream_ordered = h.mode == "stream-ordered RCCL"
        self.mode = h.mode
        self._d = h._device_plan(self.plan, 5)

    def enable_solves(self):
        """`solve_emi` / `solve_knp` of this rank's DGProblem become solves of the GLOBAL systems
        (knpemi_dg_set_distributed; the reference runs its KSP solves under MPI, SURVEY §8 e): owned rows only, ghost
        refresh of every SpMV argument, all-reduced dot products, per-rank auxiliary-space AMG.  Transports: the
        `torch.distributed` ones of the field halo (stream-ordered RCCL, host-synchronised RCCL, gloo host-staged)."""
        if self._d is None and self.world > 1:
            raise RuntimeError("DGSlab.enable_solves: call attach() first")
        h, dp, torch = self._halo, self.dp, self._halo.torch
        KS, n = dp.K - 1, dp.n
        sh = dp.lib.knpemi_dg_solver_handle(dp.h)
        if not sh:
            L.check(-1)
        self._solver_handle = C.c_void_p(sh)
        knp = lambda g: np.concatenate([np.asarray(g, np.int64) + k * n for k in range(KS)])
        mapped = {key: dict(nb=pl["nb"], send=knp(pl["send"]), recv=knp(pl["recv"])) for key, pl in self.plan.items()}
        self._vec = {L.B_EMI: h._device_plan(self.plan, 1), L.B_KNP: h._device_plan(mapped, 1)}
        h._red = torch.zeros(8 + 64, dtype=torch.float64, device=h._device)
        self._hook_error = None

        def allreduce(ctx, m):
            try:
                h._allreduce(m)
                return 0
            except Exception as exc:       # noqa: BLE001 -- must not propagate through the C frame
                self._hook_error = exc
                return -1

        def halo(ctx, vec, which):
            try:
                d = self._vec[which]
                if d is not None:
                    L.check(dp.lib.knpemi_vec_gather(self._solver_handle, vec, d["send_idx"].data_ptr(), d["send_idx"].numel(),
                                                     d["send_buf"].data_ptr()))
                    native, self._halo._native = getattr(self._halo, "_native", False), False   # (field halo may be library RCCL)
                    try:
                        self._halo._transfer(d)
                    finally:
                        self._halo._native = native
                    L.check(dp.lib.knpemi_vec_scatter(self._solver_handle, vec, d["recv_idx"].data_ptr(), d["recv_idx"].numel(),
                                                      d["recv_buf"].data_ptr()))
                return 0
            except Exception as exc:       # noqa: BLE001
                self._hook_error = exc
                return -1
        own = np.ascontiguousarray(np.repeat(self.owned_cells.astype(np.uint8), dp.nv))
        self._cb = (L.ALLREDUCE_FN(allreduce), L.HALO_FN(halo))    # keep the callbacks alive
        L.check(dp.lib.knpemi_dg_set_distributed(dp.h, own.ctypes.data_as(L.c_u8_p), h._red.data_ptr(),
                                                 C.cast(self._cb[0], C.c_void_p), C.cast(self._cb[1], C.c_void_p), None))

    def exchange(self):
        """Refresh the ghost cells' concentrations and potential from their owners."""
        d, dp = self._d, self.dp
        if d is None:
            return
        L.check(dp.lib.knpemi_dg_halo_pack(dp.h, d["send_idx"].data_ptr(), d["send_idx"].numel(), d["send_buf"].data_ptr()))
        if self._native:
            i64 = C.POINTER(C.c_int64)
            L.check(dp.lib.knpemi_dg_comm_sendrecv(
                dp.h, d["send_buf"].data_ptr(), d["recv_buf"].data_ptr(), len(d["peer"]), L.iptr(d["peer"]),
                d["send_off"].ctypes.data_as(i64), d["send_cnt"].ctypes.data_as(i64),
                d["recv_off"].ctypes.data_as(i64), d["recv_cnt"].ctypes.data_as(i64)))
        else:
            self._halo._transfer(d)
        L.check(dp.lib.knpemi_dg_halo_unpack(dp.h, d["recv_idx"].data_ptr(), d["recv_idx"].numel(), d["recv_buf"].data_ptr()))

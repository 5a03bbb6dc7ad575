"""ctypes front end of oracle/knpemi_cpu.cpp (TEST INFRASTRUCTURE / timed CPU baseline).

`CpuPort(P, params, ions)` flattens an `OracleProblem` into the global-index arrays the C++ port works
on; the CSR patterns are taken from one oracle assembly.  Two cellular sub-domains at most in the
membrane tables is enough for the idealized runs (ECS + one cell type)."""
import ctypes as C
import os
import subprocess

import numpy as np

import knpemi_oracle as o

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "_build", "libknpemi_cpu.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", HERE])
        # keep the OpenMP team spinning between the parallel regions of a step: the regions are short and
        # separated by Python work, and waking sleeping threads costs more than the regions themselves
        os.environ.setdefault("OMP_WAIT_POLICY", "ACTIVE")
        os.environ.setdefault("GOMP_SPINCOUNT", "2000000")   # a few ms, then sleep (do not starve later serial legs)
        _lib = C.CDLL(so)
        _lib.cpu_ode_sweep.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class CpuPort:
    def __init__(self, P, params, ions, A_pattern, Ak_pattern):
        self.P, self.params, self.ions = P, params, ions
        self.gdim = P.gdim
        tags = P.tags
        self.nv = P.sub[0]["cells"].shape[1]
        self.ntot = P.Ntot
        self.x = np.ascontiguousarray(np.concatenate([P.sub[t]["x"] for t in tags]))
        self.cells = np.ascontiguousarray(np.concatenate([P.sub[t]["cells"] + P.off[t] for t in tags]).astype(np.int32))
        self.cell_sub = np.concatenate([np.full(len(P.sub[t]["cells"]), s, np.int32) for s, t in enumerate(tags)])
        self.vsub = np.concatenate([np.full(P.N[t], s, np.int32) for s, t in enumerate(tags)])
        F, psi = params["F"], params["psi"]
        z = np.array([i["z"] for i in ions])
        D = np.array([[i["D"][t] for i in ions] for t in tags])
        self.kap = np.ascontiguousarray(F * z * z * D * psi)
        self.sig = np.ascontiguousarray(F * z * D)
        self.D = np.ascontiguousarray(D)
        self.zpsiD = np.ascontiguousarray(z * psi * D)
        self.az2D = np.ascontiguousarray(D * z * z)
        self.z = np.ascontiguousarray(z)
        A = A_pattern.tocsr()
        A.sort_indices()
        Ak = Ak_pattern.tocsr()
        Ak.sort_indices()
        self.rp, self.ci = A.indptr.astype(np.int32), A.indices.astype(np.int32)
        self.krp, self.kci = Ak.indptr.astype(np.int32), Ak.indices.astype(np.int32)
        boff, _ = o.knp_block_offsets(P, 2)
        self.krow = np.concatenate([np.concatenate([boff[(t, k)] + np.arange(P.N[t]) for t in tags])
                                    for k in range(2)]).astype(np.int32)
        # membrane facets of all cells, global ids; Q dofs offset per cell
        fe, fi, fq, fs, q2e, q2i = [], [], [], [], [], []
        qoff = 0
        for s, t in enumerate(tags[1:], start=1):
            m = P.mem[t]
            fe.append(m["e"] + P.off[0]); fi.append(m["i"] + P.off[t]); fq.append(m["q"] + qoff)
            fs.append(np.full(len(m["e"]), s, np.int32))
            q2e.append(m["q2e"] + P.off[0]); q2i.append(m["q2i"] + P.off[t])
            qoff += P.NQ[t]
        self.NQ = qoff
        cat = lambda L: np.ascontiguousarray(np.concatenate(L).astype(np.int32))
        self.fe, self.fi, self.fq, self.fsub = cat(fe), cat(fi), cat(fq), cat(fs)
        self.q2e, self.q2i = cat(q2e), cat(q2i)
        self.nf = self.fe.shape[1]
        pts, wts = o.quadrature(P.facet_type, 6)
        phi, _ = o.tabulate(P.facet_type, pts)
        ref = 1.0 if self.nf == 2 else 0.5
        self.qw = np.ascontiguousarray(wts if self.nf == 4 else wts / ref * (1.0 if self.nf == 2 else 0.5))
        self.qN = np.ascontiguousarray(phi)
        self.qxi = np.ascontiguousarray(pts if self.nf == 4 else np.zeros((len(wts), 2)))
        self.A = np.zeros(A.nnz); self.Pm = np.zeros(A.nnz); self.b = np.zeros(self.ntot)
        self.Ak = np.zeros(Ak.nnz); self.bk = np.zeros(2 * self.ntot)

    def flat(self, per_tag):
        return np.ascontiguousarray(np.concatenate([per_tag[t] for t in self.P.tags]))

    def assemble_emi(self, c_all, phiM, Ich, splitting=True):
        c = [self.flat({t: c_all[t][k] for t in self.P.tags}) for k in range(3)]
        pm = self.flat_q(phiM)
        isum = np.ascontiguousarray(Ich.sum(axis=0))
        lib().cpu_assemble_emi(
            C.c_int(self.gdim), C.c_int(self.nv), C.c_int(len(self.cells)), _p(self.cells), _p(self.cell_sub), _p(self.x),
            _p(c[0]), _p(c[1]), _p(c[2]), _p(self.kap), _p(self.sig), _p(self.rp), _p(self.ci), C.c_int64(len(self.A)),
            C.c_int(self.ntot), _p(self.A), _p(self.Pm), _p(self.b), C.c_int(len(self.fe)), C.c_int(self.nf), _p(self.fe),
            _p(self.fi), _p(self.fq), _p(pm), _p(isum), C.c_double(self.params["C_phi"]), C.c_int(int(splitting)))
        return self.A, self.Pm, self.b

    def flat_q(self, per_tag):
        return np.ascontiguousarray(np.concatenate([per_tag[t] for t in self.P.tags[1:]]))

    def assemble_knp(self, c_all, phi, phiM, Ich, splitting=True):
        c = [self.flat({t: c_all[t][k] for t in self.P.tags}) for k in range(3)]
        ph = self.flat(phi)
        pm = self.flat_q(phiM)
        Ich = np.ascontiguousarray(Ich)
        lib().cpu_assemble_knp(
            C.c_int(self.gdim), C.c_int(self.nv), C.c_int(len(self.cells)), _p(self.cells), _p(self.cell_sub), _p(self.x),
            _p(c[0]), _p(c[1]), _p(c[2]), _p(ph), _p(self.D), _p(self.zpsiD), _p(self.az2D), _p(self.krow), _p(self.krp),
            _p(self.kci), C.c_int64(len(self.Ak)), C.c_int(self.ntot), _p(self.Ak), _p(self.bk), C.c_double(self.params["dt"]),
            C.c_int(len(self.fe)), C.c_int(self.nf), _p(self.fe), _p(self.fi), _p(self.fq), _p(self.fsub), _p(pm), _p(Ich),
            C.c_int(self.NQ), C.c_double(self.params["C_M"]), C.c_double(self.params["F"]), _p(self.z),
            C.c_int(len(self.qw)), _p(self.qw), _p(self.qN), _p(self.qxi), C.c_int(int(splitting)))
        return self.Ak, self.bk

    def ode_sweep(self, model_id, states, params, t0, dt, mask, stim_idx, stim_val):
        nrhs = C.c_int64()
        mask = np.ascontiguousarray(mask.astype(np.uint8))
        si = np.ascontiguousarray(np.asarray(stim_idx, np.int32))
        sv = np.ascontiguousarray(np.asarray(stim_val, np.float64))
        failed = lib().cpu_ode_sweep(C.c_int(model_id), C.c_int(states.shape[0]), C.c_int(states.shape[1]),
                                     C.c_int(params.shape[1]), _p(states), _p(params), C.c_double(t0), C.c_double(dt),
                                     C.c_double(1e-8), C.c_double(1e-10), _p(mask), C.c_int(len(si)), _p(si), _p(sv),
                                     C.byref(nrhs))
        return failed, nrhs.value

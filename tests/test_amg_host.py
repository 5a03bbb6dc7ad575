"""CPU tests of the PRODUCT's AMG set-up code (csrc/amg_host.h, host build in tests/native): the pieces the device solves'
hierarchies are made of -- aggregation, the aggregation that keeps strongly positively coupled unknowns apart, the split
of given aggregates along their strong couplings (DG auxiliary space), prolongators with and without the filter, the
Galerkin product and the block-wise dense inverse of the coarsest level (pdeSolver.py:24-35,99-110 delegates all of this
to hypre BoomerAMG)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(HERE, "native", "_build", "libamg_host.so")
    src = os.path.join(HERE, "native", "amg_host_check.cpp")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-shared", "-fPIC", "-o", so, src])
    L = C.CDLL(so)
    ip, dp, bp = C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_ubyte)
    L.amg_host_aggregate.argtypes = [C.c_int, ip, ip, dp, C.c_double, C.c_int, ip]
    L.amg_host_split.argtypes = [C.c_int, ip, ip, dp, C.c_double, bp, ip, C.c_int]
    L.amg_host_prolongator.argtypes = [C.c_int, ip, ip, dp, ip, C.c_int, C.c_double, C.c_double, dp]
    L.amg_host_galerkin.argtypes = [C.c_int, ip, ip, dp, ip, C.c_int, C.c_double, C.c_double, dp]
    L.amg_host_dense_inverse.argtypes = [C.c_int, ip, ip, dp, C.c_int, dp]
    L.amg_host_dense_inverse_blocks.argtypes = [C.c_int, ip, ip, dp, C.c_int, dp, ip]
    L.amg_host_renumber.argtypes = [C.c_int, ip, ip, dp, ip, C.c_int]
    L.amg_host_rho.argtypes = [C.c_int, ip, ip, dp]
    L.amg_host_rho.restype = C.c_double
    L.amg_host_outlier.argtypes = [C.c_int, ip, ip, dp, C.c_double, ip, ip]
    return L


def _csr(A):
    A = sp.csr_matrix(A)
    A.sort_indices()
    return (A.shape[0], np.ascontiguousarray(A.indptr, np.int32), np.ascontiguousarray(A.indices, np.int32),
            np.ascontiguousarray(A.data, np.float64))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _args(A):
    n, rp, ci, v = _csr(A)
    return n, rp, ci, v, (n, _p(rp, C.c_int), _p(ci, C.c_int), _p(v, C.c_double))


def _laplace_2d(nx, ny, ex=1.0, ey=1.0):
    """5-point operator with couplings ex along x, ey along y (Dirichlet ends: non-singular)."""
    Tx = sp.diags([-np.ones(nx - 1), 2 * np.ones(nx), -np.ones(nx - 1)], [-1, 0, 1])
    Ty = sp.diags([-np.ones(ny - 1), 2 * np.ones(ny), -np.ones(ny - 1)], [-1, 0, 1])
    return sp.csr_matrix(ex * sp.kron(sp.identity(ny), Tx) + ey * sp.kron(Ty, sp.identity(nx)))


def test_aggregation_follows_the_strong_direction(lib):
    """100 : 1 anisotropy: with theta = 0.08 only the couplings along y are strong and no aggregate spans two x-columns."""
    nx, ny = 12, 12
    n, rp, ci, v, a = _args(_laplace_2d(nx, ny, ex=0.01, ey=1.0))
    agg = np.zeros(n, np.int32)
    na = lib.amg_host_aggregate(*a, 0.08, 0, _p(agg, C.c_int))
    assert 0 < na < n and agg.min() == 0 and agg.max() == na - 1
    x = np.arange(n) % nx
    for g in range(na):
        assert len(set(x[agg == g])) == 1
    # isotropic: aggregates of a vertex and its four neighbours spread both ways
    n, rp, ci, v, a = _args(_laplace_2d(nx, ny))
    na_iso = lib.amg_host_aggregate(*a, 0.08, 0, _p(agg, C.c_int))
    assert na_iso < n / 3


def test_positively_coupled_unknowns_stay_in_different_aggregates(lib):
    """Two copies of a plane of vertices (the two ends of a layer of stretched Q1 cells in the split DG space): in-plane
    couplings -0.14 to the eight neighbours, +0.48 to the partner in the other copy, -0.09 -- just above the threshold -- to
    the partner's in-plane neighbours.  The plain greedy pass glues the copies together; aggregate_apart never puts a
    positive pair into one aggregate and gives most aggregates to one copy alone."""
    nx = ny = 9
    m = nx * ny
    idx = lambda i, j: j * nx + i
    A = sp.lil_matrix((2 * m, 2 * m))
    for c in range(2):
        for j in range(ny):
            for i in range(nx):
                r = c * m + idx(i, j)
                A[r, r] = 1.0
                A[r, (1 - c) * m + idx(i, j)] = 0.48
                for dj in (-1, 0, 1):
                    for di in (-1, 0, 1):
                        if (di or dj) and 0 <= i + di < nx and 0 <= j + dj < ny:
                            A[r, c * m + idx(i + di, j + dj)] = -0.14
                            A[r, (1 - c) * m + idx(i + di, j + dj)] = -0.09
    n, rp, ci, v, a = _args(A.tocsr())
    plain, apart = np.zeros(n, np.int32), np.zeros(n, np.int32)
    lib.amg_host_aggregate(*a, 0.08, 0, _p(plain, C.c_int))
    na = lib.amg_host_aggregate(*a, 0.08, 2, _p(apart, C.c_int))
    assert any(plain[i] == plain[i + m] for i in range(m))
    assert apart.min() == 0 and apart.max() == na - 1
    for i in range(m):
        assert apart[i] != apart[i + m]
    # what the rule promises is about direct positive partners (a leftover at the rim of an aggregate may still join the
    # other copy's aggregate next door through its weak negative couplings): the copies mostly get aggregates of their own
    pure = lambda ag: sum(1 for g in range(ag.max() + 1) if len({int(i >= m) for i in np.flatnonzero(ag == g)}) == 1)
    assert pure(plain) == 0 and pure(apart) >= na // 2, (pure(apart), na)
    # BOTH copies seed aggregates: a vertex becomes a root when the neighbours it would TAKE are free (the in-plane ones),
    # not all its strong neighbours -- with the earlier rule (every strong neighbour free) the vertices of one copy waited
    # for the other copy's in-plane neighbours, never became roots and joined the other copy's aggregates as leftovers
    # (25 aggregates, none of them in copy 0 alone, 9 mixed; now 18 = 9 + 9, none mixed)
    per_copy = [sum(1 for g in range(na) if {int(i >= m) for i in np.flatnonzero(apart == g)} == {c}) for c in (0, 1)]
    assert min(per_copy) >= na // 3 and sum(per_copy) == na, (per_copy, na)


def test_given_aggregates_split_along_their_strong_couplings(lib):
    """Four coincident dofs per vertex, tied pairwise by a strong coupling (-0.5) and across the pairs by a weak one
    (-0.005): every given aggregate splits into its two strongly connected halves; dofs that are not owned stay together."""
    nvtx = 30
    blk = np.array([[1.0, -0.5, -0.005, 0.0], [-0.5, 1.0, 0.0, -0.005], [-0.005, 0.0, 1.0, -0.5], [0.0, -0.005, -0.5, 1.0]])
    A = sp.block_diag([blk] * nvtx, format="lil")
    for k in range(nvtx - 1):                     # strong couplings between neighbouring vertices must not matter
        A[4 * k, 4 * k + 4] = A[4 * k + 4, 4 * k] = -0.4
    n, rp, ci, v, a = _args(A.tocsr())
    agg = np.repeat(np.arange(nvtx), 4).astype(np.int32)
    na = lib.amg_host_split(*a, 0.1, None, _p(agg, C.c_int), nvtx)
    assert na == 2 * nvtx
    for k in range(nvtx):
        g = agg[4 * k:4 * k + 4]
        assert g[0] == g[1] and g[2] == g[3] and g[0] != g[2]
    owned = np.ones(n, np.uint8)
    owned[:8] = 0                                 # the dofs of the first two vertices belong to ghost cells
    agg = np.repeat(np.arange(nvtx), 4).astype(np.int32)
    na = lib.amg_host_split(*a, 0.1, _p(owned, C.c_ubyte), _p(agg, C.c_int), nvtx)
    assert na == 2 * (nvtx - 2) + 2 and len(set(agg[:4])) == 1 and len(set(agg[4:8])) == 1


def test_prolongators_and_galerkin_product(lib):
    nx, ny = 10, 9
    A = _laplace_2d(nx, ny, ex=0.02, ey=1.0)
    n, rp, ci, v, a = _args(A)
    agg = np.zeros(n, np.int32)
    na = lib.amg_host_aggregate(*a, 0.08, 0, _p(agg, C.c_int))
    rho = lib.amg_host_rho(*a)
    lam = np.linalg.eigvalsh((sp.diags(1.0 / A.diagonal()) @ A).toarray() if False else
                             (np.diag(A.diagonal() ** -0.5) @ A.toarray() @ np.diag(A.diagonal() ** -0.5))).max()
    assert lam <= rho * 1.001 and rho < 1.3 * lam             # an upper estimate of rho(D^-1 A), not a wild one
    w = 4.0 / (3.0 * rho)
    T = np.zeros((n, na))
    T[np.arange(n), agg] = 1.0
    P0, P1, PF = np.zeros((n, na)), np.zeros((n, na)), np.zeros((n, na))
    lib.amg_host_prolongator(*a, _p(agg, C.c_int), na, 0.0, 0.0, _p(P0, C.c_double))
    lib.amg_host_prolongator(*a, _p(agg, C.c_int), na, w, 0.0, _p(P1, C.c_double))
    lib.amg_host_prolongator(*a, _p(agg, C.c_int), na, w, 0.02, _p(PF, C.c_double))
    Ad = A.toarray()
    assert np.array_equal(P0, T)
    assert np.allclose(P1, T - w * (Ad / A.diagonal()[:, None]) @ T, atol=1e-14)
    # the filter drops the entries below 0.02 sqrt(a_ii a_jj) (here: the couplings along x) and lumps them into the diagonal
    d = A.diagonal()
    keep = np.abs(Ad) >= 0.02 * np.sqrt(np.outer(d, d))
    AF = np.where(keep, Ad, 0.0)
    AF[np.arange(n), np.arange(n)] = d + np.where(keep, 0.0, Ad).sum(1)
    assert np.allclose(PF, T - w * (AF / AF.diagonal()[:, None]) @ T, atol=1e-14)
    assert np.count_nonzero(PF) < np.count_nonzero(P1)
    assert np.allclose(PF.sum(1), 1.0 - w * AF.sum(1) / AF.diagonal())      # row sums: constants are kept where A 1 = 0
    Ac = np.zeros((na, na))
    lib.amg_host_galerkin(*a, _p(agg, C.c_int), na, w, 0.0, _p(Ac, C.c_double))
    assert np.allclose(Ac, P1.T @ Ad @ P1, rtol=1e-12, atol=1e-14)


def test_dense_inverse_of_the_coarsest_level(lib):
    rng = np.random.default_rng(5)
    # two independent blocks (the K - 1 ion systems of the concentration matrix), interleaved numbering
    m = 37
    B1 = _laplace_2d(m, 1).toarray() + np.diag(rng.random(m))
    B2 = _laplace_2d(m, 1, ex=3.0).toarray() + 0.1 * np.triu(rng.random((m, m)), 1) * (np.abs(np.subtract.outer(np.arange(m), np.arange(m))) == 1)
    A = np.zeros((2 * m, 2 * m))
    A[0::2, 0::2], A[1::2, 1::2] = B1, B2
    n, rp, ci, v, a = _args(sp.csr_matrix(A))
    inv = np.zeros((n, n))
    assert lib.amg_host_dense_inverse(*a, 0, _p(inv, C.c_double)) == 1
    assert np.allclose(inv @ A, np.eye(n), atol=1e-10)
    assert np.all(inv[0::2, 1::2] == 0.0) and np.all(inv[1::2, 0::2] == 0.0)      # inverted block by block
    # interleaved components are not contiguous ranges: the block-wise storage for the device falls back to one block
    invb, nb = np.zeros((n, n)), C.c_int()
    assert lib.amg_host_dense_inverse_blocks(*a, 0, _p(invb, C.c_double), C.byref(nb)) == n * n and nb.value == 1
    assert np.allclose(invb, inv, rtol=1e-12, atol=1e-14)
    # ... the same two systems one after the other (what renumber_by_component arranges on every coarse level): two blocks,
    # half the values, the same inverse
    A2 = np.zeros((2 * m, 2 * m))
    A2[:m, :m], A2[m:, m:] = B1, B2
    n2, rp2, ci2, v2, a2 = _args(sp.csr_matrix(A2))
    invb = np.zeros((n2, n2))
    assert lib.amg_host_dense_inverse_blocks(*a2, 0, _p(invb, C.c_double), C.byref(nb)) == 2 * m * m and nb.value == 2
    assert np.allclose(invb @ A2, np.eye(n2), atol=1e-10) and np.all(invb[:m, m:] == 0.0)
    # one connected non-singular system: the plain elimination
    A1 = _laplace_2d(9, 7)
    n, rp, ci, v, a = _args(A1)
    inv = np.zeros((n, n))
    assert lib.amg_host_dense_inverse(*a, 0, _p(inv, C.c_double)) == 1
    assert np.allclose(inv @ A1.toarray(), np.eye(n), atol=1e-10)
    invb, nb = np.zeros((n, n)), C.c_int()
    assert lib.amg_host_dense_inverse_blocks(*a, 0, _p(invb, C.c_double), C.byref(nb)) == n * n and nb.value == 1
    assert np.array_equal(invb, inv)
    # singular (constant null space, the potential system): the inverse of A + s 1 1^T / n inverts A on the zero-mean vectors
    Tx = sp.diags([-np.ones(8), np.r_[1.0, 2 * np.ones(7), 1.0], -np.ones(8)], [-1, 0, 1])
    AN = sp.csr_matrix(sp.kron(sp.identity(6), Tx) + sp.kron(sp.diags([-np.ones(5), np.r_[1.0, 2 * np.ones(4), 1.0], -np.ones(5)], [-1, 0, 1]), sp.identity(9)))
    n, rp, ci, v, a = _args(AN)
    inv = np.zeros((n, n))
    assert lib.amg_host_dense_inverse(*a, 1, _p(inv, C.c_double)) == 1
    b = rng.standard_normal(n)
    b -= b.mean()
    x = inv @ b
    assert np.linalg.norm(AN @ x - b) < 1e-9 * np.linalg.norm(b) and abs(x.mean()) < 1e-9 * np.abs(x).max()
    # a singular matrix without the flag is refused
    Z = sp.csr_matrix(np.array([[1.0, 1.0], [1.0, 1.0]]))
    n, rp, ci, v, a = _args(Z)
    assert lib.amg_host_dense_inverse(*a, 0, _p(np.zeros((2, 2)), C.c_double)) == 0


def test_an_outlier_entry_is_found_and_named(lib):
    """One finite but absurd entry (1e300, or merely 1e6 times its row's scale) passed the round-3 set-up, which only refused
    non-finite values: it collapses the damping 4 / (3 rho) of the level and changes every coarser operator.  The set-up
    now refuses |a_ij| > factor * sqrt(|a_ii a_jj|) and names the entry (kn_amg_host::find_outlier)."""
    A = _laplace_2d(8, 8).tolil()
    n, rp, ci, v, a = _args(A.tocsr())
    i, j = C.c_int(-1), C.c_int(-1)
    assert lib.amg_host_outlier(*a, 1e3, C.byref(i), C.byref(j)) == 0
    for bad in (1e300, -4e6):
        B = A.copy()
        B[19, 20] = bad
        n, rp, ci, v, a = _args(B.tocsr())
        assert lib.amg_host_outlier(*a, 1e3, C.byref(i), C.byref(j)) == 1 and (i.value, j.value) == (19, 20)
    # the positive couplings of stretched cells (0.48 sqrt(a_ii a_jj)) and a drift-skewed row are nowhere near the bound
    B = A.copy()
    B[5, 6], B[6, 5] = -3.0, 1.0
    n, rp, ci, v, a = _args(B.tocsr())
    assert lib.amg_host_outlier(*a, 1e3, C.byref(i), C.byref(j)) == 0


def test_aggregates_are_renumbered_component_by_component(lib):
    """Two independent systems with interleaved unknowns: after renumber_by_component the aggregates of the first system
    come first, in their old order, so the coarse operator is block diagonal with contiguous blocks."""
    T = _laplace_2d(6, 5)
    A = sp.lil_matrix((60, 60))
    A[0::2, 0::2], A[1::2, 1::2] = T, 2.0 * T
    n, rp, ci, v, a = _args(A.tocsr())
    agg = np.zeros(n, np.int32)
    na = lib.amg_host_aggregate(*a, 0.08, 0, _p(agg, C.c_int))
    old = agg.copy()
    lib.amg_host_renumber(*a, _p(agg, C.c_int), na)
    sys_of = lambda ag: np.array([np.flatnonzero(ag == g)[0] % 2 for g in range(na)])
    s_new = sys_of(agg)
    assert sorted(set(agg)) == list(range(na)) and np.all(np.diff(s_new) >= 0) and 0 < s_new.sum() < na
    for sysid in (0, 1):          # same partition, same relative order inside a system
        members_old = [tuple(np.flatnonzero(old == g)) for g in range(na) if np.flatnonzero(old == g)[0] % 2 == sysid]
        members_new = [tuple(np.flatnonzero(agg == g)) for g in range(na) if s_new[g] == sysid]
        assert members_old == members_new


def test_row_parallel_set_up_does_not_depend_on_the_number_of_threads(tmp_path):
    """The sparse products, the prolongator and the power iteration of the set-up run on the host's threads above 20 000
    rows (csrc/amg_host.h: build_rows, sum_rows); a hierarchy must be reproducible from one machine to the next, so the
    results are the same BITS with one thread and with several (tools/probes/amg_host_bench.cpp runs both and prints
    digests of P and of the Galerkin operator)."""
    exe = str(tmp_path / "amg_host_bench")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(os.path.dirname(HERE), "knp-emi-fenics-x_amd", "csrc"),
                           os.path.join(os.path.dirname(HERE), "tools", "probes", "amg_host_bench.cpp"), "-o", exe])
    env = dict(os.environ, KNPEMI_AMG_THREADS="4")
    out = subprocess.run([exe, "30"], env=env, capture_output=True, text=True, check=True).stdout.strip().splitlines()
    assert len(out) == 2, out
    tails = [line.split("| rho")[1] for line in out]          # rho to 17 digits and the two digests
    threads = sorted(int(line.split("threads")[1].split("|")[0]) for line in out)
    assert threads == [1, 4] and tails[0] == tails[1], out

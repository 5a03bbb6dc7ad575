"""Stream-ordered halo exchange rehearsal on ONE GPU with real RCCL: a world-size-1 nccl group sends the packed
halo buffer to itself.  Verifies that pack (library stream) -> RCCL send/recv -> unpack (library stream) is correctly
ordered WITHOUT host synchronisation when torch's current stream is the library's stream (ExternalStream)."""
import contextlib, io, os, sys, time
import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knp-emi-fenics-x_amd"))
sys.path.insert(0, os.path.join(ROOT, "examples", "idealized_geometries"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from knpemi import _lib as L
from setup_problem import Setup
with contextlib.redirect_stdout(io.StringIO()):
    s = Setup("tet", 0)
dp = s.a_emi.dp
lib = dp.lib
dev = torch.device("cuda", 0)
nv = int(dp.n_vert.sum())
rng = np.random.default_rng(0)
n = 3000
send_idx = torch.from_numpy(rng.choice(nv, n, replace=False).astype(np.int32)).to(dev)
recv_idx = torch.from_numpy(rng.permutation(nv)[:n].astype(np.int32)).to(dev)
w = 4
send_buf = torch.empty(n * w, dtype=torch.float64, device=dev)
recv_buf = torch.empty(n * w, dtype=torch.float64, device=dev)
ext = torch.cuda.ExternalStream(lib.knpemi_stream(dp.h), device=dev)


def exchange():
    L.check(lib.knpemi_halo_pack(dp.h, 0, send_idx.data_ptr(), n, send_buf.data_ptr()))
    with torch.cuda.stream(ext):
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, send_buf, 0), dist.P2POp(dist.irecv, recv_buf, 0)]):
            req.wait()
    L.check(lib.knpemi_halo_unpack(dp.h, 0, recv_idx.data_ptr(), n, recv_buf.data_ptr()))


phi0 = s.phi[0]
n0 = phi0.x.array.shape[0]
ok = True
t0 = time.perf_counter()
for it in range(300):
    phi0.x.array[:] = float(it) + np.arange(n0) * 1e-6
    dp.push(L.F_PHI, 0, 0, phi0)        # async upload on the library stream
    exchange()
    if it % 50 == 49 or it < 3:
        dp.sync()
        got = recv_buf.cpu().numpy().reshape(n, w)[:, 3]
        si = send_idx.cpu().numpy()
        exp = np.where(si < n0, float(it) + si * 1e-6, got)
        good = np.array_equal(got[si < n0], exp[si < n0])
        ok &= bool(good)
        print("iteration", it, "phi column of the received halo correct:", good, flush=True)
dp.sync()
el = time.perf_counter() - t0
print(f"{300 / el:.0f} exchanges/s including upload", "ALL OK" if ok else "MISMATCH")


def exchange_sync():
    L.check(lib.knpemi_halo_pack(dp.h, 0, send_idx.data_ptr(), n, send_buf.data_ptr()))
    dp.sync()
    for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, send_buf, 0), dist.P2POp(dist.irecv, recv_buf, 0)]):
        req.wait()
    torch.cuda.current_stream().synchronize()
    L.check(lib.knpemi_halo_unpack(dp.h, 0, recv_idx.data_ptr(), n, recv_buf.data_ptr()))


for name, fn in (("stream-ordered", exchange), ("host-synchronised", exchange_sync)):
    for _ in range(20):
        fn()
    dp.sync()
    t0 = time.perf_counter()
    for _ in range(500):
        fn()
    dp.sync()
    print(f"{name}: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per exchange", flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)

"""PCIe-inclusive rate: the same chain fed from host memory (SoA planes; pageable NumPy arrays and pinned torch
tensors), batch 8 x 3 contexts, rolling submission, calling the C ABI directly (no per-call packing in Python).
Reported in DESIGN.md 5; never the bench's `value`."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from pointcloud_obstacle_processing_amd import capi, synth
p = synth.params_of("C3")
B, NC, N = 8, 3, 1_000_000
src = [np.ascontiguousarray(synth.make_cloud("C3", seed=3000 + i).T.reshape(-1)) for i in range(4)]   # SoA planes
for kind in ("pageable", "pinned"):
    if kind == "pinned":
        bufs = [torch.from_numpy(a).pin_memory() for a in src]
        addrs = [b.data_ptr() for b in bufs]
    else:
        bufs = src
        addrs = [a.ctypes.data for a in src]
    ctxs = []
    for ci in range(NC):
        c = capi.Context(N, max_frames=B); c.set_params(publish_clouds=0, **p)
        ctxs.append(c)
    ptrs = (C.c_void_p * B)(*[C.c_void_p(addrs[i % len(addrs)]) for i in range(B)])
    cnt = (C.c_size_t * B)(*[N] * B)
    res = (capi.Result * B)()
    def submit(c):
        c._chk(c.lib.pcop_submit_batch(c.h, ptrs, capi.LAYOUT_SOA, cnt, B, 1, capi.MEM_HOST))
    def wait(c):
        c._chk(c.lib.pcop_wait_batch(c.h, res, B))
    for c in ctxs:
        submit(c); wait(c)
    steps = 10
    for c in ctxs: submit(c)
    t0 = time.perf_counter()
    for _ in range(steps):
        for c in ctxs:
            wait(c); submit(c)
    dt = time.perf_counter() - t0
    for c in ctxs: wait(c)
    fps = steps * B * NC / dt
    print("host input, %s (12 MB per frame): %.0f frames/s = %.1f GB/s over PCIe" % (kind, fps, fps * 12e6 / 1e9), flush=True)
    for c in ctxs: c.close()

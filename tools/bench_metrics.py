"""Micro-benchmark (GPU box): device validation metrics vs the numpy oracle (port of the reference's host loops), LIDC shape."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import metrics as om
from tests.helpers import metrics_case
from phiseg_code_amd import runtime as rt
L = rt.lib()
I, N, M, X, C = 16, 100, 4, 128, 2
sm0, gt0 = metrics_case(31, N, M, X, X, C, "plain")
sm = torch.as_tensor(np.stack([sm0] * I)).cuda(); gt = torch.as_tensor(np.stack([gt0] * I)).cuda(); sr = gt[:, 0].contiguous()
wsb = int(L.validation_metrics_ws_bytes(I, N, M, X * X, C)); ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
out = torch.empty(I, 10, device="cuda"); st = torch.cuda.current_stream().cuda_stream
fn = lambda: L.validation_metrics(sm.data_ptr(), gt.data_ptr(), sr.data_ptr(), ws.data_ptr(), wsb, I, N, M, X * X, C, 1, out.data_ptr(), st)
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
t0 = time.time(); o = om.validation_metrics(sm0, gt0, gt0[0], C); cpu = time.time() - t0
print("device: %.3f ms per %d images (%.1f us/image, %d samples x %d annotators, %dx%d, %.1f GB/s of soft-max read); "
      "numpy oracle (1 host core): %.1f ms/image -> %.0fx" % (ms, I, 1e3 * ms / I, N, M, X, X, I * N * X * X * C * 4 / ms / 1e6, 1e3 * cpu, 1e3 * cpu / (ms / I)))
print("device", out[0, :4].tolist(), "oracle", o)

#!/usr/bin/env python3
"""dev probe (round 6): bytes per clock and CU of the global -> LDS path by where the lines come from (one shared L2-resident slab,
per-block slabs inside the 32 MB of L2, inside the 256 MB Infinity Cache, from HBM), register-staged against LDS-DMA, by the number of CUs
pulling.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/staging_probe.hip -o tools/probes/libstaging_probe.so"""
import ctypes
import os

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libstaging_probe.so"))
lib.probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
src = torch.randn(512 << 20 >> 2, device="cuda")          # 512 MB
out = torch.zeros(4096, device="cuda")
S = torch.cuda.current_stream().cuda_stream
GHZ = 2.4


def run(span, stride, slab, reps, mode, blocks, threads):
    def go():
        rc = lib.probe_launch(src.data_ptr(), out.data_ptr(), span, stride, slab, reps, mode, blocks, threads, S)
        assert rc == 0, rc
    go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    tot = float(blocks) * slab * reps
    return tot / ms / 1e9, ms          # TB/s


print("slab 64 KB per block and repetition; B/clk/CU at a nominal 2.4 GHz (the chip clocks lower under load: read the TB/s)")
for name, span, stride in [("one shared 64 KB slab (L2 hit, every CU the same lines)", 64 << 10, 0), ("per-block slabs within 16 MB (L2)", 16 << 20, 64 << 10),
                           ("per-block slabs within 128 MB (Infinity Cache)", 128 << 20, 64 << 10), ("per-block slabs within 512 MB (HBM)", 512 << 20, 1 << 20)]:
    for mode, mname in ((0, "registers"), (1, "LDS-DMA")):
        for blocks, threads in ((64, 256), (128, 256), (256, 256), (512, 256), (256, 512)):
            reps = 200
            tbs, ms = run(span, stride, 64 << 10, reps, mode, blocks, threads)
            cus = min(blocks, 256)
            print("%-58s %-9s %3d blocks x %3d thr: %6.2f TB/s  %5.1f B/clk/CU  (%.3f ms)" % (name, mname, blocks, threads, tbs, tbs * 1e12 / (cus * GHZ * 1e9), ms), flush=True)

"""Worker of test_distributed_gpu.test_native_rccl_allreduce_two_devices: one rank per GPU, the library's own RCCL communicator
(phx_comm_* bootstrapped through DistContext._native) sums a seeded fp32 buffer in buckets on a HIP stream of its own."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, n = sys.argv[1], int(sys.argv[2])
    from phiseg_code_amd import distributed
    from phiseg_code_amd import runtime as rt
    ctx = distributed.DistContext()
    assert ctx.world >= 2 and torch.cuda.device_count() >= ctx.world
    native = ctx._native()
    assert native, "the native RCCL communicator did not come up"
    L, comm = native
    st = ctypes.c_void_p()
    L.stream_create(ctypes.byref(st))
    x = torch.from_numpy(np.random.default_rng(100 + ctx.rank).standard_normal(n).astype(np.float32)).cuda()
    torch.cuda.synchronize()
    for _ in range(2):                                   # twice: the communicator is reusable, sums compose
        L.comm_allreduce_sum_f32(comm, x.data_ptr(), x.numel(), 1 << 16, st)
    L.stream_sync(st)
    np.save(os.path.join(out_dir, "rank%d.npy" % ctx.rank), x.cpu().numpy())
    ctx.barrier()
    ctx.shutdown()


if __name__ == "__main__":
    main()

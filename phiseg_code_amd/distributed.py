"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl") / gloo on CPU.

The reference is single-process / single-device (phiseg_model.py:151-157); data parallelism is new design
(SURVEY.md section 8(e)): the global batch is sharded over ranks, every rank holds a full replica of the flat
parameter arena, and ONE all-reduce (sum) of the flat fp32 gradient arena per step is the only exchange.  The
loss kernels already scale by 1/(B_local * world), so the sum IS the global-batch mean gradient; the Philox
noise is keyed by the global sample index (rank * B_local + b), so results do not depend on the sharding.
Batch-norm statistics stay per replica (standard DP semantics); group / instance norm are exactly
sharding-invariant."""
import os

import torch
import torch.distributed as dist


class DistContext:
    def __init__(self, backend=None, force=False):
        """force=True initialises the process group even for a single rank (exercises the RCCL path on one GPU)."""
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.cuda = torch.cuda.is_available()
        if self.cuda:
            torch.cuda.set_device(self.local_rank % torch.cuda.device_count())
        self.active = self.world > 1 or force
        if self.active and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            # PHX_DIST_BACKEND=gloo: several ranks on ONE GPU (RCCL refuses duplicate devices) -- protocol tests only
            backend = backend or os.environ.get("PHX_DIST_BACKEND") or ("nccl" if self.cuda else "gloo")
            kw = {}
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, **kw)

    def allreduce_sum(self, tensor, plan=None, bucket_elems=8 << 20):
        """Sum `tensor` (the flat gradient arena) over ranks, in ~32 MB buckets so RCCL pipelines them over
        the xGMI links.  `plan`: the engine plan whose stream produced the gradients and will consume the sums.
        On the GPU nothing blocks the host: the collectives are enqueued against the plan's own HIP stream (wrapped
        as a torch ExternalStream), so RCCL waits for the backward graph and the optimizer graph waits for RCCL."""
        if not self.active:
            return
        flat = tensor.view(-1)
        if self.cuda and plan is not None:
            ext = torch.cuda.ExternalStream(int(plan.stream_handle()))
            with torch.cuda.stream(ext):
                works = [dist.all_reduce(flat[i:i + bucket_elems], op=dist.ReduceOp.SUM, async_op=True)
                         for i in range(0, flat.numel(), bucket_elems)]
                for w in works:
                    w.wait()                  # stream-side wait: the plan's stream waits for RCCL
            return
        if plan is not None:
            plan.sync()
        works = [dist.all_reduce(flat[i:i + bucket_elems], op=dist.ReduceOp.SUM, async_op=True)
                 for i in range(0, flat.numel(), bucket_elems)]
        for w in works:
            w.wait()
        if self.cuda:
            torch.cuda.current_stream().synchronize()

    def barrier(self):
        if self.active:
            dist.barrier()

    def max_float(self, v):
        if not self.active:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device="cuda" if self.cuda else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_float(self, v):
        if not self.active:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device="cuda" if self.cuda else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def broadcast_(self, tensor, src=0):
        if self.active:
            dist.broadcast(tensor, src=src)

    def shutdown(self):
        if self.active and dist.is_initialized():
            dist.destroy_process_group()

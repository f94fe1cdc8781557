"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl") / gloo on CPU.

The reference is single-process / single-device (phiseg_model.py:151-157); data parallelism is new design
(SURVEY.md section 8(e)): the global batch is sharded over ranks, every rank holds a full replica of the flat
parameter arena, and ONE all-reduce (sum) of the flat fp32 gradient arena per step is the only exchange.  The
loss kernels already scale by 1/(B_local * world), so the sum IS the global-batch mean gradient; the Philox
noise is keyed by the global sample index (rank * B_local + b), so results do not depend on the sharding.
Batch-norm statistics stay per replica (standard DP semantics); group / instance norm are exactly
sharding-invariant."""
import ctypes
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
        self._comm = None
        if self.active and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            # PHX_DIST_BACKEND=gloo: several ranks on ONE GPU (RCCL refuses duplicate devices) -- protocol tests only
            backend = backend or os.environ.get("PHX_DIST_BACKEND") or ("nccl" if self.cuda else "gloo")
            kw = {}
            if backend == "nccl":
                kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, **kw)

    def _native(self):
        """The library's own RCCL communicator (phx_comm_*): rank 0 creates the rendezvous id, torch.distributed -- already up
        for process bootstrap -- carries it to the other ranks.  Used on the GPU with the nccl backend (PHX_COMM=torch keeps
        torch.distributed's collectives; gloo = several ranks on one GPU, where RCCL refuses duplicate devices)."""
        if self._comm is None:
            self._comm = False
            if (self.active and self.cuda and dist.get_backend() == "nccl"
                    and os.environ.get("PHX_COMM", "native") == "native"):
                from phiseg_code_amd import runtime as rt
                L = rt.lib()
                # Every step that can fail on ONE rank is followed by a MIN-all-reduce of an ok flag BEFORE the next collective
                # step, so a failing rank never leaves the others blocked inside a broadcast or inside ncclCommInitRank (which is
                # itself collective): (1) the RCCL entry points load on every rank, (2) rank 0 has a rendezvous id.
                def all_ok(ok):
                    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    return int(flag.item()) == 1
                ok, err = 1, ""
                try:
                    L.comm_load_api()
                except rt.PhxError as e:
                    ok, err = 0, str(e)
                idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
                if ok and self.rank == 0:
                    try:
                        buf = ctypes.create_string_buffer(128)
                        L.comm_unique_id(buf)
                        idt.copy_(torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8))
                    except rt.PhxError as e:
                        ok, err = 0, str(e)
                comm = ctypes.c_void_p()
                if all_ok(ok):
                    dist.broadcast(idt, src=0)
                    raw = bytes(idt.cpu().numpy().tobytes())
                    try:
                        L.comm_init(ctypes.byref(comm), self.world, self.rank, raw)
                    except rt.PhxError as e:       # (e.g. an RCCL build the dlopen'd entry points do not match)
                        ok, err = 0, str(e)
                else:
                    ok, err = 0, err or "RCCL unavailable on another rank"
                # every rank must take the same path: if ANY rank failed, all fall back to torch.distributed's RCCL collectives
                # (same exchange, issued through torch instead of phx_comm_*) -- loudly, it is still the HIP extension that computes
                flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 1:
                    self._comm = (L, comm)
                else:
                    import sys
                    if ok:
                        L.comm_destroy(comm)
                    print("[phiseg_code_amd.distributed] rank %d: phx_comm_init unavailable (%s); gradient exchange through "
                          "torch.distributed (RCCL) instead" % (self.rank, err or "failed on another rank"), file=sys.stderr)
        return self._comm

    def comm_path(self):
        """Which transport `allreduce_sum` uses for the gradient arena of a GPU plan: "phx_comm_rccl" (the library's own RCCL
        communicator on the plan's stream), "torch_rccl" (torch.distributed's RCCL collectives: the loud fallback of _native()),
        "torch_gloo_host_staged" (several ranks on one GPU: protocol tests) or "none" (single process)."""
        if not self.active:
            return "none"
        if dist.get_backend() == "gloo":
            return "torch_gloo_host_staged" if self.cuda else "torch_gloo"
        return "phx_comm_rccl" if (self.cuda and self._native()) else "torch_rccl"

    def allreduce_sum(self, tensor, plan=None, bucket_elems=8 << 20):
        """Sum `tensor` (the live part of the flat gradient arena) over ranks, in ~32 MB buckets so RCCL pipelines them over
        the xGMI links.  `plan`: the engine plan whose stream produced the gradients and will consume the sums.
        On the GPU nothing blocks the host: the collective is enqueued ON the plan's own HIP stream (phx_comm_allreduce_sum_f32),
        so RCCL runs after the backward graph and the optimizer graph after RCCL."""
        if not self.active:
            return
        flat = tensor.view(-1)
        native = self._native() if (self.cuda and plan is not None and flat.dtype == torch.float32) else False
        if native:
            L, comm = native
            L.comm_allreduce_sum_f32(comm, flat.data_ptr(), flat.numel(), bucket_elems, plan.stream_handle())
            return
        if dist.get_backend() == "gloo" and flat.is_cuda:
            # protocol tests (several ranks on one GPU): stage through the host
            if plan is not None:
                plan.sync()
            torch.cuda.synchronize()
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
            torch.cuda.synchronize()
            return
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

    def broadcast_flags(self, flags, src=0):
        """-> rank `src`'s list of small integers on every rank (decisions that must be rank-invariant)."""
        if not self.active:
            return list(flags)
        t = torch.tensor([int(f) for f in flags], dtype=torch.int32, device=self._scalar_device())
        dist.broadcast(t, src=src)
        return [int(v) for v in t.cpu().tolist()]

    def barrier(self):
        if self.active:
            dist.barrier()

    def max_float(self, v):
        if not self.active:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device=self._scalar_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_float(self, v):
        if not self.active:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device=self._scalar_device())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def gather_floats(self, v):
        """-> [v of rank 0, v of rank 1, ...] on every rank."""
        if not self.active:
            return [float(v)]
        t = torch.zeros(self.world, dtype=torch.float64, device=self._scalar_device())
        t[self.rank] = float(v)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(a) for a in t.cpu().tolist()]

    def _scalar_device(self):
        return "cuda" if (self.cuda and dist.get_backend() == "nccl") else "cpu"

    def broadcast_(self, tensor, src=0):
        if not self.active:
            return
        if dist.get_backend() == "gloo" and tensor.is_cuda:
            host = tensor.cpu()
            dist.broadcast(host, src=src)
            tensor.copy_(host)
            torch.cuda.synchronize()
            return
        dist.broadcast(tensor, src=src)

    def shutdown(self):
        if self._comm:
            L, comm = self._comm
            L.comm_destroy(comm)
            self._comm = None
        if self.active and dist.is_initialized():
            dist.destroy_process_group()

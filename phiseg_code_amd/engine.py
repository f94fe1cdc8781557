"""Lowering of a symbolic PHiSeg graph to a fixed HIP launch list + hipGraph replay.

This is the run-time half of the TF1 replacement: where the reference calls
``sess.run([train_step, loss_tot], feed_dict)`` (phiseg/phiseg_model.py:194) this module

* keeps all variables in flat fp32 device arenas (parameters, gradients, Adam m / v) so the optimiser and
  the data-parallel gradient all-reduce are single flat operations (``ParamStore``);
* compiles (fetches, loss) for one batch size / training flag into a list of libphx launches: forward of
  the live graph, reverse-mode backward (Appendix C of SURVEY.md), TF1 Adam (``Plan``);
* captures the list into a hipGraph and replays it per step (device-side step counter / learning rate, so a
  replay needs no host-side argument patching).

torch is used for device memory and, in ``distributed.py``, for torch.distributed -- plumbing only; every
arithmetic operation is a libphx kernel and a missing library is a hard error.
"""
import ctypes
import os

import numpy as np
import torch

from phiseg_code_amd import graph as G
from phiseg_code_amd import runtime as rt
from phiseg_code_amd.tfwrapper import normalisation as tfnorm

F32, BF16, U8 = rt.F32, rt.BF16, 2
_TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16, U8: torch.uint8}
_NP_DT = {F32: np.float32, U8: np.uint8}
_ESIZE = {F32: 4, BF16: 2, U8: 1}
# two lanes: likelihood chains of levels <= this go to the prior's lane; the coarsest chain stays on lane 0, in front of the top-down
# path that starts with it: lane 0 then enters the likelihood without waiting for lane 1 (with all five chains there it started only
# when the LAST of them was done, whatever the order: 12.38 vs 11.95 ms)
_LIK_SIDE_MAXLVL = 3
_WGRAD_DEFER_BLOCKS = 96      # pixel-tile split target of a deferred layer (measured 16 .. 192: fewer slices are long tail blocks)
_NREP = 4                     # accumulator replicas of the norm backward reduction (re-measured with the LDS-shared prologues; 8 before)
_NREP_MINP = 4096
_STAMPS = os.environ.get("PHX_STAMPS", "0") == "1"           # dev: per-operator device time stamps (tools/lane_timeline.py)
_DETERMINISTIC = os.environ.get("PHX_DETERMINISTIC", "0") not in ("0", "")   # fixed summation orders everywhere (libphx reads the same variable)
_BN_SMALL = 1024              # one-launch batch norm up to this many pixels
_BN_SMALL_F32 = os.environ.get("PHX_BN_SMALL_F32", "1") != "0"      # dev A/B (tools/convergence_study.py): fp32 pre-normalisation tensor of those layers


def _fgn_mode():
    # conv + bias + group norm + activation in one launch on maps <= 16 x 16 (phx_conv3x3_mfma_bf16_fgn).  1: group norm (16-channel
    # groups); 2: instance norm too (per-channel statistics: every lane adds to the LDS table -- measured 11.24 vs 11.15 ms, so not
    # by default); 0: off.  Group norm, phiseg_7_5 B = 64: 11.28 vs 11.28 - 11.30 ms with 57 launches fewer.
    return int(os.environ.get("PHX_FGN", "1"))


def _dual_enabled():
    return os.environ.get("PHX_DUAL", "1") == "1"      # concat -> conv3x3 edges without the concatenated tensor (A/B hook; read when a plan is built)


def _noop():
    pass


def _device():
    if not torch.cuda.is_available():
        raise rt.PhxError("no GPU visible: the PHiSeg engine has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def live_variables(loss):
    """Names of the variables the scalar `loss` depends on (TF: the variables optimizer.minimize gets a gradient for)."""
    seen, live, stack = set(), set(), [loss.op]
    while stack:
        op = stack.pop()
        if op in seen:
            continue
        seen.add(op)
        if op.type == "l2_weights":            # weight decay reaches every member of the collection (never-consumed branches too)
            live.update(v.name for v in op.attrs["vars"])
        if op.type == "conv_unit":
            a = op.attrs
            for v in (a["W"], a["b"]):
                if v is not None:
                    live.add(v.name)
            for v in (a.get("norm_vars") or {}).values():
                live.add(v.name)
        if op.type == "norm_act":
            for v in (op.attrs.get("norm_vars") or {}).values():
                live.add(v.name)
        stack.extend(i.op for i in op.inputs)
    return live


def device_sync():
    """Order work enqueued through torch (parameter loads, lr / step updates, broadcasts) before plan replays, which run on
    the plans' own non-blocking HIP streams."""
    torch.cuda.synchronize()


class Buf:
    """A device buffer: torch owns the memory, libphx sees the raw pointer."""

    def __init__(self, shape, dt, zero=False, like=None):
        self.shape = tuple(int(s) for s in shape)
        self.dt = dt
        n = int(np.prod(self.shape)) if self.shape else 1
        self.n = n
        if like is not None:
            self.t = like
        else:
            self.t = (torch.zeros if zero else torch.empty)(max(n, 1), dtype=_TORCH_DT[dt], device=_device())
        self.ptr = self.t.data_ptr()
        self.shift = 0           # nearest-neighbour view: logical size = stored size << shift

    @property
    def nbytes(self):
        return self.n * _ESIZE[self.dt]

    def numpy(self):
        torch.cuda.synchronize()
        a = self.t[:self.n].float().cpu().numpy() if self.dt != U8 else self.t[:self.n].cpu().numpy()
        return a.reshape(self.shape)


class DualBuf:
    """The value of tf.concat([a, b], axis=3) whose only reader is a 3x3 convolution: never materialised -- the convolution reads
    the two tensors in place (struct Dual in csrc/conv_mfma.hip), its data gradient writes their two gradients directly."""

    def __init__(self, a, b):
        self.a, self.b = a, b
        self.shape = tuple(a.shape[:-1]) + (a.shape[-1] + b.shape[-1],)
        self.dt, self.ptr, self.n, self.shift = a.dt, a.ptr, a.n + b.n, 0
        self.k1 = a.shape[-1]


class HeadGrad:
    """Placeholder for the gradient of a = act(norm(y)) whose only reader is a 1x1 head: dA = dy_head w_head^T is never materialised,
    the producer's norm backward launches form it on the fly (phx_norm_bwd_reduce_head / phx_norm_bwd_apply_fused_head)."""

    def __init__(self, like, dy, w_ptr, nout):
        self.shape, self.dt, self.n = like.shape, like.dt, like.n
        self.dy, self.w_ptr, self.nout = dy, w_ptr, nout


class ParamStore:
    """Flat arenas for every variable of a graph (created once, shared by all plans of a model)."""

    def __init__(self, graph, seed=0, live=None):
        """live: names of the trainable variables the loss depends on (live_variables()).  They are laid out FIRST, so the
        data-parallel exchange sums grads[:n_live] only -- the never-consumed up-sampling branches of the reference's zoo
        (887 808 parameters) get no gradient and need no reduction."""
        self.graph = graph
        self.offset, self.state_offset = {}, {}
        off = soff = 0
        names = list(graph.variables)
        if live is not None:
            names = [n for n in names if n in live] + [n for n in names if n not in live]
        self.n_live = None
        for name in names:
            v = graph.variables[name]
            if v.trainable:
                if live is not None and name not in live and self.n_live is None:
                    self.n_live = off
                self.offset[name] = off
                off += (v.size + 3) // 4 * 4
            else:
                self.state_offset[name] = soff
                soff += (v.size + 3) // 4 * 4
        self.n_train, self.n_state = off, soff
        if self.n_live is None:
            self.n_live = off
        dev = _device()
        self.params = torch.zeros(max(off, 4), dtype=torch.float32, device=dev)
        self.grads = torch.zeros_like(self.params)
        self.adam_m = torch.zeros_like(self.params)
        self.adam_v = torch.zeros_like(self.params)
        self.state = torch.zeros(max(soff, 4), dtype=torch.float32, device=dev)
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)          # optimiser step t-1 (also the noise step)
        self.noise_step = torch.zeros(1, dtype=torch.int32, device=dev)    # Philox step word of sampling plans
        self.lr = torch.full((1,), 1e-3, dtype=torch.float32, device=dev)
        self.initialize(seed)

    def initialize(self, seed=0):
        """tf.global_variables_initializer() (phiseg_model.py:175)."""
        self.load({name: v.initial_value(seed) for name, v in self.graph.variables.items()})
        self.adam_m.zero_()
        self.adam_v.zero_()
        self.step.zero_()

    def _slot(self, name):
        v = self.graph.variables[name]
        if v.trainable:
            return self.params, self.offset[name], v
        return self.state, self.state_offset[name], v

    def ptr(self, var):
        arena, off, _ = self._slot(var.name)
        return arena.data_ptr() + 4 * off

    def grad_ptr(self, var):
        assert var.trainable
        return self.grads.data_ptr() + 4 * self.offset[var.name]

    def load(self, values):
        for name, val in values.items():
            arena, off, v = self._slot(name)
            a = torch.as_tensor(np.asarray(val, dtype=np.float32).reshape(-1))
            assert a.numel() == v.size, "shape mismatch for %s" % name
            arena[off:off + v.size] = a.to(arena.device)
        torch.cuda.synchronize()

    def export(self, grads=False):
        torch.cuda.synchronize()
        out = {}
        for name, v in self.graph.variables.items():
            if grads and not v.trainable:
                continue
            arena, off, _ = (self.grads, self.offset[name], v) if grads else self._slot(name)
            out[name] = arena[off:off + v.size].cpu().numpy().reshape(v.shape)
        return out

    def set_lr(self, lr):
        self.lr.fill_(float(lr))
        device_sync()

    def set_step(self, step):
        self.step.fill_(int(step))
        device_sync()

    def export_adam(self):
        """-> {variable name: (m, v)} for the trainable variables (TF's '<var>/Adam', '<var>/Adam_1' slots)."""
        torch.cuda.synchronize()
        out = {}
        for name, v in self.graph.variables.items():
            if v.trainable:
                off = self.offset[name]
                out[name] = (self.adam_m[off:off + v.size].cpu().numpy().reshape(v.shape),
                             self.adam_v[off:off + v.size].cpu().numpy().reshape(v.shape))
        return out

    def load_adam(self, slots):
        for name, (m, vv) in slots.items():
            v = self.graph.variables.get(name)
            if v is None or not v.trainable:
                continue
            off = self.offset[name]
            for arena, val in ((self.adam_m, m), (self.adam_v, vv)):
                a = torch.as_tensor(np.asarray(val, dtype=np.float32).reshape(-1))
                assert a.numel() == v.size, "shape mismatch for the Adam slot of %s" % name
                arena[off:off + v.size] = a.to(arena.device)
        torch.cuda.synchronize()

    def reset_optimizer(self):
        self.adam_m.zero_()
        self.adam_v.zero_()
        self.set_step(0)


class Plan:
    """One compiled (fetches, loss) program for a fixed batch size / training flag / compute dtype."""

    def __init__(self, store, fetches, loss=None, batch=1, training=True, compute_dtype="f32", optimize=True,
                 rng_seed=42, sample_offset=0, loss_inv_batch=None, stream=None, use_hip_graph=True,
                 split_optimizer=False, n_lanes=None):
        self.L = rt.lib()
        self.store = store
        self.graph = store.graph
        self.B = int(batch)
        self.training = bool(training)
        self.act_dt = {"f32": F32, "bf16": BF16}[compute_dtype]
        self.rng_seed = int(rng_seed)
        self.sample_offset = int(sample_offset)
        self.inv_batch = float(loss_inv_batch) if loss_inv_batch is not None else 1.0 / self.B
        self.loss = loss
        self.optimize = bool(optimize and loss is not None)
        self.split_optimizer = split_optimizer      # data-parallel: [fwd+bwd] | all-reduce | [adam]
        self.use_hip_graph = use_hip_graph and os.environ.get("PHX_HIP_GRAPH", "1") == "1"     # 0: replay the launch list on the lane streams
        # Lanes: independent sub-graphs (posterior / prior encoders, the per-level likelihood chains) are enqueued on
        # separate HIP streams so the many small-map kernels overlap; cross-lane dependencies are HIP events.  The
        # whole multi-stream launch sequence is captured into ONE hipGraph (fork from / join into lane 0).
        self._lanes = []
        if n_lanes is None:
            n_lanes = int(os.environ.get("PHX_LANES", "2"))
        if stream is None:
            for _ in range(max(1, int(n_lanes))):
                st = ctypes.c_void_p()
                self.L.stream_create(ctypes.byref(st))
                self._lanes.append(st)
            self._own_stream = True
        else:
            self._lanes, self._own_stream = [stream], False
        self._lane = 0
        self._events = []
        self._last_rec, self._lane_seq = {}, {}
        self._ev_order, self._waited = {}, {}
        self.launches, self.opt_launches = [], []
        self._cur = self.launches
        self.val, self.grad, self.saved = {}, {}, {}
        self.tags = {}
        self.feeds = {}
        self._keep = []
        self._wpk = {}
        self._tail_jobs = []          # launches that follow the deferred ones on lane 0 (padded-filter folds)
        self._wgm_jobs = {}           # deferred small-map filter gradients by kernel variant: records, total blocks, LDS bytes
        self._headw_jobs = {}         # deferred 1x1-head filter gradients by (x dtype, nout): (x, dy, dw, db, npix, C, PL, chunk, grid, lds)
        self._wgr_jobs = []           # deferred filter-gradient reductions: (ws, dw, nslice, Cin, Cout, tci, tco, gx, gy)
        self._pack_jobs = []          # (w, wpk_fwd, wpk_dgrad, Cin, Cin_pad, Cout): ONE multi-filter pack launch per run
        self._bninfer_jobs = []       # (gamma, beta, moving_mean, moving_var, scale, shift, C, eps): ONE launch per run
        self._zarena = torch.zeros(8 << 20, dtype=torch.float32, device=_device())     # 32 MB of per-step accumulators
        self._zused = 0
        self.fetches = list(fetches)
        self.n_launch_fwd = self.n_launch_bwd = 0
        self._build()
        torch.cuda.synchronize()
        self._graph_exec = self._graph_exec_opt = None

    @property
    def stream(self):
        return self._lanes[self._lane]

    def _new_event(self):
        ev = ctypes.c_void_p()
        self.L.event_create(ctypes.byref(ev))
        self._events.append(ev)
        return ev

    def _record(self, lane):
        """Record an event at the current tail of `lane`; returns (event, lane).  Nothing enqueued on the lane since its
        previous record -> that event is reused (two back-to-back records of distinct events on one stream, one of them
        waited on by another stream, make hipStreamEndCapture segfault on ROCm 7.2)."""
        last = self._last_rec.get(lane)
        if last is not None and last[1] == self._lane_seq.get(lane, 0) and last[2] is self._cur:
            return (last[0], lane)
        ev = self._new_event()
        self._cur.append((self.L.event_record, (ev, self._lanes[lane])))
        self._last_rec[lane] = (ev, self._lane_seq.get(lane, 0), self._cur)
        self._ev_order[ev.value] = (id(self._cur), len(self._cur))      # position in the launch list = order on its lane
        return (ev, lane)

    def _wait(self, evl):
        """Make the current lane wait for an (event, lane) pair recorded elsewhere."""
        if evl is not None and evl[1] != self._lane:
            # events of one lane are ordered: a wait for an event recorded BEFORE one this lane already waited for adds
            # nothing but a graph edge (~3 us each in the replayed hipGraph)
            lst, pos = self._ev_order.get(evl[0].value, (None, -1))
            key = (self._lane, evl[1], lst)
            if pos >= 0 and self._waited.get(key, -1) >= pos:
                return
            self._waited[key] = pos
            self._cur.append((self.L.stream_wait_event, (self.stream, evl[0])))
            self._lane_seq[self._lane] = self._lane_seq.get(self._lane, 0) + 1

    # ---- PHX_STAMPS=1: device wall-clock stamps around every operator (tools/lane_timeline.py) ----
    def _stamp_begin(self):
        if _STAMPS:
            if not hasattr(self, "_stamp_buf"):
                self._stamp_buf = torch.zeros(16384, dtype=torch.int64, device=_device())
                self.stamps = []                  # (phase, op name, lane, index of the begin stamp)
            self._stamp_i = len(self.stamps) * 2
            self._emit(self.L.stamp, self._stamp_buf.data_ptr() + 8 * self._stamp_i, self.stream)

    def _stamp_end(self, phase, op, n0):
        if _STAMPS:
            if len(self._cur) == n0 + 1:          # the operator launched nothing: drop its begin stamp
                self._cur.pop()
                return
            self._emit(self.L.stamp, self._stamp_buf.data_ptr() + 8 * (self._stamp_i + 1), self.stream)
            self.stamps.append((phase, op.name, self._lane, self._stamp_i))

    def _emit_deferred(self):
        """Launch everything the backward pass has deferred so far (filter gradients of the small / mid-size maps, the sums
        over partial filters, padded-filter folds, head filter gradients) on the current lane, and clear the lists."""
        for variant, grp in sorted(self._wgm_jobs.items()):
            desc = torch.frombuffer(bytearray(b"".join(grp["recs"])), dtype=torch.uint8).to(_device())
            self._keep.append(desc)
            self._emit(self.L.conv3x3_wgrad_multi, desc.data_ptr(), len(grp["recs"]), grp["blocks"], variant,
                       grp["lds"], self.stream)
        if self._wgr_jobs:
            rec = np.zeros(len(self._wgr_jobs), dtype=[("ws", "<u8"), ("dw", "<u8"), ("nslice", "<i4"), ("cin", "<i4"), ("cout", "<i4"),
                                                       ("tci", "<i4"), ("tco", "<i4"), ("gx", "<i4"), ("gy", "<i4"), ("blk0", "<i4")])
            blk = 0
            for i, j in enumerate(self._wgr_jobs):
                rec[i] = tuple(j) + (blk,)
                blk += j[7] * j[8]
            desc = torch.from_numpy(rec.view(np.uint8).copy()).to(_device())
            self._keep.append(desc)
            self._emit(self.L.wgrad_reduce_multi, desc.data_ptr(), len(self._wgr_jobs), blk, self.stream)
        for fn, args in self._tail_jobs:
            self._emit(fn, *args, self.stream)
        for (xdt, nout), jobs in self._headw_jobs.items():
            rec = np.zeros(len(jobs), dtype=[("x", "<u8"), ("dy", "<u8"), ("dw", "<u8"), ("db", "<u8"), ("npix", "<u8"), ("C", "<i4"),
                                             ("PL", "<i4"), ("chunk", "<i4"), ("blk0", "<i4")])
            blk = lds = 0
            for i, j in enumerate(jobs):
                rec[i] = (j[0], j[1], j[2], j[3], j[4], j[5], j[6], j[7], blk)
                blk += j[8]
                lds = max(lds, j[9])
            desc = torch.from_numpy(rec.view(np.uint8).copy()).to(_device())
            self._keep.append(desc)
            self._emit(self.L.head1x1_wgrad_multi, desc.data_ptr(), len(jobs), blk, xdt, nout, lds, self.stream)
        self._wgm_jobs, self._wgr_jobs, self._tail_jobs, self._headw_jobs = {}, [], [], {}

    def _prune_dead_event_records(self):
        """Every gradient contribution records an event in case another lane folds it in; most are consumed on the lane
        that produced them and nobody waits.  An unwaited record is still a node of the captured graph (214 records against
        60 waits in the PHiSeg training plan): replace those by no-ops -- the launch list keeps its indices."""
        for lst in (self.launches, self.opt_launches):
            waited = {a[1].value for f, a in lst if f is self.L.stream_wait_event}
            for i, (f, a) in enumerate(lst):
                if f is self.L.event_record and a[0].value not in waited:
                    lst[i] = (_noop, ())

    def _lane_of(self, op):
        """Lane plan.  Lane 0 (the capture's origin stream): posterior, the likelihood's top-down fusion path, losses.
        Lane 1: prior.  Lanes 2..: the independent per-level likelihood chains (z{i}_post_*, preups_{i}).  Every
        cross-lane dependency then has lane 0 on one side: on ROCm 7.2 an event wait between two NON-origin streams of a
        multi-stream capture makes hipStreamEndCapture segfault (found by bisecting the launch list)."""
        n = len(self._lanes)
        if n == 1:
            return 0
        name = op.name
        if name.startswith("prior/"):
            return 1
        if name.startswith("likelihood/"):
            import re
            m = re.match(r"likelihood/(?:z(\d+)_post_|preups_(\d+)/)", name)
            if m:
                lvl = int(m.group(1) or m.group(2))
                if n == 2:
                    return 1 if lvl <= _LIK_SIDE_MAXLVL else 0
                return 2 + lvl % (n - 2)
        return 0

    # ---------------------------------------------------------------------------------------------
    def _emit(self, fn, *args, tag=None, flops=0.0, shape=None):
        if tag is not None:
            self.tags[(id(self._cur), len(self._cur))] = (tag, float(flops), shape)
        self._cur.append((fn, args))
        self._lane_seq[self._lane] = self._lane_seq.get(self._lane, 0) + 1

    def _alloc(self, shape, dt, zero=False):
        b = Buf(shape, dt, zero=zero)
        self._keep.append(b)
        return b

    def _alloc_zeroed(self, n):
        """fp32 accumulator that must be zero at the start of every run: carved from one arena, ONE memset per run."""
        n4 = (int(n) + 3) // 4 * 4
        assert self._zused + n4 <= self._zarena.numel(), "zero arena exhausted"
        b = Buf((int(n),), F32, like=self._zarena[self._zused:self._zused + n4])
        self._zused += n4
        self._keep.append(b)
        return b

    def _dt_of(self, t):
        return {G.KIND_ACT: self.act_dt, G.KIND_F32: F32, G.KIND_U8: U8}[t.kind]

    def _cshape(self, t):
        return tuple(self.B * getattr(t, "bmul", 1) if s is None else s for s in t.shape)

    def _alloc_like(self, t, zero=False):
        return self._alloc(self._cshape(t), self._dt_of(t), zero=zero)

    def _noise_step_ptr(self):
        """Training plans key the noise by the optimiser step; sampling plans by their own counter."""
        return (self.store.step if self.loss is not None else self.store.noise_step).data_ptr()

    def _needed_ops(self):
        want = set()
        stack = [t.op for t in self.fetches] + ([self.loss.op] if self.loss is not None else [])
        while stack:
            op = stack.pop()
            if op in want:
                continue
            want.add(op)
            stack.extend(i.op for i in op.inputs)
        ops = [op for op in self.graph.ops if op in want]
        return self._coarse_first(ops)

    def _backward_order(self, ops, opset):
        """Emission order of the backward pass: the reverse of the forward order."""
        return list(reversed(ops))

    @staticmethod
    def _coarse_first(ops):
        """The likelihood's per-level chains (z{i}_post_*, preups_{i}: independent of each other, created finest level first,
        likelihoods.py:186-206) are emitted COARSEST level first: the top-down fusion path on the other lane starts at the coarsest
        level and needs chain i only when it reaches level i, so it no longer waits for all five chains (0.9 ms with the 128 x 128
        chain in front); the backward pass -- reverse emission order -- then reaches the chains in the order the top-down backward
        releases their gradients (finest first).  Results do not depend on the order."""
        import re
        pat = re.compile(r"likelihood/(?:z(\d+)_post_|preups_(\d+)/)")
        lvl = {}
        for op in ops:
            m = pat.match(op.name)
            if m:
                lvl[op] = int(m.group(1) or m.group(2))
        if not lvl:
            return ops
        idx = [i for i, op in enumerate(ops) if op in lvl]
        first, last = idx[0], idx[-1]
        inner = ops[first:last + 1]
        side = [op for op in inner if op in lvl]
        rest = [op for op in inner if op not in lvl]           # (anything interleaved keeps its place after the chains' inputs)
        chain_ops = set(side)
        for op in rest:                                        # only safe if nothing in between consumes a chain
            if any(i.op in chain_ops for i in op.inputs):
                return ops
        side.sort(key=lambda op: -lvl[op])                     # stable: a chain's own order is kept
        return ops[:first] + rest + side + ops[last + 1:]

    def _build(self):
        ops = self._needed_ops()
        self.ops = ops
        # which tensors depend on trainable variables
        self.req = {}
        for op in ops:
            r = op.type in ("conv_unit", "norm_act") or any(self.req.get(i, False) for i in op.inputs)
            for o in op.outputs:
                self.req[o] = r
        self.loss_weight = {}
        if self.loss is not None:
            assert self.loss.op.type == "weighted_sum", "loss must be built with graph.weighted_sum"
            for t, w in zip(self.loss.op.inputs, self.loss.op.attrs["weights"]):
                self.loss_weight[t] = w
        with_bw = self.loss is not None
        self._lane = 0
        self._emit(self.L.memset, self._zarena.data_ptr(), 0, 4, self.stream)       # size patched below
        self._emit(self.L.memset, self._zarena.data_ptr(), 0, 4, self.stream)       # slot 1: multi-filter pack, patched below
        self._emit(_noop)                                                            # slot 2: inference-mode batch-norm scale / shift of all layers
        if with_bw:
            self._emit(self.L.memset, self.store.grads.data_ptr(), 0, self.store.grads.numel() * 4, self.stream)
        nl = len(self._lanes)
        self.op_lane = {op: self._lane_of(op) for op in ops}
        opset = set(ops)
        self._opset = opset
        self._lat = self._find_latent_heads(ops)     # op -> record of a fused (mu head, sigma head[, sample]) group
        self._kl_group = None
        kls = [op for op in ops if op.type == "kl"]
        if len(kls) >= 2 and len(kls) <= 8:
            ws = [self.loss_weight.get(op.outputs[0], 0.0) for op in kls]
            i0, i1 = ops.index(kls[0]), ops.index(kls[-1])
            between = [op for op in ops[i0:i1 + 1] if op.type != "kl"]
            # one launch at the last level's operator: nothing in between may read a level's loss, one lane, one loss weight
            if (len({self.op_lane[op] for op in kls}) == 1 and all(abs(w - ws[0]) < 1e-12 for w in ws)
                    and not any(i.op in kls for b in between for i in b.inputs)):
                self._kl_group = dict(ops=kls, recs=[], gscale=ws[0])
        self._bw_skip = set()
        self._norm_head = {}          # 1x1 head op -> the conv unit whose apply pass computed it (phx_norm_apply_fused_head)
        fork = self._record(0) if nl > 1 else None          # lanes 1.. join the capture / wait for the memsets
        for ln in range(1, nl):
            self._lane = ln
            self._wait(fork)
        if nl > 1:
            # ROCm 7.2's graph executor starts a forked branch only when the origin stream first WAITS for it (measured with
            # PHX_STAMPS: the prior lane, ready at t = 0, begins after lane 0's whole forward, alone on the GPU).  With
            # lane 0 waiting right here for a token kernel at the head of every other lane, the prior encoder runs FIRST and
            # the posterior after it (the executor still does not overlap them), which takes the prior's forward off the
            # end of the forward pass: +0.7 % measured.
            self._touch = torch.zeros(64, dtype=torch.int64, device=_device())
            self._keep.append(self._touch)
            toks = []
            for ln in range(1, nl):
                self._lane = ln
                self._emit(self.L.stamp, self._touch.data_ptr() + 8 * ln, self.stream)
                toks.append(self._record(ln))
            self._lane = 0
            for evl in toks:
                self._wait(evl)
        self.fw_event = {}
        for op in ops:
            ln = self._lane = self.op_lane[op]
            for t in op.inputs:                               # forward dependencies produced on other lanes
                self._wait(self.fw_event.get(self._real_producer(t)))
            n0 = len(self._cur)
            self._stamp_begin()
            getattr(self, "_fw_" + op.type)(op, with_bw)
            self._stamp_end("fw", op, n0)
            if nl > 1 and len(self._cur) > n0:
                cross = any(self.op_lane.get(c, ln) != ln for o in op.outputs for c in self._real_consumers(o, opset))
                rec = self._lat.get(op)
                if rec is not None and rec["last"] is op:
                    # the group's one launch was emitted here: it stands for all of its operators (their readers on other lanes wait
                    # for THIS point, not for the place where the mu head alone would have run)
                    gops = [o for o in (rec["mu"], rec["sig"], rec["add"]) if o is not None]
                    if any(self.op_lane.get(c, ln) != ln for g in gops for o in g.outputs for c in self._real_consumers(o, opset)):
                        ev = self._record(ln)
                        for g in gops:
                            self.fw_event[g] = ev
                        cross = False
                if cross or op.type in ("residual_ce", "kl", "weighted_sum"):
                    self.fw_event[op] = self._record(ln)
        self.n_launch_fwd = len(self.launches)
        self.pending = {}                                    # tensor -> [(buf, (event, lane))]: late grad contributions
        if with_bw:
            # (Launching what the likelihood and the prior have deferred on the prior's lane as soon as their backward is
            # done, beside the posterior's backward chain, was measured 7 % slower than one batch after the join.)
            bw_ops = self._backward_order(ops, opset)
            for op in bw_ops:
                if op in self._bw_skip:
                    continue                              # (its backward ran inside a fused group's launch)
                if any(o in self.grad for o in op.outputs) or op.type in ("residual_ce", "kl", "l2_weights"):
                    self._lane = self.op_lane[op]
                    self._cur_bw_op = op
                    self._wait(self.fw_event.get(op))        # forward of this op may live on another lane's past
                    n0 = len(self._cur)
                    self._stamp_begin()
                    for o in op.outputs:
                        self._finalize_grad(o)
                    getattr(self, "_bw_" + op.type)(op)
                    self._stamp_end("bw", op, n0)
            self.n_launch_bwd = len(self.launches) - self.n_launch_fwd
        if nl > 1:                                            # join: lane 0 waits for every other lane
            tails = [self._record(ln) for ln in range(1, nl)]
            self._lane = 0
            for evl in tails:
                self._wait(evl)
        self._lane = 0
        self._emit_deferred()
        self._prune_dead_event_records()
        self.launches[0] = (self.L.memset, (self._zarena.data_ptr(), 0, max(self._zused, 1) * 4, self.stream))
        if self._pack_jobs:           # slot 1 was reserved before the fork: refresh every packed bf16 filter in one launch
            rec = np.zeros(len(self._pack_jobs), dtype=[("w", "<u8"), ("wf", "<u8"), ("wd", "<u8"), ("cin", "<i4"),
                                                         ("cpad", "<i4"), ("cout", "<i4"), ("pad", "<i4")])
            for i, j in enumerate(self._pack_jobs):
                rec[i] = (j[0], j[1], j[2], j[3], j[4], j[5], j[6] if len(j) > 6 else 0)
            self._pack_desc = torch.from_numpy(rec.view(np.uint8).copy()).to(_device())
            self._keep.append(self._pack_desc)
            self.launches[1] = (self.L.pack_conv3x3_bf16_multi, (self._pack_desc.data_ptr(), len(self._pack_jobs), self.stream))
        if self._bninfer_jobs:        # slot 2: scale / shift of every inference-mode batch-norm layer in one launch
            rec = np.zeros(len(self._bninfer_jobs), dtype=[("gamma", "<u8"), ("beta", "<u8"), ("mm", "<u8"), ("mv", "<u8"),
                                                            ("scale", "<u8"), ("shift", "<u8"), ("C", "<i4"), ("eps", "<f4")])
            for i, j in enumerate(self._bninfer_jobs):
                rec[i] = j
            self._bninfer_desc = torch.from_numpy(rec.view(np.uint8).copy()).to(_device())
            self._keep.append(self._bninfer_desc)
            self.launches[2] = (self.L.bn_infer_scale_shift_multi, (self._bninfer_desc.data_ptr(), len(self._bninfer_jobs), self.stream))
        if self.optimize:
            if self.split_optimizer:
                self._cur = self.opt_launches
            s = self.store
            self._emit(self.L.adam_tf1, s.params.data_ptr(), s.grads.data_ptr(), s.adam_m.data_ptr(),
                       s.adam_v.data_ptr(), s.n_train, s.lr.data_ptr(), 0.9, 0.999, 1e-8, s.step.data_ptr(),
                       self.stream)
            self._emit(self.L.step_increment, s.step.data_ptr(), self.stream)
            self._cur = self.launches

    # ---- forward emitters -----------------------------------------------------------------------
    def _fw_placeholder(self, op, bw):
        t = op.outputs[0]
        b = self._alloc_like(t, zero=True)
        self.val[t] = b
        self.feeds[op.name.rsplit("/", 1)[-1]] = b

    def _fw_constant(self, op, bw):
        b = self._alloc((), F32, zero=True)
        if op.attrs["value"] != 0.0:
            b.t.fill_(op.attrs["value"])
        self.val[op.outputs[0]] = b

    def _fw_l2_weights(self, op, bw):
        st = self.store
        if not hasattr(st, "decay_mask"):
            m = torch.zeros_like(st.params)
            for v in op.attrs["vars"]:
                off = st.offset[v.name]
                m[off:off + v.size] = 1.0
            st.decay_mask = m
            device_sync()
        out = self._alloc((), F32)
        work = self._alloc((256,), F32)
        self.val[op.outputs[0]] = out
        # data parallel (loss_inv_batch = 1 / (B * world)): every rank evaluates the term on the full parameter set, the scalar fetches
        # and the gradient arena are SUMMED over the ranks -> each rank carries a 1 / world share of the term and of its gradient
        share = self.inv_batch * self.B
        self._emit(self.L.l2_masked, st.params.data_ptr(), st.decay_mask.data_ptr(), st.n_train, op.attrs["scale"] * share, work.ptr, out.ptr,
                   self.stream)
        if bw:
            self._l2_weight = self.loss_weight.get(op.outputs[0], 0.0) * op.attrs["scale"] * share

    def _bw_l2_weights(self, op):
        st = self.store
        self._emit(self.L.axpy_masked, st.grads.data_ptr(), st.params.data_ptr(), st.decay_mask.data_ptr(), st.n_train,
                   float(self._l2_weight), self.stream)

    def _fw_one_hot(self, op, bw):
        pass            # virtual: consumed by the fused posterior-input kernel / the loss kernel

    def _fw_sub_const(self, op, bw):
        pass

    def _fw_nn_resize(self, op, bw):
        src = self.val[op.inputs[0]]
        v = Buf(src.shape, src.dt, like=src.t)
        v.shift = op.attrs["shift"]
        self.val[op.outputs[0]] = v

    def _fw_random_normal(self, op, bw):
        pass

    def _fw_mul(self, op, bw):
        pass

    def _fw_concat(self, op, bw):
        a, b = op.inputs
        ot = op.outputs[0]
        va, vb = self.val.get(a), self.val.get(b)
        cons = self._real_consumers(ot, self._opset)
        if (_dual_enabled() and self.act_dt == BF16 and self._dt_of(ot) == BF16 and isinstance(va, Buf) and isinstance(vb, Buf) and va.dt == BF16 and vb.dt == BF16
                and len(va.shape) == 4 and va.shape[-1] % 32 == 0 and vb.shape[-1] % 32 == 0 and len(cons) == 1 and ot not in self.fetches):
            c = cons[0]
            ca = c.attrs if c.type == "conv_unit" else None
            if (ca is not None and ca["ksize"] == 3 and ca.get("transposed") is None and ca.get("general") is None
                    and ca["W"].shape[-1] % 32 == 0 and self.op_lane.get(c) == self.op_lane.get(op) and c not in self._lat):
                # concat-free: the one reader, a 3x3 convolution on the MFMA path, takes the two tensors as they are (no launch here)
                self.val[ot] = DualBuf(va, vb)
                return
        out = self._alloc_like(ot)
        self.val[ot] = out
        npix = int(np.prod(out.shape[:-1]))
        if b.op.type == "sub_const" and b.op.inputs[0].op.type == "one_hot":
            # concat[x, one_hot(s) - 0.5] (posteriors.py:87) in one kernel
            oh = b.op.inputs[0].op
            assert abs(b.op.attrs["c"] - 0.5) < 1e-12 and a.shape[-1] == 1
            xb, sb = self.val[a], self.val[oh.inputs[0]]
            self._emit(self.L.posterior_input, xb.ptr, sb.ptr, out.ptr, out.dt, npix, oh.attrs["depth"], self.stream)
            return
        ab, bb = self._as_dt(self.val[a], out.dt), self._as_dt(self.val[b], out.dt)
        self._emit(self.L.concat2, ab.ptr, ab.shape[-1], bb.ptr, bb.shape[-1], out.ptr, npix, out.dt, self.stream)

    def _as_dt(self, buf, dt):
        if buf.dt == dt:
            return buf
        c = self._alloc(buf.shape, dt)
        self._emit(self.L.cast, buf.ptr, buf.dt, c.ptr, dt, buf.n, self.stream)
        return c

    def _packed(self, W):
        """bf16 packed copies of a 3x3 filter, refreshed at the head of every run (after Adam moved W)."""
        if W.name not in self._wpk:
            kh, kw, cin, cout = W.shape
            wf, wd = self._alloc((9 * cin * cout,), BF16), self._alloc((9 * cin * cout,), BF16)
            self._wpk[W.name] = (wf, wd)
            self._pack_jobs.append((self.store.ptr(W), wf.ptr, wd.ptr, cin, cin, cout))
        return self._wpk[W.name]

    def _fw_tconv_unit(self, op, bw):
        """tf.nn.conv2d_transpose -> [bias] -> [norm] -> act (tfwrapper/layers.py:197-258) on the direct kernels of tconv.hip;
        the normalisation runs as statistics pass + fused apply on the up-sampled tensor."""
        a = op.attrs
        x = self.val[op.inputs[0]]
        W, b = a["W"], a["b"]
        B, H, Wd = x.shape[0], x.shape[1], x.shape[2]
        S, Lb = self.stream, self.L
        if a.get("general") is not None:
            # strided / dilated SAME convolution on the direct kernels of gconv.hip (conv2D with strides, dilated_conv2D,
            # dense_layer as a 1x1 convolution of the flattened input); filter HWIO (a dense layer's [F, U] is [1][1][F][U])
            geo = (B, H, Wd, W.shape[-2], W.shape[-1]) + tuple(a["general"])
            cin, cout = W.shape[-2], W.shape[-1]
            conv_fwd = Lb.gconv2d_fwd
        else:
            kh, kw, sh, sw = a["transposed"]
            cout, cin = W.shape[2], W.shape[3]
            geo = (B, H, Wd, cin, cout, kh, kw, sh, sw)
            conv_fwd = Lb.tconv2d_fwd
        out = self._alloc_like(op.outputs[0])
        self.val[op.outputs[0]] = out
        Ho, Wo = out.shape[1], out.shape[2]
        act = rt.ACT_CODES[a["act"]]
        norm = a["norm"]
        training = a["training"] if isinstance(a["training"], bool) else self.training
        wptr, bptr = self.store.ptr(W), (self.store.ptr(b) if b is not None else None)
        st = dict(x=x, out=out, mfma=False, norm=norm, padded=False, cin_eff=cin, k1=False, head1x1=False,
                  transposed=a.get("transposed"), general=a.get("general"), geo=geo)
        if norm is None:
            self._emit(conv_fwd, x.ptr, x.dt, wptr, bptr, out.ptr, out.dt, *geo, act, S)
            self.saved[op] = st
            return
        nv = a["norm_vars"]
        gptr, beptr = self.store.ptr(nv["gamma"]), self.store.ptr(nv["beta"])
        y = self._alloc(out.shape, out.dt)
        self._emit(conv_fwd, x.ptr, x.dt, wptr, bptr, y.ptr, y.dt, *geo, 0, S)
        if norm == "batch":
            NS, P, Gn = 1, B * Ho * Wo, cout
        else:
            Gn = cout if norm == "instance" else (a["num_groups"] or max(2, cout // 16))
            NS, P = B, Ho * Wo
        scale, shift = self._alloc((NS * cout,), F32), self._alloc((NS * cout,), F32)
        mean, rstd = self._alloc((NS * Gn,), F32), self._alloc((NS * Gn,), F32)
        eps = tfnorm.EPS[norm]
        if norm == "batch" and not training:
            self._emit(Lb.bn_infer_scale_shift, gptr, beptr, self.store.ptr(nv["moving_mean"]),
                       self.store.ptr(nv["moving_variance"]), eps, cout, scale.ptr, shift.ptr, S)
            self._emit(Lb.affine_act, y.ptr, y.dt, scale.ptr, shift.ptr, out.ptr, out.dt, NS, P, cout, act, S)
        else:
            sums = self._alloc_zeroed(NS * cout * 2)
            pivot = self._alloc((NS * cout,), F32)
            self._emit(Lb.norm_stats, y.ptr, y.dt, sums.ptr, pivot.ptr, NS, P, cout, S)
            upd = norm == "batch" and training and self.loss is not None
            self._emit(Lb.norm_apply_fused, y.ptr, y.dt, sums.ptr, pivot.ptr, gptr, beptr, eps, out.ptr, out.dt, mean.ptr, rstd.ptr,
                       scale.ptr, shift.ptr, self.store.ptr(nv["moving_mean"]) if upd else None,
                       self.store.ptr(nv["moving_variance"]) if upd else None, (1.0 - tfnorm.BN_DECAY) if upd else 0.0,
                       NS, P, cout, Gn, act, S)
        st.update(y=y, scale=scale, shift=shift, mean=mean, rstd=rstd, NS=NS, P=P, G=Gn)
        self.saved[op] = st

    # ---- fused latent heads: mu = conv1x1(x), sigma = softplus(conv1x1(x)), z = mu + sigma * eps (posteriors.py:125-128,
    # priors.py:117-120) as one launch forward (phx_latent_heads_fwd) and one backward (phx_latent_heads_bwd) ----------------------
    def _find_latent_heads(self, ops):
        pos = {op: i for i, op in enumerate(ops)}
        opset = set(ops)
        out = {}

        def is_head(op, act):
            a = op.attrs
            if op.type != "conv_unit" or a.get("transposed") is not None or a.get("general") is not None:
                return False
            W = a["W"]
            return (a["ksize"] == 1 and a["norm"] is None and a["b"] is not None and a["act"] == act and W.shape[-1] in (2, 4, 6)
                    and W.shape[-2] % 8 == 0 and op.outputs[0].kind == G.KIND_F32)
        for mu in ops:
            if mu in out or not is_head(mu, "identity"):
                continue
            x = mu.inputs[0]
            sib = [c for c in x.consumers if c in opset and c is not mu and is_head(c, "softplus")
                   and c.attrs["W"].shape == mu.attrs["W"].shape and c not in out]
            if len(sib) != 1:
                continue
            sig = sib[0]
            add = None
            for c in mu.outputs[0].consumers:
                if (c in opset and c.type == "add" and c.inputs[0] is mu.outputs[0] and c.inputs[1].op.type == "mul"
                        and c.inputs[1].op.inputs[0] is sig.outputs[0] and c.inputs[1].op.inputs[1].op.type == "random_normal"):
                    add = c
            members = [o for o in (mu, sig, add) if o is not None]
            last = max(members, key=lambda o: pos[o])
            virt = {add.inputs[1].op, add.inputs[1].op.inputs[1].op} if add is not None else set()
            # nothing may read mu / sigma before the group's launch, the heads must sit on one lane, and neither may be a fetch
            ok = all(pos.get(c, 1 << 30) > pos[last] or c in members or c in virt
                     for t in (mu.outputs[0], sig.outputs[0]) for c in t.consumers if c in opset)
            ok = ok and len({self.op_lane[o] for o in members}) == 1
            if not ok:
                continue
            rec = dict(mu=mu, sig=sig, add=add, last=last, x=x)
            for o in members:
                out[o] = rec
        return out

    def _fw_latent_group(self, rec):
        mu_op, sig_op, add_op = rec["mu"], rec["sig"], rec["add"]
        x = self.val[rec["x"]]
        mu, sigma = self.val[mu_op.outputs[0]], self.val[sig_op.outputs[0]]
        z = self.val[add_op.outputs[0]] if add_op is not None else None
        cin, zd = mu_op.attrs["W"].shape[-2], mu_op.attrs["W"].shape[-1]
        npix = int(np.prod(x.shape[:-1]))
        hw = npix // x.shape[0]
        sid = add_op.inputs[1].op.inputs[1].op.attrs["stream"] if add_op is not None else 0
        st = self.store
        self._emit(self.L.latent_heads_fwd, x.ptr, x.dt, st.ptr(mu_op.attrs["W"]), st.ptr(mu_op.attrs["b"]), st.ptr(sig_op.attrs["W"]),
                   st.ptr(sig_op.attrs["b"]), mu.ptr, sigma.ptr, z.ptr if z is not None else None, npix, cin, zd, hw, self.rng_seed,
                   self._noise_step_ptr(), sid, self.sample_offset, self.stream)
        rec.update(npix=npix, hw=hw, sid=sid, cin=cin, zd=zd)

    def _bw_latent_group(self, rec):
        """Called at the group's LAST operator (the first one the backward sweep meets): every contribution to the gradients of mu,
        sigma and z has been registered by then (their readers come later in the forward order)."""
        mu_op, sig_op, add_op = rec["mu"], rec["sig"], rec["add"]
        mu_t, sig_t = mu_op.outputs[0], sig_op.outputs[0]
        for t in (mu_t, sig_t):
            if t in self.grad:
                self._finalize_grad(t)
        dz = self.grad.get(add_op.outputs[0]) if add_op is not None else None
        dmu, dsg = self.grad.get(mu_t), self.grad.get(sig_t)
        for o in (mu_op, sig_op, add_op):
            if o is not None:
                self._bw_skip.add(o)
        if dz is None and dmu is None and dsg is None:
            return
        x_t = rec["x"]
        x, sigma = self.val[x_t], self.val[sig_t]
        npix, cin, zd = rec["npix"], rec["cin"], rec["zd"]
        gmu, gsig = self._alloc((npix, zd), F32), self._alloc((npix, zd), F32)
        st, Lb = self.store, self.L
        wmu, wsg = mu_op.attrs["W"], sig_op.attrs["W"]
        if not self.req.get(x_t, False):
            raise NotImplementedError("latent heads on a tensor without gradient")

        def wr(g):
            self._emit(Lb.latent_heads_bwd, dz.ptr if dz is not None else None, dmu.ptr if dmu is not None else None,
                       dsg.ptr if dsg is not None else None, sigma.ptr, st.ptr(wmu), st.ptr(wsg), g.ptr, g.dt, gmu.ptr, gsig.ptr, npix,
                       cin, zd, rec["hw"], self.rng_seed, self._noise_step_ptr(), rec["sid"], self.sample_offset, self.stream)
        self._add_grad(x_t, write_fn=wr)
        for hop, gy in ((mu_op, gmu), (sig_op, gsig)):      # the two filter / bias gradients: leaves, one launch for all heads later
            W, b = hop.attrs["W"], hop.attrs["b"]
            if cin % 8 == 0:
                plan4 = (ctypes.c_int * 4)()
                Lb.head1x1_wgrad_plan(npix, cin, zd, plan4)
                self._headw_jobs.setdefault((x.dt, zd), []).append((x.ptr, gy.ptr, st.grad_ptr(W), st.grad_ptr(b), npix, cin, plan4[0],
                                                                    plan4[1], plan4[2], plan4[3]))
            else:
                self._emit(Lb.head1x1_wgrad, x.ptr, x.dt, gy.ptr, st.grad_ptr(W), st.grad_ptr(b), npix, cin, zd, self.stream)

    def _norm_head_consumer(self, op):
        """The 1x1 head (bias, no norm, identity, fp32 out, 2 / 4 outputs) that is the ONLY reader of this unit's output, or None."""
        if self.act_dt != BF16:
            return None
        out = op.outputs[0]
        if out in self.fetches:
            return None
        cons = self._real_consumers(out, self._opset)
        if len(cons) != 1 or cons[0].type != "conv_unit" or cons[0] in self._lat:
            return None
        c, ca = cons[0], cons[0].attrs
        if (ca.get("transposed") is not None or ca.get("general") is not None or ca["ksize"] != 1 or ca["norm"] is not None
                or ca["b"] is None or ca["act"] != "identity" or c.inputs[0] is not out or c.outputs[0].kind != G.KIND_F32
                or c.outputs[0] in self.fetches or self.op_lane.get(c) != self.op_lane.get(op)):
            return None
        return c

    def _fw_conv_unit(self, op, bw):
        a = op.attrs
        if a.get("transposed") is not None or a.get("general") is not None:
            return self._fw_tconv_unit(op, bw)
        if op in self._norm_head:                # its forward ran inside the producer's apply pass
            x = self.val[op.inputs[0]]
            W = a["W"]
            self.saved[op] = dict(x=x, out=self.val[op.outputs[0]], mfma=False, norm=None, padded=False, cin_eff=W.shape[-2], k1=False,
                                  head1x1=True, norm_head=True)
            return
        rec = self._lat.get(op)
        if rec is not None:                      # a latent head: its arithmetic runs in the group's one launch
            self.val[op.outputs[0]] = self._alloc_like(op.outputs[0])
            self.saved[op] = dict(latent=True)
            if rec["last"] is op:
                self._fw_latent_group(rec)
            return
        x = self.val[op.inputs[0]]
        W, b = a["W"], a["b"]
        k, (_, _, cin, cout) = a["ksize"], W.shape
        B, H, Wd = x.shape[0], x.shape[1], x.shape[2]
        out = self._alloc_like(op.outputs[0])
        self.val[op.outputs[0]] = out
        act = rt.ACT_CODES[a["act"]]
        training = a["training"] if isinstance(a["training"], bool) else self.training
        mfma = (self.act_dt == BF16 and x.dt == BF16 and out.dt == BF16 and k == 3 and cin % 32 == 0
                and cout % 32 == 0)
        S, Lb = self.stream, self.L
        dual = x if isinstance(x, DualBuf) else None
        assert dual is None or mfma, "concat-free input reached a convolution off the MFMA path"

        def mfma_conv(y, bias_p, oscale_p, act_code, stats, stats_mode, ws, wsb):
            """One forward launch on the bf16 MFMA path (plain or concat-free input): phx_conv3x3_mfma_bf16_dual takes every option"""
            self._emit(Lb.conv3x3_mfma_bf16_dual, x.ptr, dual.b.ptr if dual is not None else None, dual.k1 if dual is not None else 0,
                       wf.ptr, y.ptr if y is not None else None, None, 0, bias_p, oscale_p, act_code,
                       stats.ptr if stats is not None else None, stats_mode, ws.ptr if ws is not None else None, wsb,
                       B, H, Wd, cin_eff, cout, S, tag="conv3x3_mfma_fwd", flops=18.0 * cin * cout * B * H * Wd)
        cin_eff = cin
        # Convolutions the 3x3 MFMA kernels do not take as they are: input channels not a multiple of 32 (image Cin = 1 / 3,
        # latent Cin = 2, prob_unet2D's feature + z concat) are zero-padded, and 1x1 filters (prob_unet2D's recombination
        # layers, likelihoods.py) run as the centre tap of a 3x3 -- 9x the FLOPs on the matrix cores still beats the fp32
        # direct kernel by 30x.  Both get their own packed filter copies ("padded" path).
        k1 = k == 1 and cout % 32 == 0
        padded = (self.act_dt == BF16 and out.dt == BF16 and cout % 32 == 0 and
                  ((k == 3 and (cin % 32 != 0 or x.dt != BF16)) or k1))
        if padded:
            cin_eff = (cin + 31) // 32 * 32
            if cin_eff != cin or x.dt != BF16:        # (with cin_eff == cin the pad kernel is just the cast to bf16)
                xp = self._alloc((B, H, Wd, cin_eff), BF16)
                self._emit(Lb.pad_channels_bf16, x.ptr, x.dt, cin, xp.ptr, cin_eff, B * H * Wd, S)
                x = xp
            mfma = True
        st = dict(x=x, out=out, mfma=mfma, norm=a["norm"], padded=padded, cin_eff=cin_eff, k1=bool(padded and k1))
        wptr, bptr = self.store.ptr(W), (self.store.ptr(b) if b is not None else None)
        if padded:
            wf = self._alloc((9 * cin_eff * cout,), BF16)
            need_dgrad = bw and self.req.get(op.inputs[0], False)
            wdp = self._alloc((9 * cin_eff * cout,), BF16) if need_dgrad else None
            st["wd_pad"] = wdp
            self._pack_jobs.append((wptr, wf.ptr, wdp.ptr if wdp else 0, cin, cin_eff, cout, 1 if k1 else 0))
        elif mfma:
            wf, _ = self._packed(W)

        head1x1 = (k == 1 and out.dt == F32 and cout in (2, 4, 6, 8) and a["norm"] is None and b is not None)
        st["head1x1"] = head1x1

        def tiles_fn():
            if dual is not None:
                return int(Lb.conv3x3_mfma_bf16_tiles_dual(B, H, Wd, cin_eff, cout))
            return int(Lb.conv3x3_mfma_bf16_tiles(B, H, Wd, cin_eff, cout))

        def conv_into(y, act_code, stats_direct=None, stats_part=None, stats_atomic=None):
            if stats_atomic is not None:
                mfma_conv(y, bptr, None, act_code, stats_atomic, 2, None, 0)
            elif head1x1:
                self._emit(Lb.head1x1_fwd, x.ptr, x.dt, wptr, bptr, y.ptr, B * H * Wd, cin, cout, act_code, S)
            elif mfma:
                wsb = int(Lb.conv3x3_mfma_ws_bytes(B, H, Wd, cin_eff, cout)) if stats_part is None else 0
                ws = self._alloc((wsb // 4,), F32) if wsb else None          # split-K slices (small maps)
                mfma_conv(y, bptr, None, act_code, stats_part, 1 if stats_part is not None else 0, ws, wsb)
            else:
                self._emit(Lb.conv2d_direct, x.ptr, x.dt, wptr, bptr, y.ptr, y.dt, B, H, Wd, cin, cout, k, act_code,
                           0, stats_direct.ptr if stats_direct is not None else None, S)

        norm = a["norm"]
        if norm is None:
            conv_into(out, act)
            self.saved[op] = st
            return
        nv = a["norm_vars"]
        gptr, beptr = self.store.ptr(nv["gamma"]), self.store.ptr(nv["beta"])
        y = self._alloc(out.shape, out.dt)
        if norm == "batch":
            NS, P, Gn = 1, B * H * Wd, cout
        else:
            Gn = cout if norm == "instance" else (a["num_groups"] or max(2, cout // 16))
            NS, P = B, H * Wd
        scale, shift = self._alloc((NS * cout,), F32), self._alloc((NS * cout,), F32)
        mean, rstd = self._alloc((NS * Gn,), F32), self._alloc((NS * Gn,), F32)
        eps = tfnorm.EPS[norm]
        if norm == "batch" and not training and mfma and not head1x1 and not bw:
            # inference-mode batch norm + activation folded into the convolution's epilogue (phx_conv3x3_mfma_bf16_affine):
            # one launch where the reference runs conv2d, batch_norm and relu; the scale / shift vectors of all layers come
            # from one launch at the head of the run
            self._bninfer_jobs.append((gptr, beptr, self.store.ptr(nv["moving_mean"]), self.store.ptr(nv["moving_variance"]),
                                       scale.ptr, shift.ptr, cout, eps))
            wsb = int(Lb.conv3x3_mfma_ws_bytes(B, H, Wd, cin_eff, cout))
            ws = self._alloc((wsb // 4,), F32) if wsb else None
            mfma_conv(out, shift.ptr, scale.ptr, act, None, 0, ws, wsb)
            st.update(scale=scale, shift=shift, NS=NS, P=P, G=Gn)
            self.saved[op] = st
            return
        if norm == "batch" and not training:
            self._emit(Lb.bn_infer_scale_shift, gptr, beptr, self.store.ptr(nv["moving_mean"]),
                       self.store.ptr(nv["moving_variance"]), eps, cout, scale.ptr, shift.ptr, S)
            conv_into(y, 0)
            self._emit(Lb.affine_act, y.ptr, y.dt, scale.ptr, shift.ptr, out.ptr, out.dt, NS, P, cout, act, S)
        else:
            # H <= 8 levels: the whole batch-norm layer in one launch (phx_bn_small_fwd / _bwd; csrc/elementwise.hip)
            # (policy P <= 1024, the H <= 4 levels: at P = 4096 the single launch measured no faster than the chain)
            bn_small = (norm == "batch" and y.dt == BF16 and out.dt == BF16 and P <= _BN_SMALL
                        and Lb.bn_small_supported(P, cout, BF16))
            if bn_small:
                upd = training and self.loss is not None
                mm = self.store.ptr(nv["moving_mean"]) if upd else None
                mv = self.store.ptr(nv["moving_variance"]) if upd else None
                mom = (1.0 - tfnorm.BN_DECAY) if upd else 0.0
                if mfma and not head1x1 and _BN_SMALL_F32 and Lb.conv3x3_mfma_f32out_supported(B, H, Wd, cin_eff, cout):
                    # the 2 x 2 / 4 x 4 levels: the pre-normalisation tensor stays in fp32 (the split-K kernel's accumulators, summed) --
                    # a channel is normalised from a few dozen to a few hundred values here, and the bf16 rounding of y (2^-9 of the
                    # channel mean) is blown up with their spread: the two coarsest KL terms trained 40 % high
                    y = self._alloc(out.shape, F32)
                    wsb = int(Lb.conv3x3_mfma_ws_bytes(B, H, Wd, cin_eff, cout))
                    ws = self._alloc((wsb // 4,), F32) if wsb else None
                    self._emit(Lb.conv3x3_mfma_bf16_f32out, x.ptr, dual.b.ptr if dual is not None else None,
                               dual.k1 if dual is not None else 0, wf.ptr, y.ptr, ws.ptr if ws is not None else None, wsb,
                               B, H, Wd, cin_eff, cout, S, tag="conv3x3_mfma_fwd", flops=18.0 * cin * cout * B * H * Wd)
                else:
                    conv_into(y, 0)
                self._emit(Lb.bn_small_fwd, y.ptr, y.dt, gptr, beptr, eps, out.ptr, mean.ptr, rstd.ptr, scale.ptr, shift.ptr,
                           mm, mv, mom, P, cout, act, S,
                           tag="bytes_norm_apply", flops=float(y.nbytes + out.nbytes))
                st.update(y=y, scale=scale, shift=shift, mean=mean, rstd=rstd, NS=NS, P=P, G=Gn, bn_small=True)
                self.saved[op] = st
                return
            if (norm != "batch" and mfma and not head1x1 and dual is None and not _DETERMINISTIC and (_fgn_mode() >= 2 or (_fgn_mode() == 1 and Gn != cout)) and y.dt == BF16 and out.dt == BF16
                    and x.dt == BF16 and Lb.conv3x3_fgn_supported(B, H, Wd, cin_eff, cout, Gn)
                    and Lb.norm_small_supported(NS, P, cout, Gn, BF16)):
                # maps of at most 16 x 16: convolution, bias, group / instance norm and activation in ONE launch (a block holds whole
                # samples and whole groups: no cross-block step); the backward pass is phx_norm_small_bwd's
                self._emit(Lb.conv3x3_mfma_bf16_fgn, x.ptr, wf.ptr, y.ptr, out.ptr, bptr, gptr, beptr, eps, Gn, act, mean.ptr, rstd.ptr,
                           scale.ptr, shift.ptr, B, H, Wd, cin_eff, cout, S,
                           tag="conv3x3_mfma_fwd", flops=18.0 * cin * cout * B * H * Wd, shape=("fgn", B, H, Wd, cin_eff, cout))
                st.update(y=y, scale=scale, shift=shift, mean=mean, rstd=rstd, NS=NS, P=P, G=Gn, norm_small=True)
                self.saved[op] = st
                return
            # group / instance norm on maps of up to 256 pixels: the whole layer in one launch as well (phx_norm_small_fwd / _bwd: a
            # wave per (sample, 16-channel slice)); a split-K convolution hands over its slices and its bias
            if (norm != "batch" and y.dt == BF16 and out.dt == BF16
                    and Lb.norm_small_supported(NS, P, cout, Gn, BF16)):
                conv_into(y, 0)
                self._emit(Lb.norm_small_fwd, y.ptr, None, 0, None, gptr, beptr, eps, out.ptr, mean.ptr, rstd.ptr,
                           scale.ptr, shift.ptr, NS, P, cout, Gn, act, S,
                           tag="bytes_norm_apply", flops=float(y.nbytes + out.nbytes))
                st.update(y=y, scale=scale, shift=shift, mean=mean, rstd=rstd, NS=NS, P=P, G=Gn, norm_small=True)
                self.saved[op] = st
                return
            sums = self._alloc_zeroed(NS * cout * 2)
            pivot = None
            # shifted (pivot) sums in a stand-alone pass: always on the fp32 parity path, and on the bf16 path when
            # a statistic has few samples (cheap there); otherwise the sums come from the conv epilogue.
            small = P <= 16384 or self.act_dt == F32
            nrep_fw = 1
            if norm == "batch" and mfma and not small:
                ntile = tiles_fn()
                part = self._alloc((ntile * 2 * cout,), F32)
                conv_into(y, 0, stats_part=part)
                self._emit(Lb.norm_reduce_partials, part.ptr, ntile, cout, sums.ptr, S)
            elif (norm == "batch" and mfma and small and not _DETERMINISTIC and not head1x1 and self.act_dt == BF16
                  and Lb.conv3x3_mfma_stats_atomic_supported(B, H, Wd, cin_eff, cout)):
                # few pixel tiles (the H <= 16 levels): the convolution adds its statistics straight into `sums` -- no pass over y
                conv_into(y, 0, stats_atomic=sums)
            elif (norm != "batch" and mfma and not head1x1 and self.act_dt == BF16 and H % 16 == 0 and Wd % 16 == 0
                  and tiles_fn() % B == 0):
                # group / instance norm on maps of at least 16 x 16: a pixel tile lies inside one sample, so the convolution's per-tile
                # sums reduce to per-sample sums without another pass over y (phx_norm_reduce_partials_ns)
                ntile = tiles_fn()
                part = self._alloc((ntile * 2 * cout,), F32)
                conv_into(y, 0, stats_part=part)
                self._emit(Lb.norm_reduce_partials_ns, part.ptr, ntile // B, B, cout, sums.ptr, S)
            elif norm == "batch" and not small and not _DETERMINISTIC:
                conv_into(y, 0, stats_direct=sums)        # (direct kernels add their tiles' sums atomically)
            else:
                pivot = self._alloc((NS * cout,), F32)
                conv_into(y, 0)
                self._emit(Lb.norm_stats, y.ptr, y.dt, sums.ptr, pivot.ptr, NS, P, cout, S)
            upd = norm == "batch" and training and self.loss is not None
            mmp = self.store.ptr(nv["moving_mean"]) if upd else None
            mvp = self.store.ptr(nv["moving_variance"]) if upd else None
            mom = (1.0 - tfnorm.BN_DECAY) if upd else 0.0
            apply_args = (y.ptr, y.dt, sums.ptr, nrep_fw, pivot.ptr if pivot is not None else None, gptr, beptr, eps, out.ptr, out.dt,
                          mean.ptr, rstd.ptr, scale.ptr, shift.ptr, mmp, mvp, mom, NS, P, cout, Gn, act)
            if True:
                hop = self._norm_head_consumer(op) if (y.dt == BF16 and out.dt == BF16) else None
                if hop is not None and Lb.norm_head_supported(cout, hop.attrs["W"].shape[-1], y.dt, out.dt):
                    # the head rides on the apply pass (phx_norm_apply_fused_head): no pass of its own over a
                    hW, hb = hop.attrs["W"], hop.attrs["b"]
                    yh = self._alloc_like(hop.outputs[0])
                    self.val[hop.outputs[0]] = yh
                    self._emit(Lb.norm_apply_fused_head, *apply_args, self.store.ptr(hW), self.store.ptr(hb), hW.shape[-1], yh.ptr, S,
                               tag="bytes_norm_apply", flops=float(y.nbytes + out.nbytes))
                    self._norm_head[hop] = op
                else:
                    self._emit(Lb.norm_apply_fused_rep, *apply_args, S, tag="bytes_norm_apply", flops=float(y.nbytes + out.nbytes))
        st.update(y=y, scale=scale, shift=shift, mean=mean, rstd=rstd, NS=NS, P=P, G=Gn)
        if norm != "batch":
            st.update(fsums=sums, fpivot=pivot)          # forward per-channel sums: the bias gradient is closed-form from them
        self.saved[op] = st

    def _fw_maxpool(self, op, bw):
        x = self.val[op.inputs[0]]
        out = self._alloc(self._cshape(op.outputs[0]), x.dt)
        self.val[op.outputs[0]] = out
        self._emit(self.L.maxpool2x2_fwd, x.ptr, x.dt, out.ptr, x.shape[0], x.shape[1], x.shape[2], x.shape[3], self.stream)

    def _fw_spatial_window(self, op, bw):
        x = self.val[op.inputs[0]]
        out = self._alloc(self._cshape(op.outputs[0]), x.dt)
        self.val[op.outputs[0]] = out
        oy, ox = op.attrs["off"]
        self._emit(self.L.spatial_window, x.ptr, out.ptr, x.dt, x.shape[0], x.shape[1], x.shape[2], out.shape[1], out.shape[2],
                   x.shape[3], oy, ox, self.stream)

    def _dropout_on(self, op):
        tr = op.attrs["training"]
        return (tr if isinstance(tr, bool) else self.training) and op.attrs["keep_prob"] < 1.0

    def _fw_dropout(self, op, bw):
        x = self.val[op.inputs[0]]
        if not self._dropout_on(op):
            self.val[op.outputs[0]] = x                      # inference: identity (layers.py:659-661)
            return
        out = self._alloc(x.shape, x.dt)
        self.val[op.outputs[0]] = out
        self._emit(self.L.dropout, x.ptr, out.ptr, x.dt, x.n // x.shape[0], x.shape[0], op.attrs["keep_prob"], self.rng_seed,
                   self._noise_step_ptr(), op.attrs["stream"], self.sample_offset, self.stream)

    def _fw_window4(self, op, bw):
        x = self.val[op.inputs[0]]
        out = self._alloc(self._cshape(op.outputs[0]), x.dt)
        self.val[op.outputs[0]] = out
        (sy, sx), (oy, ox, oc) = op.attrs["stride"], op.attrs["off"]
        self._emit(self.L.window4_fwd, x.ptr, out.ptr, x.dt, x.shape[0], x.shape[1], x.shape[2], x.shape[3], out.shape[1],
                   out.shape[2], out.shape[3], sy, sx, oy, ox, oc, self.stream)

    def _fw_add_act(self, op, bw):
        a, b = self.val[op.inputs[0]], self.val[op.inputs[1]]
        out = self._alloc(a.shape, a.dt)
        self.val[op.outputs[0]] = out
        b = self._as_dt(b, a.dt)
        self._emit(self.L.add_act, a.ptr, b.ptr, out.ptr, a.dt, a.n, rt.ACT_CODES[op.attrs["act"]], self.stream)

    def _fw_norm_act(self, op, bw):
        """Stand-alone act(normalisation(x)): statistics pass + fused apply (the generic path of the convolution units)."""
        a = op.attrs
        x = self.val[op.inputs[0]]
        out = self._alloc(x.shape, x.dt)
        self.val[op.outputs[0]] = out
        B, C = x.shape[0], x.shape[3]
        HW = x.shape[1] * x.shape[2]
        act = rt.ACT_CODES[a["act"]]
        norm = a["norm"]
        training = a["training"] if isinstance(a["training"], bool) else self.training
        S, Lb = self.stream, self.L
        if norm is None:
            ones = Buf((C,), F32, like=torch.ones(C, dtype=torch.float32, device=_device()))
            zeros = Buf((C,), F32, like=torch.zeros(C, dtype=torch.float32, device=_device()))
            self._keep += [ones, zeros]
            self._emit(Lb.affine_act, x.ptr, x.dt, ones.ptr, zeros.ptr, out.ptr, out.dt, 1, B * HW, C, act, S)
            self.saved[op] = dict(norm=None, out=out)
            return
        nv = a["norm_vars"]
        gptr, beptr = self.store.ptr(nv["gamma"]), self.store.ptr(nv["beta"])
        if norm == "batch":
            NS, P, Gn = 1, B * HW, C
        else:
            Gn = C if norm == "instance" else (a["num_groups"] or max(2, C // 16))
            NS, P = B, HW
        scale, shift = self._alloc((NS * C,), F32), self._alloc((NS * C,), F32)
        mean, rstd = self._alloc((NS * Gn,), F32), self._alloc((NS * Gn,), F32)
        eps = tfnorm.EPS[norm]
        st = dict(norm=norm, y=x, out=out, scale=scale, shift=shift, mean=mean, rstd=rstd, NS=NS, P=P, G=Gn)
        if norm == "batch" and not training:
            self._emit(Lb.bn_infer_scale_shift, gptr, beptr, self.store.ptr(nv["moving_mean"]),
                       self.store.ptr(nv["moving_variance"]), eps, C, scale.ptr, shift.ptr, S)
            self._emit(Lb.affine_act, x.ptr, x.dt, scale.ptr, shift.ptr, out.ptr, out.dt, NS, P, C, act, S)
            st["inference"] = True
        else:
            sums = self._alloc_zeroed(NS * C * 2)
            pivot = self._alloc((NS * C,), F32)
            self._emit(Lb.norm_stats, x.ptr, x.dt, sums.ptr, pivot.ptr, NS, P, C, S)
            upd = norm == "batch" and training and self.loss is not None
            self._emit(Lb.norm_apply_fused, x.ptr, x.dt, sums.ptr, pivot.ptr, gptr, beptr, eps, out.ptr, out.dt, mean.ptr, rstd.ptr,
                       scale.ptr, shift.ptr, self.store.ptr(nv["moving_mean"]) if upd else None,
                       self.store.ptr(nv["moving_variance"]) if upd else None, (1.0 - tfnorm.BN_DECAY) if upd else 0.0,
                       NS, P, C, Gn, act, S)
        self.saved[op] = st

    def _fw_flatten(self, op, bw):
        x = self.val[op.inputs[0]]
        self.val[op.outputs[0]] = Buf(self._cshape(op.outputs[0]), x.dt, like=x.t)      # same memory, new shape

    def _fw_avgpool(self, op, bw):
        x = self.val[op.inputs[0]]
        out = self._alloc(self._cshape(op.outputs[0]), x.dt)
        self.val[op.outputs[0]] = out
        self._emit(self.L.avgpool2x2_fwd, x.ptr, x.dt, out.ptr, x.shape[0], x.shape[1], x.shape[2], x.shape[3],
                   self.stream)

    def _fw_bilinear_up(self, op, bw):
        x = self.val[op.inputs[0]]
        out = self._alloc(self._cshape(op.outputs[0]), x.dt)
        self.val[op.outputs[0]] = out
        self._emit(self.L.bilinear_up2x_fwd, x.ptr, x.dt, out.ptr, x.shape[0], x.shape[1], x.shape[2], x.shape[3],
                   self.stream)

    def _fw_add(self, op, bw):
        mu_t, m = op.inputs
        if m.op.type != "mul" or m.op.inputs[1].op.type != "random_normal":
            raise NotImplementedError("only z = mu + sigma * random_normal(...) is on the hot path")
        rec = self._lat.get(op)
        if rec is not None:
            self.val[op.outputs[0]] = self._alloc(self.val[mu_t].shape, F32)
            self.saved[op] = dict(latent=True)
            if rec["last"] is op:
                self._fw_latent_group(rec)
            return
        sigma_t, eps_t = m.op.inputs
        mu, sigma = self.val[mu_t], self.val[sigma_t]
        z = self._alloc(mu.shape, F32)
        self.val[op.outputs[0]] = z
        per = mu.n // mu.shape[0]
        stream_id = eps_t.op.attrs["stream"]
        self._emit(self.L.reparam_fwd, mu.ptr, sigma.ptr, z.ptr, mu.shape[0], per, self.rng_seed,
                   self._noise_step_ptr(), stream_id, self.sample_offset, self.stream)
        self.saved[op] = dict(mu_t=mu_t, sigma_t=sigma_t, per=per, stream_id=stream_id)

    def _fw_tile_batch(self, op, bw):
        x = self.val[op.inputs[0]]
        out = self._alloc(self._cshape(op.outputs[0]), x.dt)
        self.val[op.outputs[0]] = out
        n = op.attrs["tile"]
        assert out.shape[0] == x.shape[0] * n
        self._emit(self.L.repeat_batch, x.ptr, out.ptr, x.shape[0], (x.n // x.shape[0]) * _ESIZE[x.dt], n, self.stream)

    def _bw_tile_batch(self, op):
        raise NotImplementedError("tile_batch is part of the sampling path only")

    def _fw_global_avgpool(self, op, bw):
        x = self._as_dt(self.val[op.inputs[0]], F32)
        out = self._alloc_like(op.outputs[0])
        self.val[op.outputs[0]] = out
        self._emit(self.L.global_avgpool_fwd, x.ptr, out.ptr, x.shape[0], x.shape[1] * x.shape[2], x.shape[3],
                   self.stream)

    def _fw_tile_pixels(self, op, bw):
        z = self.val[op.inputs[0]]
        out = self._alloc_like(op.outputs[0])
        self.val[op.outputs[0]] = out
        self._emit(self.L.broadcast_pixels_fwd, z.ptr, out.ptr, out.dt, out.shape[0], out.shape[1] * out.shape[2],
                   out.shape[3], self.stream)

    def _level_args(self, tensors):
        bufs = [self.val[t] for t in tensors]
        for b in bufs:
            assert b.dt == F32, "logit levels are fp32 heads"
        return bufs, rt.ptr_array([b.ptr for b in bufs]), rt.int_array([b.shift for b in bufs])

    def _fw_residual_ce(self, op, bw):
        Ls = op.attrs["L"]
        s_t, lab_t = op.inputs[:Ls], op.inputs[Ls]
        bufs, sp, shp = self._level_args(s_t)
        lab = self.val[lab_t]
        B, H, W = lab.shape
        C = bufs[0].shape[3]
        losses = self._alloc((8 + 512,), F32)
        s_out = self._alloc_like(op.outputs[Ls])
        self.val[op.outputs[Ls]] = s_out
        for l in range(Ls):
            v = Buf((), F32, like=losses.t[l:l + 1])
            self._keep.append(v)
            self.val[op.outputs[l]] = v
        dsp, dbufs, w = None, None, 0.0
        if bw:
            ws = [self.loss_weight.get(op.outputs[l], 0.0) for l in range(Ls)]
            assert all(abs(x - ws[0]) < 1e-12 for x in ws), "one weight for all residual-CE levels"
            w = ws[0]
            dbufs = []
            for b in bufs:
                if b.shape[1] != H:        # coarse levels are accumulated atomically -> zero every run
                    zb = self._alloc_zeroed(b.n)
                    zb.shape = b.shape
                    dbufs.append(zb)
                else:
                    dbufs.append(self._alloc(b.shape, F32))
            dsp = rt.ptr_array([b.ptr for b in dbufs])
            self.saved[op] = dict(dbufs=dbufs, src=[t.op.inputs[0] if t.op.type == "nn_resize" else t for t in s_t])
        self._emit(self.L.residual_ce, sp, dsp, shp, Ls, lab.ptr, B, H, W, C, w, self.inv_batch, losses.ptr,
                   s_out.ptr, None, self.stream)

    def _fw_aggregate(self, op, bw):
        Ls = op.attrs["L"]
        bufs, sp, shp = self._level_args(op.inputs)
        s_out, sm = self._alloc_like(op.outputs[0]), self._alloc_like(op.outputs[1])
        self.val[op.outputs[0]], self.val[op.outputs[1]] = s_out, sm
        B, H, W, C = s_out.shape
        self._emit(self.L.residual_ce, sp, None, shp, Ls, None, B, H, W, C, 0.0, 1.0, None, s_out.ptr, sm.ptr,
                   self.stream)

    def _fw_kl(self, op, bw):
        mu0, s0, mu1, s1 = [self.val[t] for t in op.inputs]
        grp = self._kl_group
        if grp is not None and op in grp["ops"]:
            # every level of the hierarchical KL term in ONE launch, emitted at the last level's operator (phx_kl_diag_gauss_multi);
            # the loss scalars live in the per-step zero arena (accumulated atomically: no memset node per level)
            loss = self._alloc_zeroed(1)
            loss.shape = ()
            self.val[op.outputs[0]] = loss
            gs = [self._alloc(mu0.shape, F32) for _ in range(4)] if bw else [None] * 4
            if bw:
                self.saved[op] = dict(gs=gs)
            grp["recs"].append((mu0, s0, mu1, s1, gs, loss, op.attrs["level_weight"]))
            if op is grp["ops"][-1]:
                recs = grp["recs"]
                ptrs = rt.ptr_array([p for r in recs for p in ([r[0].ptr, r[1].ptr, r[2].ptr, r[3].ptr] +
                                                               [g.ptr if g is not None else None for g in r[4]] + [r[5].ptr])])
                ns = (ctypes.c_size_t * len(recs))(*[r[0].n for r in recs])
                lws = (ctypes.c_float * len(recs))(*[r[6] for r in recs])
                self._keep += [ptrs, ns, lws]
                self._emit(self.L.kl_diag_gauss_multi, ptrs, ctypes.cast(ns, ctypes.c_void_p), ctypes.cast(lws, ctypes.c_void_p), len(recs),
                           self.inv_batch, grp["gscale"] if bw else 0.0, self.stream)
            return
        loss = self._alloc((), F32)
        self.val[op.outputs[0]] = loss
        gs = [None] * 4
        gscale = 0.0
        if bw:
            gscale = self.loss_weight.get(op.outputs[0], 0.0)
            gs = [self._alloc(mu0.shape, F32) for _ in range(4)]
            self.saved[op] = dict(gs=gs)
        self._emit(self.L.kl_diag_gauss, mu0.ptr, s0.ptr, mu1.ptr, s1.ptr, mu0.n, op.attrs["level_weight"],
                   self.inv_batch, gscale, loss.ptr, *[g.ptr if g is not None else None for g in gs], self.stream)

    def _fw_weighted_sum(self, op, bw):
        out = self._alloc((), F32)
        self.val[op.outputs[0]] = out
        ptrs = rt.ptr_array([self.val[t].ptr for t in op.inputs])
        ws = (ctypes.c_float * len(op.inputs))(*op.attrs["weights"])
        self._keep.append(ws)
        self._emit(self.L.weighted_sum, ptrs, ctypes.cast(ws, ctypes.c_void_p), len(op.inputs), out.ptr, self.stream)

    # ---- backward -------------------------------------------------------------------------------
    def _add_grad(self, t, write_fn=None, buf=None, accum_fn=None):
        """Accumulate a gradient contribution for tensor t: either `buf` (already complete) or produced by
        write_fn(target).  The first contribution owns the buffer; later ones are added in place -- by accum_fn(owner buffer) when
        the contributing kernel has an accumulating form (no buffer of its own, no add pass), else by phx_add_inplace."""
        if not self.req.get(t, False):
            return
        if accum_fn is not None and t in self.grad:
            g = self.grad[t]
            own = [evl for b, evl in self.pending.get(t, []) if b is g]
            # (only behind contributions of THIS lane: waiting here for another lane's write would tie the two backward chains
            # together early -- measured 3 % slower than leaving that contribution in a buffer of its own for the finaliser)
            if g.dt == self.val[t].dt and own and all(evl is None or evl[1] == self._lane for evl in own):
                accum_fn(g)
                evl = self._record(self._lane) if len(self._lanes) > 1 else None
                self.pending[t].append((g, evl))             # (same buffer: the finaliser only waits for it)
                return
        if buf is None:
            buf = self._alloc(self.val[t].shape, self.val[t].dt)
            write_fn(buf)
        if t not in self.grad:
            self.grad[t] = buf
        # the producer's backward (possibly on another lane) folds this contribution in: _finalize_grad
        evl = self._record(self._lane) if len(self._lanes) > 1 else None
        self.pending.setdefault(t, []).append((buf, evl))

    _VIRTUAL = ("one_hot", "sub_const", "random_normal", "mul", "nn_resize")

    def _real_producer(self, t):
        """Producer op whose launches create the data behind tensor t (looks through launch-less view ops)."""
        op = t.op
        while op is not None and op.type in self._VIRTUAL and op.inputs:
            op = op.inputs[0].op
        return op

    def _real_consumers(self, t, opset):
        out = []
        for c in t.consumers:
            if c not in opset:
                continue
            if c.type in self._VIRTUAL:
                for o in c.outputs:
                    out.extend(self._real_consumers(o, opset))
            else:
                out.append(c)
        return out

    def _finalize_grad(self, t):
        """Called on the producer's lane before its backward: wait for every contribution to grad[t] (they were
        written on the consumers' lanes) and fold the late ones into the primary buffer."""
        for buf, evl in self.pending.pop(t, []):
            self._wait(evl)
            g = self.grad[t]
            if buf is not g:
                assert g.dt == buf.dt and g.n == buf.n
                self._emit(self.L.add_inplace, g.ptr, buf.ptr, g.n, g.dt, self.stream)

    def _bw_placeholder(self, op):
        pass

    _bw_one_hot = _bw_sub_const = _bw_random_normal = _bw_mul = _bw_weighted_sum = _bw_aggregate = _bw_constant = _bw_placeholder

    def _bw_nn_resize(self, op):
        raise NotImplementedError("nearest-resized logits only feed the fused loss kernel")

    def _bw_residual_ce(self, op):
        sv = self.saved.get(op)
        if sv:
            for t, d in zip(sv["src"], sv["dbufs"]):
                self._add_grad(t, buf=d)

    def _bw_kl(self, op):
        sv = self.saved.get(op)
        if sv:
            for t, g in zip(op.inputs, sv["gs"]):
                self._add_grad(t, buf=g)

    def _bw_add(self, op):
        if op in self._lat:
            return self._bw_latent_group(self._lat[op])
        sv, dz = self.saved[op], self.grad[op.outputs[0]]
        self._add_grad(sv["mu_t"], buf=dz)
        B = dz.shape[0]

        def wr(target):
            self._emit(self.L.reparam_bwd, dz.ptr, target.ptr, B, sv["per"], self.rng_seed,
                       self._noise_step_ptr(), sv["stream_id"], self.sample_offset, self.stream)
        self._add_grad(sv["sigma_t"], write_fn=wr)

    def _bw_concat(self, op):
        a, b = op.inputs
        d = self.grad[op.outputs[0]]
        if b.op.type == "sub_const":
            return                       # posterior input: x and s are data
        npix = int(np.prod(d.shape[:-1]))
        ca, cb = self.val[a].shape[-1], self.val[b].shape[-1]
        da = self._alloc(self.val[a].shape, d.dt) if self.req.get(a) else None
        db = self._alloc(self.val[b].shape, d.dt) if self.req.get(b) else None
        self._emit(self.L.split2, d.ptr, da.ptr if da else None, ca, db.ptr if db else None, cb, npix, d.dt,
                   self.stream)
        for t, g in ((a, da), (b, db)):
            if g is not None:
                self._add_grad(t, buf=self._as_dt(g, self.val[t].dt))

    def _bw_maxpool(self, op):
        x, d = self.val[op.inputs[0]], self.grad[op.outputs[0]]
        self._add_grad(op.inputs[0], write_fn=lambda g: self._emit(
            self.L.maxpool2x2_bwd, x.ptr, d.ptr, d.dt, g.ptr, x.shape[0], x.shape[1], x.shape[2], x.shape[3], self.stream))

    def _bw_spatial_window(self, op):
        x, d = self.val[op.inputs[0]], self.grad[op.outputs[0]]
        oy, ox = op.attrs["off"]
        self._add_grad(op.inputs[0], write_fn=lambda g: self._emit(
            self.L.spatial_window, d.ptr, g.ptr, d.dt, x.shape[0], d.shape[1], d.shape[2], x.shape[1], x.shape[2], x.shape[3],
            -oy, -ox, self.stream))

    def _bw_dropout(self, op):
        d = self.grad[op.outputs[0]]
        if not self._dropout_on(op):
            self._add_grad(op.inputs[0], buf=d)
            return
        self._add_grad(op.inputs[0], write_fn=lambda g: self._emit(
            self.L.dropout, d.ptr, g.ptr, d.dt, d.n // d.shape[0], d.shape[0], op.attrs["keep_prob"], self.rng_seed,
            self._noise_step_ptr(), op.attrs["stream"], self.sample_offset, self.stream))

    def _bw_window4(self, op):
        x, d = self.val[op.inputs[0]], self.grad[op.outputs[0]]
        (sy, sx), (oy, ox, oc) = op.attrs["stride"], op.attrs["off"]
        self._add_grad(op.inputs[0], write_fn=lambda g: self._emit(
            self.L.window4_bwd, d.ptr, g.ptr, d.dt, x.shape[0], x.shape[1], x.shape[2], x.shape[3], d.shape[1], d.shape[2],
            d.shape[3], sy, sx, oy, ox, oc, self.stream))

    def _bw_add_act(self, op):
        d, out = self.grad[op.outputs[0]], self.val[op.outputs[0]]
        act = rt.ACT_CODES[op.attrs["act"]]
        for t in op.inputs:                                   # (one buffer per input: later contributions are added in place)
            if act != rt.ACT_ID:
                self._add_grad(t, write_fn=lambda g: self._emit(self.L.act_bwd, d.ptr, d.dt, out.ptr, out.dt, g.ptr, g.dt, d.n, act,
                                                                self.stream))
            else:
                self._add_grad(t, write_fn=lambda g: self._emit(self.L.memcpy_d2d, g.ptr, d.ptr, d.nbytes, self.stream))

    def _bw_norm_act(self, op):
        a, sv = op.attrs, self.saved[op]
        dA = self.grad[op.outputs[0]]
        act = rt.ACT_CODES[a["act"]]
        S, Lb = self.stream, self.L
        if sv["norm"] is None:
            out = sv["out"]
            self._add_grad(op.inputs[0], write_fn=lambda g: self._emit(Lb.act_bwd, dA.ptr, dA.dt, out.ptr, out.dt, g.ptr, g.dt, dA.n,
                                                                       act, S))
            return
        if sv.get("inference"):
            raise NotImplementedError("backward through inference-mode batch norm is not on the hot path")
        nv = a["norm_vars"]
        y, NS, P, Gn = sv["y"], sv["NS"], sv["P"], sv["G"]
        C = y.shape[3]
        nrep = _NREP if P >= _NREP_MINP else 1
        sums2 = self._alloc_zeroed(nrep * NS * C * 2)

        def wr(g):
            self._emit(Lb.norm_bwd_reduce, dA.ptr, dA.dt, y.ptr, y.dt, sv["scale"].ptr, sv["shift"].ptr, sv["mean"].ptr,
                       sv["rstd"].ptr, sums2.ptr, NS, P, C, Gn, act, nrep, S)
            self._emit(Lb.norm_bwd_apply_fused, dA.ptr, dA.dt, y.ptr, y.dt, sv["scale"].ptr, sv["shift"].ptr, sv["mean"].ptr,
                       sv["rstd"].ptr, self.store.ptr(nv["gamma"]), sums2.ptr, g.ptr, g.dt, self.store.grad_ptr(nv["gamma"]),
                       self.store.grad_ptr(nv["beta"]), NS, P, C, Gn, act, nrep, S)
        self._add_grad(op.inputs[0], write_fn=wr)

    def _bw_flatten(self, op):
        x, d = self.val[op.inputs[0]], self.grad[op.outputs[0]]
        self._add_grad(op.inputs[0], buf=Buf(x.shape, d.dt, like=d.t))

    def _bw_avgpool(self, op):
        x, d = self.val[op.inputs[0]], self.grad[op.outputs[0]]
        self._add_grad(op.inputs[0], write_fn=lambda g: self._emit(
            self.L.avgpool2x2_bwd, d.ptr, d.dt, g.ptr, x.shape[0], x.shape[1], x.shape[2], x.shape[3], self.stream),
            accum_fn=(lambda g: self._emit(self.L.avgpool2x2_bwd_acc, d.ptr, d.dt, g.ptr, x.shape[0], x.shape[1], x.shape[2], x.shape[3],
                                           self.stream)) if d.dt == x.dt else None)

    def _bw_bilinear_up(self, op):
        x, d = self.val[op.inputs[0]], self.grad[op.outputs[0]]
        self._add_grad(op.inputs[0], write_fn=lambda g: self._emit(
            self.L.bilinear_up2x_bwd, d.ptr, d.dt, g.ptr, x.shape[0], x.shape[1], x.shape[2], x.shape[3], self.stream),
            accum_fn=(lambda g: self._emit(self.L.bilinear_up2x_bwd_acc, d.ptr, d.dt, g.ptr, x.shape[0], x.shape[1], x.shape[2],
                                           x.shape[3], self.stream)) if d.dt == x.dt else None)

    def _bw_global_avgpool(self, op):
        x, d = self.val[op.inputs[0]], self.grad[op.outputs[0]]
        self._add_grad(op.inputs[0], write_fn=lambda g: self._emit(
            self.L.global_avgpool_bwd, d.ptr, g.ptr, x.shape[0], x.shape[1] * x.shape[2], x.shape[3], self.stream))

    def _bw_tile_pixels(self, op):
        d = self.grad[op.outputs[0]]
        self._add_grad(op.inputs[0], write_fn=lambda g: self._emit(
            self.L.broadcast_pixels_bwd, d.ptr, d.dt, g.ptr, d.shape[0], d.shape[1] * d.shape[2], d.shape[3],
            self.stream))

    def _bw_conv_unit(self, op):
        if op in self._lat:
            return self._bw_latent_group(self._lat[op])
        a, sv = op.attrs, self.saved[op]
        dA = self.grad[op.outputs[0]]
        x, out = sv["x"], sv["out"]
        W, b = a["W"], a["b"]
        k, cin, cout = a["ksize"], W.shape[-2], W.shape[-1]
        if sv.get("transposed") is not None:
            cout, cin = W.shape[2], W.shape[3]
        B, H, Wd = x.shape[0], x.shape[1], x.shape[2]
        act = rt.ACT_CODES[a["act"]]
        S, Lb = self.stream, self.L
        db_done = False
        if sv["norm"] is not None:
            if "y" not in sv or "mean" not in sv or (sv["norm"] == "batch" and not self.training):
                raise NotImplementedError("backward through inference-mode batch norm is not on the hot path")
            nv = a["norm_vars"]
            y, NS, P, Gn = sv["y"], sv["NS"], sv["P"], sv["G"]
            if sv.get("bn_small") and dA.dt == BF16:
                dY = self._alloc(y.shape, BF16)
                self._emit(Lb.bn_small_bwd, dA.ptr, y.ptr, y.dt, sv["scale"].ptr, sv["shift"].ptr, sv["mean"].ptr, sv["rstd"].ptr,
                           self.store.ptr(nv["gamma"]), dY.ptr, self.store.grad_ptr(nv["gamma"]),
                           self.store.grad_ptr(nv["beta"]), P, cout, act, S,
                           tag="bytes_norm_bwd_apply", flops=float(dA.nbytes + y.nbytes + dY.nbytes))
            elif sv.get("norm_small") and dA.dt == BF16:
                dY = self._alloc(y.shape, y.dt)
                self._emit(Lb.norm_small_bwd, dA.ptr, y.ptr, sv["scale"].ptr, sv["shift"].ptr, sv["mean"].ptr, sv["rstd"].ptr,
                           self.store.ptr(nv["gamma"]), dY.ptr, self.store.grad_ptr(nv["gamma"]),
                           self.store.grad_ptr(nv["beta"]), self.store.grad_ptr(b) if b is not None else None,
                           NS, P, cout, Gn, act, S,
                           tag="bytes_norm_bwd_apply", flops=float(dA.nbytes + y.nbytes + dY.nbytes))
                db_done = True
            else:
                nrep = _NREP if P >= _NREP_MINP else 1   # replicated accumulators: see k_norm_bwd_reduce
                if _DETERMINISTIC and P >= _NREP_MINP:
                    nrep = 64                             # one block per replica there: more replicas = more blocks
                sums2 = self._alloc_zeroed(nrep * NS * cout * 2)
                Sg = self._alloc((NS * Gn * 2,), F32)
                dY = self._alloc(y.shape, y.dt)
                hg = dA if isinstance(dA, HeadGrad) else None
                if hg is not None:
                    self._emit(Lb.norm_bwd_reduce_head, hg.dy.ptr, hg.w_ptr, hg.nout, y.ptr, sv["scale"].ptr, sv["shift"].ptr,
                               sv["mean"].ptr, sv["rstd"].ptr, sums2.ptr, NS, P, cout, Gn, act, nrep, S,
                               tag="bytes_norm_bwd_reduce", flops=float(y.nbytes))
                else:
                    self._emit(Lb.norm_bwd_reduce, dA.ptr, dA.dt, y.ptr, y.dt, sv["scale"].ptr, sv["shift"].ptr,
                               sv["mean"].ptr, sv["rstd"].ptr, sums2.ptr, NS, P, cout, Gn, act, nrep, S,
                               tag="bytes_norm_bwd_reduce", flops=float(dA.nbytes + y.nbytes))
                # group / instance norm keep the convolution bias: its gradient (the per-channel sum of dY) comes out of this
                # launch in closed form instead of a pass over dY (phx_norm_bwd_apply_fused_bias)
                fs = sv.get("fsums") if b is not None else None
                if fs is not None:
                    db_done = True
                if hg is not None:
                    self._emit(Lb.norm_bwd_apply_fused_head, hg.dy.ptr, hg.w_ptr, hg.nout, y.ptr, sv["scale"].ptr, sv["shift"].ptr,
                               sv["mean"].ptr, sv["rstd"].ptr, self.store.ptr(nv["gamma"]), sums2.ptr, dY.ptr,
                               self.store.grad_ptr(nv["gamma"]), self.store.grad_ptr(nv["beta"]),
                               fs.ptr if fs is not None else None,
                               sv["fpivot"].ptr if (fs is not None and sv.get("fpivot") is not None) else None,
                               self.store.grad_ptr(b) if fs is not None else None, NS, P, cout, Gn, act, nrep, S,
                               tag="bytes_norm_bwd_apply", flops=float(y.nbytes + dY.nbytes))
                else:
                    self._emit(Lb.norm_bwd_apply_fused_bias, dA.ptr, dA.dt, y.ptr, y.dt, sv["scale"].ptr, sv["shift"].ptr,
                               sv["mean"].ptr, sv["rstd"].ptr, self.store.ptr(nv["gamma"]), sums2.ptr, dY.ptr, dY.dt,
                               self.store.grad_ptr(nv["gamma"]), self.store.grad_ptr(nv["beta"]),
                               fs.ptr if fs is not None else None,
                               sv["fpivot"].ptr if (fs is not None and sv.get("fpivot") is not None) else None,
                               self.store.grad_ptr(b) if fs is not None else None, NS, P, cout, Gn, act, nrep, S,
                               tag="bytes_norm_bwd_apply", flops=float(dA.nbytes + y.nbytes + dY.nbytes))
        elif act != rt.ACT_ID:
            dY = self._alloc(out.shape, dA.dt)
            self._emit(Lb.act_bwd, dA.ptr, dA.dt, out.ptr, out.dt, dY.ptr, dY.dt, dA.n, act, S)
        else:
            dY = dA
        dw = self.store.grad_ptr(W)
        db = self.store.grad_ptr(b) if (b is not None and not db_done) else None
        if sv.get("general") is not None:
            geo = sv["geo"]
            self._emit(Lb.gconv2d_wgrad, x.ptr, x.dt, dY.ptr, dY.dt, dw, *geo, S)
            if db is not None:
                self._emit(Lb.channel_sum_accumulate, dY.ptr, dY.dt, db, dY.n // cout, cout, S)
            xin = op.inputs[0]
            if self.req.get(xin, False):
                self._add_grad(xin, write_fn=lambda g: self._emit(Lb.gconv2d_dgrad, dY.ptr, dY.dt, self.store.ptr(W), g.ptr, g.dt,
                                                                   *geo, S))
            return
        if sv.get("transposed") is not None:
            kh, kw, sh, sw = sv["transposed"]
            geo = (B, H, Wd, cin, cout, kh, kw, sh, sw)
            self._emit(Lb.tconv2d_wgrad, x.ptr, x.dt, dY.ptr, dY.dt, dw, *geo, S)
            if db is not None:
                self._emit(Lb.channel_sum_accumulate, dY.ptr, dY.dt, db, dY.n // cout, cout, S)
            xin = op.inputs[0]
            if self.req.get(xin, False):
                self._add_grad(xin, write_fn=lambda g: self._emit(Lb.tconv2d_dgrad, dY.ptr, dY.dt, self.store.ptr(W), g.ptr, g.dt,
                                                                   *geo, S))
            return
        # (The filter gradient is a leaf of the backward graph; moving these launches to another lane, beside the data-
        # gradient chain, was measured 20 % SLOWER: both are bound by the same global->LDS path, so the kernel on the
        # critical path just gets half of it.)
        if sv.get("head1x1") and cin % 8 == 0 and db is not None:
            # a leaf of the backward graph: all heads share one launch after the lanes have joined (phx_head1x1_wgrad_multi)
            plan4 = (ctypes.c_int * 4)()
            Lb.head1x1_wgrad_plan(B * H * Wd, cin, cout, plan4)
            self._headw_jobs.setdefault((x.dt, cout), []).append((x.ptr, dY.ptr, dw, db, B * H * Wd, cin, plan4[0], plan4[1],
                                                                   plan4[2], plan4[3]))
        elif sv.get("head1x1"):
            self._emit(Lb.head1x1_wgrad, x.ptr, x.dt, dY.ptr, dw, db, B * H * Wd, cin, cout, S)
        elif sv.get("padded") or sv["mfma"]:
            # padded layers (zero-padded input channels / 1x1 as centre tap): the gradient goes to a padded filter buffer
            # first and a small kernel folds it into dw afterwards
            padded = bool(sv.get("padded"))
            ce = sv["cin_eff"] if padded else cin
            tgt = self._alloc_zeroed(9 * ce * cout).ptr if padded else dw
            dual = x if isinstance(x, DualBuf) else None       # concat-free input: the filter gradient reads the two tensors in place
            k1d = dual.k1 if dual is not None else 0
            wsb = int(Lb.conv3x3_wgrad_ws_bytes_dual(B, H, Wd, ce, cout, k1d))
            wsp = self._alloc((wsb // 4,), F32)      # per-layer workspace of partial filters (no cross-lane sharing)
            plan6 = (ctypes.c_int * 6)()
            Lb.conv3x3_wgrad_reduce_plan_dual(B, H, Wd, ce, cout, k1d, plan6)
            rjob = (wsp.ptr, tgt, plan6[1], ce, cout, plan6[2], plan6[3], plan6[4], plan6[5])
            wargs = (x.ptr, dY.ptr, tgt, wsp.ptr, wsb, B, H, Wd, ce, cout)
            dargs = (x.ptr, dual.b.ptr if dual is not None else None, k1d) + wargs[1:]      # (x, x2, K1, dy, ...)
            wflops = 18.0 * cin * cout * B * H * Wd
            deferred = False
            if True:
                # The filter gradients are leaves of the backward graph.  Small and mid-size maps: the launch itself is
                # deferred -- one launch per kernel variant runs all such layers side by side after the lanes have joined
                # (phx_conv3x3_wgrad_multi); their latency leaves the posterior / prior / likelihood chains.
                nb = int(Lb.conv3x3_wgrad_multi_job_bytes())
                jb, info = ctypes.create_string_buffer(nb), (ctypes.c_int * 9)()
                Lb.conv3x3_wgrad_multi_job_dual(*dargs, _WGRAD_DEFER_BLOCKS, 0, jb, info)
                if info[0]:
                    grp = self._wgm_jobs.setdefault(int(info[0]), dict(recs=[], blocks=0, lds=0))
                    Lb.conv3x3_wgrad_multi_job_dual(*dargs, _WGRAD_DEFER_BLOCKS, grp["blocks"], jb, info)
                    grp["recs"].append(jb.raw)
                    grp["blocks"] += int(info[1])
                    grp["lds"] = max(grp["lds"], int(info[2]))
                    if info[3]:
                        self._wgr_jobs.append((wsp.ptr, tgt, info[4], ce, cout, info[5], info[6], info[7], info[8]))
                    deferred = True
            if deferred:
                pass
            elif plan6[0]:
                # large maps: the launch stays here, only the sum over its partial filters is deferred to ONE launch for all
                # layers (phx_wgrad_reduce_multi)
                if dual is not None:
                    self._emit(Lb.conv3x3_wgrad_mfma_bf16_dual, *dargs, 0, S, tag="conv3x3_mfma_wgrad", flops=wflops)
                else:
                    self._emit(Lb.conv3x3_wgrad_mfma_bf16_partial, *wargs, S, tag="conv3x3_mfma_wgrad", flops=wflops)
                self._wgr_jobs.append(rjob)
                deferred = True
            elif dual is not None:
                self._emit(Lb.conv3x3_wgrad_mfma_bf16_dual, *dargs, 1, S, tag="conv3x3_mfma_wgrad", flops=wflops)
            else:
                self._emit(Lb.conv3x3_wgrad_mfma_bf16, *wargs, S, tag="conv3x3_mfma_wgrad", flops=wflops)
            if padded:
                unpad = (Lb.unpad_filter_grad_center if sv.get("k1") else Lb.unpad_filter_grad_accumulate, (tgt, dw, cin, ce, cout))
                if deferred:
                    self._tail_jobs.append(unpad)             # after the deferred launches, on lane 0
                else:
                    self._emit(unpad[0], *unpad[1], S)
            if db is not None:
                self._emit(Lb.channel_sum_accumulate, dY.ptr, dY.dt, db, B * H * Wd, cout, S)
        else:
            if _DETERMINISTIC:
                # ordered partial filters: the fixed summation order at full parallelism (the plain entry point's deterministic
                # launch is one block per channel block -- 0.47 s instead of 0.1 s per fp32 training step at n0 = 32, batch 12)
                wsb = int(Lb.conv2d_direct_wgrad_ordered_ws_bytes(B, H, Wd, cin, cout, k))
                ws = self._alloc((wsb // 4,), F32) if wsb else None
                self._emit(Lb.conv2d_direct_wgrad_ordered, x.ptr, x.dt, dY.ptr, dY.dt, dw, db, ws.ptr if ws is not None else None, wsb,
                           B, H, Wd, cin, cout, k, S)
            else:
                self._emit(Lb.conv2d_direct_wgrad, x.ptr, x.dt, dY.ptr, dY.dt, dw, db, B, H, Wd, cin, cout, k, S)
        xin = op.inputs[0]
        if isinstance(x, DualBuf) and self.req.get(xin, False):
            # concat-free: the two halves of d(concat) are written straight to the gradients of the concatenated tensors
            ta, tb = xin.op.inputs
            _, wd = self._packed(W)
            g1, g2 = self._alloc(x.a.shape, BF16), self._alloc(x.b.shape, BF16)
            wsb = int(Lb.conv3x3_mfma_ws_bytes(B, H, Wd, cout, cin))
            ws = self._alloc((wsb // 4,), F32) if wsb else None      # split-K slices (small maps)
            self._emit(Lb.conv3x3_mfma_bf16_dual, dY.ptr, None, 0, wd.ptr, g1.ptr, g2.ptr, x.k1, None, None, 0, None, 0,
                       ws.ptr if ws else None, wsb, B, H, Wd, cout, cin, S,
                       tag="conv3x3_mfma_dgrad", flops=18.0 * cin * cout * B * H * Wd)
            for t, gb in ((ta, g1), (tb, g2)):
                if self.req.get(t, False):
                    self._add_grad(t, buf=gb)
        elif self.req.get(xin, False):
            if sv.get("norm_head"):              # no data-gradient launch: the producer's norm backward forms dA = dY W^T itself
                self._add_grad(xin, buf=HeadGrad(self.val[xin], dY, self.store.ptr(W), cout))
            elif sv.get("head1x1"):
                self._add_grad(xin, write_fn=lambda g: self._emit(
                    Lb.head1x1_dgrad, dY.ptr, self.store.ptr(W), g.ptr, g.dt, B * H * Wd, cin, cout, S))
            elif sv.get("padded"):
                ce, wdp = sv["cin_eff"], sv["wd_pad"]

                def wr(g):
                    gp = self._alloc((B, H, Wd, ce), BF16)
                    self._emit(Lb.conv3x3_mfma_bf16, dY.ptr, wdp.ptr, gp.ptr, None, 0, None, B, H, Wd, cout, ce, S,
                               tag="conv3x3_mfma_dgrad", flops=18.0 * cin * cout * B * H * Wd)
                    self._emit(Lb.unpad_channels_bf16, gp.ptr, g.ptr, g.dt, cin, ce, B * H * Wd, S)
                if ce == cin and self.val[xin].dt == BF16:      # nothing to strip / cast: the data gradient is written in place
                    self._add_grad(xin, write_fn=lambda g: self._emit(
                        Lb.conv3x3_mfma_bf16, dY.ptr, wdp.ptr, g.ptr, None, 0, None, B, H, Wd, cout, ce, S,
                        tag="conv3x3_mfma_dgrad", flops=18.0 * cin * cout * B * H * Wd))
                else:
                    self._add_grad(xin, write_fn=wr)
            elif sv["mfma"]:
                _, wd = self._packed(W)

                def wr_mfma(g):
                    wsb = int(Lb.conv3x3_mfma_ws_bytes(B, H, Wd, cout, cin))
                    ws = self._alloc((wsb // 4,), F32) if wsb else None      # split-K slices (small maps)
                    self._emit(Lb.conv3x3_mfma_bf16_ws, dY.ptr, wd.ptr, g.ptr, None, 0, None, ws.ptr if ws else None, wsb,
                               B, H, Wd, cout, cin, S, tag="conv3x3_mfma_dgrad", flops=18.0 * cin * cout * B * H * Wd)
                self._add_grad(xin, write_fn=wr_mfma)
            else:
                self._add_grad(xin, write_fn=lambda g: self._emit(
                    Lb.conv2d_direct, dY.ptr, dY.dt, self.store.ptr(W), None, g.ptr, g.dt, B, H, Wd, cin, cout, k, 0,
                    1, None, S))

    # ---- execution ------------------------------------------------------------------------------
    def set_input(self, name, array):
        b = self.feeds[name]
        a = np.ascontiguousarray(array, dtype=_NP_DT[b.dt]).reshape(-1)
        assert a.size == b.n, "feed %s: got %d elements, plan expects %d" % (name, a.size, b.n)
        self.L.memcpy_h2d(b.ptr, a.ctypes.data, a.nbytes, self.stream)
        self.L.stream_sync(self.stream)

    def _run_list(self, lst):
        for fn, args in lst:
            fn(*args)

    def run_eager(self, opt=True):
        self._run_list(self.launches)
        if opt:
            self._run_list(self.opt_launches)

    def _capture(self, lst):
        self.L.stream_sync(self.stream)
        self.L.graph_begin_capture(self.stream)
        try:
            self._run_list(lst)
        finally:
            ge = ctypes.c_void_p()
            self.L.graph_end_capture(self.stream, ctypes.byref(ge))
        return ge

    def run(self, sync=False):
        """One step.  First call runs eagerly (sets kernel attributes, validates), the second captures the
        launch list into a hipGraph, later calls replay it."""
        if not self.use_hip_graph:
            self.run_eager()
        elif self._graph_exec is None:
            if not getattr(self, "_warm", False):
                self.run_eager()
                self._warm = True
            else:
                self._graph_exec = self._capture(self.launches)
                self.L.graph_launch(self._graph_exec, self.stream)
                if self.opt_launches:
                    self._graph_exec_opt = self._capture(self.opt_launches)
                    self.L.graph_launch(self._graph_exec_opt, self.stream)
        else:
            self.L.graph_launch(self._graph_exec, self.stream)
            if self._graph_exec_opt is not None:
                self.L.graph_launch(self._graph_exec_opt, self.stream)
        if sync:
            self.sync()

    def run_main(self):
        """Data-parallel use: forward+backward only (graph 1); run_opt() applies Adam after the all-reduce."""
        if not self.use_hip_graph or not getattr(self, "_warm", False):
            self._run_list(self.launches)
            self._warm = True
        else:
            if self._graph_exec is None:
                self._graph_exec = self._capture(self.launches)
            self.L.graph_launch(self._graph_exec, self.stream)

    def run_opt(self):
        if not self.use_hip_graph or not getattr(self, "_warm_opt", False):
            self._run_list(self.opt_launches)
            self._warm_opt = True
        else:
            if self._graph_exec_opt is None:
                self._graph_exec_opt = self._capture(self.opt_launches)
            self.L.graph_launch(self._graph_exec_opt, self.stream)

    def sync(self):
        self.L.stream_sync(self.stream)

    def kernel_launch_count(self):
        """Entries of the launch lists that call a launching entry point of the C ABI (kernels and fills): everything but the event
        records / waits between the lanes and the slots blanked by _prune_dead_event_records.  (A rocprofv3 kernel trace of one step
        counts a few more: a split-K convolution's call launches its finishing kernel as well.)"""
        skip = (_noop, self.L.event_record, self.L.stream_wait_event)
        return sum(1 for lst in (self.launches, self.opt_launches) for fn, _ in lst if not any(fn is f for f in skip))

    def stream_handle(self):
        """hipStream_t (as int) of the origin lane: graph launches are enqueued on it and every lane joins into it."""
        return int(self._lanes[0].value or 0)

    def time_tagged_kernels(self, repeats=3):
        """Per tagged launch (the bf16 MFMA convolutions): average GPU duration over `repeats` back-to-back
        re-launches, bracketed by HIP events on THIS plan's stream.  -> list of (tag, flops, ms, args-shape)."""
        ev0, ev1 = ctypes.c_void_p(), ctypes.c_void_p()
        self.L.event_create(ctypes.byref(ev0))
        self.L.event_create(ctypes.byref(ev1))
        out = []
        ms = ctypes.c_float()
        for idx, (fn, args) in enumerate(self.launches):
            tg = self.tags.get((id(self.launches), idx))
            if tg is None:
                continue
            st = args[-1]                               # the lane (HIP stream) this launch is enqueued on
            fn(*args)                                   # warm
            self.L.event_record(ev0, st)
            for _ in range(repeats):
                fn(*args)
            self.L.event_record(ev1, st)
            self.L.event_sync(ev1)
            self.L.event_elapsed_ms(ev0, ev1, ctypes.byref(ms))
            out.append((tg[0], tg[1], ms.value / repeats,
                        tg[2] if tg[2] is not None else tuple(a for a in args if isinstance(a, int) and a < 1 << 20)))
        self.L.event_destroy(ev0)
        self.L.event_destroy(ev1)
        return out

    def fetch(self, t):
        self.sync()
        b = self.val[t]
        a = b.numpy()
        if b.shift:                     # nearest-neighbour view (likelihoods.py:221): expand on the host
            f = 1 << b.shift
            a = np.repeat(np.repeat(a, f, axis=1), f, axis=2)
        return a

    def fetch_grad(self, t):
        self.sync()
        return self.grad[t].numpy()

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

from phiseg_code_amd.engine_common import *  # noqa: F401,F403  (dtype codes, schedule constants, Buf / DualBuf / HeadGrad)
from phiseg_code_amd.engine_common import (_BN_SMALL, _BN_SMALL_F32, _DETERMINISTIC, _NREP, _NREP_MINP, _fgn_mode, _dual_enabled, _noop,  # noqa: F401
                                           _device, _TORCH_DT, _NP_DT, _ESIZE, _LIK_SIDE_MAXLVL, _WGRAD_DEFER_BLOCKS, _STAMPS)
from phiseg_code_amd.engine_backward import BackwardLowering
from phiseg_code_amd.engine_forward import ForwardLowering


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


class Plan(ForwardLowering, BackwardLowering):
    """One compiled (fetches, loss) program for a fixed batch size / training flag / compute dtype."""

    def __init__(self, store, fetches, loss=None, batch=1, training=True, compute_dtype="f32", optimize=True,
                 rng_seed=42, sample_offset=0, loss_inv_batch=None, stream=None, use_hip_graph=True,
                 split_optimizer=False, n_lanes=None, stamp_tagged=False):
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
        # stamp_tagged: every tagged launch (the convolution family) sits between two wall-clock stamps on its lane -- bench.py's
        # measurement of the family's time INSIDE the replayed two-lane step (tagged_in_situ())
        self._stamp_tagged = bool(stamp_tagged)
        self.tag_stamps, self._tstamp_n = [], 16
        self._tstamp_buf = torch.zeros(8192, dtype=torch.int64, device=_device()) if self._stamp_tagged else None
        self.feeds = {}
        self._keep = []
        self._wpk = {}
        self._tail_jobs = []          # launches that follow the deferred ones on lane 0 (padded-filter folds)
        self._wgm_jobs = {}           # deferred small-map filter gradients by kernel variant: records, total blocks, LDS bytes
        self._headw_jobs = {}         # deferred 1x1-head filter gradients by (x dtype, nout): (x, dy, dw, db, npix, C, PL, chunk, grid, lds)
        self._wgr_jobs = []           # deferred filter-gradient reductions: (ws, dw, nslice, Cin, Cout, tci, tco, gx, gy)
        self._pack_jobs = []          # (w, wpk_fwd, wpk_dgrad, Cin, Cin_pad, Cout): ONE multi-filter pack launch per run
        self._barriers = []           # grid-barrier words of the one-launch kernels (zero arena): word 288 != 0 = that barrier timed out
        self._wpk32, self._pack32_jobs = {}, []     # fp32 plans: packed fp32 filters of the fp32 matrix kernels, [w, wpk_fwd, wpk_dgrad, Cin, Cout]
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
                                             ("PL", "<i4"), ("chunk", "<i4"), ("blk0", "<i4"), ("xscale", "<u8"), ("xshift", "<u8"),
                                             ("xact", "<i4"), ("pad", "<i4")])
            blk = lds = 0
            for i, j in enumerate(jobs):
                xf = j[10] if len(j) > 10 else (0, 0, 0)      # (scale ptr, shift ptr, act): x is a pre-normalisation tensor
                rec[i] = (j[0], j[1], j[2], j[3], j[4], j[5], j[6], j[7], blk, xf[0], xf[1], xf[2], 0)
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
        st = tag is not None and self._stamp_tagged and self._cur is self.launches and self._tstamp_n + 2 <= self._tstamp_buf.numel()
        if st:
            i = self._tstamp_n
            self._tstamp_n += 2
            self._cur.append((self.L.stamp, (self._tstamp_buf.data_ptr() + 8 * i, self.stream)))
        if tag is not None:
            self.tags[(id(self._cur), len(self._cur))] = (tag, float(flops), shape)
        self._cur.append((fn, args))
        if st:
            self._cur.append((self.L.stamp, (self._tstamp_buf.data_ptr() + 8 * (i + 1), self.stream)))
            self.tag_stamps.append((tag, float(flops), shape, i))
        self._lane_seq[self._lane] = self._lane_seq.get(self._lane, 0) + 1

    def tagged_in_situ(self):
        """stamp_tagged plans, after at least one replay: -> (rows, gap_us); rows = [(tag, flops, us, shape)] with us = the wall-clock
        time between the stamps around each tagged launch in the LAST replay minus gap_us, the median distance of back-to-back
        stamps (eight calibration pairs at the head of lane 0): the launch's duration plus one launch boundary, with whatever the
        other lane ran beside it."""
        self.sync()
        t = self._tstamp_buf.cpu().numpy().astype(np.int64)          # 100 MHz wall clock
        gap = float(np.median([(t[2 * k + 1] - t[2 * k]) / 100.0 for k in range(8)]))
        return [(tag, fl, max((t[i + 1] - t[i]) / 100.0 - gap, 0.0), shp) for tag, fl, shp, i in self.tag_stamps], gap

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
        if self._stamp_tagged:            # calibration: eight back-to-back stamp pairs
            for k in range(16):
                self._cur.append((self.L.stamp, (self._tstamp_buf.data_ptr() + 8 * k, self.stream)))
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
        self._pool_done = set()       # avgpool ops whose output the producer's apply pass wrote (phx_norm_apply_pool)
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
        if self._pack32_jobs:         # fp32 plans: the packed fp32 filters of csrc/conv_f32_mfma.hip, same slot (a plan packs one kind)
            assert not self._pack_jobs, "a plan packs either bf16 or fp32 filters"
            rec = np.zeros(len(self._pack32_jobs), dtype=[("w", "<u8"), ("wf", "<u8"), ("wd", "<u8"), ("cin", "<i4"), ("cout", "<i4")])
            for i, j in enumerate(self._pack32_jobs):
                rec[i] = tuple(j)
            self._pack_desc = torch.from_numpy(rec.view(np.uint8).copy()).to(_device())
            self._keep.append(self._pack_desc)
            self.launches[1] = (self.L.pack_conv3x3_f32_multi, (self._pack_desc.data_ptr(), len(self._pack32_jobs), self.stream))
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


    _VIRTUAL = ("one_hot", "sub_const", "random_normal", "mul", "nn_resize")




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

    def barrier_timeouts(self):
        """Number of in-kernel grid barriers of the LAST replay that gave up waiting (phx_bn_bwd_onepass: bounded spin).  0 on a healthy
        run; anything else means that launch's result was wrong -- callers that time or publish results check it."""
        self.sync()
        return sum(int(b.t[288].item() != 0) for b in self._barriers)

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

"""Reverse-mode backward pass of the plan (engine.Plan mixin): one `_bw_<op type>` method per operator type emits the launches that
turn the gradient of the operator's outputs into gradients of its inputs and variables -- what `optimizer.minimize` derives in the
reference (phiseg/phiseg_model.py:135-141) -- plus the gradient bookkeeping (`_add_grad`, `_finalize_grad`)."""
import ctypes
import os

import numpy as np
import torch

from phiseg_code_amd import graph as G
from phiseg_code_amd import runtime as rt
from phiseg_code_amd import upconv
from phiseg_code_amd.tfwrapper import normalisation as tfnorm
from phiseg_code_amd.engine_common import *  # noqa: F401,F403
from phiseg_code_amd.engine_common import _BN_SMALL, _BN_SMALL_F32, _BN_WIDE, _BN_WIDE_MAXLINES, _DETERMINISTIC, _NREP, _NREP_MINP, _fgn_mode, _dual_enabled, _noop, _device, _TORCH_DT, _NP_DT, _ESIZE, _LIK_SIDE_MAXLVL, _WGRAD_DEFER_BLOCKS, _STAMPS  # noqa: F401


class BackwardLowering:
    def _bw_l2_weights(self, op):
        st = self.store
        self._emit(self.L.axpy_masked, st.grads.data_ptr(), st.params.data_ptr(), st.decay_mask.data_ptr(), st.n_train,
                   float(self._l2_weight), self.stream)

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

    def _bw_tile_batch(self, op):
        raise NotImplementedError("tile_batch is part of the sampling path only")

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

    def _slice_grad_ok(self, op, xin, B, H, Wd, K, N):
        """May the data gradient of convolution `op` (reduction channels K = its Cout, N = its Cin) hand its split-K slices to the
        producer of its input?  -- that producer is a one-launch wide batch-norm layer, this convolution is the input's ONLY
        reader (so the slices are the whole gradient), on the same lane, nobody fetches the gradient, and the launch does run split-K."""
        if _BN_WIDE < 2 or xin in self.fetches or self.act_dt != BF16:
            return False
        prod = self._real_producer(xin)
        if prod is None or prod.type != "conv_unit" or not (self.saved.get(prod) or {}).get("bn_wide") or prod.outputs[0] is not xin:
            return False
        if self.op_lane.get(prod) != self.op_lane.get(op):
            return False
        cons = self._real_consumers(xin, self._opset)
        if len(cons) != 1 or cons[0] is not op or xin in self.grad or self.pending.get(xin):
            return False
        nzd = int(self.L.conv3x3_mfma_ksplit(B, H, Wd, K, N))
        return nzd > 1 and B * H * Wd * nzd <= _BN_WIDE_MAXLINES

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
        upc = sv.get("upconv")
        if upc is not None:
            # phase form (upconv.py): y and everything downstream of it live in the PACKED pixel order; dA is a hi-res map -- the two
            # norm-backward passes read it through the space-to-depth permutation
            assert isinstance(dA, Buf) and dA.dt == BF16 and not sv.get("bn_small") and not sv.get("norm_small")
        if sv["norm"] is not None:
            if "y" not in sv or "mean" not in sv or (sv["norm"] == "batch" and not self.training):
                raise NotImplementedError("backward through inference-mode batch norm is not on the hot path")
            nv = a["norm_vars"]
            y, NS, P, Gn = sv["y"], sv["NS"], sv["P"], sv["G"]
            if sv.get("bn_wide") and dA.dt == BF16:
                dY = self._alloc(y.shape, BF16)
                sg = dA if isinstance(dA, SliceGrad) else None      # the consumer's split-K data gradient left its slices: summed here
                self._emit(Lb.bn_wide_bwd, None if sg is not None else dA.ptr, sg.ws.ptr if sg is not None else None,
                           sg.nz if sg is not None else 0, y.ptr, sv["scale"].ptr, sv["shift"].ptr, sv["mean"].ptr, sv["rstd"].ptr,
                           self.store.ptr(nv["gamma"]), dY.ptr, self.store.grad_ptr(nv["gamma"]),
                           self.store.grad_ptr(nv["beta"]), P, cout, act, S,
                           tag="bytes_norm_bwd_apply", flops=float(dY.nbytes + y.nbytes + dY.nbytes))
            elif sv.get("bn_small") and dA.dt == BF16:
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
                fs0 = sv.get("fsums") if b is not None else None
                onepass = False
                if (upc is None and hg is None and fs0 is None and sv["norm"] == "batch" and NS == 1 and Gn == cout and isinstance(dA, Buf)
                        and dA.dt == BF16 and y.dt == BF16 and _onepass_enabled() and Lb.bn_bwd_onepass_supported(P, cout, act)):
                    # mid-size batch-norm layers: ONE launch, (dA, y) read once and held in registers across a grid barrier
                    bar = self._alloc_zeroed(int(Lb.bn_bwd_onepass_barrier_words()))
                    self._barriers.append(bar)
                    onepass = True
                    self._emit(Lb.bn_bwd_onepass, dA.ptr, y.ptr, sv["scale"].ptr, sv["shift"].ptr, sv["mean"].ptr, sv["rstd"].ptr,
                               self.store.ptr(nv["gamma"]), sums2.ptr, bar.ptr, dY.ptr, self.store.grad_ptr(nv["gamma"]),
                               self.store.grad_ptr(nv["beta"]), P, cout, act, nrep, S,
                               tag="bytes_norm_bwd_onepass", flops=float(dA.nbytes + y.nbytes + dY.nbytes))
                elif upc is not None:
                    self._emit(Lb.norm_bwd_reduce_s2d, dA.ptr, dA.dt, y.ptr, y.dt, sv["scale"].ptr, sv["shift"].ptr, sv["mean"].ptr,
                               sv["rstd"].ptr, sums2.ptr, NS, P, cout, Gn, act, nrep, H // 2, Wd // 2, S,
                               tag="bytes_norm_bwd_reduce", flops=float(dA.nbytes + y.nbytes))
                elif hg is not None:
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
                if onepass:
                    pass                                  # (the one launch above formed dY, dgamma and dbeta)
                elif upc is not None:
                    self._emit(Lb.norm_bwd_apply_fused_s2d, dA.ptr, dA.dt, y.ptr, y.dt, sv["scale"].ptr, sv["shift"].ptr, sv["mean"].ptr,
                               sv["rstd"].ptr, self.store.ptr(nv["gamma"]), sums2.ptr, dY.ptr, dY.dt, self.store.grad_ptr(nv["gamma"]),
                               self.store.grad_ptr(nv["beta"]), fs.ptr if fs is not None else None,
                               sv["fpivot"].ptr if (fs is not None and sv.get("fpivot") is not None) else None,
                               self.store.grad_ptr(b) if fs is not None else None, NS, P, cout, Gn, act, nrep, H // 2, Wd // 2, S,
                               tag="bytes_norm_bwd_apply", flops=float(dA.nbytes + y.nbytes + dY.nbytes))
                elif hg is not None:
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
        if upc is not None:
            src = x.src                                  # the low-resolution tensor bilinear_upsample2D read
            h, w = H // 2, Wd // 2
            if db is not None:                           # (not reached with the closed-form bias gradient of the norm backward; kept exact)
                self._emit(Lb.channel_sum_accumulate, dY.ptr, dY.dt, db, B * H * Wd, cout, S)
            upconv.backward_prepare(self._emit, self._alloc, Lb, S, upc, dY, B, h, w, cout)
            # (filter gradients before or after the data gradients: same step time, measured three alternating pairs)
            upconv.backward_filters(self._emit, self._alloc, self._alloc_zeroed, Lb, S, upc, src, dY, dw, B, h, w, cin, cout)
            xin = op.inputs[0].op.inputs[0]              # the gradient goes straight to the resize's input (its adjoint is part of the form)
            if self.req.get(xin, False):
                _, wd_w = self._packed(W)
                self._add_grad(xin, write_fn=lambda g: upconv.backward_data(self._emit, self._alloc, Lb, S, upc, dY, wd_w, g, B, h, w, cin, cout))
            return
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
            prod = self._norm_head.get(op)
            au = self.saved[prod].get("a_unwritten") if prod is not None else None
            if au is not None:      # the producer never wrote a = act(bn(y)): the job re-forms it from y (phx_head1x1_wgrad_multi, xscale)
                self._headw_jobs.setdefault((au["y"].dt, cout), []).append((au["y"].ptr, dY.ptr, dw, db, B * H * Wd, cin, plan4[0], plan4[1],
                                                                            plan4[2], plan4[3], (au["scale"].ptr, au["shift"].ptr, au["act"])))
            else:
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
            xf = x if isinstance(x, XfBuf) else None           # unmaterialised input activation: re-formed from y by the kernel's loader
            k1d = dual.k1 if dual is not None else 0
            wsb = int(Lb.conv3x3_wgrad_ws_bytes_dual(B, H, Wd, ce, cout, k1d))
            wsp = self._alloc((wsb // 4,), F32)      # per-layer workspace of partial filters (no cross-lane sharing)
            plan6 = (ctypes.c_int * 6)()
            Lb.conv3x3_wgrad_reduce_plan_dual(B, H, Wd, ce, cout, k1d, plan6)
            rjob = (wsp.ptr, tgt, plan6[1], ce, cout, plan6[2], plan6[3], plan6[4], plan6[5])
            if xf is None:
                wargs = (x.ptr, dY.ptr, tgt, wsp.ptr, wsb, B, H, Wd, ce, cout)
                dargs = (x.ptr, dual.b.ptr if dual is not None else None, k1d) + wargs[1:]      # (x, x2, K1, dy, ...)
            wflops = 18.0 * cin * cout * B * H * Wd
            deferred = False
            # The filter gradients are leaves of the backward graph.  Small and mid-size maps: the launch itself is
            # deferred -- one launch per kernel variant runs all such layers side by side after the lanes have joined
            # (phx_conv3x3_wgrad_multi); their latency leaves the posterior / prior / likelihood chains.
            nb = int(Lb.conv3x3_wgrad_multi_job_bytes())
            jb, info = ctypes.create_string_buffer(nb), (ctypes.c_int * 9)()
            if xf is None:       # (an unmaterialised input only exists on maps too large for the deferred launches: _xf_edge_ok)
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
                if xf is not None:
                    self._emit(Lb.conv3x3_wgrad_mfma_bf16_partial_xf, xf.y.ptr, xf.scale.ptr, xf.shift.ptr, dY.ptr, tgt, wsp.ptr, wsb,
                               B, H, Wd, ce, cout, S, tag="conv3x3_mfma_wgrad", flops=wflops)
                elif dual is not None:
                    self._emit(Lb.conv3x3_wgrad_mfma_bf16_dual, *dargs, 0, S, tag="conv3x3_mfma_wgrad", flops=wflops)
                else:
                    self._emit(Lb.conv3x3_wgrad_mfma_bf16_partial, *wargs, S, tag="conv3x3_mfma_wgrad", flops=wflops)
                self._wgr_jobs.append(rjob)
                deferred = True
            elif xf is not None:
                raise rt.PhxError("filter gradient of an unmaterialised input activation without a workspace plan (see _xf_edge_ok)")
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
        elif (sv.get("f32m") and k == 3 and x.dt == F32 and dY.dt == F32 and isinstance(x, Buf)
              and Lb.conv3x3_f32_mfma_wgrad_supported(B, H, Wd, cin, cout)):
            # fp32 plans: the filter gradient on the fp32 matrix instruction; partial filters in a workspace, summed in slice order
            # (a fixed order in every mode)
            wsb = int(Lb.conv3x3_f32_mfma_wgrad_ws_bytes(B, H, Wd, cin, cout, 1 if db is not None else 0))
            ws = self._alloc((wsb // 4,), F32)
            self._emit(Lb.conv3x3_f32_mfma_wgrad, x.ptr, dY.ptr, dw, db, ws.ptr, wsb, B, H, Wd, cin, cout, S,
                       tag="conv3x3_f32_mfma_wgrad", flops=18.0 * cin * cout * B * H * Wd)
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
            elif sv["mfma"] and self._slice_grad_ok(op, xin, B, H, Wd, cout, cin):
                # 2 x 2 / 4 x 4 levels: the split-K data gradient leaves its fp32 slices for the producer's one-launch batch-norm
                # backward (phx_bn_wide_bwd sums them): no finishing launch, no bf16 gradient tensor
                _, wd = self._packed(W)
                wsb = int(Lb.conv3x3_mfma_ws_bytes(B, H, Wd, cout, cin))
                ws = self._alloc((wsb // 4,), F32)
                self._emit(Lb.conv3x3_mfma_bf16_ws, dY.ptr, wd.ptr, None, None, 0, None, ws.ptr, wsb,
                           B, H, Wd, cout, cin, S, tag="conv3x3_mfma_dgrad", flops=18.0 * cin * cout * B * H * Wd)
                self._add_grad(xin, buf=SliceGrad(self.val[xin], ws, int(Lb.conv3x3_mfma_ksplit(B, H, Wd, cout, cin))))
            elif sv["mfma"]:
                _, wd = self._packed(W)

                def wr_mfma(g):
                    wsb = int(Lb.conv3x3_mfma_ws_bytes(B, H, Wd, cout, cin))
                    ws = self._alloc((wsb // 4,), F32) if wsb else None      # split-K slices (small maps)
                    self._emit(Lb.conv3x3_mfma_bf16_ws, dY.ptr, wd.ptr, g.ptr, None, 0, None, ws.ptr if ws else None, wsb,
                               B, H, Wd, cout, cin, S, tag="conv3x3_mfma_dgrad", flops=18.0 * cin * cout * B * H * Wd)
                self._add_grad(xin, write_fn=wr_mfma)
            elif sv.get("f32m") and cin % 32 == 0 and dY.dt == F32 and self._wpk32.get(W.name, (None, None))[1] is not None:
                wd32 = self._wpk32[W.name][1]

                def wr_f32m(g):
                    assert g.dt == F32
                    self._emit(Lb.conv3x3_f32_mfma, dY.ptr, wd32.ptr, None, g.ptr, B, H, Wd, cout, cin, 0, S,
                               tag="conv3x3_f32_mfma_dgrad", flops=18.0 * cin * cout * B * H * Wd)
                self._add_grad(xin, write_fn=wr_f32m)
            else:
                self._add_grad(xin, write_fn=lambda g: self._emit(
                    Lb.conv2d_direct, dY.ptr, dY.dt, self.store.ptr(W), None, g.ptr, g.dt, B, H, Wd, cin, cout, k, 0,
                    1, None, S))

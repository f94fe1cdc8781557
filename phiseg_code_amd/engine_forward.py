"""Forward lowering of the symbolic graph (engine.Plan mixin): one `_fw_<op type>` method per operator type emits the libphx launches
of that operator into the plan's launch list and records what the backward pass needs (`self.saved`).  Replaces the forward half of
what TensorFlow builds for `sess.run` in the reference (phiseg/phiseg_model.py:20-157, tfwrapper/layers.py, model_zoo/*.py)."""
import ctypes
import os

import numpy as np
import torch

from phiseg_code_amd import graph as G
from phiseg_code_amd import runtime as rt
from phiseg_code_amd import upconv
from phiseg_code_amd.tfwrapper import normalisation as tfnorm
from phiseg_code_amd.engine_common import *  # noqa: F401,F403
from phiseg_code_amd.engine_common import _BN_SMALL, _BN_SMALL_F32, _BN_WIDE, _BN_WIDE_MAXLINES, _SKIP_HEAD_A, _DETERMINISTIC, _NREP, _NREP_MINP, _fgn_mode, _dual_enabled, _noop, _device, _TORCH_DT, _NP_DT, _ESIZE, _LIK_SIDE_MAXLVL, _WGRAD_DEFER_BLOCKS, _STAMPS  # noqa: F401


class ForwardLowering:
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

    def _packed_f32(self, W, need_dgrad):
        """fp32 packed copies of a 3x3 filter for the fp32 matrix kernels (csrc/conv_f32_mfma.hip: [K8 / 8][9][N][8]), refreshed at the
        head of every run; the data-gradient copy only where a data gradient is taken."""
        kh, kw, cin, cout = W.shape
        rec = self._wpk32.get(W.name)
        if rec is None:
            wf = self._alloc((int(self.L.conv3x3_f32_mfma_packed_floats(cin, cout)),), F32)
            rec = self._wpk32[W.name] = [wf, None, len(self._pack32_jobs)]
            self._pack32_jobs.append([self.store.ptr(W), wf.ptr, 0, cin, cout])
        if need_dgrad and rec[1] is None:
            rec[1] = self._alloc((int(self.L.conv3x3_f32_mfma_packed_floats(cout, cin)),), F32)
            self._pack32_jobs[rec[2]][2] = rec[1].ptr
        return rec[0], rec[1]

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

    def _xf_edge_ok(self, op, B, H, Wd, C):
        """May conv unit `op` leave its activation a = relu(bn(y)) unwritten?  -- every real reader is a 3x3 conv unit with batch norm
        (no bias, statistics epilogue) that reads it directly (not through a concat), trains, and whose forward and filter-gradient
        kernels take the input transform at this shape (phx_conv3x3_xf_supported / phx_conv3x3_wgrad_xf_supported: the HBM-bound 32 -> 32
        layers of the 128 x 128 level -- on the matrix-bound shapes the transform costs more than the apply pass it deletes,
        DESIGN.md section 5); nobody fetches it."""
        out = op.outputs[0]
        if out in self.fetches:
            return False
        cons = self._real_consumers(out, self._opset)
        if not cons:
            return False
        for c in cons:
            ca = c.attrs
            if (c.type != "conv_unit" or c in self._lat or ca.get("transposed") is not None or ca.get("general") is not None
                    or ca["ksize"] != 3 or ca["norm"] != "batch" or ca["b"] is not None or c.inputs[0] is not out):
                return False
            tr = ca["training"] if isinstance(ca["training"], bool) else self.training
            cin, cout = ca["W"].shape[-2], ca["W"].shape[-1]
            if (not tr or cin != C or not self.L.conv3x3_xf_supported(B, H, Wd, cin, cout)
                    or not self.L.conv3x3_wgrad_xf_supported(B, H, Wd, cin, cout)):
                return False
        return True

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
        up = x if isinstance(x, UpBuf) else None      # bilinear_upsample2D of up.src, never written: this unit runs in the phase form (upconv.py)
        xf = x if isinstance(x, XfBuf) else None      # the producer's activation was never written: this launch re-forms it (see _xf_edge_ok)
        assert xf is None or (mfma and a["norm"] == "batch" and b is None), "unmaterialised activation reached a convolution that cannot re-form it"

        def mfma_conv(y, bias_p, oscale_p, act_code, stats, stats_mode, ws, wsb):
            """One forward launch on the bf16 MFMA path (plain or concat-free input): phx_conv3x3_mfma_bf16_dual takes every option"""
            if xf is not None:
                assert bias_p is None and oscale_p is None and act_code == 0 and stats_mode in (0, 1) and ws is None
                self._emit(Lb.conv3x3_mfma_bf16_xf, xf.y.ptr, xf.scale.ptr, xf.shift.ptr, wf.ptr, y.ptr,
                           stats.ptr if stats is not None else None, B, H, Wd, cin_eff, cout, S,
                           tag="conv3x3_mfma_fwd", flops=18.0 * cin * cout * B * H * Wd)
                return
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
        # fp32 plans: the 3x3 convolution on the fp32 matrix instruction (same arithmetic class as the direct kernel: an fp32 FMA chain)
        f32m = bool(not mfma and self.act_dt == F32 and x.dt == F32 and out.dt == F32 and k == 3 and cout % 32 == 0 and _f32_mfma_enabled()
                    and isinstance(x, Buf) and Lb.conv3x3_f32_mfma_supported(B, H, Wd, cin, cout))
        st["f32m"] = f32m

        def tiles_fn():
            if dual is not None:
                return int(Lb.conv3x3_mfma_bf16_tiles_dual(B, H, Wd, cin_eff, cout))
            return int(Lb.conv3x3_mfma_bf16_tiles(B, H, Wd, cin_eff, cout))

        def conv_into(y, act_code, stats_direct=None, stats_part=None, stats_atomic=None):
            if up is not None:
                # y <- the hi-res pre-normalisation map in PACKED pixel order [B, h, w, (a, b, cout)]: the per-channel norm kernels do not
                # care about the order of the pixels; the apply pass's output is permuted to hi-res below
                assert act_code == 0 and stats_direct is None and stats_part is None and stats_atomic is None
                st["upconv"] = upconv.forward(self._emit, self._alloc, Lb, S, up.src, wptr, wf, y, B, H // 2, Wd // 2, cin, cout,
                                              need_dgrad=bool(bw and self.req.get(op.inputs[0].op.inputs[0], False)), bias_ptr=bptr)
                return
            if stats_atomic is not None:
                mfma_conv(y, bptr, None, act_code, stats_atomic, 2, None, 0)
            elif head1x1:
                self._emit(Lb.head1x1_fwd, x.ptr, x.dt, wptr, bptr, y.ptr, B * H * Wd, cin, cout, act_code, S)
            elif mfma:
                wsb = int(Lb.conv3x3_mfma_ws_bytes(B, H, Wd, cin_eff, cout)) if stats_part is None else 0
                ws = self._alloc((wsb // 4,), F32) if wsb else None          # split-K slices (small maps)
                mfma_conv(y, bptr, None, act_code, stats_part, 1 if stats_part is not None else 0, ws, wsb)
            elif f32m and stats_direct is None and y.dt == F32:
                need_dgrad = bool(bw and self.req.get(op.inputs[0], False) and cin % 32 == 0)
                w32, _ = self._packed_f32(W, need_dgrad)
                self._emit(Lb.conv3x3_f32_mfma, x.ptr, w32.ptr, bptr, y.ptr, B, H, Wd, cin, cout, act_code, S,
                           tag="conv3x3_f32_mfma_fwd", flops=18.0 * cin * cout * B * H * Wd)
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
                    nz = int(Lb.conv3x3_mfma_ksplit(B, H, Wd, cin_eff, cout))
                    # ... and the batch-norm launch is the split-K finishing pass as well (phx_bn_wide_fwd: four channels per block,
                    # sums the slices in slice order): one launch fewer per layer, 48 blocks instead of 12 on a 192-channel layer
                    # (a block of that launch pulls P x nz 128-byte lines through ONE CU whatever its channel count: measured + 5 us per
                    # layer at 2 x 2 (P = 256, six slices), - 8 us at 4 x 4 (P = 1 024, three slices) against finishing pass + phx_bn_small_fwd)
                    wide = bool(_BN_WIDE and Lb.bn_wide_supported(P, cout) and P * nz <= _BN_WIDE_MAXLINES)
                    self._emit(Lb.conv3x3_mfma_bf16_f32out, x.ptr, dual.b.ptr if dual is not None else None,
                               dual.k1 if dual is not None else 0, wf.ptr, y.ptr, 0 if wide else 1, ws.ptr if ws is not None else None, wsb,
                               B, H, Wd, cin_eff, cout, S, tag="conv3x3_mfma_fwd", flops=18.0 * cin * cout * B * H * Wd)
                    if wide:
                        self._emit(Lb.bn_wide_fwd, ws.ptr if nz > 1 else y.ptr, nz, y.ptr, gptr, beptr, eps, out.ptr, mean.ptr, rstd.ptr,
                                   scale.ptr, shift.ptr, mm, mv, mom, P, cout, act, S,
                                   tag="bytes_norm_apply", flops=float(y.nbytes + out.nbytes))
                        st.update(y=y, scale=scale, shift=shift, mean=mean, rstd=rstd, NS=NS, P=P, G=Gn, bn_small=True, bn_wide=True)
                        self.saved[op] = st
                        return
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
            xf_producer_ok = False
            if up is not None:
                # (the frame of the packed map is written after the phase convolution: its statistics epilogue cannot be used)
                pivot = self._alloc((NS * cout,), F32)
                conv_into(y, 0)
                self._emit(Lb.norm_stats, y.ptr, y.dt, sums.ptr, pivot.ptr, NS, P, cout, S)
            elif norm == "batch" and mfma and not small:
                ntile = tiles_fn()
                part = self._alloc((ntile * 2 * cout,), F32)
                conv_into(y, 0, stats_part=part)
                self._emit(Lb.norm_reduce_partials, part.ptr, ntile, cout, sums.ptr, S)
                xf_producer_ok = bool(training and act == rt.ACT_RELU and y.dt == BF16 and out.dt == BF16 and not head1x1 and _xf_enabled())
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
            if (bw and xf_producer_ok and pivot is None and self._xf_edge_ok(op, B, H, Wd, cout)):
                # every reader of a = relu(bn(y)) is a large-map 3x3 convolution (and its filter gradient): no apply pass, no tensor a --
                # the statistics are finalised by a one-block launch and the readers transform y in their loaders (XfBuf)
                self._emit(Lb.norm_finalize, sums.ptr, None, gptr, beptr, eps, NS, P, cout, Gn, mean.ptr, rstd.ptr, scale.ptr, shift.ptr,
                           mmp, mvp, mom, S)
                self.val[op.outputs[0]] = XfBuf(out, y, scale, shift)
                st.update(y=y, scale=scale, shift=shift, mean=mean, rstd=rstd, NS=NS, P=P, G=Gn, out=None)
                self.saved[op] = st
                return
            apply_args = (y.ptr, y.dt, sums.ptr, pivot.ptr if pivot is not None else None, gptr, beptr, eps, out.ptr, out.dt,
                          mean.ptr, rstd.ptr, scale.ptr, shift.ptr, mmp, mvp, mom, NS, P, cout, Gn, act)
            hop = self._norm_head_consumer(op) if (y.dt == BF16 and out.dt == BF16 and up is None) else None
            if up is not None:
                # y is in the packed pixel order, the readers of a want hi-res rows: the apply pass writes them (depth-to-space on the fly)
                self._emit(Lb.norm_apply_fused_d2s, *apply_args[:16], NS, P, cout, Gn, act, H // 2, Wd // 2, S, tag="bytes_norm_apply",
                           flops=float(y.nbytes + out.nbytes))
            elif hop is not None and Lb.norm_head_supported(cout, hop.attrs["W"].shape[-1], y.dt, out.dt):
                # the head rides on the apply pass (phx_norm_apply_fused_head): no pass of its own over a
                hW, hb = hop.attrs["W"], hop.attrs["b"]
                yh = self._alloc_like(hop.outputs[0])
                self.val[hop.outputs[0]] = yh
                # training plan, batch norm: a itself is never written -- its one other reader, the head's filter gradient (a leaf of the
                # backward graph), re-forms it from y with this layer's scale / shift (phx_head1x1_wgrad_multi, xscale): for the
                # likelihood's top layer (128 channels @ 128 x 128) 268 MB less to write on the critical lane
                skip_a = bool(bw and norm == "batch" and NS == 1 and _SKIP_HEAD_A)
                if skip_a:
                    apply_args = apply_args[:7] + (None,) + apply_args[8:]
                    st["a_unwritten"] = dict(y=y, scale=scale, shift=shift, act=act)
                self._emit(Lb.norm_apply_fused_head, *apply_args, self.store.ptr(hW), self.store.ptr(hb), hW.shape[-1], yh.ptr, S,
                           tag="bytes_norm_apply", flops=float(y.nbytes + (0 if skip_a else out.nbytes)))
                self._norm_head[hop] = op
            else:
                pop = self._pool_consumer(op, H, Wd, cout) if (y.dt == BF16 and out.dt == BF16) else None
                if pop is not None:
                    # one of the readers is averagepool2D (the next encoder level): the apply pass writes the pooled tensor too
                    pooled = self._alloc(self._cshape(pop.outputs[0]), out.dt)
                    self.val[pop.outputs[0]] = pooled
                    self._pool_done.add(pop)
                    self._emit(Lb.norm_apply_pool, y.ptr, sums.ptr, pivot.ptr if pivot is not None else None, gptr, beptr, eps, out.ptr,
                               pooled.ptr, mean.ptr, rstd.ptr, scale.ptr, shift.ptr, mmp, mvp, mom, NS, P, cout, Gn, H, Wd, act, S,
                               tag="bytes_norm_apply", flops=float(y.nbytes + out.nbytes + pooled.nbytes))
                else:
                    self._emit(Lb.norm_apply_fused, *apply_args, S, tag="bytes_norm_apply", flops=float(y.nbytes + out.nbytes))
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

    def _pool_consumer(self, op, H, Wd, C):
        """The averagepool2D op that reads this conv unit's output directly, on the same lane, on an even map -- or None."""
        if not _POOL_FUSE or not self.L.norm_apply_pool_supported(H, Wd, C):
            return None
        out = op.outputs[0]
        ln = self.op_lane.get(op)
        for c in self._real_consumers(out, self._opset):
            if c.type == "avgpool" and c.inputs[0] is out and self.op_lane.get(c) == ln and c not in self._pool_done:
                # the pool then emits no launch of its own and records no forward event: every reader of the pooled tensor has to sit
                # on the producer's lane too (stream order is its only ordering), and the pooled tensor may not be a fetch
                if c.outputs[0] in self.fetches or any(self.op_lane.get(r) != ln for r in self._real_consumers(c.outputs[0], self._opset)):
                    continue
                return c
        return None

    def _fw_avgpool(self, op, bw):
        if op in self._pool_done:                # the producer's apply pass wrote it (phx_norm_apply_pool)
            return
        x = self.val[op.inputs[0]]
        out = self._alloc(self._cshape(op.outputs[0]), x.dt)
        self.val[op.outputs[0]] = out
        self._emit(self.L.avgpool2x2_fwd, x.ptr, x.dt, out.ptr, x.shape[0], x.shape[1], x.shape[2], x.shape[3],
                   self.stream)

    def _upconv_consumer(self, op, x, bw):
        """The 3x3 conv unit (batch norm, training plan) that is the ONLY reader of this resize and runs in the phase form -- or None."""
        mh = _upconv_min_h()
        if not (mh > 0 and bw and self.loss is not None and self.act_dt == BF16 and isinstance(x, Buf) and x.dt == BF16 and len(x.shape) == 4):
            return None
        ot = op.outputs[0]
        cons = self._real_consumers(ot, self._opset)
        # (the small-map forms of _fw_conv_unit -- one-launch batch norm, fused group norm, wave-per-sample norm: maps up to 16 x 16 --
        # read x.ptr and an UpBuf has none: the phase form starts at 32 x 32 low-resolution maps whatever PHX_UPCONV says)
        if len(cons) != 1 or ot in self.fetches or x.shape[1] < max(mh, 32) or x.shape[2] < max(mh, 32):
            return None
        c = cons[0]
        a = c.attrs if c.type == "conv_unit" else None
        if (a is None or c.inputs[0] is not ot or a["ksize"] != 3 or a.get("transposed") is not None or a.get("general") is not None
                or a["norm"] not in ("batch", "group", "instance") or (a["norm"] == "batch") != (a["b"] is None)
                or self.op_lane.get(c) != self.op_lane.get(op) or c in self._lat):
            return None
        training = a["training"] if isinstance(a["training"], bool) else self.training
        cin, cout = a["W"].shape[-2], a["W"].shape[-1]
        if not training or cin != x.shape[3] or not self.L.upconv_supported(x.shape[0], x.shape[1], x.shape[2], cin, cout):
            return None
        # Measured per edge at batch 64 (profiles/r05_ab_upconv.txt): 192 -> 32 from 64 x 64 + 1.0 % of the step; 192 -> 64 from 32 x 32 - 1.3 %,
        # 32 -> 32 from 64 x 64 - 1 % (the form trades the 4 Cin-channel hi-res tensor for ten more launches and a 4 Cout-column filter
        # gradient: it pays where the up-sampled tensor dominates the layer's bytes)
        if cin < 4 * cout:
            return None
        return c

    def _fw_bilinear_up(self, op, bw):
        x = self.val[op.inputs[0]]
        if self._upconv_consumer(op, x, bw) is not None:
            # no launch, no hi-res tensor: the reader convolves the low-resolution map with the phase filters (_fw_conv_unit, upconv.py)
            self.val[op.outputs[0]] = UpBuf(x, self._cshape(op.outputs[0]))
            return
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

"""The PHiSeg model: graph wiring, ELBO, train loop, sampling API.

Counterpart of the reference's phiseg/phiseg_model.py with the same public surface
(``phiseg(exp_config)``, ``.train(data)``, ``.predict*``, ``.generate_prior_samples``, ``.load_weights``,
attributes ``x_inp``, ``s_inp``, ``training_pl``, ``lr_pl``, ``z_list``, ``s_out_list``, ``s_out_eval``,
``s_out_eval_sm``, ``loss_dict``, ``loss_tot``, ``train_step``, ``sess``) -- but ``sess.run`` lowers the
requested fetches to hand-written HIP kernels replayed from a hipGraph (phiseg_code_amd.engine) instead of
dispatching TensorFlow ops.

Deliberate, documented differences from the TF1 graph:
* Q1/Q4: only the live graph is executed per training step (the reference also runs the generation-mode
  prior, the evaluation likelihood and the never-consumed up-sampling branches because their batch-norm
  update ops sit in UPDATE_OPS); batch-norm moving statistics are updated once per step from the training
  graph instance.
* Q10: noise is a seeded Philox4x32-10 stream keyed by (seed, step, net, level, global sample index).
"""
import logging
import os
import time

import numpy as np

from phiseg_code_amd import engine
from phiseg_code_amd import graph as G
from phiseg_code_amd import utils

logging.basicConfig(level=logging.INFO, format='%(asctime)s %(message)s')


class _Flag:
    """Scalar placeholder fed through feed_dict (training_pl, lr_pl)."""

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "<placeholder %s>" % self.name


class TrainStep:
    """Handle returned as ``model.train_step``: fetching it runs backward + Adam (optimizer.minimize)."""

    def __init__(self, loss):
        self.loss = loss


class Session:
    """``sess.run(fetches, feed_dict)`` over compiled plans (one per fetch set / batch size / training flag)."""

    def __init__(self, model, compute_dtype, rng_seed=42, dist=None):
        self.model = model
        self.compute_dtype = compute_dtype
        self.rng_seed = rng_seed
        self.dist = dist
        self.store = None
        self.plans = {}
        self._lr = None

    def _ensure_store(self):
        if self.store is None:
            self.store = engine.ParamStore(self.model.graph, seed=self.model.init_seed,
                                           live=engine.live_variables(self.model.loss_tot))
            if self.dist is not None and self.dist.active:
                # identical replicas (the Philox initialiser already gives every rank the same values; this also covers
                # weights loaded on one rank only)
                self.dist.broadcast_(self.store.params)
                self.dist.broadcast_(self.store.state)
                engine.device_sync()
        return self.store

    def plan_for(self, fetch_tensors, train, batch, training, stamp_tagged=False):
        key = (tuple(id(t) for t in fetch_tensors), bool(train), int(batch), bool(training)) + (("stamped",) if stamp_tagged else ())
        if key not in self.plans:
            store = self._ensure_store()
            world, rank = (self.dist.world, self.dist.rank) if self.dist else (1, 0)
            self.plans[key] = engine.Plan(
                store, fetch_tensors, loss=self.model.loss_tot if train else None, batch=batch, training=training,
                compute_dtype=self.compute_dtype, rng_seed=self.rng_seed, sample_offset=rank * batch,
                loss_inv_batch=1.0 / (batch * world), split_optimizer=bool(self.dist and self.dist.active), stamp_tagged=stamp_tagged)
        return self.plans[key]

    def run(self, fetches, feed_dict=None):
        feed_dict = feed_dict or {}
        m = self.model
        single = not isinstance(fetches, (list, tuple))
        flat, spec = [], []

        def walk(f):
            if isinstance(f, (list, tuple)):
                return [walk(x) for x in f]
            flat.append(f)
            return len(flat) - 1
        spec = walk([fetches] if single else list(fetches))
        train = any(isinstance(f, TrainStep) for f in flat)
        tensors = [f for f in flat if isinstance(f, G.Tensor)]
        x = feed_dict.get(m.x_inp)
        if x is None:
            raise ValueError("feed_dict must provide x_inp")
        x = np.asarray(x)
        training = bool(feed_dict.get(m.training_pl, False))
        if train and training and m.bn_double_update:
            tensors = tensors + [m.s_out_eval]         # Q4: the second graph instances run (and update moving statistics) too
        plan = self.plan_for(tensors, train, x.shape[0], training)
        plan.set_input("x_input", x)
        if "s_input" in plan.feeds:
            if m.s_inp not in feed_dict:
                raise ValueError("these fetches need s_inp")
            plan.set_input("s_input", feed_dict[m.s_inp])
        if m.lr_pl in feed_dict and float(feed_dict[m.lr_pl]) != self._lr:
            self._lr = float(feed_dict[m.lr_pl])
            self.store.set_lr(self._lr)               # (synchronises: the plans replay on their own HIP streams)
        dp = self.dist is not None and self.dist.active
        if train and dp:
            plan.run_main()
            self.dist.allreduce_sum(self.store.grads[:self.store.n_live], plan)
            plan.run_opt()
        else:
            plan.run()
        vals = [plan.fetch(f) if isinstance(f, G.Tensor) else None for f in flat]
        if dp and "s_input" in plan.feeds:
            # the loss kernels scale every term by 1 / (B_local * world): a rank's scalar is its SHARE of the global-batch mean
            # (phiseg_model.py:221,236 take the mean over the whole batch) -- sum the shares
            for i, f in enumerate(flat):
                if isinstance(f, G.Tensor) and f.shape == () and vals[i] is not None:
                    vals[i] = np.float32(self.dist.sum_float(float(vals[i])))

        def build(s):
            return [build(x) for x in s] if isinstance(s, list) else vals[s]
        out = build(spec)
        return out[0] if single else out


class phiseg():

    def __init__(self, exp_config, dist=None, init_seed=0, rng_seed=42, bn_double_update=None):
        """bn_double_update (default: $PHX_BN_DOUBLE_UPDATE == 1): reproduce the TF graph's UPDATE_OPS behaviour (SURVEY.md Q4,
        phiseg_model.py:135-141): every training step also runs the generation-mode prior and the evaluation likelihood in
        training mode, so the batch-norm moving statistics of the layers they share with the training instances are updated
        TWICE per step.  Off by default: only the live graph runs (1.9x less forward work)."""
        self.exp_config = exp_config
        self.bn_double_update = (os.environ.get("PHX_BN_DOUBLE_UPDATE", "0") == "1") if bn_double_update is None else bool(bn_double_update)
        self.init_seed = init_seed
        self.checks()
        self.graph = G.reset_default_graph()
        L = exp_config.latent_levels

        # placeholders (phiseg_model.py:26-32)
        self.x_inp = G.placeholder(G.KIND_F32, [None] + list(exp_config.image_size), name='x_input')
        self.s_inp = G.placeholder(G.KIND_U8, [None] + list(exp_config.image_size[0:2]), name='s_input')
        self.s_inp_oh = G.one_hot(self.s_inp, depth=exp_config.nlabels)
        self.training_pl = _Flag('training_time')
        self.lr_pl = _Flag('learning_rate')

        # networks (phiseg_model.py:37-98)
        net_kw = dict(n0=exp_config.n0, resolution_levels=exp_config.resolution_levels, latent_levels=L,
                      norm=exp_config.layer_norm)
        self.z_list, self.mu_list, self.sigma_list = exp_config.posterior(
            self.x_inp, self.s_inp_oh, exp_config.zdim0, training=self.training_pl, **net_kw)
        self.prior_z_list, self.prior_mu_list, self.prior_sigma_list = exp_config.prior(
            self.z_list, self.x_inp, zdim_0=exp_config.zdim0, n_classes=exp_config.nlabels,
            training=self.training_pl, generation_mode=False, **net_kw)
        self.prior_z_list_gen, self.prior_mu_list_gen, self.prior_sigma_list_gen = exp_config.prior(
            self.z_list, self.x_inp, zdim_0=exp_config.zdim0, n_classes=exp_config.nlabels,
            training=self.training_pl, generation_mode=True, scope_reuse=True, **net_kw)
        self.s_out_list = exp_config.likelihood(self.z_list, self.training_pl, n_classes=exp_config.nlabels,
                                                image_size=exp_config.image_size, x=self.x_inp, **net_kw)
        self.s_out_eval_list = exp_config.likelihood(self.prior_z_list_gen, self.training_pl, scope_reuse=True,
                                                     n_classes=exp_config.nlabels, image_size=exp_config.image_size,
                                                     x=self.x_inp, **net_kw)
        # aggregated outputs (phiseg_model.py:106-109)
        self.s_out_eval, self.s_out_eval_sm = G.aggregate_logits(self.s_out_eval_list)

        # losses (phiseg_model.py:113-130)
        self.loss_dict = {}
        terms, weights = [], []
        self.s_out = None
        if getattr(exp_config, 'residual_multinoulli_loss_weight', None) is not None:
            logging.info(' - Adding residual multinoulli loss')
            w = exp_config.residual_multinoulli_loss_weight
            ce, self.s_out = G.residual_multinoulli(self.s_out_list, self.s_inp, w)
            for ii in reversed(range(L)):
                self.loss_dict['residual_multinoulli_loss_lvl%d' % ii] = ce[ii]
                terms.append(ce[ii])
                weights.append(w)
        if getattr(exp_config, 'KL_divergence_loss_weight', None) is not None:
            logging.info(' - Adding hierarchical KL loss')
            w = exp_config.KL_divergence_loss_weight
            lw = [4 ** i for i in range(L)] if exp_config.exponential_weighting else [1] * L
            for ii in reversed(range(L)):
                kl = G.kl_two_gauss(self.mu_list[ii], self.sigma_list[ii], self.prior_mu_list[ii],
                                    self.prior_sigma_list[ii], lw[ii], w)
                self.loss_dict['KL_divergence_loss_lvl%d' % ii] = kl
                terms.append(kl)
                weights.append(w)
        if getattr(exp_config, 'weight_decay_weight', None) is not None:
            logging.info(' - Adding weight decay')                      # add_weight_decay (phiseg_model.py:126-128, 290-299)
            wd = G.l2_of_collection(float(exp_config.weight_decay_weight), 'weight_variables')
            self.loss_dict['weight_decay'] = wd
            terms.append(wd)
            weights.append(1.0)
        self.loss_tot = G.weighted_sum(terms, weights)
        self.loss_dict['total_loss'] = self.loss_tot
        self.train_step = TrainStep(self.loss_tot)

        self._multi = {}
        self.keep_checkpoint_every_n_hours = 3.0          # tf.train.Saver(max_to_keep=1, keep_checkpoint_every_n_hours=3) (phiseg_model.py:144)
        self._ckpt_permanent, self._ckpt_last_permanent = {}, time.time()
        self._ckpt_written, self._ckpt_order = {}, {}      # per saver prefix: the steps this instance wrote (set / in write order)
        self.sess = Session(self, getattr(exp_config, 'compute_dtype', 'f32'), rng_seed=rng_seed, dist=dist)
        self.dist = dist

    def checks(self):
        pass

    # ---- training loop (phiseg_model.py:166-207) -------------------------------------------------
    def _is_writer(self):
        return self.dist is None or not self.dist.active or self.dist.rank == 0

    def _setup_log_dir_and_continue_mode(self, log_dir=None):
        """phiseg_model.py:821-845: log dir = <log_root>/<log_dir_name>/<experiment_name>; when it already holds a
        model.ckpt-N checkpoint the run continues from the highest N and logs into <log_dir>_cont."""
        from phiseg_code_amd.config import system as sys_config
        from phiseg_code_amd.tfwrapper import utils as tfutils
        cfg = self.exp_config
        self.log_dir = log_dir or os.path.join(sys_config.log_root, cfg.log_dir_name, cfg.experiment_name)
        self.init_checkpoint_path = None
        self.continue_run = False
        self.init_step = 0
        if os.path.isdir(self.log_dir):
            ckpt = tfutils.get_latest_model_checkpoint_path(self.log_dir, 'model.ckpt')
            if ckpt is not False:
                self.init_checkpoint_path = ckpt
                self.continue_run = True
                self.init_step = int(os.path.basename(ckpt).split('-')[-1])
                self.log_dir += '_cont'
                logging.info('--------------------------- Continuing previous run --------------------------------')
                logging.info('Checkpoint path: %s' % self.init_checkpoint_path)
                logging.info('Latest step was: %d' % self.init_step)
        if self._is_writer():
            os.makedirs(self.log_dir, exist_ok=True)

    def train(self, data, num_iter=None, log_every=100, log_dir=None):
        """The reference's train(data) (phiseg_model.py:166-207): continue mode, lr schedule, one ELBO step per iteration,
        `_do_validation` (checkpoint + metrics + best-of checkpoints) every `validation_frequency` steps.  TensorBoard
        summaries (199-203) are out of scope.  `log_dir=None` keeps everything in memory (no checkpoints, no validation)
        unless the config's log root exists; pass a directory to get the reference's on-disk behaviour.
        Data-parallel: `data` should draw rank-dependent batches (SyntheticLIDC(cfg, seed=1234 + rank)); the returned /
        logged loss is the global-batch mean; only rank 0 writes files."""
        cfg = self.exp_config
        num_iter = cfg.num_iter if num_iter is None else num_iter
        on_disk = log_dir is not None
        self.init_step, self.continue_run = 0, False
        if on_disk:
            self._setup_log_dir_and_continue_mode(log_dir)
            if self.continue_run:
                self.load_weights(self.init_checkpoint_path)
        self.best_dice, self.best_loss, self.best_ged, self.best_ncc = -1, np.inf, np.inf, -1
        losses = []
        t0 = time.time()
        for step in range(self.init_step, num_iter):
            lr_key, _ = utils.find_floor_in_list(cfg.lr_schedule_dict.keys(), step)
            lr = cfg.lr_schedule_dict[lr_key]
            x_b, s_b = data.train.next_batch(cfg.batch_size)
            _, loss_tot_eval = self.sess.run([self.train_step, self.loss_tot],
                                             feed_dict={self.x_inp: x_b, self.s_inp: s_b, self.training_pl: True,
                                                        self.lr_pl: lr})
            losses.append(float(loss_tot_eval))
            if log_every and step % log_every == 0:
                world = self.dist.world if (self.dist is not None and self.dist.active) else 1
                logging.info('step %d  loss %.4f  (%.1f img/s)', step, losses[-1],
                             (step - self.init_step + 1) * cfg.batch_size * world / max(time.time() - t0, 1e-9))
            vf = getattr(cfg, 'validation_frequency', None)
            if on_disk and vf and step % vf == 0:
                self._do_validation(data)
        return losses

    # ---- validation (phiseg_model.py:530-660): N Monte-Carlo samples per image, metrics on the device ---------------
    def _do_validation(self, data):
        """For every validation image: `validation_samples` segmentation samples + the ELBO against one randomly chosen
        annotation (phiseg_model.py:570-586), then GED over the foreground labels, variance-NCC and the Dice of the mean
        prediction (586-613).  The reference scores these in Python loops on the host (N*M + N^2 + M^2 IoU evaluations per
        image); here they are one libphx call per image (utils.validation_metrics -> phx_validation_metrics).
        -> dict(loss, dice, per_structure_dice, ged, ncc), the averages the reference logs (615-634)."""
        cfg = self.exp_config
        store = self.sess._ensure_store()
        global_step = int(store.step.cpu().item()) - 1               # tf global_step - 1 (phiseg_model.py:532)
        dp = self.dist is not None and self.dist.active
        # Data parallel: every decision below that leads to a collective (save_weights averages the batch-norm moving statistics
        # over the replicas) must be the same on all ranks -- the per-rank metrics differ (noise offsets, random annotators).  Rank 0
        # decides; its flags are broadcast.  The replicas' moving statistics are averaged ONCE, here, so that the best-of saves
        # further down run no collective at all.
        save = getattr(self, 'log_dir', None) is not None and os.path.isdir(getattr(self, 'log_dir', '') or '')
        if dp:
            save = bool(self.dist.broadcast_flags([1 if save else 0])[0])
            self._average_replica_state()
        if save:
            self.save_weights(os.path.join(self.log_dir, 'model.ckpt-%d' % global_step), format=self._ckpt_format(),
                              keep_prefix='model.ckpt', max_to_keep=1, average_state=False)
        if hasattr(data.validation, 'next_batch'):                   # BATCH VALIDATION of every loss term (537-556)
            names = list(self.loss_dict.keys())
            val_x, val_s = data.validation.next_batch(cfg.batch_size)
            val_out = self.sess.run(list(self.loss_dict.values()),
                                    feed_dict={self.x_inp: val_x, self.s_inp: val_s, self.training_pl: False})
            train_x, train_s = data.train.next_batch(cfg.batch_size)
            train_out = self.sess.run(list(self.loss_dict.values()),
                                      feed_dict={self.x_inp: train_x, self.s_inp: train_s, self.training_pl: False})
            logging.info('----- Step: %d ------' % global_step)
            logging.info('BATCH VALIDATION:')
            for ii, loss_name in enumerate(names):
                logging.info('%s | training: %f | validation: %f' % (loss_name, train_out[ii], val_out[ii]))
        imgs, labs = data.validation.images, data.validation.labels
        n_img = imgs.shape[0] if cfg.num_validation_images == 'all' else min(cfg.num_validation_images, imgs.shape[0])
        ns = cfg.validation_samples
        dice_list, elbo_list, ged_list, ncc_list = [], [], [], []
        for ii in range(n_img):
            x = imgs[ii, ...].reshape([1] + list(cfg.image_size))
            s_gt_arr = labs[ii, ...]                                   # [X, Y, num annotators]
            s = s_gt_arr[:, :, np.random.choice(cfg.annotator_range)]
            x_b, s_b = np.tile(x, [ns, 1, 1, 1]), np.tile(s, [ns, 1, 1])
            fd = {self.training_pl: False, self.x_inp: x_b, self.s_inp: s_b}
            sm_arr, elbo = self.sess.run([self.s_out_eval_sm, self.loss_tot], feed_dict=fd)
            self._advance_noise()
            gts = np.ascontiguousarray(s_gt_arr.transpose((2, 0, 1)))  # num annotators x X x Y
            ged, ncc, dice = utils.validation_metrics(sm_arr[None], gts[None], s[None], cfg.nlabels)
            dice_list.append(dice[0])
            elbo_list.append(float(elbo))
            ged_list.append(float(ged[0]))
            ncc_list.append(float(ncc[0]))
        dice_arr = np.asarray(dice_list)
        out = dict(loss=utils.list_mean(elbo_list), dice=float(np.mean(dice_arr)), per_structure_dice=dice_arr.mean(axis=0),
                   ged=utils.list_mean(ged_list), ncc=utils.list_mean(ncc_list))
        logging.info('FULL VALIDATION (%d images):' % n_img)
        logging.info(' - Mean foreground dice: %.4f' % np.mean(out['per_structure_dice']))
        logging.info(' - Mean (neg.) ELBO: %.4f' % out['loss'])
        logging.info(' - Mean GED: %.4f' % out['ged'])
        logging.info(' - Mean NCC: %.4f' % out['ncc'])
        # best-of checkpoints (phiseg_model.py:638-660)
        if not hasattr(self, 'best_dice'):
            self.best_dice, self.best_loss, self.best_ged, self.best_ncc = -1, np.inf, np.inf, -1
        mean_dice = float(np.mean(out['per_structure_dice']))
        cands = (('dice', mean_dice, mean_dice >= self.best_dice, 'New best validation Dice! (%.3f)'),
                 ('loss', out['loss'], out['loss'] <= self.best_loss, 'New best validation loss! (%.3f)'),
                 ('ged', out['ged'], out['ged'] <= self.best_ged, 'New best GED score! (%.3f)'),
                 ('ncc', out['ncc'], out['ncc'] >= self.best_ncc, 'New best NCC score! (%.3f)'))
        flags = [1 if c[2] else 0 for c in cands]
        if dp:
            flags = self.dist.broadcast_flags(flags)                   # rank 0's metrics decide on every rank
        for (key, value, _, fmt), better in zip(cands, flags):
            if better:
                setattr(self, 'best_' + key, value)
                logging.info(fmt % value)
                if save:                                               # (the reference: Saver(max_to_keep=2) per best-of saver)
                    self.save_weights(os.path.join(self.log_dir, 'model_best_%s.ckpt-%d' % (key, global_step)), format=self._ckpt_format(),
                                      keep_prefix='model_best_%s.ckpt' % key, max_to_keep=2, average_state=False)
        return out

    def _average_replica_state(self):
        """Data parallel: batch-norm moving statistics are per replica; average them over the ranks."""
        store = self.sess._ensure_store()
        if self.dist is not None and self.dist.active and store.n_state:
            avg = store.state.clone()
            self.dist.allreduce_sum(avg)
            avg /= self.dist.world
            store.state.copy_(avg)
            engine.device_sync()

    def _note_checkpoint(self, keep_prefix, path):
        import re
        m = re.search(re.escape(keep_prefix) + r'-(\d+)(?:\.npz)?$', os.path.basename(path))
        if m:
            st = int(m.group(1))
            self._ckpt_written.setdefault(keep_prefix, set()).add(st)
            order = self._ckpt_order.setdefault(keep_prefix, [])
            if st in order:
                order.remove(st)
            order.append(st)

    def _prune_checkpoints(self, directory, keep_prefix, max_to_keep):
        """tf.train.Saver(max_to_keep=...) (phiseg_model.py:144-148): delete all but the newest `max_to_keep` checkpoints named
        <keep_prefix>-<step>(.npz | .index + .data-*); -> the retained prefixes, oldest first."""
        import glob
        import re
        found = {}
        # like Saver._last_checkpoints: only checkpoints THIS instance has written are rotated -- files an earlier run left in the
        # directory (a fresh run into an old log_dir writes model.ckpt-0 while model.ckpt-5000 is still there) are never touched
        mine = self._ckpt_written.setdefault(keep_prefix, set())
        for f in glob.glob(os.path.join(directory, keep_prefix + '-*')):
            m = re.match(re.escape(keep_prefix) + r'-(\d+)(\.npz|\.index|\.data-\d+-of-\d+)$', os.path.basename(f))
            if m and int(m.group(1)) in mine:
                found.setdefault(int(m.group(1)), []).append(f)
        steps = [st for st in self._ckpt_order.get(keep_prefix, []) if st in found]      # write order, as tf.train.Saver keeps it
        # keep_checkpoint_every_n_hours (3 for the training saver): a checkpoint that falls out of the max_to_keep window is kept for
        # good when that many hours of training have passed since the last one kept this way
        keep_h = self.keep_checkpoint_every_n_hours if keep_prefix == 'model.ckpt' else None
        perm = self._ckpt_permanent.setdefault(keep_prefix, set())
        for s in steps[:-max_to_keep] if max_to_keep > 0 else []:
            if s in perm:
                continue
            if keep_h is not None and time.time() - self._ckpt_last_permanent >= keep_h * 3600.0:
                perm.add(s)
                self._ckpt_last_permanent = time.time()
                continue
            for f in found[s]:
                try:
                    os.remove(f)
                except OSError:
                    pass
        kept = [s for s in steps if s in perm or s in steps[-max_to_keep:]]
        return [os.path.join(directory, '%s-%d' % (keep_prefix, s)) for s in kept]

    # ---- checkpoints (npz keyed by the TF variable names of SURVEY.md Appendix B) -------------------
    def _ckpt_format(self):
        """exp_config.checkpoint_format: 'npz' (default) or 'tf' (TensorFlow tensor bundles, as the reference's Saver writes)"""
        return getattr(self.exp_config, 'checkpoint_format', 'npz')

    def save_weights(self, path, format='npz', keep_prefix=None, max_to_keep=0, average_state=True):
        """What tf.train.Saver writes for this graph (phiseg_model.py:144-148, 534-535): every variable, the Adam slots under
        TF's names '<var>/Adam' (m) and '<var>/Adam_1' (v), and the step (TF keeps beta1_power / beta2_power and global_step;
        one integer carries the same information).  File: <path>.npz, or with format='tf' a TensorFlow tensor-bundle checkpoint
        <path>.index + <path>.data-00000-of-00001 (tfwrapper/tf_checkpoint.py) that tf.train.Saver.restore of the reference
        graph accepts: same variable names, beta1_power / beta2_power / global_step included.  Data-parallel: batch-norm
        moving statistics are averaged over the replicas first (per-replica statistics); rank 0 writes."""
        store = self.sess._ensure_store()
        if average_state:              # (a collective: every rank must call save_weights -- _do_validation averages once itself)
            self._average_replica_state()
        if not self._is_writer():
            return
        blob = dict(store.export())
        for name, (m, v) in store.export_adam().items():
            blob[name + '/Adam'] = m
            blob[name + '/Adam_1'] = v
        step = int(store.step.cpu().numpy()[0])
        if format == 'tf':
            from phiseg_code_amd import optimizers
            from phiseg_code_amd.tfwrapper import tf_checkpoint
            if path.endswith('.npz'):
                path = path[:-4]
            # TF 1.x AdamOptimizer's non-slot variables: beta_power = beta^(t + 1) after t updates; minimize() counts global_step
            b1, b2 = optimizers.AdamOptimizer.beta1, optimizers.AdamOptimizer.beta2
            blob['beta1_power'] = np.asarray(b1 ** (step + 1), dtype=np.float32)
            blob['beta2_power'] = np.asarray(b2 ** (step + 1), dtype=np.float32)
            blob['global_step'] = np.asarray(step, dtype=np.int64)
            tf_checkpoint.write(path, blob)
            d = os.path.dirname(os.path.abspath(path))
            if keep_prefix:
                self._note_checkpoint(keep_prefix, path)
            kept = self._prune_checkpoints(d, keep_prefix, max_to_keep) if keep_prefix else None
            tf_checkpoint.update_checkpoint_state(d, os.path.basename(path), all_paths=[os.path.basename(k) for k in kept] if kept else None)
            return
        if format != 'npz':
            raise ValueError("save_weights: format is 'npz' or 'tf'")
        if not path.endswith('.npz'):
            path += '.npz'
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        tmp = path + '.tmp.npz'
        np.savez(tmp, __step__=np.asarray([step], dtype=np.int32), **blob)
        os.replace(tmp, path)
        if keep_prefix:
            self._note_checkpoint(keep_prefix, path)
            self._prune_checkpoints(os.path.dirname(os.path.abspath(path)), keep_prefix, max_to_keep)

    def load_weights(self, log_dir=None, type='latest', **kwargs):
        """phiseg_model.py:505-525 (+ 'best_ncc', which the reference writes but cannot load -- SURVEY.md Q9).  `log_dir` may
        also be a checkpoint file / prefix.  Restores variables, Adam slots and the step; a checkpoint without Adam slots
        (weights only) resets the optimiser state and the step.  A prefix with a `.index` file next to it is a TensorFlow
        tensor-bundle checkpoint -- one written by the reference's tf.train.Saver or by save_weights(format='tf') -- and is
        read directly (tfwrapper/tf_checkpoint.py); variables the checkpoint lacks keep their values, as Saver.restore of a
        sub-graph would."""
        from phiseg_code_amd.tfwrapper import utils as tfutils
        if not log_dir:
            log_dir = getattr(self, 'log_dir', None)
        path = log_dir
        if os.path.isdir(log_dir):
            names = {'latest': 'model.ckpt', 'best_dice': 'model_best_dice.ckpt', 'best_loss': 'model_best_loss.ckpt',
                     'best_ged': 'model_best_ged.ckpt', 'best_ncc': 'model_best_ncc.ckpt'}
            if type == 'iter':
                assert 'iteration' in kwargs, "argument 'iteration' must be provided for type='iter'"
                path = os.path.join(log_dir, 'model.ckpt-%d' % kwargs['iteration'])
            elif type in names:
                path = tfutils.get_latest_model_checkpoint_path(log_dir, names[type])
                if path is False:
                    raise FileNotFoundError('no %s checkpoint in %s' % (type, log_dir))
            else:
                raise ValueError('Argument type=%s is unknown. type can be latest/iter.' % type)
        store = self.sess._ensure_store()
        names = set(self.graph.variables)
        if os.path.exists(path + '.index') and not os.path.exists(path + '.npz'):
            from phiseg_code_amd import optimizers
            from phiseg_code_amd.tfwrapper import tf_checkpoint
            ck = tf_checkpoint.read(path)
            files = list(ck)
            if 'global_step' in ck:
                step = int(ck['global_step'])
            elif 'beta1_power' in ck:            # beta1^(t + 1) after t updates
                step = max(0, int(round(np.log(float(ck['beta1_power'])) / np.log(optimizers.AdamOptimizer.beta1))) - 1)
            else:
                step = 0
        else:
            if not os.path.exists(path) and os.path.exists(path + '.npz'):
                path += '.npz'
            ck = np.load(path)
            files = ck.files
            step = int(ck['__step__'][0]) if '__step__' in files else 0
        store.load({k: ck[k] for k in files if k in names})
        missing = sorted(names - set(files))
        if missing:                      # (a name-mapping error would otherwise pass silently: those variables keep their values)
            logging.warning('load_weights: %d of %d graph variables are not in %s and keep their values (first: %s)',
                            len(missing), len(names), path, ', '.join(missing[:3]))
        self.last_load_missing = missing
        slots = {k[:-len('/Adam')]: (ck[k], ck[k + '_1']) for k in files if k.endswith('/Adam') and k + '_1' in files}
        if slots:
            store.load_adam(slots)
            store.set_step(step)
        else:
            store.reset_optimizer()
        self.sess._lr = None
        if self.dist is not None and self.dist.active:
            for t in (store.params, store.state, store.adam_m, store.adam_v):
                self.dist.broadcast_(t)
            engine.device_sync()

    def set_weights(self, values):
        self.sess._ensure_store().load(values)

    # ---- inference API (phiseg_model.py:313-375) -------------------------------------------------
    def generate_prior_samples(self, x_in, return_params=False):
        fd = {self.training_pl: False, self.x_inp: x_in}
        if return_params:
            z, mu, sg = self.sess.run([self.prior_z_list_gen, self.prior_mu_list_gen, self.prior_sigma_list_gen], fd)
            return z, mu, sg
        return self.sess.run(self.prior_z_list_gen, fd)

    def sampling_graph(self, num_samples):
        """(s_out_eval, s_out_eval_sm) of a graph instance that draws `num_samples` segmentations PER fed image in one pass:
        the prior's encoder -- a function of x alone -- runs once per image, its features are repeated num_samples times
        (graph.tile_batch) and the latent path + likelihood run at batch B * num_samples.  Same variables (scope reuse) and
        the same Philox stream as s_out_eval_sm; output rows b * num_samples + k = sample k of image b."""
        if num_samples not in self._multi:
            cfg = self.exp_config
            net_kw = dict(n0=cfg.n0, resolution_levels=cfg.resolution_levels, latent_levels=cfg.latent_levels, norm=cfg.layer_norm)
            G.set_default_graph(self.graph)
            z_gen, _, _ = cfg.prior(self.z_list, self.x_inp, zdim_0=cfg.zdim0, n_classes=cfg.nlabels, training=self.training_pl,
                                    generation_mode=True, scope_reuse=True, tile_samples=num_samples, **net_kw)
            s_list = cfg.likelihood(z_gen, self.training_pl, scope_reuse=True, n_classes=cfg.nlabels, image_size=cfg.image_size,
                                    x=self.x_inp, **net_kw)
            self._multi[num_samples] = G.aggregate_logits(s_list)
        return self._multi[num_samples]

    def predict(self, x_in, num_samples=50, return_softmax=False):
        """phiseg_model.py:337-354: mean soft-max over num_samples prior samples, arg-max.  One pass per call when the prior
        has an x-only encoder to share (sampling_graph); prob_unet2D draws z per image, so it keeps the reference's loop."""
        fd = {self.training_pl: False, self.x_inp: x_in}
        if num_samples > 1 and getattr(self.exp_config.prior, '__name__', '') == 'phiseg':
            _, sm = self.sampling_graph(num_samples)
            sm_all = self.sess.run(sm, feed_dict=fd)                          # [B * n, X, Y, C]
            self._advance_noise()
            cumsum_sm = sm_all.reshape((-1, num_samples) + sm_all.shape[1:]).sum(axis=1)
            if return_softmax:
                return np.argmax(cumsum_sm, axis=-1), cumsum_sm / num_samples
            return np.argmax(cumsum_sm, axis=-1)
        cumsum_sm = self.sess.run(self.s_out_eval_sm, feed_dict=fd)
        for _ in range(num_samples - 1):
            self._advance_noise()
            cumsum_sm = cumsum_sm + self.sess.run(self.s_out_eval_sm, feed_dict=fd)
        self._advance_noise()
        if return_softmax:
            return np.argmax(cumsum_sm, axis=-1), cumsum_sm / num_samples
        return np.argmax(cumsum_sm, axis=-1)

    def predict_segmentation_sample(self, x_in, return_softmax=False):
        fd = {self.training_pl: False, self.x_inp: x_in}
        out = self.sess.run(self.s_out_eval_sm if return_softmax else self.s_out_eval, feed_dict=fd)
        self._advance_noise()
        return out if return_softmax else np.argmax(out, axis=-1)

    def predict_segmentation_sample_levels(self, x_in, return_softmax=False):
        fd = {self.training_pl: False, self.x_inp: x_in}
        lv = self.sess.run(self.s_out_eval_list, feed_dict=fd)
        self._advance_noise()
        if return_softmax:
            e = [np.exp(v - v.max(axis=-1, keepdims=True)) for v in lv]
            return [v / v.sum(axis=-1, keepdims=True) for v in e]
        return lv

    def _advance_noise(self):
        """Every sampling call must see fresh noise (TF's stateful RNG): bump the Philox step word."""
        store = self.sess._ensure_store()
        store.noise_step += 1
        engine.device_sync()          # torch's stream -> the plans' own HIP streams

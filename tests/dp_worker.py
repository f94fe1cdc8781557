"""Worker of tests/test_distributed_gpu.py: one rank of a data-parallel training step of the ENGINE (not the oracle) on
its shard of a global batch; several such ranks share cuda:0 (PHX_DIST_BACKEND=gloo: RCCL refuses duplicate devices)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, case, dtype, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    from oracle import init as oinit
    from oracle import train as otrain
    from phiseg_code_amd import distributed
    from phiseg_code_amd.phiseg import phiseg_model
    from tests.helpers import load_golden
    from tests.test_graph_cpu import make_config
    ctx = distributed.DistContext()
    g, cfg, var_order = load_golden(case)
    b = cfg["B"]                                         # images per rank; global batch = world * b
    gb = b * ctx.world
    c = make_config(cfg, dtype)
    if os.environ.get("PHX_TEST_WD"):                    # weight decay term (phiseg_model.py:126-128): data-parallel share test
        c.weight_decay_weight = float(os.environ["PHX_TEST_WD"])
    if os.environ.get("PHX_TEST_TRAIN_DIR"):             # model.train(data, log_dir) through several validations on every rank
        return train_with_validation(ctx, c, os.environ["PHX_TEST_TRAIN_DIR"], out_dir)
    model = phiseg_model.phiseg(c, dist=ctx if ctx.active else None, rng_seed=cfg["eps_seed"])
    params = otrain.make_params(var_order, cfg["weight_seed"], torch.float64, perturbed=True)
    if ctx.rank == 0:                                    # only rank 0 holds the weights: the broadcast at store creation /
        model.set_weights({k: v.detach().numpy() for k, v in params.items()})       # load time must replicate them
        if ctx.active:
            for t in (model.sess.store.params, model.sess.store.state):
                ctx.broadcast_(t)
    elif ctx.active:
        model.sess._ensure_store()
        for t in (model.sess.store.params, model.sess.store.state):
            ctx.broadcast_(t)
    x, s = oinit.synthetic_batch(gb, cfg["H"], cfg["nlabels"], cfg["data_seed"])
    lo = ctx.rank * b
    losses = []
    keys = sorted(model.loss_dict)
    for _ in range(steps):
        out = model.sess.run([model.train_step] + [model.loss_dict[k] for k in keys],
                             {model.x_inp: x[lo:lo + b], model.s_inp: s[lo:lo + b], model.training_pl: True,
                              model.lr_pl: 1e-5})
        losses.append([float(v) for v in out[1:]])
    store = model.sess.store
    blob = {"losses": np.array(losses), "keys": np.array(keys), "n_live": store.n_live, "n_train": store.n_train}
    for k, v in store.export(grads=True).items():
        blob["grad/" + k] = v
    for k, v in store.export().items():
        blob["param/" + k] = v
    np.savez(os.path.join(out_dir, "rank%d.npz" % ctx.rank), **blob)
    ctx.barrier()
    ctx.shutdown()


def train_with_validation(ctx, c, log_dir, out_dir):
    """train() with validation_frequency = 1: every validation takes the best-of decisions and writes checkpoints -- the ranks'
    metrics differ (noise offsets, random annotators), the decisions and the collectives they lead to must not."""
    from phiseg_code_amd.data import synthetic
    from phiseg_code_amd.phiseg import phiseg_model
    c.batch_size = 2
    c.validation_frequency = 1
    c.validation_samples = 4
    c.num_validation_images = 2
    c.annotator_range = range(4)
    c.lr_schedule_dict = {0: 1e-3}
    np.random.seed(100 + ctx.rank)                       # (the reference draws the validation annotator from the global numpy state)
    model = phiseg_model.phiseg(c, dist=ctx if ctx.active else None)
    data = synthetic.SyntheticLIDC(c, seed=1234 + ctx.rank, n_validation=3)
    losses = model.train(data, num_iter=4, log_every=0, log_dir=log_dir)
    np.savez(os.path.join(out_dir, "rank%d.npz" % ctx.rank), losses=np.array(losses),
             best=np.array([model.best_dice, model.best_loss, model.best_ged, model.best_ncc]))
    ctx.barrier()
    ctx.shutdown()


if __name__ == "__main__":
    main()

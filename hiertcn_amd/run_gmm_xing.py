"""Run loop of the mixture-output models (model.py:74-77: --model_type tcn_gmm | gru_gmm) in the shape of run_nn / evaluate_nn
(run_nn_xing.py:10-98,101-192): single-level loader, no carried state, one scalar per batch -- the masked lane-mean (model.py:111-117) of
the likelihood loss args.loss in {gmm_prob, gmm_prob_max} (loss.py:247-257) -- averaged over the batches of ALL ranks (pooled once per epoch).

What the reference itself does with these models on XING cannot be run: its graph feeds the [B, T, M (1 + D)] head output to calc_loss /
calc_metric_fast against the [B, T, item_num] one-hot targets (model.py:105-132), whose last dimensions differ, and calc_loss_gmm has no
caller.  This loop therefore reports the mixture loss only; the rank metrics of results.txt do not exist for a density model and are
written as nan.  Training (--train): model_tcn_gmm and model_gru_gmm (fixed or learned scale) step through csrc/gmm_train.hip (the gradient of
this loop's own loss: the decode is the build's, so is its derivative -- checked against fp64 autograd, tests/test_gpu_gmm_train.py).

--model_type hier --model_low_type tcn_gmm | gru_gmm (model_hier.py:64-65,68-69): run_hier_gmm / evaluate_hier_gmm below, run_hier's cadence
(run_hier_xing.py:210-350: hierarchical loader, one carried state per loader) around the same scalar.  The reference's run_hier cannot build
that graph either (same shape mismatch); the loop exists so that every model of the dispatch (model.py:66-77) is reachable from run_xing.
"""
from __future__ import annotations

import time

from .run_common import SCALARS, EpochMeter, ScalarLog


def evaluate_gmm(args, loader, sess, prefix="test"):
    """-> (summary, time_load, time_run).  `sess`: a TcnGmmModel / GruGmmModel (anything with eval_step(x, y) -> {'sums': 2 doubles})."""
    from .distributed import is_main
    model = sess
    time_load = time_run = 0.0
    meter = EpochMeter(args.log_path + prefix + "/results.txt" if getattr(args, "log_path", None) and is_main() else None, final_header=False)
    nan = float("nan")

    def add(sums):
        s = [float(v) for v in sums.cpu()]
        step = {k: nan for k, _ in SCALARS}
        step["loss"] = s[0] / s[1] if s[1] > 0 else nan
        meter.add(step)

    pending = None
    while True:
        t1 = time.time()
        batch_x, batch_y, _info = loader.get_batch()
        t2 = time.time()
        sums = model.eval_step(batch_x, batch_y)["sums"]
        if pending is not None:       # a batch's two doubles are read while the next one runs
            add(pending)
        pending = sums
        t3 = time.time()
        time_load += t2 - t1
        time_run += t3 - t2
        if loader.done:
            break
    add(pending)
    # data-parallel evaluation: every rank walked its own shard of the users -- shards may differ in their number of batches, so there is
    # no collective inside the loop (a rank that finished early would leave the others waiting in it); the epoch value is the mean over
    # the batches of all ranks, pooled once here exactly as evaluate_hier / evaluate_nn pool theirs (EpochMeter.finish)
    m = meter.finish()
    return m, time_load, time_run


def run_gmm(args, loader_train, loader_validate, loader_test, sess=None):
    """run_nn's cadence (run_nn_xing.py:101-192) for --model_type tcn_gmm / gru_gmm: per epoch args.epoch_batches_train optimizer steps at the
    scheduled learning rate (run_hier_xing.py:265-269), a checkpoint every args.save_epoch, validation / test every args.test_epoch -- or one
    evaluation pass with the evaluation flags (args.py:193).  Training is built for model_tcn_gmm / model_gru_gmm
    (csrc/gmm_train.hip)."""
    from . import checkpoint
    from .run_common import scheduled_lr
    if sess is None:
        if args.model_type == "tcn_gmm":
            from .model_tcn_gmm import TcnGmmModel
            sess = TcnGmmModel(args, None, "tcn_gmm")
        else:
            from .model_gru_gmm import GruGmmModel
            sess = GruGmmModel(args, None, "rnn_gmm")
    model = sess
    epoch = 0
    if args.load:
        fname = args.load_path + str(args.load_epoch) + ".ckpt"
        try:
            params = checkpoint.load(fname)
        except FileNotFoundError:
            print("No ckpt found: {}, {}".format(fname, "train from scratch" if args.train else "evaluating the initial weights"))
        else:
            model.params = params
            model._upload(params)
            print("Model restored: {}".format(fname))
            epoch = args.load_epoch + 1
    if getattr(args, "train", False):
        model.enable_training()      # (raises for the models / options the training step is not built for)
    from .distributed import barrier, is_main, rank_world
    log = args.log_path if is_main() else args.log_path + "rank{}/".format(rank_world()[0])
    train_writer = ScalarLog(log + "train/") if args.train else None
    validate_writer = ScalarLog(log + "validate/")
    test_writer = ScalarLog(log + "test/")
    while epoch <= args.epoch_max:
        if args.train:
            lr = scheduled_lr(args, epoch)
            sums = []
            for _ in range(args.epoch_batches_train):
                batch_x, batch_y, _info = loader_train.get_batch()
                sums.append(model.train_step(batch_x, batch_y, lr)["sums"])
            import torch
            s = torch.stack(sums).cpu().numpy()                      # one read-back per epoch: the steps themselves never wait for the host
            training_loss_epoch = float((s[:, 0] / s[:, 1]).mean())
            train_writer.add_summary({"loss_epoch": training_loss_epoch}, epoch)
            if args.save and epoch % args.save_epoch == 0 and epoch != 0:
                fname = args.save_path + str(epoch) + ".ckpt"
                if is_main():
                    model.params = model.export("weights")
                    checkpoint.save(fname, model.params)
                    print("Model saved in path: %s" % fname)
                barrier()
        if (epoch % args.test_epoch == 0 and epoch != 0) or not args.train:
            if args.validate:
                summary, _, _ = evaluate_gmm(args, loader_validate, model, prefix="validate")
                validate_writer.add_summary({"loss_epoch": summary["loss_epoch"]}, epoch)
            if args.test:
                summary, t_load, t_run = evaluate_gmm(args, loader_test, model, prefix="test")
                test_writer.add_summary({"loss_epoch": summary["loss_epoch"]}, epoch)
                print("Step {}: test load {:.2f}s, test run {:.2f}s, {}".format(epoch, t_load, t_run, args.name))
            if not args.train:
                break
        epoch += 1
    if args.train:
        model.params = model.export("weights")
    return model


def _dev_batch(model, x, y, mask):
    import numpy as np
    import torch
    return tuple(torch.from_numpy(np.ascontiguousarray(v)).to(model.device) for v in (x, y, mask))


def evaluate_hier_gmm(args, loader, state, sess, prefix="test"):
    """-> (summary, time_load, time_run, state): evaluate_hier's flow (run_hier_xing.py:83-207) for a mixture low level -- packed hierarchical
    batches, the carried state handed from batch to batch ON the device, one loss scalar per batch (read back one batch late), pooled over ranks."""
    from .distributed import is_main
    model = sess
    time_load = time_run = 0.0
    meter = EpochMeter(args.log_path + prefix + "/results.txt" if getattr(args, "log_path", None) and is_main() else None, final_header=False)
    nan = float("nan")

    def add(sums):
        s = [float(v) for v in sums.cpu()]
        step = {k: nan for k, _ in SCALARS}
        step["loss"] = s[0] / s[1] if s[1] > 0 else nan
        meter.add(step)

    # the high level runs on the main handle, whose across-session scan is a persistent launch at these batch sizes: its health word is polled where
    # evaluate_hier polls it (every 20 batches, without a synchronisation) and read once at the end -- a timed-out launch never reports metrics
    from .run_hier_xing import _HealthWatch, _check_persistent_health
    health = _HealthWatch(model, "evaluation epoch '%s'" % prefix)
    pending = None
    while True:
        t1 = time.time()
        x, y, mask, _widths = loader.get_packed_batch()
        t2 = time.time()
        xd, yd, md = _dev_batch(model, x, y, mask)
        out = model.eval_step_hier(xd, yd, md, state)
        state = out["state"]
        health.after_batch()
        if pending is not None:
            add(pending)
        pending = out["sums"]
        t3 = time.time()
        time_load += t2 - t1
        time_run += t3 - t2
        if loader.done:
            break
    add(pending)
    _check_persistent_health(model, "evaluation epoch '%s'" % prefix)
    return meter.finish(), time_load, time_run, state


def run_hier_gmm(args, loader_train, loader_validate, loader_test, sess=None):
    """run_hier's cadence (run_hier_xing.py:210-350) for --model_type hier --model_low_type tcn_gmm | gru_gmm: per epoch args.epoch_batches_train
    optimizer steps of HierGmmModel.train_step_hier (both levels, one step on the mixture handle), the carried state of each loader kept on the
    device across batches, evaluations and epochs (:241-243,332,343), a checkpoint every args.save_epoch, validation / test every args.test_epoch."""
    import torch
    from . import checkpoint
    from .distributed import barrier, is_main, rank_world
    from .model_hier_gmm import HierGmmModel
    from .run_common import scheduled_lr
    model = sess if sess is not None else HierGmmModel(args, None, "hier")
    epoch = 0
    if args.load:
        fname = args.load_path + str(args.load_epoch) + ".ckpt"
        try:
            params = checkpoint.load(fname)
        except FileNotFoundError:
            print("No ckpt found: {}, {}".format(fname, "train from scratch" if args.train else "evaluating the initial weights"))
        else:
            model.set_parameters(params)
            print("Model restored: {}".format(fname))
            epoch = args.load_epoch + 1
    if getattr(args, "train", False):
        model.enable_training()
    log = args.log_path if is_main() else args.log_path + "rank{}/".format(rank_world()[0])
    train_writer = ScalarLog(log + "train/") if args.train else None
    validate_writer = ScalarLog(log + "validate/")
    test_writer = ScalarLog(log + "test/")
    HT = int(args.hidden_dim) * int(args.num_layer)
    zeros = lambda: torch.zeros((int(args.batch_size), HT), dtype=torch.float32, device=model.device)  # noqa: E731
    state, state_validate, state_test = zeros(), zeros(), zeros()
    while epoch <= args.epoch_max:
        if args.train:
            lr = scheduled_lr(args, epoch)
            sums = []
            for _ in range(args.epoch_batches_train):
                x, y, mask, _widths = loader_train.get_packed_batch()
                xd, yd, md = _dev_batch(model, x, y, mask)
                out = model.train_step_hier(xd, yd, md, state, lr)
                state = out["state"]
                sums.append(out["sums"])
            s = torch.stack(sums).cpu().numpy()                      # one read-back per epoch
            train_writer.add_summary({"loss_epoch": float((s[:, 0] / s[:, 1]).mean())}, epoch)
            if args.save and epoch % args.save_epoch == 0 and epoch != 0:
                fname = args.save_path + str(epoch) + ".ckpt"
                if is_main():
                    model.params = model.export("weights")
                    checkpoint.save(fname, model.params)
                    print("Model saved in path: %s" % fname)
                barrier()
        if (epoch % args.test_epoch == 0 and epoch != 0) or not args.train:
            if args.validate:
                summary, _, _, state_validate = evaluate_hier_gmm(args, loader_validate, state_validate, model, prefix="validate")
                validate_writer.add_summary({"loss_epoch": summary["loss_epoch"]}, epoch)
            if args.test:
                summary, t_load, t_run, state_test = evaluate_hier_gmm(args, loader_test, state_test, model, prefix="test")
                test_writer.add_summary({"loss_epoch": summary["loss_epoch"]}, epoch)
                print("Step {}: test load {:.2f}s, test run {:.2f}s, {}".format(epoch, t_load, t_run, args.name))
            if not args.train:
                break
        epoch += 1
    if args.train:
        model.params = model.export("weights")
    return model

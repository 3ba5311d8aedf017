"""Run loop of the mixture-output models (model.py:74-77: --model_type tcn_gmm | gru_gmm) in the shape of run_nn / evaluate_nn
(run_nn_xing.py:10-98,101-192): single-level loader, no carried state, one scalar per batch -- the masked lane-mean (model.py:111-117) of
the likelihood loss args.loss in {gmm_prob, gmm_prob_max} (loss.py:247-257) -- averaged over the batches of ALL ranks (pooled once per epoch).

What the reference itself does with these models on XING cannot be run: its graph feeds the [B, T, M (1 + D)] head output to calc_loss /
calc_metric_fast against the [B, T, item_num] one-hot targets (model.py:105-132), whose last dimensions differ, and calc_loss_gmm has no
caller.  This loop therefore reports the mixture loss only; the rank metrics of results.txt do not exist for a density model and are
written as nan.  Evaluation only: the head has no backward pass.
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
    while True:
        t1 = time.time()
        batch_x, batch_y, _info = loader.get_batch()
        t2 = time.time()
        sums = model.eval_step(batch_x, batch_y)["sums"]
        s = [float(v) for v in sums.cpu()]
        step = {k: nan for k, _ in SCALARS}
        step["loss"] = s[0] / s[1] if s[1] > 0 else nan
        meter.add(step)
        t3 = time.time()
        time_load += t2 - t1
        time_run += t3 - t2
        if loader.done:
            break
    # data-parallel evaluation: every rank walked its own shard of the users -- shards may differ in their number of batches, so there is
    # no collective inside the loop (a rank that finished early would leave the others waiting in it); the epoch value is the mean over
    # the batches of all ranks, pooled once here exactly as evaluate_hier / evaluate_nn pool theirs (EpochMeter.finish)
    m = meter.finish()
    return m, time_load, time_run


def run_gmm(args, loader_train, loader_validate, loader_test, sess=None):
    """run_nn's cadence (run_nn_xing.py:101-192) for --model_type tcn_gmm / gru_gmm, evaluation flow."""
    from . import checkpoint
    if getattr(args, "train", False):
        raise NotImplementedError("training of the mixture-output models is not built (no backward pass for csrc/gmm.hip): run with the "
                                  "evaluation flags (args.py:193)")
    if sess is None:
        if args.model_type == "tcn_gmm":
            from .model_tcn_gmm import TcnGmmModel
            sess = TcnGmmModel(args, None, "tcn_gmm")
        else:
            from .model_gru_gmm import GruGmmModel
            sess = GruGmmModel(args, None, "rnn_gmm")
    model = sess
    if args.load:
        fname = args.load_path + str(args.load_epoch) + ".ckpt"
        try:
            params = checkpoint.load(fname)
        except FileNotFoundError:
            print("No ckpt found: {}, evaluating the initial weights".format(fname))
        else:
            model.params = params
            model._upload(params)
            print("Model restored: {}".format(fname))
    from .distributed import is_main, rank_world
    log = args.log_path if is_main() else args.log_path + "rank{}/".format(rank_world()[0])
    validate_writer = ScalarLog(log + "validate/")
    test_writer = ScalarLog(log + "test/")
    if args.validate:
        summary, _, _ = evaluate_gmm(args, loader_validate, model, prefix="validate")
        validate_writer.add_summary({"loss_epoch": summary["loss_epoch"]}, 0)
    if args.test:
        summary, t_load, t_run = evaluate_gmm(args, loader_test, model, prefix="test")
        test_writer.add_summary({"loss_epoch": summary["loss_epoch"]}, 0)
        print("Step 0: test load {:.2f}s, test run {:.2f}s, {}".format(t_load, t_run, args.name))
    return model

"""run_hier / evaluate_hier with the reference's control flow (run_hier_xing.py:83-207, 210-350);
the TF session is replaced by a model handle (hiertcn_amd.model.HierTCNModel or anything with the
same `eval_step` / `train_step`).

What one evaluation batch does (run_hier_xing.py:112-165): loader.get_batch() -> feed the S slot
arrays, masks, y_id = concat(y) and the carried state -> fetch loss, state, the five lane-mean
metrics, ranks_float and mask_y -> accumulate; results.txt every 20 batches and at the end; stop
when loader.done.  The carried state belongs to the loader it was produced with and survives across
evaluations and epochs (:241-243,332,343).
"""
from __future__ import annotations

import os
import time
from typing import Sequence

import numpy as np

from .run_common import EpochMeter, ScalarLog, scheduled_lr, warm_start_mask


def ranks_analysis(ranks_float_np, batch_x_list: Sequence, batch_mask_y, ranks_temp, ranks_temp_count):
    """Per-position-in-session rank profile (run_hier_xing.py:11-31): position `i` of every slot
    that is at least i+1 wide contributes its column sums.  Vectorised over slots."""
    widths = np.array([np.shape(x)[1] for x in batch_x_list])
    first = np.concatenate([[0], np.cumsum(widths)[:-1]])
    col_rank = np.asarray(ranks_float_np).sum(axis=0)
    col_cnt = np.asarray(batch_mask_y).sum(axis=0)
    for w, f in zip(widths, first):
        ranks_temp[:w] += col_rank[f:f + w]
        ranks_temp_count[:w] += col_cnt[f:f + w]
    return ranks_temp, ranks_temp_count


def evaluate_hier(args, loader, state_np, sess, y_slice_np=None, topic_state_np=None, prefix="test", warm_start=None):
    """-> (summary, summary_val, time_load, time_run, state_np), the tuple of run_hier_xing.py:207.
    `sess` is the model handle.  `summary` is a dict tag -> value (the reference builds a
    tf.Summary with the same tags, :189-196); `summary_val` is the last batch's step dict."""
    # warm-start runs (args.warm_start, run_hier_xing.py:125-129): only the events after WARM_START_TIME count; the mask comes from the loader's
    # info rows, so those runs take the slot-list batches (get_batch) one blocking step at a time
    warm = bool(warm_start if warm_start is not None else getattr(args, "warm_start", False))
    model = sess
    time_load = time_run = 0.0
    from .distributed import is_main
    meter = EpochMeter(args.log_path + prefix + "/results.txt" if getattr(args, "log_path", None) and is_main() else None, final_header=True)
    ranks_temp = np.zeros(args.max_activity_len + 1)
    ranks_temp_count = np.zeros(args.max_activity_len + 1)
    step = None
    packed = hasattr(loader, "get_packed_batch") and hasattr(model, "eval_step_packed") and not warm
    pre = None
    if packed and int(getattr(args, "load_worker_num", 0) or 0) > 0:
        # the packer runs ahead on a background thread (args.load_worker_num > 0, the reference's loader-worker flag, args.py:75):
        # one Prefetcher per loader for its whole life, so no batch it has already drawn is ever dropped between evaluations
        from .data_loader import Prefetcher
        pre = getattr(loader, "_prefetcher", None)
        if pre is None:
            pre = loader._prefetcher = Prefetcher(loader.get_packed_batch, depth=2, owner=loader)
    # the device is left running (model.py, PendingEval): batch i is enqueued -- carried state device to device -- before the host turns to
    # batch i - 1's ranks and metrics.  HIERTCN_EVAL_SERIAL=1: one blocking sess.run-shaped call per batch, as before round 4.
    pipelined = packed and hasattr(model, "eval_step_packed_async") and not os.environ.get("HIERTCN_EVAL_SERIAL")
    health = _HealthWatch(model, "evaluation epoch '%s'" % prefix)
    if pipelined:
        model.eval_begin(state_np)
        pending = None

        def take(p):
            st = p[0].result()
            meter.add(st)
            return st, ranks_analysis(st["ranks_float"], [np.empty((0, w)) for w in p[1]], st["mask_y"], ranks_temp, ranks_temp_count)

        while True:
            t1 = time.time()
            x, y, mask, widths = pre.get() if pre is not None else loader.get_packed_batch()
            done = pre.done if pre is not None else loader.done
            t2 = time.time()
            nxt = (model.eval_step_packed_async(x, y, mask, widths), widths)
            health.after_batch()
            if pending is not None:
                step, (ranks_temp, ranks_temp_count) = take(pending)
            pending = nxt
            t3 = time.time()
            time_load += t2 - t1
            time_run += t3 - t2
            if done:
                break
        t2 = time.time()
        step, (ranks_temp, ranks_temp_count) = take(pending)
        state_np = model.eval_end()
        time_run += time.time() - t2
    while not pipelined:  # :112
        t1 = time.time()
        if packed:   # same batch, already in the C-ABI layout (tests/test_data_loader.py)
            x, y, mask, widths = pre.get() if pre is not None else loader.get_packed_batch()
            t2 = time.time()
            step = model.eval_step_packed(x, y, mask, widths, state_np)
            slot_shapes = [np.empty((0, w)) for w in widths]
        else:
            batch_x_list, batch_y_list, batch_mask_list, batch_info_list = loader.get_batch()
            t2 = time.time()
            if warm:
                step = model.eval_step(batch_x_list, batch_y_list, batch_mask_list, state_np, mask_warmstart=warm_start_mask(batch_info_list, train=False))
            else:
                step = model.eval_step(batch_x_list, batch_y_list, batch_mask_list, state_np)
            slot_shapes = batch_x_list
        state_np = step["state"]  # :147 -- carried into the next batch
        health.after_batch()
        meter.add(step)
        ranks_temp, ranks_temp_count = ranks_analysis(step["ranks_float"], slot_shapes, step["mask_y"], ranks_temp, ranks_temp_count)
        t3 = time.time()
        time_load += t2 - t1
        time_run += t3 - t2
        if (pre.done if pre is not None else loader.done):  # :177
            break
    _check_persistent_health(model, "evaluation epoch '%s'" % prefix)
    summary = meter.finish()
    summary_val = {k: v for k, v in step.items() if np.ndim(v) == 0}
    summary_val["ranks_by_position"] = ranks_temp / (ranks_temp_count + 1e-6)  # :183
    return summary, summary_val, time_load, time_run, state_np


class _HealthWatch:
    """The persistent launches' health word, read where the reference writes results.txt (run_hier_xing.py:170-176: every 20 batches) WITHOUT
    stalling the pipelined loop: every 20th batch an 8-byte copy of the word into pinned memory is enqueued behind the batch's launches
    (Engine.health_probe -> hiertcn_persistent_health_async); every later batch asks, without waiting, whether it has landed, and a non-zero
    word raises there -- a time-out (another process on the GPU, a CU mask) costs at most ~20 batches of metrics instead of the epoch."""
    EVERY = 20

    def __init__(self, model, what: str):
        self.eng = getattr(model, "engine", None)
        self.what, self.n, self.probe = what, 0, None

    def after_batch(self):
        if self.eng is None or not hasattr(self.eng, "health_probe"):
            return
        self.n += 1
        if self.probe is not None and self.probe.done():
            bad, self.probe = self.probe.timed_out(), None
            if bad:
                _check_persistent_health_raise(self.eng, "%s, by batch %d" % (self.what, self.n))
        if self.n % self.EVERY == 0 and self.probe is None:
            self.probe = self.eng.health_probe()


def _check_persistent_health_raise(eng, what: str):
    eng.persistent_timed_out()     # (synchronises; resets the handle's barrier counters so that it is usable again)
    raise RuntimeError(f"{what}: a persistent GRU-scan launch timed out (its workgroups were not co-resident); the states and metrics since the "
                       "last check are invalid.  On a GPU shared with another process run the scan as a chain of GEMM launches: "
                       "HIERTCN_GRU_CHAIN=1 (before the model is created) or Engine.set_tuning(gru_scan_persistent=0, lgru_persistent=0).")


def _check_persistent_health(model, what: str):
    """One read of hiertcn_persistent_health per epoch (it synchronises the device): a bounded wait inside the persistent GRU scan that ran
    out -- workgroups not co-resident because another process, rank or CU mask holds the CUs -- finishes its launch with WRONG carried
    states instead of hanging; metrics built on them must not be reported."""
    eng = getattr(model, "engine", None)
    if eng is not None and eng.persistent_timed_out():
        raise RuntimeError(f"{what}: a persistent GRU-scan launch timed out (its workgroups were not co-resident); the states and metrics of "
                           "this epoch are invalid.  On a shared GPU: HIERTCN_GRU_CHAIN=1 before the model is created, or "
                           "Engine.set_tuning(gru_scan_persistent=0, lgru_persistent=0).")


def _rank() -> int:
    from .distributed import rank_world
    return rank_world()[0]


def run_hier(args, loader_train, loader_validate, loader_test, warm_start=None, sess=None):
    """run_hier_xing.py:210-350.  `sess` = model handle; when None a HierTCNModel is created for
    `args` (glorot / zeros / gate-bias-1 initialisation, i.e. `sess.run(init)`, :219) and, with
    args.load, restored from args.load_path (:220-229).  Returns the model handle."""
    from . import checkpoint
    from .model import HierTCNModel

    from .distributed import barrier, is_main
    model = sess if sess is not None else HierTCNModel(args)
    warm = bool(warm_start if warm_start is not None else getattr(args, "warm_start", False))   # :286-289: train on the events up to WARM_START_TIME
    epoch = 0
    if args.load:  # :220-229
        fname = args.load_path + str(args.load_epoch) + ".ckpt"
        try:
            params, extra = checkpoint.load(fname, with_extra=True)
        except FileNotFoundError:   # the reference's bare `except` (:227) hides every failure; only a missing file means "from scratch"
            print("No ckpt found: {}, train from scratch".format(fname))
            epoch = 0
        else:
            model.set_parameters(params)   # a shape / name mismatch propagates: never continue on half-restored weights
            if getattr(model, "trainable", False) and "adam_m" in extra:   # optimizer slots, as tf.train.Saver keeps them
                model.engine.load_optimizer_state(extra)
            print("Model restored: {}".format(fname))
            epoch = args.load_epoch + 1
    # data-parallel runs (torch.distributed.run): every rank walks its own shard of the users (the loaders were built with
    # rank / world, run_xing.py) and steps in lockstep; rank 0 owns the files
    log = args.log_path if is_main() else args.log_path + "rank{}/".format(_rank())
    train_writer = ScalarLog(log + "train/")
    validate_writer = ScalarLog(log + "validate/")
    test_writer = ScalarLog(log + "test/")
    HT = args.hidden_dim * args.num_layer
    # :241-243: one carried state per loader, zero at every (re)start as in the reference -- the lanes of a freshly built loader
    # hold other users than the ones a checkpointed state belonged to
    state_np = np.zeros((args.batch_size, HT))
    state_np_validate = np.zeros((args.batch_size, HT))
    state_np_test = np.zeros((args.batch_size, HT))
    while epoch <= args.epoch_max:  # :261
        if args.train:
            lr = scheduled_lr(args, epoch)
            training_loss_epoch = 0.0
            if (hasattr(loader_train, "get_packed_batch") and hasattr(model, "train_step_packed_async") and not os.environ.get("HIERTCN_TRAIN_SERIAL")
                    and args.epoch_batches_train > 0 and not warm):
                # the device is left running (model.py, PendingTrain): the carried state goes device to device, a step's loss is read one step later
                model.train_begin(state_np)
                pending = None
                health = _HealthWatch(model, "training epoch %d" % epoch)   # (the training step runs persistent launches too: scan, its backward, lgru_pipe)
                for _ in range(args.epoch_batches_train):  # :275
                    x, y, mask, _widths = loader_train.get_packed_batch()
                    nxt = model.train_step_packed_async(x, y, mask, lr)
                    health.after_batch()
                    if pending is not None:
                        training_loss_epoch += float(pending.result()["loss"])
                    pending = nxt
                training_loss_epoch += float(pending.result()["loss"])
                state_np = model.train_end()
            else:
              for _ in range(args.epoch_batches_train):  # :275
                batch_x_list, batch_y_list, batch_mask_list, batch_info_list = loader_train.get_batch()
                if warm:
                    out = model.train_step(batch_x_list, batch_y_list, batch_mask_list, state_np, lr, mask_warmstart=warm_start_mask(batch_info_list, train=True))
                else:
                    out = model.train_step(batch_x_list, batch_y_list, batch_mask_list, state_np, lr)
                state_np = out["state"]
                training_loss_epoch += float(out["loss"])
            _check_persistent_health(model, "training epoch %d" % epoch)   # (the training step runs persistent launches too: scan, its backward, lgru_pipe)
            training_loss_epoch /= args.epoch_batches_train
            train_writer.add_summary({"loss_epoch": training_loss_epoch}, epoch)
            if args.save and epoch % args.save_epoch == 0 and epoch != 0:  # :317-321
                fname = args.save_path + str(epoch) + ".ckpt"
                if is_main():   # replicas are bit-identical: one writer (concurrent np.savez into one temp file would corrupt it)
                    extra = {}
                    if getattr(model, "trainable", False):
                        extra.update(model.engine.optimizer_state())
                    checkpoint.save(fname, model.params, extra=extra)  # .params reads the device masters back
                    print("Model saved in path: %s" % fname)
                barrier()
        if (epoch % args.test_epoch == 0 and epoch != 0) or not args.train:  # :324
            if args.validate:
                summary, _, _, _, state_np_validate = evaluate_hier(args, loader_validate, state_np_validate, model, prefix="validate", warm_start=warm)
                validate_writer.add_summary(summary, epoch)
            if args.test:
                summary, _, t_load, t_run, state_np_test = evaluate_hier(args, loader_test, state_np_test, model, prefix="test", warm_start=warm)
                test_writer.add_summary(summary, epoch)
                print("Step {}: test load {:.2f}s, test run {:.2f}s, {}".format(epoch, t_load, t_run, args.name))
            if not args.train:
                break  # :349 evaluate once
        epoch += 1
    return model

"""Config contract of the hot path: the reference's `args` Namespace, attribute for attribute.

Mirrors /root/reference/args.py:3-206 (flags and defaults), :208-228 (input paths), :230-291
(run name / output paths) and :293-316 (args_adjust).  Differences, all deliberate:
  * numeric flags are typed (the reference omits `type=` for most of them, args.py:13-189, so
    CLI overrides arrive as str there);
  * nothing is parsed at import time (the reference does `args = make_args()` at args.py:318);
    call make_args(argv) or use default_args();
  * `--emb_dim` is added: the reference hard-codes 128 for the item-embedding / projection width
    (model_hier.py:50,85; model_tcn.py:35); BASELINE.json's configs vary it ("d");
  * `--storage_dtype` is added: bf16 (production), fp16 (BASELINE.json configs[3]) or fp32 -- the exact evaluation mode that
    computes everything in fp32 like the reference's graph (for parity runs on trained checkpoints).
"""
from __future__ import annotations

from argparse import ArgumentParser, Namespace


def _int_list(s):
    if isinstance(s, (list, tuple)):
        return [int(v) for v in s]
    return [int(v) for v in str(s).replace("[", "").replace("]", "").split(",") if v.strip()]


def build_parser() -> ArgumentParser:
    p = ArgumentParser()
    p.add_argument("--dataset", dest="dataset", default="xing")
    p.add_argument("--warm_start", dest="warm_start", action="store_true")
    # training
    p.add_argument("--lr", dest="learning_rate", default=1e-2, type=float)
    p.add_argument("--lr_schedule_no", dest="lr_schedule", action="store_false")
    p.add_argument("--epoch_max", dest="epoch_max", default=200, type=int)
    p.add_argument("--batch_size", dest="batch_size", default=32, type=int)
    p.add_argument("--has_batchnorm", dest="has_batchnorm", action="store_true")
    p.add_argument("--has_weightnorm", dest="has_weightnorm", action="store_true")
    p.add_argument("--has_layernorm", dest="has_layernorm", action="store_true")
    # network
    p.add_argument("--model_type", dest="model_type", default="mv_xing")
    p.add_argument("--model_low_type", dest="model_low_type", default="gru")
    p.add_argument("--item_num", dest="item_num", default=20777 + 1, type=int)
    p.add_argument("--hidden_dim", dest="hidden_dim", default=128, type=int)
    p.add_argument("--num_layer", dest="num_layer", default=2, type=int)
    p.add_argument("--tcn_channel", dest="tcn_channel", default=[128, 128], type=_int_list)
    p.add_argument("--kernel_size", dest="kernel_size", default=5, type=int)
    p.add_argument("--strides", dest="strides", default=1, type=int)
    p.add_argument("--multi_num", dest="multi_num", default=3, type=int)
    p.add_argument("--dropout", dest="dropout", default=0.0, type=float)
    p.add_argument("--topic_dim", dest="topic_dim", default=50, type=int)
    p.add_argument("--sigma", dest="sigma", default=0.1, type=lambda v: None if str(v).lower() in ("none", "learn", "0", "0.0") else float(v))   # None: a learned scale
    p.add_argument("--num_mixture", dest="num_mixture", default=5, type=int)
    p.add_argument("--pred_mode", dest="pred_mode", default="gmm_score")
    p.add_argument("--mix_weight", dest="mix_weight", default=0.01, type=float)
    p.add_argument("--mv_isWeight", dest="mv_isWeight", action="store_true")
    p.add_argument("--lag", dest="lag", default=10, type=int)
    # features
    p.add_argument("--has_impression", dest="has_impression", action="store_true")
    p.add_argument("--has_gap", dest="has_gap", action="store_true")
    p.add_argument("--gap_bandwidth", dest="gap_bandwidth", default=168, type=float)
    p.add_argument("--train_gap", dest="train_gap", action="store_true")
    p.add_argument("--data_noise", dest="data_noise", default=None, type=float)
    p.add_argument("--binary_input", dest="binary_input", action="store_true")
    p.add_argument("--input_dim", dest="input_dim", default=512, type=int)
    p.add_argument("--output_dim", dest="output_dim", default=20777 + 1, type=int)
    # loss
    p.add_argument("--loss", dest="loss", default="cross_entropy")
    p.add_argument("--rank_metric", dest="rank_metric", default="l2")
    p.add_argument("--num_neg_sample", dest="num_neg_sample", default=20, type=int)
    p.add_argument("--nce_weight", dest="nce_weight", default=1, type=float)
    p.add_argument("--l2_normalize", dest="l2_normalize", action="store_true")
    p.add_argument("--hinge_delta", dest="hinge_delta", default=0.1, type=float)
    # data
    p.add_argument("--max_seq_len", dest="max_seq_len", default=500, type=int)
    p.add_argument("--max_impression_len", dest="max_impression_len", default=20, type=int)
    p.add_argument("--max_activity_len", dest="max_activity_len", default=20, type=int)
    p.add_argument("--max_session_num", dest="max_session_num", default=10, type=int)
    p.add_argument("--epoch_batches_train", dest="epoch_batches_train", default=500, type=int)
    p.add_argument("--load_worker_num", dest="load_worker_num", default=8, type=int)
    # I/O
    p.add_argument("--save", dest="save", action="store_true")
    p.add_argument("--save_no", dest="save", action="store_false")
    p.add_argument("--save_epoch", dest="save_epoch", default=20, type=int)
    p.add_argument("--test_epoch", dest="test_epoch", default=20, type=int)
    p.add_argument("--load", dest="load", action="store_true")
    p.add_argument("--load_no", dest="load", action="store_false")
    p.add_argument("--load_epoch", dest="load_epoch", default=20, type=int)
    p.add_argument("--train", dest="train", action="store_true")
    p.add_argument("--train_no", dest="train", action="store_false")
    p.add_argument("--validate", dest="validate", action="store_true")
    p.add_argument("--validate_no", dest="validate", action="store_false")
    p.add_argument("--test", dest="test", action="store_true")
    p.add_argument("--test_no", dest="test", action="store_false")
    p.add_argument("--shuffle", dest="shuffle", action="store_true")
    p.add_argument("--shuffle_no", dest="shuffle", action="store_false")
    p.add_argument("--html", dest="html", action="store_true")
    p.add_argument("--html_no", dest="html", action="store_false")
    p.add_argument("--test_random", dest="test_random", action="store_true")
    p.add_argument("--test_random_no", dest="test_random", action="store_false")
    p.add_argument("--data_type", dest="data_type", default="1m")
    p.add_argument("--filter_user", dest="filter_user", action="store_true")
    p.add_argument("--filter_user_no", dest="filter_user", action="store_false")
    p.add_argument("--load_num_train", dest="load_num_train", default=800, type=int)
    p.add_argument("--load_num_test", dest="load_num_test", default=800, type=int)
    p.add_argument("--test_data_ratio", dest="test_data_ratio", default=1, type=float)
    p.add_argument("--data_splits", dest="data_splits", default=100, type=int)
    # build-only addition (see module docstring)
    p.add_argument("--emb_dim", dest="emb_dim", default=128, type=int)
    p.add_argument("--storage_dtype", dest="storage_dtype", default="bf16", choices=["bf16", "fp16", "fp32", "split16"])
    # args.py:193 -- the "eval" default set
    p.set_defaults(save=False, load=False, train=False, validate=False, test=True, html=False,
                   test_random=False, filter_user=False, shuffle=False)
    return p


def args_make_input_path(args: Namespace) -> Namespace:
    """args.py:208-228 (Pinterest paths; unused on XING, kept because the attribute names are
    part of the contract)."""
    if args.data_type == "1m":
        args.train_path = "/data1/home/ywang/user_data_homefeed_1m_train_sort_session/"
        args.test_path = "/data1/home/ywang/user_data_homefeed_1m_test_sort_session/"
        if args.filter_user:
            args.data_splits = 20
    if args.data_type == "1m_splittime":
        args.train_path = "/data1/home/ywang/user_data_homefeed_1m_train_sort_session_splittime/"
        args.test_path = "/data1/home/ywang/user_data_homefeed_1m_test_sort_session_splittime/"
        if args.filter_user:
            args.data_splits = 20
    elif args.data_type == "10m":
        args.train_path = "/data1/home/ywang/user_data_homefeed_10m_train_sort_session/"
        args.test_path = "/data1/home/ywang/user_data_homefeed_10m_test_sort_session/"
        if not args.filter_user:
            args.load_worker_num = 1
            args.data_splits = 800
    args.train_cache_path = getattr(args, "train_path", None)
    args.test_cache_path = getattr(args, "test_path", None)
    return args


def args_make_output_path(args: Namespace) -> Namespace:
    """args.py:230-291: the run name encodes the config; log/save/load paths derive from it."""
    args.name = ("1030_final" + args.model_type + args.model_low_type + args.loss + "hinge_delta"
                 + str(args.hinge_delta) + "gap" + str(args.has_gap) + "bandwidth"
                 + str(_fmt_num(args.gap_bandwidth)) + str(args.train_gap)
                 + "impression" + str(args.has_impression) + "noise" + str(args.data_noise)
                 + "rank_metric" + args.rank_metric + "dropout" + str(args.dropout)
                 + "batchnorm" + str(args.has_batchnorm) + "weightnorm" + str(args.has_weightnorm)
                 + "kernel" + str(args.kernel_size) + "channel" + ",".join(str(i) for i in args.tcn_channel)
                 + "data" + args.data_type + "loadnum" + str(args.load_num_train) + "split"
                 + str(args.data_splits) + "filter" + str(args.filter_user))
    args.name_load = ("0913_finalcomparelosshiertcnhinge_linearhinge_delta0.1gapFalsebandwidth168False"
                      "impressionFalsenoiseNonerank_metricl2dropout0.0batchnormFalseweightnormFalse"
                      "kernel5channel128,128,128,128data1mloadnum800split100filterFalse")  # args.py:280
    if args.warm_start:
        args.name += "warm"
    args.log_path = "run/tensorboard/" + args.name + "/"
    args.html_path = "run/html/"
    args.save_path = "run/model/" + args.name + "/"
    args.load_path = "run/model/" + args.name_load + "/"
    return args


def _fmt_num(v):
    # the reference's default is the int 168 (args.py:91) and is printed as such in the name
    return int(v) if float(v).is_integer() else v


def args_adjust(args: Namespace) -> Namespace:
    """args.py:293-316."""
    if args.model_type == "gru":
        args.hidden_dim = 200
    if args.model_type == "tcn":
        args.tcn_channel = [128, 128, 128, 128, 256, 256]
    args = args_make_input_path(args)
    args = args_make_output_path(args)
    return args


def make_args(argv=None, adjust=True, **overrides) -> Namespace:
    """args.py:3-206.  `argv=None` parses an EMPTY list (not sys.argv); pass sys.argv[1:]
    explicitly from a CLI entry.  Keyword overrides are applied before args_adjust."""
    ns = build_parser().parse_args([] if argv is None else list(argv))
    for k, v in overrides.items():
        setattr(ns, k, v)
    if "item_num" in overrides and "output_dim" not in overrides:
        ns.output_dim = ns.item_num
    return args_adjust(ns) if adjust else ns


def default_args(**overrides) -> Namespace:
    return make_args([], **overrides)


def hier_tcn_args(**overrides) -> Namespace:
    """HierTCN as the paper runs it: --model_type hier --model_low_type tcn."""
    kw = dict(model_type="hier", model_low_type="tcn")
    kw.update(overrides)
    return make_args([], **kw)

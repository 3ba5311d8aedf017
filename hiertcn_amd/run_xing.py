"""CLI entry with the reference's behaviour (run_xing.py:1-45):

    python -m hiertcn_amd.run_xing --model_type hier --model_low_type tcn [flags of args.py]

uniquify the log directory, build the three loaders from dataset/interactions.csv, dispatch on
model_type.  Only the hot-path models are built: 'hier' (model_low_type 'tcn' = HierTCN, trainable; 'gru' = HierGRU,
evaluation only) and 'tcn'.
"""
from __future__ import annotations

import os
import sys

from .args import make_args
from .data_loader import Dataloader_hier_model_xing, Dataloader_single_level_model_xing


def main(argv=None):
    args = make_args(sys.argv[1:] if argv is None else argv)
    while os.path.isdir(args.log_path):  # run_xing.py:10-13
        args.log_path = args.log_path[:-1] + "1/"
        print("logdir exist, new logdir: ", args.log_path)
    os.makedirs(args.log_path)
    hier = "hier" in args.model_type
    Loader = Dataloader_hier_model_xing if hier else Dataloader_single_level_model_xing
    splits = ("train", "train", "train") if args.warm_start else ("train", "validate", "test")  # :20-35
    loader_train, loader_validate, loader_test = (Loader(args, s) for s in splits)
    if args.model_type == "tcn":
        from .run_nn_xing import run_nn
        return run_nn(args, loader_train, loader_validate, loader_test)
    if args.model_type == "hier":
        from .run_hier_xing import run_hier
        return run_hier(args, loader_train, loader_validate, loader_test)
    raise NotImplementedError(f"model_type {args.model_type!r}: only 'hier' (HierTCN) and 'tcn' are built (SURVEY.md 8)")


if __name__ == "__main__":
    main()

"""CLI entry with the reference's behaviour (run_xing.py:1-45):

    python -m hiertcn_amd.run_xing --model_type hier --model_low_type tcn [flags of args.py]

uniquify the log directory, build the three loaders from dataset/interactions.csv, dispatch on
model_type.  Only the hot-path models are built: 'hier' (model_low_type 'tcn' = HierTCN, 'gru' = HierGRU, 'tcn_gmm' / 'gru_gmm' = a mixture low
level), 'tcn', 'gru' (single level), and the mixture-output models 'tcn_gmm' / 'gru_gmm' (run_gmm_xing.py).
"""
from __future__ import annotations

import os
import sys

from .args import make_args
from .data_loader import Dataloader_hier_model_xing, Dataloader_single_level_model_xing


def main(argv=None):
    args = make_args(sys.argv[1:] if argv is None else argv)
    # under torch.distributed.run: one process per GPU, users sharded over the ranks; a plain launch is world 1
    from .distributed import init_from_env
    import torch
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    rank, world, _ = init_from_env()
    if rank == 0:
        while os.path.isdir(args.log_path):  # run_xing.py:10-13
            args.log_path = args.log_path[:-1] + "1/"
            print("logdir exist, new logdir: ", args.log_path)
        os.makedirs(args.log_path)
    if world > 1:   # one log directory for the job: rank 0 picks it
        import torch.distributed as dist
        box = [args.log_path]
        dist.broadcast_object_list(box, src=0)
        args.log_path = box[0]
    hier = "hier" in args.model_type
    Loader = Dataloader_hier_model_xing if hier else Dataloader_single_level_model_xing
    splits = ("train", "train", "train") if args.warm_start else ("train", "validate", "test")  # :20-35
    loader_train, loader_validate, loader_test = (Loader(args, s, rank=rank, world=world) for s in splits)
    if args.model_type in ("tcn", "gru"):   # model.py:66-73: the single-level models, one loop (run_nn_xing.py)
        from .run_nn_xing import run_nn
        return run_nn(args, loader_train, loader_validate, loader_test)
    if args.model_type == "hier" and args.model_low_type in ("tcn_gmm", "gru_gmm"):   # model_hier.py:64-65,68-69
        from .run_gmm_xing import run_hier_gmm
        return run_hier_gmm(args, loader_train, loader_validate, loader_test)
    if args.model_type == "hier":
        from .run_hier_xing import run_hier
        return run_hier(args, loader_train, loader_validate, loader_test)
    if args.model_type in ("tcn_gmm", "gru_gmm"):   # model.py:74-77
        from .run_gmm_xing import run_gmm
        return run_gmm(args, loader_train, loader_validate, loader_test)
    raise NotImplementedError(f"model_type {args.model_type!r}: 'hier', 'tcn', 'gru', 'tcn_gmm' and 'gru_gmm' are built (SURVEY.md 8)")


if __name__ == "__main__":
    main()

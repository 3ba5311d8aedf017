"""Capture golden config vectors from the REFERENCE's own args.py (make_args + args_adjust + the path builders).

Run in the build container only (it imports /root/reference, which never travels):
    python tests/golden/make_args_fixtures.py
Writes args_fixtures.json next to this file: for each command line below, every attribute of the Namespace the reference builds
(args.py:203-206: parse_args -> args_adjust -> input / output paths).  DATA only: no reference source is copied.
The reference parses sys.argv at import (args.py:318) and declares no `type=` for most numeric flags, so a value given on the command
line arrives there as a str; the fixture records what the reference produced, the test coerces (tests/test_args_golden.py).
"""
import importlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

CASES = {
    "defaults": [],
    "hier_tcn": ["--model_type", "hier", "--model_low_type", "tcn"],
    "hier_gru": ["--model_type", "hier", "--model_low_type", "gru"],
    "single_tcn": ["--model_type", "tcn"],
    "single_gru": ["--model_type", "gru"],
    "tcn_gmm": ["--model_type", "tcn_gmm", "--loss", "gmm_prob"],
    "gru_gmm": ["--model_type", "gru_gmm", "--loss", "gmm_prob_max"],
    "overrides": ["--model_type", "hier", "--model_low_type", "tcn", "--lr", "0.003", "--batch_size", "64", "--hidden_dim", "256", "--num_layer", "3",
                  "--kernel_size", "3", "--epoch_max", "10", "--has_gap", "--warm_start", "--dropout", "0.2"],
}


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    out = {}
    for name, argv in CASES.items():
        sys.argv = ["run_xing.py"] + argv
        sys.modules.pop("args", None)
        mod = importlib.import_module("args")      # args.py:318 builds `args` at import
        ns = vars(mod.args)
        out[name] = {"argv": argv, "namespace": {k: ns[k] for k in sorted(ns)}}
    with open(os.path.join(HERE, "args_fixtures.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", len(out), "cases;", len(out["defaults"]["namespace"]), "attributes each")


if __name__ == "__main__":
    main()

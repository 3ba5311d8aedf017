"""The config contract (SURVEY.md 8(b) "Config") against vectors captured from the REFERENCE's own args.py (tests/golden/make_args_fixtures.py
ran make_args -> args_adjust -> the path builders there for eight command lines): every attribute the reference's Namespace carries exists
here under the same name with the same value -- defaults, the args_adjust rules (args.py:306-311), the run name and the output paths
(args.py:230-291) character for character.  Deliberate differences, all stated in hiertcn_amd/args.py: numeric flags are typed here (the
reference declares no `type=`, so a value from the command line stays a str there: compared after coercion), and --emb_dim /
--storage_dtype are additions."""
import json
import os

import pytest

from hiertcn_amd.args import make_args

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "args_fixtures.json")))
ADDED = {"emb_dim", "storage_dtype"}   # this build's additions (hiertcn_amd/args.py header)


def _same(mine, ref):
    if isinstance(ref, str) and not isinstance(mine, str):   # a CLI override of an untyped flag: '64' there, 64 here
        return type(mine)(ref) == mine if not isinstance(mine, bool) else str(mine) == ref
    if isinstance(ref, (list, tuple)):
        return list(mine) == list(ref)
    if isinstance(ref, float) or isinstance(mine, float):
        return mine is not None and ref is not None and float(mine) == float(ref)
    return mine == ref


@pytest.mark.parametrize("case", sorted(FIX))
def test_namespace_equals_the_reference(case):
    ref = FIX[case]["namespace"]
    mine = vars(make_args(FIX[case]["argv"]))
    missing = sorted(set(ref) - set(mine))
    assert not missing, f"attributes of the reference's args that this build lacks: {missing}"
    extra = sorted(set(mine) - set(ref) - ADDED)
    assert not extra, f"attributes this build adds without saying so: {extra}"
    wrong = {k: (mine[k], ref[k]) for k in ref if not _same(mine[k], ref[k])}
    assert not wrong, wrong


def test_the_fixture_covers_the_adjust_rules_and_the_name():
    assert FIX["single_tcn"]["namespace"]["tcn_channel"] == [128, 128, 128, 128, 256, 256]     # args.py:310-311
    assert FIX["single_gru"]["namespace"]["hidden_dim"] == 200                                  # args.py:306-307
    assert FIX["overrides"]["namespace"]["name"].endswith("warm")                               # args.py:282-283
    assert len(FIX["defaults"]["namespace"]) >= 70


def test_sigma_none_selects_the_learned_scale():
    """`if not args.sigma` (model_tcn_gmm.py:39, model_gru_gmm.py:50) is the reference's switch for a learned scale; the reference's untyped
    flag can only express it by editing the default -- here `--sigma none` (or 0) parses to None, any number to a float."""
    assert make_args(["--sigma", "none"]).sigma is None and make_args(["--sigma", "0"]).sigma is None
    assert make_args(["--sigma", "0.25"]).sigma == 0.25 and make_args([]).sigma == 0.1

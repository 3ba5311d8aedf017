"""Capture golden loader vectors from the REFERENCE's own data_prepare.py / data_loader.py.

Run in the build container only (it imports /root/reference, which never travels):
    python tests/golden/make_loader_fixtures.py
Writes, next to this file:
    interactions_synth.csv     the synthetic XING-format input (tab separated, README.md:11)
    loader_fixtures.npz        load_xing outputs + get_batch() sequences of both loaders

What is captured is DATA (inputs and the reference's outputs); no reference source is copied.
`fastparquet` is imported but never used by the two reference files (data_prepare.py:1,
data_loader.py:1) and is not installed here; an empty placeholder module object satisfies the
import (SURVEY.md 8(c) [probe]).
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def make_csv(path, seed=0):
    rng = np.random.default_rng(seed)
    rows = []
    n_items_hot, n_items_cold = 40, 25  # cold items have < 50 events and must be filtered out
    t0 = 1_445_000_000
    for u in range(90):
        if u % 11 == 0:
            n = int(rng.integers(2, 9))  # < 10 events: filtered user
        elif u % 17 == 0:
            n = 700  # > 500 events after the type-4 filter: filtered user
        else:
            n = int(rng.integers(12, 70))
        t = t0 + int(rng.integers(0, 10_000))
        for _ in range(n):
            gap = int(rng.integers(5, 900)) if rng.random() < 0.8 else int(rng.integers(1800, 90_000))
            t += gap
            if rng.random() < 0.97:
                item = 1000 + (int(rng.zipf(1.3)) + int(rng.integers(0, 3)) * 13) % n_items_hot
            else:
                item = 5000 + int(rng.integers(0, n_items_cold))
            itype = int(rng.choice([1, 2, 3, 4, 5], p=[0.6, 0.15, 0.1, 0.1, 0.05]))
            rows.append((1000 + u * 7, item, itype, t))
    rows = [rows[i] for i in rng.permutation(len(rows))]  # the reference sorts; start unsorted
    with open(path, "w") as f:
        f.write("user_id\titem_id\tinteraction_type\tcreated_at\n")
        for r in rows:
            f.write("\t".join(str(v) for v in r) + "\n")


def main():
    csv_path = os.path.join(HERE, "interactions_synth.csv")
    make_csv(csv_path)
    sys.modules.setdefault("fastparquet", types.SimpleNamespace(ParquetFile=None))
    import matplotlib
    matplotlib.use("Agg")
    sys.argv = ["fixture"]  # args.py:318 parses sys.argv at import
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    cwd = os.getcwd()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "dataset"))
        import shutil
        shutil.copy(csv_path, os.path.join(tmp, "dataset", "interactions.csv"))  # data_loader.py:54,140
        os.chdir(tmp)
        try:
            import args as ref_args_mod
            import data_loader as ref_loader
            import data_prepare as ref_prepare
            a = ref_args_mod.args
            vt, vd = ref_prepare.load_xing("dataset/interactions.csv")
            out["viewer_table"] = np.asarray(vt, dtype=np.float64)
            out["viewer_data"] = np.asarray(vd, dtype=np.float64)
            # hier loader: B=4 lanes, S=3 slots, max 6 events/session
            a.batch_size, a.max_session_num, a.max_activity_len, a.shuffle = 4, 3, 6, False
            for split in ("train", "test"):
                ld = ref_loader.Dataloader_hier_model_xing(a, split)
                out[f"hier_{split}_table"] = np.asarray(ld.viewer_table, dtype=np.float64)
                n_batches = 14 if split == "train" else 16
                dones = []
                for b in range(n_batches):
                    x, y, m, info = ld.get_batch()
                    dones.append(ld.done)
                    for s in range(a.max_session_num):
                        out[f"hier_{split}_b{b}_x{s}"] = x[s]
                        out[f"hier_{split}_b{b}_y{s}"] = y[s]
                        out[f"hier_{split}_b{b}_m{s}"] = m[s]
                        out[f"hier_{split}_b{b}_i{s}"] = info[s]
                out[f"hier_{split}_done"] = np.array(dones)
            # single-level loader: B=5
            a.batch_size = 5
            ld = ref_loader.Dataloader_single_level_model_xing(a, "train")
            dones = []
            for b in range(16):
                x, y, info = ld.get_batch()
                dones.append(ld.done)
                out[f"single_b{b}_x"] = x
                out[f"single_b{b}_y"] = y
                out[f"single_b{b}_i"] = info
            out["single_done"] = np.array(dones)
        finally:
            os.chdir(cwd)
    out["meta"] = np.array([4, 3, 6, 5, 14, 16, 16])  # B_hier, S, max_act, B_single, n_train, n_test, n_single
    np.savez_compressed(os.path.join(HERE, "loader_fixtures.npz"), **out)
    print("viewer_table", out["viewer_table"].shape, "viewer_data", out["viewer_data"].shape,
          "items", int(out["viewer_data"][:, 1].max()) + 1)
    print("hier train done flags", out["hier_train_done"].astype(int))
    print("single done flags", out["single_done"].astype(int))


if __name__ == "__main__":
    main()

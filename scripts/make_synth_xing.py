"""A synthetic `dataset/interactions.csv` in the XING format (README.md:11 of the reference: tab-separated
user_id, item_id, interaction_type, created_at) for end-to-end runs of `python -m hiertcn_amd.run_xing`:
Zipf item popularity, 12-120 events per user, 80 % short gaps / 20 % session breaks (> 30 min), 10 % type-4 rows.
usage: python scripts/make_synth_xing.py <users> <out_dir>   (writes <out_dir>/dataset/interactions.csv)"""
import os
import sys

import numpy as np

users, out_dir = int(sys.argv[1]), sys.argv[2]
os.makedirs(os.path.join(out_dir, "dataset"), exist_ok=True)
rng = np.random.default_rng(0)
n_items = 20000
w = 1 / np.arange(1, n_items + 1) ** 1.05
cdf = np.cumsum(w / w.sum())
t0 = 1_445_000_000
with open(os.path.join(out_dir, "dataset", "interactions.csv"), "w") as f:
    f.write("user_id\titem_id\tinteraction_type\tcreated_at\n")
    for u in range(users):
        n = int(rng.integers(12, 120))
        gaps = np.where(rng.random(n) < 0.8, rng.integers(5, 900, n), rng.integers(1800, 90000, n))
        t = t0 + np.cumsum(gaps)
        items = 1000 + np.searchsorted(cdf, rng.random(n))
        ty = rng.choice([1, 2, 3, 4, 5], size=n, p=[0.6, 0.15, 0.1, 0.1, 0.05])
        f.write("".join(f"{1000 + u * 7}\t{items[i]}\t{ty[i]}\t{t[i]}\n" for i in range(n)))

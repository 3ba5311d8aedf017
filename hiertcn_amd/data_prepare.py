"""XING ingestion: `interactions.csv` -> (viewer_table, viewer_data), the arrays the loaders walk.

Stands where /root/reference/data_prepare.py:15-93 (`load_xing`) stands and produces the same two
arrays bit for bit (pinned by tests/golden/loader_fixtures.npz, which was captured from the
reference's own loader), but is written as whole-array NumPy passes: the reference walks every
event in Python three times (item re-index :56-57, session split :70-79, masks :34-35,50-51);
here the filters are boolean masks over counts, the re-index is a `searchsorted`, and the session
index is a segmented cumulative sum.

Output contract:
  viewer_data  float64 [events, 5] = user_id, item_id (re-indexed 1..N-1), interaction_type,
               created_at, session index within the user; sorted by (user_id, created_at)
  viewer_table int64 [users, 3]    = first row, number of events, user_id
"""
from __future__ import annotations

import numpy as np

SESSION_GAP_S = 1800  # data_prepare.py:74: a gap of >= 30 min starts a new session


def read_interactions(path: str) -> np.ndarray:
    """Tab-separated, header row, columns user_id item_id interaction_type created_at
    (data_prepare.py:17, README.md:11) -> int64 [events, 4]."""
    import pandas as pd  # host-side only; the GPU path never imports pandas

    df = pd.read_csv(path, sep="\t")
    return df[["user_id", "item_id", "interaction_type", "created_at"]].to_numpy(dtype=np.int64)


def _runs(sorted_keys: np.ndarray):
    """start, length, key of each run of equal keys in an already sorted vector."""
    keys, start, count = np.unique(sorted_keys, return_index=True, return_counts=True)
    return start, count, keys


def prepare_events(ev: np.ndarray, min_activity=10, max_activity=500, min_item=50, max_item=1e7):
    """int64 [events, 4] -> (viewer_table, viewer_data) as documented above."""
    ev = ev[ev[:, 2] != 4]                                    # data_prepare.py:21 drop 'delete' interactions
    # items with 50 <= events <= 1e7 survive (:24-36).  The reference sorts by item first with
    # pandas' default (numpy introsort); the later (user, time) sort is stable, so that first order
    # decides how exact (user, time) ties are laid out -- reproduce both steps with the same sorts.
    ev = ev[np.argsort(ev[:, 1], kind="quicksort")]
    start, count, _ = _runs(ev[:, 1])
    keep_item = (count >= min_item) & (count <= max_item)
    ev = ev[np.repeat(keep_item, count)]
    # users with 10 <= events <= max_seq_len survive (:39-52), counted AFTER the item filter
    ev = ev[np.lexsort((ev[:, 3], ev[:, 0]))]                 # stable, primary user_id, secondary created_at
    start, count, _ = _runs(ev[:, 0])
    keep_user = (count >= min_activity) & (count <= max_activity)
    ev = ev[np.repeat(keep_user, count)]
    # surviving items are re-indexed 1..N-1 in id order; 0 stays the null id (:54-57)
    items = np.unique(ev[:, 1])
    ev[:, 1] = np.searchsorted(items, ev[:, 1]) + 1
    start, count, users = _runs(ev[:, 0])
    viewer_table = np.stack([start, count, users], axis=1).astype(np.int64)
    # session index within the user: +1 at every gap >= 30 min that is not a user boundary (:66-79)
    n = ev.shape[0]
    new_session = np.zeros(n, np.int64)
    if n > 1:
        new_session[1:] = (ev[1:, 3] - ev[:-1, 3]) // SESSION_GAP_S > 0
    new_session[start] = 0
    running = np.cumsum(new_session)
    session = running - np.repeat(running[start], count)
    viewer_data = np.concatenate([ev.astype(np.float64), session[:, None].astype(np.float64)], axis=1)
    return viewer_table, viewer_data


def load_xing(path, min_activity=10, max_activity=500, min_item=50, max_item=1e7):
    """data_prepare.py:15 signature (max_activity defaults to args.max_seq_len = 500, args.py:126)."""
    return prepare_events(read_interactions(path), min_activity, max_activity, min_item, max_item)

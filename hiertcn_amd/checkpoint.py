"""Checkpoints keyed by the reference's TF variable names (`hier/emb/kernel`,
`hier/tcn/temporal_conv_net/tblock_0/conv1/kernel`, `hier/multi_rnn_cell/cell_0/gru_cell/gates/kernel`
...), so that weights exported from a TF-1.x run of the reference (tf.train.Saver, model.py:146-147;
saved at run_hier_xing.py:317-321 as `<save_path><epoch>.ckpt`) map one-to-one.

Format: one `.npz` per checkpoint (`<name>.ckpt.npz`), every variable under its TF name in the
reference's layout (fp32).  Two things the reference forgets are handled here: the directory is
created (save_path is never makedirs'd there) and the carried GRU state can ride along under
`__extra__/state_np` (it is re-zeroed on restart in the reference, run_hier_xing.py:241).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np

EXTRA = "__extra__/"


def _path(fname: str) -> str:
    return fname if fname.endswith(".npz") else fname + ".npz"


def save(fname: str, params: Dict[str, np.ndarray], extra: Optional[Dict[str, np.ndarray]] = None) -> str:
    path = _path(fname)
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    blob = {k: np.asarray(v, dtype=np.float32) for k, v in params.items()}
    for k, v in (extra or {}).items():
        blob[EXTRA + k] = np.asarray(v)
    tmp = path + ".tmp.npz"
    np.savez(tmp, **blob)
    os.replace(tmp, path)  # a crash mid-write never leaves a truncated checkpoint behind
    return path


def load(fname: str, with_extra: bool = False):
    """`fname` = what was passed to save() -- or the prefix of a checkpoint written by TensorFlow's own tf.train.Saver
    (`<prefix>.index` + `<prefix>.data-*`: hiertcn_amd.tf_checkpoint reads it without TensorFlow; Adam slots are skipped)."""
    if not os.path.exists(_path(fname)) and os.path.exists(fname + ".index"):
        from .tf_checkpoint import read_tf_checkpoint
        allv = read_tf_checkpoint(fname)
        params = {k: v for k, v in allv.items() if v.dtype == np.float32 and k != "global_step"}
        skipped = sorted(set(allv) - set(params))
        if skipped:   # said out loud: a model variable that arrived as a non-float tensor would otherwise vanish silently
            print("checkpoint %s: %d non-float32 variable(s) not loaded: %s" % (fname, len(skipped), ", ".join(skipped)))
        return (params, {}) if with_extra else params
    with np.load(_path(fname)) as z:
        params = {k: z[k] for k in z.files if not k.startswith(EXTRA)}
        extra = {k[len(EXTRA):]: z[k] for k in z.files if k.startswith(EXTRA)}
    return (params, extra) if with_extra else params

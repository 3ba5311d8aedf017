"""Capture golden vectors of the REFERENCE's own `ranks_analysis` (run_hier_xing.py:11-31), the one NumPy-only function of the run loop.

Run in the build container only (it reads /root/reference, which never travels):
    python tests/golden/make_run_loop_fixtures.py
Writes run_loop_fixtures.npz next to this file: for each case the slot widths, ranks_float, mask_y, the accumulators at entry and the
reference's accumulators at exit.

`import run_hier_xing` would import TensorFlow (absent here), so the function is taken out of the file's syntax tree and compiled on its own
-- the reference's code, executed as it is; nothing of it is stored: what is captured is DATA (inputs and the reference's outputs).
"""
import ast
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/run_hier_xing.py"


def reference_function(name):
    tree = ast.parse(open(REF).read())
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = {"np": np}
    exec(compile(ast.Module(body=[node], type_ignores=[]), REF, "exec"), ns)   # noqa: S102  (the reference's own function)
    return ns[name]


def main():
    ranks_analysis = reference_function("ranks_analysis")
    rng = np.random.default_rng(11)
    out = {}
    cases = [[3, 5, 2], [1], [7, 7, 7, 7], [2, 1, 9, 4, 1, 6], [20] * 10, [1, 1, 1]]
    for c, widths in enumerate(cases):
        B = int(rng.integers(1, 9))
        T = sum(widths)
        n = max(widths) + int(rng.integers(0, 4))                     # the run loop's accumulators are max_activity_len + 1 long
        rf = rng.random((B, T)) * 100
        mk = (rng.random((B, T)) > 0.3).astype(np.float64)
        acc0, cnt0 = rng.random(n) * 5, np.floor(rng.random(n) * 9)      # non-zero at entry: the function accumulates across batches
        xs = [np.zeros((B, w, 1)) for w in widths]                       # only .shape[1] is read (:21)
        acc, cnt = ranks_analysis(rf, xs, mk, acc0.copy(), cnt0.copy())
        out.update({f"c{c}_widths": np.array(widths), f"c{c}_ranks_float": rf, f"c{c}_mask_y": mk, f"c{c}_acc_in": acc0, f"c{c}_cnt_in": cnt0,
                    f"c{c}_acc_out": acc, f"c{c}_cnt_out": cnt})
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "run_loop_fixtures.npz"), **out)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()

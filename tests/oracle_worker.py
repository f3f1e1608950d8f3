"""Worker process of tests.helpers.oracle_box_rows: `python -m tests.oracle_worker <dir> <worker> <workers>` solves rows
worker, worker + workers, ... of <dir>/job.npz with oracle.qp.solve_exact_box and writes <dir>/out<worker>.npz.  Test
infrastructure only; never touches the GPU."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import qp as oqp  # noqa: E402


def main(d, w, nw):
    with np.load(os.path.join(d, "job.npz")) as f:
        Ps, tq, nu, N, x0, lb, ub = f["Ps"], f["tq"], int(f["nu"]), int(f["N"]), f["x0"], f["lb"], f["ub"]
    out = {"rows": np.arange(w, x0.shape[0], nw)}
    for i in out["rows"]:
        info = {"nu": nu}
        out[f"x{i}"] = oqp.solve_exact_box(Ps, tq @ x0[i], np.tile(lb[i], N), np.tile(ub[i], N), info=info)
        out[f"a{i}"] = np.asarray(info["active"], dtype=np.int64)
    np.savez(os.path.join(d, f"out{w}.npz"), **out)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))

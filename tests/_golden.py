"""Loader for tests/golden/*.npz (see tests/golden/make_golden.py for provenance)."""
import glob
import os

import numpy as np

from pagan2_msa_amd import abi

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(HERE, "*.npz")))


def load(name):
    d = np.load(os.path.join(HERE, name + ".npz"))

    def graph(p):
        return abi.Graph(d[p + "state"], d[p + "bwd_off"], d[p + "bwd_src"], d[p + "bwd_logw"], d[p + "bwd_eid"],
                         n_edges=int(d[p + "n_edges"]))
    model = abi.Model(d["table"], *d["params"])
    band = abi.Band(d["upper"], d["lower"]) if "upper" in d else None
    return graph("l_"), graph("r_"), model, band, int(d["flags"]), d


def check(result, d, what=""):
    assert result.status == int(d["status"]), what
    assert np.float64(result.score).tobytes() == np.float64(d["score"]).tobytes(), what + " score"
    if result.status == 0:
        assert tuple(result.end) == tuple(int(x) for x in d["end"]), what + " end cell"
    assert np.array_equal(result.cols, d["cols"]), what + " columns"
    assert np.array_equal(result.left_used, d["left_used"]) and np.array_equal(result.right_used, d["right_used"]), what
    assert result.cells == int(d["cells"])

"""GPU (-m gpu): randomised parity sweep -- tools/fuzz_parity.py -- over awkward shapes (empty matrices, one column, one
dense row among empty ones, repeated column indices inside rows, unsorted rows, hub columns, every bin) through every
entry point of the path (one-shot, host arrays, classification handed back in, sharded over logical shards, fused R-MCL
step), each against the CPU oracle.  4 000 cases over ten seeds ran clean when this was written; two fixed seeds here."""
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [12, 13])
def test_random_shapes_through_every_entry_point(seed):
    import __graft_entry__ as ge
    ge.build()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "150", str(seed)],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert f"fuzz ok: 150 cases, seed {seed}" in out.stdout

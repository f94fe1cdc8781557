"""PHX_DETERMINISTIC=1: every cross-block floating-point reduction of the training step takes a fixed summation order, so two
runs from the same state are bit-identical -- loss terms, gradients, parameters after Adam (lr = 1e-3, where the default mode's
atomics make two runs drift apart within a step or two).  The default mode is also run twice to show the test can tell."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digest(case, dtype, steps, det):
    env = dict(os.environ, PYTHONPATH=ROOT, PHX_DETERMINISTIC="1" if det else "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "det_worker.py"), case, dtype, str(steps)], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1].split()
    return line[1], float(line[2])


@pytest.mark.parametrize("case,dtype", [("tiny_phiseg_bn", "f32"), ("lidc_phiseg_bn", "bf16"), ("tiny_phiseg_gn4", "f32"),
                                        ("tiny_probunet_bn", "bf16")])
def test_two_runs_are_bit_identical(case, dtype):
    a = _digest(case, dtype, 4, True)
    b = _digest(case, dtype, 4, True)
    assert a == b, (a, b)

"""PHX_DETERMINISTIC=1: every cross-block floating-point reduction of the training step takes a fixed summation order, so two
runs from the same state are bit-identical -- loss terms, gradients, parameters after Adam (lr = 1e-3, where the default mode's
atomics make two runs drift apart within a step or two).  The default mode is also run twice to show the test can tell."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digest(case, dtype, steps, det, batch=0, **extra_env):
    env = dict(os.environ, PYTHONPATH=ROOT, PHX_DETERMINISTIC="1" if det else "0", **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "det_worker.py"), case, dtype, str(steps), str(batch)], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1].split()
    return line[1], float(line[2])


@pytest.mark.parametrize("case,dtype", [("tiny_phiseg_bn", "f32"), ("lidc_phiseg_bn", "bf16"), ("tiny_phiseg_gn4", "f32"),
                                        ("tiny_probunet_bn", "bf16")])
def test_two_runs_are_bit_identical(case, dtype):
    a = _digest(case, dtype, 4, True)
    b = _digest(case, dtype, 4, True)
    assert a == b, (a, b)


def test_unmaterialised_activations_are_bit_identical_to_the_plain_plan():
    """conv2d -> batch_norm -> relu -> conv2d edges of the 32-channel 128 x 128 level (round 5, engine XfBuf): the apply pass and the
    activation tensor are not made, the readers' forward / filter-gradient kernels re-form the activation in their staged patches
    (phx_conv3x3_mfma_bf16_xf, phx_conv3x3_wgrad_mfma_bf16_partial_xf) -- with the SAME roundings the apply pass has, so two training
    steps of phiseg_7_5 (n0 = 32, 128 x 128, bf16, batch norm) at BATCH 64 give the same digest of every loss term, gradient and
    parameter with the rewrite on (default) and off (PHX_XF=0) in the deterministic mode."""
    on = _digest("lidc_phiseg_bn", "bf16", 2, True, batch=64, PHX_XF="1")
    off = _digest("lidc_phiseg_bn", "bf16", 2, True, batch=64, PHX_XF="0")
    assert on == off, (on, off)

"""CPU oracle for the PHiSeg ELBO hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU (numpy + torch-CPU, fp64 or fp32), the algorithm
of the reference's hot path (baumgach/PHiSeg-code, TF 1.12 graph mode):

* ``philox``   -- Philox4x32-10 counter RNG + Box-Muller (the build's RNG contract;
                  the reference never seeds ``tf.random_normal``).
* ``tf1_ops``  -- the TF 1.12 primitive semantics the reference calls into
                  (tfwrapper/layers.py, tfwrapper/normalisation.py, phiseg_model.py).
* ``nets``     -- posterior / prior / likelihood (phiseg and prob_unet2D) + ELBO.
* ``train``    -- loss, autograd gradients, TF1-form Adam.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it; the product package (``phiseg_code_amd``) never does.

Parity pinning: the reference has NO tests / golden vectors for this path and its
arithmetic lives in TensorFlow 1.12 (not installed, not vendored).  The oracle is
pinned instead by executing the reference's OWN ``phiseg/model_zoo/*.py`` and the loss
methods of ``phiseg/phiseg_model.py`` (imported unmodified from /root/reference in the
build container) on top of a ``tensorflow`` API shim (``tools/tf1_shim``) whose
primitives are ``oracle.tf1_ops``; outputs are committed under ``tests/golden/`` by
``tools/make_goldens.py``.  So network structure / loss formulas are pinned by the
reference's code; the TF-1.12 primitive semantics themselves (legacy bilinear resize,
fused batch norm, epsilon-hat Adam) are restated from the published TF 1.12 behaviour
and cross-checked against torch-CPU where torch has the same op: **primitive-level
parity is unpinned by the reference** (it cannot be executed here or on the GPU box).
"""

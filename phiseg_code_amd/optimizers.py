"""Optimiser selectors named like the TF classes the reference's experiment configs reference
(phiseg/experiments/*.py:37, phiseg_model.py:137-140).  The update itself is the fused HIP kernel
phx_adam_tf1 over the flat parameter arena (TF 1.12 epsilon-hat Adam)."""


class AdamOptimizer:
    beta1, beta2, epsilon = 0.9, 0.999, 1e-8

    def __init__(self, learning_rate=1e-3):
        self.learning_rate = learning_rate


class MomentumOptimizer:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("MomentumOptimizer is not selected by any PHiSeg experiment (phiseg_model.py:137-138)")

"""phiseg_code_amd -- MI355X-native engine for the PHiSeg ELBO hot path.

Layout (mirrors the reference's top-level packages so it is a drop-in for that path):
  tfwrapper/   layers, normalisation, utils      (reference: tfwrapper/)
  phiseg/      model_zoo, experiments, phiseg_model   (reference: phiseg/)
  config/      system                             (reference: config/)
  graph.py engine.py runtime.py distributed.py    symbolic graph -> HIP launch plan -> hipGraph
  csrc/        hand-written HIP kernels + the C ABI (include/phx.h) -> libphx.so
"""
import os
import sys

# The plan replays one hipGraph whose branches ("lanes", engine.Plan) run on up to 6 streams next to torch's own; the HIP
# runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), read once when the runtime starts.
# Measured on MI355X (bench.py, 6 lanes): 8 queues 16.4-16.7 ms/step, 4 queues 16.9-17.0.  Only a default: the user's
# setting wins, and it has no effect if HIP was initialised before this package was imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def install_dropin_aliases():
    """Make ``import tfwrapper.layers`` / ``from phiseg.model_zoo import posteriors`` / ``import config.system``
    / ``from data.data_switch import data_switch`` / ``import utils`` resolve to this package, the way scripts written against the
    reference import them."""
    import importlib
    for sub in ("data_switch", "lidc_data", "batch_provider", "lidc_data_loader"):      # (from data.data_switch import data_switch)
        importlib.import_module("phiseg_code_amd.data." + sub)
    for top in ("tfwrapper", "phiseg", "config", "data", "utils"):
        mod = importlib.import_module("phiseg_code_amd." + top)
        sys.modules.setdefault(top, mod)
        for name, m in list(sys.modules.items()):
            if name.startswith("phiseg_code_amd.%s." % top):
                sys.modules.setdefault(name[len("phiseg_code_amd."):], m)

"""Site configuration -- counterpart of the reference's config/system.py (paths + GPU environment).

The reference hard-codes ETH cluster paths and reads $SGE_GPU at import time (config/system.py:16-39); here
the log root comes from $PHISEG_LOG_ROOT (default ./logs) and device selection from LOCAL_RANK (one process
per GPU, launched by torch.distributed.run)."""
import os

project_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
log_root = os.environ.get("PHISEG_LOG_ROOT", os.path.join(os.getcwd(), "logs"))


def setup_GPU_environment():
    """Bind this process to its GPU (LOCAL_RANK) -- replaces the CUDA_VISIBLE_DEVICES juggling."""
    import torch
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())

"""bf16 vs fp32 convergence, many noise seeds (GPU box).  Runs tests/convergence_worker.py for SEEDS Philox noise seeds per dtype in the
mode the environment selects (default: the benchmarked atomics mode; PHX_DETERMINISTIC=1 for the ordered one), PAR workers side by
side, and prints per loss term: mean over seeds, standard error, bf16 / fp32 and the standard error of that ratio.

    python tools/convergence_study.py [SEEDS=8] [STEPS=200] [TAIL=50] [PAR=8] [arms=f32,bf16]

An arm is a dtype, optionally with environment overrides for its workers: "bf16@PHX_BN_SMALL_F32=0" (the first arm is the reference
of the ratios).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.convergence_lib import run_all, summarise  # noqa: E402


if __name__ == "__main__":
    a = sys.argv[1:]
    seeds = int(a[0]) if len(a) > 0 else 8
    steps = int(a[1]) if len(a) > 1 else 200
    tail = int(a[2]) if len(a) > 2 else 50
    par = int(a[3]) if len(a) > 3 else 8
    dts = tuple(a[4].split(",")) if len(a) > 4 else ("f32", "bf16")
    jobs = [(dt, so) for so in range(seeds) for dt in dts]
    res = run_all(jobs, par, steps, tail)
    print("mode:", "deterministic" if os.environ.get("PHX_DETERMINISTIC", "0") not in ("0", "") else "default (atomics)",
          " steps", steps, " tail", tail, " extra env:", {k: v for k, v in os.environ.items() if k.startswith("PHX_")})
    summarise(res, dts)
    print("STUDY " + json.dumps({"%s/%d" % k: dict(first=v["first"], tail=v["tail"]) for k, v in res.items()}))

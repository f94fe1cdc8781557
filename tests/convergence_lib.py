"""Runner / statistics of the bf16-vs-fp32 convergence experiments (tests/test_model_gpu.py, tools/convergence_study.py): starts
tests/convergence_worker.py processes side by side -- one (arm, noise seed) each -- and summarises the means over the last steps.
An arm is a dtype, optionally with environment overrides for its workers ("bf16@PHX_DETERMINISTIC=1")."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_all(jobs, par, steps, tail):
    env = dict(os.environ, PYTHONPATH=ROOT)
    out, running = {}, []
    jobs = list(jobs)
    while jobs or running:
        while jobs and len(running) < par:
            dt, so = jobs.pop(0)
            log = tempfile.TemporaryFile(mode="w+")
            wenv = dict(env)
            for kv in dt.split("@")[1:]:
                k, v = kv.split("=", 1)
                wenv[k] = v
            pr = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "convergence_worker.py"), str(steps), str(tail), dt.split("@")[0], str(so)],
                                  env=wenv, cwd=ROOT, stdout=log, stderr=subprocess.STDOUT, text=True)
            running.append((dt, so, log, pr))
        dt, so, log, pr = running.pop(0)
        pr.wait(timeout=3000)
        log.seek(0)
        txt = log.read()
        log.close()
        if pr.returncode != 0:
            print("worker failed", dt, so, txt[-2000:])
            continue
        out[(dt, so)] = json.loads([l for l in txt.splitlines() if l.startswith("CONVERGENCE ")][-1][len("CONVERGENCE "):])
    return out


def summarise(res, dts=("f32", "bf16")):
    keys = next(iter(res.values()))["keys"]
    tails = {dt: np.array([r["tail"] for (d, _), r in sorted(res.items()) if d == dt]) for dt in dts}
    kl = [i for i, k in enumerate(keys) if k.startswith("KL")]
    ce = [i for i, k in enumerate(keys) if k.startswith("residual")]
    it = keys.index("total_loss")
    cols = [("ELBO", lambda a: a[:, it]), ("CE sum", lambda a: a[:, ce].sum(1)), ("ELBO - CE (weighted KL)", lambda a: a[:, it] - a[:, ce].sum(1)),
            ("KL sum (unweighted)", lambda a: a[:, kl].sum(1))]
    cols += [(keys[i], (lambda i: lambda a: a[:, i])(i)) for i in range(len(keys)) if i != it]
    rows = []
    for name, fn in cols:
        v = {dt: fn(tails[dt]) for dt in dts}
        m = {dt: v[dt].mean() for dt in dts}
        se = {dt: v[dt].std(ddof=1) / np.sqrt(len(v[dt])) if len(v[dt]) > 1 else float("nan") for dt in dts}
        line = "%-36s" % name + "".join("  %s %10.2f +- %7.2f (n=%d)" % (dt, m[dt], se[dt], len(v[dt])) for dt in dts)
        for dt in dts[1:]:
            if m[dts[0]] != 0:
                r = m[dt] / m[dts[0]]
                rse = abs(r) * np.sqrt((se[dts[0]] / m[dts[0]]) ** 2 + (se[dt] / m[dt]) ** 2)
                line += "   %s/%s = %.3f +- %.3f" % (dt, dts[0], r, rse)
                rows.append((name, dt, float(r), float(rse)))
        print(line)
    return rows

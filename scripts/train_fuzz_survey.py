"""Error of the fused training path vs torch's fp32 evaluation over many random level shapes (the cases of
tests/test_train_fuzz_gpu.py): prints the cases whose worst error exceeds 1e-5 and a summary.
usage: python scripts/train_fuzz_survey.py [ncases]"""
import contextlib
import importlib.util
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts import train_mlp_check as T  # noqa: E402

spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_train_fuzz_gpu.py"))
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)
from pointnet2_amd import train_mlp  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    over, ratios, errors = 0, [], 0
    for seed in range(n):
        kw, env = fz._case(seed)
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf), train_mlp.options(**env):
                worst = T.run_case("fuzz %d" % seed, seed=seed, fp32_baseline=True, **kw)
        except Exception as exc:                          # noqa: BLE001 -- a case the library refuses is a finding, not the end of the survey
            print("seed %3d EXCEPTION %s  %s %s" % (seed, str(exc)[:120], kw, env), flush=True)
            errors += 1
            continue
        base = T.run_case.baseline
        ratios.append(worst / max(base, 1e-9))
        if worst > 1e-5:
            over += 1
            rows = kw["b"] * (kw["n"] if kw.get("plain_cin") or kw.get("group_all") else kw["m"] * kw["ns"])
            print("seed %3d rows %5d widths %-22s worst %.2e torch-fp32 %.2e ratio %5.1f  %s" % (
                seed, rows, kw["widths"], worst, base, worst / max(base, 1e-9), env), flush=True)
    ratios.sort()
    print("cases %d, exceptions %d, worst > 1e-5: %d; ratio to torch fp32: median %.2f, 90%% %.2f, max %.2f" % (
        n, errors, over, ratios[len(ratios) // 2], ratios[int(len(ratios) * 0.9)], ratios[-1]))


if __name__ == "__main__":
    main()

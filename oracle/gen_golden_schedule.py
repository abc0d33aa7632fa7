"""TEST INFRASTRUCTURE (runs only in the build container, never on the GPU box): imports the reference's
``helpers/ramp.py`` from /root/reference (one ``sys.modules`` stub: sacred, which the file imports and does not use for
these functions) and writes the learning-rate factors the reference's LambdaLR would apply, epoch by epoch, for the two
schedule modes of ``Module.get_scheduler_lambda`` (models/module.py:213-226) -> tests/golden/g10_lr_schedule.npz.

    python oracle/gen_golden_schedule.py [out_dir]
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "tests", "golden")


def main():
    sacred = types.ModuleType("sacred")
    sacred.Ingredient = type("Ingredient", (), {"__init__": lambda self, *a, **k: None})
    sys.modules["sacred"] = sacred
    sys.path.insert(0, REF)
    from helpers import ramp
    epochs = np.arange(0, 160, dtype=np.int64)
    cases = {}
    # (warm_up_len, ramp_down_start, ramp_down_len, last_lr_value): the reference default and two other settings
    for tag, (w, s, l, last) in {"default": (5, 50, 50, 0.01), "b": (10, 30, 90, 0.001), "c": (1, 0, 7, 0.5)}.items():
        f = ramp.exp_warmup_linear_down(w, l, s, last)             # argument order of module.py:219-221
        cases[f"exp_lin_{tag}_args"] = np.array([w, s, l, last], np.float64)
        cases[f"exp_lin_{tag}"] = np.array([f(int(e)) for e in epochs], np.float64)
        with contextlib.redirect_stdout(io.StringIO()):            # cosine_cycle prints its adjusted start
            g = ramp.cosine_cycle(w, s, last)                      # module.py:222-223
        cases[f"cos_cyc_{tag}"] = np.array([g(int(e)) for e in epochs], np.float64)
    np.savez(os.path.join(OUT, "g10_lr_schedule.npz"), epochs=epochs, **cases)
    print("wrote g10_lr_schedule.npz:", sorted(cases))


if __name__ == "__main__":
    main()

"""Where the HOST time of a training step goes (run on the GPU box): cProfile around bench.py --config train.

    python tools/profile_train_host.py [steps]
"""
import cProfile
import io
import os
import pstats
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "8"
sys.argv = ["bench.py", "--config", "train", "--no-cpu-baseline", "--steps", steps]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
for key, n in (("tottime", 40), ("cumtime", 70)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(n)
    print(s.getvalue()[:14000])

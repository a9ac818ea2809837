"""Same-box A/B of a module-level test hook: python scripts/ab_hook.py ait_amd.rpn _RPN_HEADS_KERNEL [bench args...]
runs bench.py's step with the hook True and False alternately (2 x 2 runs) and prints ms/step of each."""
import importlib, io, json, os, sys, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mod, hook, rest = sys.argv[1], sys.argv[2], sys.argv[3:]
import bench
m = importlib.import_module(mod)
for rep in range(2):
    for val in (True, False):
        setattr(m, hook, val)
        sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-ab"] + rest
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                bench.main()
            d = json.loads(buf.getvalue().strip().splitlines()[-1])
            print("%s.%s = %-5s  %.3f ms/step  %.2f pairs/s" % (mod, hook, val, d["ms_per_step"], d["value"]), flush=True)
        except SystemExit as e:
            print("%s.%s = %-5s  bench refused: %s" % (mod, hook, val, e), flush=True)

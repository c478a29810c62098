"""bench.py whose ranks die AFTER their first measured split (tests/test_bench_launch.py, CPU): as a rank (WORLD_SIZE
set) main() is replaced by one that fills the provisional line on rank 0 and raises - what a collective timing out in
the tuned run looks like to run_main(); as the launcher (no WORLD_SIZE) the real main() runs and starts the ranks as
this script. bench.py carries no test hook."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
ns = {"__name__": "bench_wrapped", "__file__": BENCH}
exec(compile(open(BENCH).read(), BENCH, "exec"), ns)

if "WORLD_SIZE" in os.environ:
    def main():
        if int(os.environ.get("RANK", "0")) == 0:
            ns["_PROVISIONAL"].update({"metric": "training steps/sec (test)", "value": 123.0,
                                       "n_gpus": int(os.environ["WORLD_SIZE"]), "argv": sys.argv[1:]})
        raise RuntimeError("collective timed out (injected)")
    ns["main"] = main
ns["run_main"]()

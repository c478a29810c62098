"""bench.py with a rank that dies in the auto-tuned attempt (tests/test_bench_launch.py): the last rank of a run whose
exchange is still to be tuned (no --dp-exchange on its command line) exits with code 3 before doing anything; the
launcher's retry passes --dp-exchange and goes through. The launcher starts its ranks as the script it was itself started
as, so this wrapper is what every rank runs - bench.py carries no test hook."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if "WORLD_SIZE" in os.environ and "--dp-exchange" not in sys.argv and \
        int(os.environ.get("RANK", "0")) == int(os.environ["WORLD_SIZE"]) - 1:
    raise SystemExit(3)
# (not runpy.run_path: it rewrites sys.argv[0] to bench.py's path, and the launcher starts the ranks as sys.argv[0])
BENCH = os.path.join(ROOT, "bench.py")
exec(compile(open(BENCH).read(), BENCH, "exec"), {"__name__": "__main__", "__file__": BENCH})

"""cProfile of the timed loop of bench.py only (enabled / disabled at the two thread_cpu_seconds() calls that bracket it)."""
import cProfile, pstats, sys, io
sys.path.insert(0, ".")
sys.argv = ["bench.py", "--steps", "600", "--warmup", "10"] + sys.argv[1:]
import bench
prof = cProfile.Profile()
orig = bench.thread_cpu_seconds
state = {"n": 0}
def hook():
    r = orig()
    state["n"] += 1
    if state["n"] == 1:
        prof.enable()
    elif state["n"] == 2:
        prof.disable()
    return r
bench.thread_cpu_seconds = hook
bench.main()
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue(), file=sys.stderr)

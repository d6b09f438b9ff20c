"""Where does the main thread of bench.py spend its time?  A sampler thread reads sys._current_frames() of the main
thread every 0.5 ms during the run and counts (file:line function) of the innermost 3 Python frames."""
import collections, os, sys, threading, time, traceback
sys.path.insert(0, ".")
main_id = threading.main_thread().ident
counts = collections.Counter()
stop = False
def sampler():
    while not stop:
        f = sys._current_frames().get(main_id)
        if f is not None:
            st = traceback.extract_stack(f)[-3:]
            counts[" <- ".join(f"{os.path.basename(s.filename)}:{s.lineno} {s.name}" for s in reversed(st))] += 1
        time.sleep(0.0005)
th = threading.Thread(target=sampler, daemon=True); th.start()
sys.argv = ["bench.py", "--steps", "600", "--warmup", "10"] + sys.argv[1:]
import bench
bench.main()
stop = True
tot = sum(counts.values())
for k, v in counts.most_common(25):
    print(f"{100*v/tot:5.1f}%  {k}", file=sys.stderr)

import os, sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
os.environ.setdefault("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM", "0")
os.environ.setdefault("RMEM_TEST_WATCHDOG", "80")
import torch.multiprocessing as mp
import test_driver as T
if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=T._sharded_hip_worker, args=(r, world, 34567, q, 4, 8, 97, 129, True)) for r in range(world)]
    t0 = time.time()
    for p in procs: p.start()
    for p in procs: p.join(timeout=150)
    print("world", world, "exit codes", [p.exitcode for p in procs], "seconds", round(time.time() - t0, 1))
    for p in procs:
        if p.is_alive(): p.kill()

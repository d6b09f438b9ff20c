"""How much CPU does a host thread burn while it waits for the GPU?  200 x (a ~5 ms GPU spin kernel, event record,
event.synchronize) under different wait settings.  Prints wall and CPU seconds per thread."""
import ctypes, os, sys, time, threading
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
import torch
if mode.startswith("devflag"):
    hip = ctypes.CDLL("libamdhip64.so")
    rc = hip.hipSetDeviceFlags(int(mode[7:] or 4))
    print("hipSetDeviceFlags rc", rc)
torch.cuda.init()
x = torch.zeros(1, device="cuda")
if mode.startswith("late"):
    hip = ctypes.CDLL("libamdhip64.so")
    rc = hip.hipSetDeviceFlags(int(mode[4:] or 4))
    print("late hipSetDeviceFlags rc", rc)
def thr():
    tick = os.sysconf("SC_CLK_TCK"); out = {}
    for tid in os.listdir("/proc/self/task"):
        st = open(f"/proc/self/task/{tid}/stat").read()
        f = st[st.rindex(")") + 2:].split()
        out[tid] = (int(f[11]) + int(f[12])) / tick
    return out
cycles = int(1.2e7)      # torch.cuda._sleep spins on the shader clock: about 5 ms
torch.cuda._sleep(cycles); torch.cuda.synchronize()
t0 = time.perf_counter(); c0 = thr()
for i in range(200):
    torch.cuda._sleep(cycles)
    ev = torch.cuda.Event(blocking=(mode == "blocking_event"))
    ev.record()
    if mode == "stream_sync":
        torch.cuda.current_stream().synchronize()
    elif mode == "poll_sleep":
        while not ev.query():
            time.sleep(0.0002)
    else:
        ev.synchronize()
wall = time.perf_counter() - t0; c1 = thr()
print(mode, "wall", round(wall, 3), "cpu by thread", {k: round(v - c0.get(k, 0), 2) for k, v in c1.items() if v - c0.get(k, 0) > 0.02})

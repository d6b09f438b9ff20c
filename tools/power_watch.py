"""Board power and shader clock while a command runs: polls the amdgpu hwmon files (power1_average / power1_input in uW,
freq1_input in Hz, power1_cap) every 50 ms.   python tools/power_watch.py -- python bench.py --steps 1500 ..."""
import glob
import json
import subprocess
import sys
import time


def rd(p):
    try:
        return int(open(p).read().strip())
    except (OSError, ValueError):
        return None


def device_hwmon():
    """hwmon directory of the GPU this process sees as device 0 (by PCI address); every card's when that fails."""
    try:
        import torch
        q = torch.cuda.get_device_properties(0)
        bdf = f"{getattr(q, 'pci_domain_id', 0):04x}:{q.pci_bus_id:02x}:{q.pci_device_id:02x}.0"
        found = sorted(glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*"))
        if found:
            return found
    except Exception:
        pass
    return sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))


hw = device_hwmon()
cmd = sys.argv[sys.argv.index("--") + 1:]
p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
samples = []
t0 = time.time()
while p.poll() is None:
    row = {"t": round(time.time() - t0, 2)}
    for h in hw[:1]:
        row["power_w"] = next((v / 1e6 for v in (rd(h + "/power1_average"), rd(h + "/power1_input")) if v), None)
        row["sclk_mhz"] = (rd(h + "/freq1_input") or 0) / 1e6
        row["cap_w"] = (rd(h + "/power1_cap") or 0) / 1e6
    samples.append(row)
    time.sleep(0.05)
out = p.stdout.read()
val = None
for line in out.splitlines():
    if line.startswith("{"):
        val = json.loads(line).get("value")
busy = [s for s in samples if s.get("power_w") and s["power_w"] > 0.6 * max(x["power_w"] or 0 for x in samples)]
print(json.dumps({"cmd": " ".join(cmd), "value": val, "hwmon": hw[:1], "n_samples": len(samples),
                  "cap_w": samples[-1].get("cap_w") if samples else None,
                  "power_w_max": max((s["power_w"] or 0) for s in samples) if samples else None,
                  "power_w_mean_busy": sum(s["power_w"] for s in busy) / len(busy) if busy else None,
                  "sclk_mhz_mean_busy": sum(s["sclk_mhz"] for s in busy) / len(busy) if busy else None,
                  "sclk_mhz_min_busy": min(s["sclk_mhz"] for s in busy) if busy else None,
                  "tail": samples[-60::6]}))

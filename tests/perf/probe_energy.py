"""Energy of one hourglass forward: python tests/perf/probe_energy.py [dtype] [views] [seconds]
Runs engine.forward back to back for `seconds` while a thread samples the socket power (hwmon power1_average / power1_input in
microwatts when the box exposes it, else `rocm-smi --showpower`), and prints ms per forward, mean watts over the steady part
(the first 25 % of the samples are dropped: clocks and power are still ramping) and joules per forward.  With DF3D_LIB set to an
ablation build (scripts/build_variant.sh) the DIFFERENCE of joules per forward from the product library prices the ablated term."""
import glob, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deepfly3d_amd.hourglass import HourglassEngine
from deepfly3d_amd.synthetic import synthetic_state_dict

dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
views = int(sys.argv[2]) if len(sys.argv) > 2 else 896
seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0


def pci_address():
    """PCI address of HIP device 0 (the box's other GPUs have hwmon nodes too: pick OUR card)."""
    import ctypes

    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
            return buf.value.decode().lower()
    except OSError:
        pass
    return None


def power_source():
    addr = pci_address()
    cands = []
    for pat in ("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        for f in sorted(glob.glob(pat)):
            dev = os.path.realpath(os.path.join(os.path.dirname(f), "..", ".."))
            try:
                v = int(open(f).read())
            except (OSError, ValueError):
                continue
            cands.append((f, dev, v))
    for f, dev, v in cands:
        if addr and dev.lower().endswith(addr):
            return f
    return None


src = power_source()
samples, clocks, stop = [], [], False


def sampler():
    while not stop:
        t = time.perf_counter()
        if src:
            try:
                samples.append((t, int(open(src).read()) * 1e-6))
            except (OSError, ValueError):
                pass
            time.sleep(0.01)
        else:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
            m = re.search(r"Power \(W\): ([0-9.]+)", out)
            if m:
                samples.append((t, float(m.group(1))))
            c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            if c:
                clocks.append(int(c.group(1)))


dev = torch.device("cuda:0")
eng = HourglassEngine(synthetic_state_dict(0), dtype=dtype, device=dev)
img = torch.rand((views, 256, 512, 3), device=dev)
for _ in range(5):
    eng.forward(img)
torch.cuda.synchronize()
th = threading.Thread(target=sampler, daemon=True)
th.start()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < seconds:
    for _ in range(8):
        eng.forward(img)
    torch.cuda.synchronize()
    n += 8
t1 = time.perf_counter()
stop = True
th.join(timeout=5)
inside = [w for (t, w) in samples if t0 + 0.25 * (t1 - t0) <= t <= t1]
ms = 1e3 * (t1 - t0) / n
watts = sum(inside) / max(1, len(inside))
print(f"[{pci_address()}] {os.path.basename(os.environ.get('DF3D_LIB', 'product'))} {dtype} {views} views: {ms:.3f} ms per forward, {watts:.0f} W mean of {len(inside)} samples"
      f" ({'hwmon ' + os.path.basename(src) if src else 'rocm-smi'}), {ms * 1e-3 * watts:.3f} J per forward" + (f", sclk median {sorted(clocks)[len(clocks) // 2]} MHz" if clocks else ""))

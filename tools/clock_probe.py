"""Shader clock and board power while the in-LDS kernel, the external kernel and nothing run (hwmon of the card that runs
the kernels, sampled every 50 ms for ~3 s each): the evidence for the clock figure in DESIGN.md section 4.
    python tools/clock_probe.py > profiles/rNN_clocks.txt"""
import ctypes
import glob
import os
import sys
import threading
import time

import torch  # noqa: E402  (before the library: one HIP runtime in the process, torch's)

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import smfft_amd as sm  # noqa: E402

TOTAL = 1 << 29


def my_card():
    """sysfs node of the card this process computes on (the node shows every card of the machine)"""
    p = torch.cuda.get_device_properties(0)
    want = "%04x:%02x:%02x." % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    for d in glob.glob("/sys/class/drm/card*/device"):
        if os.path.basename(os.path.realpath(d)).startswith(want):
            return d
    return None


MY_CARD = my_card()
A, B = sm.DeviceBuffer(TOTAL * 8), sm.DeviceBuffer(TOTAL * 8)
sm.lib.smfft_memset(A.ptr, 0, TOTAL * 8)


def hwmon_files():
    out = {}
    for card in ([MY_CARD] if MY_CARD else sorted(glob.glob("/sys/class/drm/card*/device"))):
        for h in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
            for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input"):
                f = os.path.join(h, name)
                if os.path.exists(f):
                    out.setdefault(card, {})[name] = f
    return out


def read(f):
    try:
        return int(open(f).read().strip())
    except Exception:
        return None


def sample_while(work, seconds=3.0):
    stop = threading.Event()

    def loop():
        while not stop.is_set():
            work()
    th = threading.Thread(target=loop)
    samples = {}
    th.start()
    t0 = time.time()
    time.sleep(0.5)
    while time.time() - t0 < seconds:
        for card, files in hwmon_files().items():
            for name, f in files.items():
                v = read(f)
                if v is not None:
                    samples.setdefault((card, name), []).append(v)
        time.sleep(0.05)
    stop.set()
    th.join()
    return samples


def report(label, samples):
    print(label)
    for (card, name), vs in sorted(samples.items()):
        scale, unit = (1e-6, "MHz") if name.startswith("freq") else (1e-6, "W") if name.startswith("power") else (1e-3, "C")
        print(f"  {card} {name}: mean {sum(vs) / len(vs) * scale:.0f} {unit}, min {min(vs) * scale:.0f}, max {max(vs) * scale:.0f} ({len(vs)} samples)")


def in_lds():
    t = ctypes.c_double(0)
    sm.lib.smfft_ct_multiple_benchmark(A.ptr, B.ptr, 1024, min(TOTAL // 1024 * 10, 2**31 - 1), 0, 1, ctypes.byref(t))


def external():
    t = ctypes.c_double(0)
    sm.lib.smfft_ct_external_benchmark(A.ptr, B.ptr, 1024, TOTAL // 1024, 0, 1, ctypes.byref(t))


report("idle", sample_while(lambda: time.sleep(0.01)))
report("in-LDS path, N=1024 reorder, saturating batch (x10), back to back", sample_while(in_lds))
report("external path, N=1024, 4 GiB + 4 GiB, back to back", sample_while(external))
report("in-LDS path again", sample_while(in_lds))

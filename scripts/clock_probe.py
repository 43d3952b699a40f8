"""Shader clock and package power while the training bench runs (sysfs, no HIP in this process):
python scripts/clock_probe.py [bench args].  Prints the distribution of the samples taken inside the timed region."""
import glob
import os
import subprocess
import sys
import threading
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return ""


def sclk_mhz(dev):
    for line in read(dev + "/pp_dpm_sclk").splitlines():
        if line.rstrip().endswith("*"):
            return int(line.split(":")[1].strip().split("M")[0])
    return -1


def main():
    devs = [d for d in glob.glob("/sys/class/drm/card*/device") if os.path.exists(d + "/pp_dpm_sclk")]
    hw = [h for d in devs for h in glob.glob(d + "/hwmon/hwmon*")]
    print("devices", devs, "hwmon", hw, flush=True)
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            row = [time.time()]
            for d in devs:
                row.append(sclk_mhz(d))
            for h in hw:
                p = read(h + "/power1_average") or read(h + "/power1_input")
                row.append(int(p) / 1e6 if p.strip().isdigit() else -1)
                f = read(h + "/freq1_input")
                row.append(int(f) / 1e6 if f.strip().isdigit() else -1)
            samples.append(row)
            time.sleep(0.02)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--no-cpu-baseline", "--no-kernel-timing", "--steps", "300",
                        "--max-blocks", "1"] + sys.argv[1:], capture_output=True, text=True)
    t_end = time.time()
    stop.set()
    th.join()
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    print(line[-1][:160] if line else p.stderr[-500:])
    # the last 12 s before the bench ended: the timed region
    busy = [r for r in samples if t_end - 14 < r[0] < t_end - 2]
    print(len(samples), "samples,", len(busy), "in the timed region")
    for col in range(1, len(busy[0]) if busy else 0):
        v = sorted(r[col] for r in busy)
        print("col", col, "min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f  mean %.1f" %
              (v[0], v[len(v) // 10], v[len(v) // 2], v[9 * len(v) // 10], v[-1], sum(v) / len(v)))


if __name__ == "__main__":
    main()

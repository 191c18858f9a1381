#!/usr/bin/env python3
"""Samples the GPU's socket power and shader clock from sysfs (hwmon) while a command runs, and prints the distribution.

    python3 tools/power_trace.py [--period-ms 10] -- python3 bench.py --steps 2000 --warmup 20 --no-cpu-baseline --no-extras

Used for DESIGN.md section 3.1 / 9: is a window bound by the matrix rate at the nominal clock, or by the energy it takes at the
board's power limit?  Reads only (an ordinary user may); the command runs as a child process, this process never touches the GPU.
"""
import glob
import json
import os
import statistics
import subprocess
import sys
import threading
import time


def read_int(path):
    try:
        with open(path) as fh:
            return int(fh.read().strip())
    except (OSError, ValueError):
        return None


def find_hwmon():
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        names = os.listdir(d)
        out.append({"dir": d, "power": next((os.path.join(d, n) for n in ("power1_input", "power1_average") if n in names), None),
                    "cap": os.path.join(d, "power1_cap") if "power1_cap" in names else None,
                    "sclk": os.path.join(d, "freq1_input") if "freq1_input" in names else None,
                    "mclk": os.path.join(d, "freq2_input") if "freq2_input" in names else None,
                    "temp": os.path.join(d, "temp1_input") if "temp1_input" in names else None})
    return out


def active_sclk_mhz(dev_dir):
    """pp_dpm_sclk marks the current level with '*' (fallback when hwmon has no freq1_input)."""
    try:
        with open(os.path.join(dev_dir, "pp_dpm_sclk")) as fh:
            for line in fh:
                if "*" in line:
                    return float(line.split(":")[1].strip().split("M")[0])
    except (OSError, ValueError, IndexError):
        pass
    return None


def summary(xs):
    if not xs:
        return None
    xs = sorted(xs)
    q = lambda f: xs[min(len(xs) - 1, int(f * len(xs)))]  # noqa: E731
    return {"n": len(xs), "mean": round(statistics.fmean(xs), 1), "p10": q(0.1), "p50": q(0.5), "p90": q(0.9), "min": xs[0], "max": xs[-1]}


def main():
    argv = sys.argv[1:]
    period = 0.01
    if argv and argv[0] == "--period-ms":
        period = float(argv[1]) / 1000.0
        argv = argv[2:]
    if argv and argv[0] == "--":
        argv = argv[1:]
    mons = [m for m in find_hwmon() if m["power"]]
    if not mons:
        print(json.dumps({"error": "no hwmon power sensor readable", "hwmon": find_hwmon()}))
    samples = {m["dir"]: {"power_w": [], "sclk_mhz": [], "t": []} for m in mons}
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            now = time.perf_counter()
            for m in mons:
                s = samples[m["dir"]]
                p = read_int(m["power"])
                f = read_int(m["sclk"]) if m["sclk"] else None
                if p is not None:
                    s["power_w"].append(p / 1e6)
                    s["t"].append(now)
                if f is not None:
                    s["sclk_mhz"].append(f / 1e6)
                elif not m["sclk"]:
                    g = active_sclk_mhz(os.path.dirname(os.path.dirname(m["dir"])))
                    if g is not None:
                        s["sclk_mhz"].append(g)
            time.sleep(period)

    th = threading.Thread(target=sampler, daemon=True)
    idle = {m["dir"]: (read_int(m["power"]) or 0) / 1e6 for m in mons}
    th.start()
    t0 = time.perf_counter()
    child = subprocess.run(argv, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    t1 = time.perf_counter()
    stop.set()
    th.join()
    line = next((ln for ln in child.stdout.splitlines()[::-1] if ln.startswith("{")), None)
    out = {"command": " ".join(argv), "returncode": child.returncode, "wall_s": round(t1 - t0, 2), "period_ms": period * 1e3, "sensors": []}
    for m in mons:
        s = samples[m["dir"]]
        # the busy part of the run: samples above the midpoint between idle and peak power
        pw = s["power_w"]
        busy = []
        if pw:
            thr = idle[m["dir"]] + 0.5 * (max(pw) - idle[m["dir"]])
            busy = [i for i, p in enumerate(pw) if p >= thr]
        out["sensors"].append({
            "hwmon": m["dir"], "power_cap_w": (read_int(m["cap"]) or 0) / 1e6 if m["cap"] else None, "idle_power_w": round(idle[m["dir"]], 1),
            "power_w_all": summary(pw), "power_w_busy": summary([pw[i] for i in busy]),
            "sclk_mhz_all": summary(s["sclk_mhz"]),
            "sclk_mhz_busy": summary([s["sclk_mhz"][i] for i in busy if i < len(s["sclk_mhz"])]),
            "busy_fraction_of_samples": round(len(busy) / max(1, len(pw)), 3)})
    if line:
        try:
            j = json.loads(line)
            out["bench"] = {k: j.get(k) for k in ("value", "unit", "ms_per_step", "median_ms_per_step", "steps")}
        except ValueError:
            out["bench_line"] = line[:200]
    else:
        out["child_tail"] = child.stdout[-600:]
    print(json.dumps(out))


if __name__ == "__main__":
    main()

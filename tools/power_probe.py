#!/usr/bin/env python3
"""Socket power and shader clock of the busy card while a command runs (hwmon / sysfs, sampled every 10 ms): is a
workload power-bound?  The whole host's cards are visible; the busy one is the card whose median power is highest.
   python3 tools/power_probe.py -- <command ...>        e.g.  -- python3 tools/profile_mll.py 4096 256 40
   python3 tools/power_probe.py --skip 2.0 -- ./tools/mfma_f64_peak     (--skip: seconds of start-up to leave out)"""
import glob
import os
import subprocess
import sys
import threading
import time


def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def main():
    args = sys.argv[1:]
    skip = 0.0
    if args and args[0] == "--skip":
        skip = float(args[1])
        args = args[2:]
    assert args and args[0] == "--", __doc__
    cmd = args[1:]
    cards = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        if read(os.path.join(hw, "power1_input")) is not None:
            cards[hw.split("/")[4]] = hw
    # the whole host's cards are in sysfs; ours is the one whose render node this user may open
    mine = []
    for rn in glob.glob("/sys/class/drm/renderD*"):
        dev = os.path.realpath(os.path.join(rn, "device"))
        if os.access("/dev/dri/" + os.path.basename(rn), os.R_OK | os.W_OK):
            mine += [c for c in cards if os.path.realpath(f"/sys/class/drm/{c}/device") == dev]
    if len(mine) == 1:
        cards = {mine[0]: cards[mine[0]]}
    print(f"  cards watched: {sorted(cards)} (render-node access picked {mine})")
    samples, stop = {c: [] for c in cards}, threading.Event()

    def sampler():
        watch = dict(cards)
        while not stop.is_set():
            t = time.perf_counter()
            for c, hw in watch.items():
                pw, fq = read(os.path.join(hw, "power1_input")), read(os.path.join(hw, "freq1_input"))
                if pw and fq:
                    samples[c].append((t, float(pw) / 1e6, float(fq) / 1e6))
            # a read costs ~10 ms: once one card stands out (> 1.6 x the others' power), watch it alone
            if len(watch) > 1 and all(len(v) >= 3 for v in samples.values()):
                last = {c: samples[c][-1][1] for c in watch}
                top = max(last, key=last.get)
                rest = sorted(v for c, v in last.items() if c != top)
                if last[top] > 900.0 and last[top] > 1.6 * rest[len(rest) // 2]:
                    watch = {top: watch[top]}
            time.sleep(0.005)

    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    out = subprocess.run(cmd, capture_output=True, text=True)
    t1 = time.perf_counter()
    stop.set()
    th.join()
    print((out.stdout.strip().splitlines() or ["(no output)"])[-1])
    if out.returncode:
        print("command failed:", out.stderr[-400:])
        sys.exit(1)
    # the busy card: highest 90th-percentile power over the run
    def p90(c):
        v = sorted(s[1] for s in samples[c])
        return v[int(0.9 * len(v))] if v else 0.0
    busy = max(cards, key=p90)
    rows = [s for s in samples[busy] if s[0] > t0 + skip]
    thr = 0.9 * p90(busy)
    hot = [s for s in rows if s[1] >= thr] or rows or samples[busy]  # samples taken while the kernels ran (the launcher's idle phases fall out)
    cap = read(os.path.join(cards[busy], "power1_cap"))
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"  busy card {busy}: cap {float(cap) / 1e6:.0f} W; {len(hot)} of {len(rows)} samples under load: power median {med([s[1] for s in hot]):.0f} W "
          f"(min {min(s[1] for s in hot):.0f}, max {max(s[1] for s in hot):.0f}); sclk median {med([s[2] for s in hot]):.0f} MHz "
          f"(min {min(s[2] for s in hot):.0f}, max {max(s[2] for s in hot):.0f}); wall {t1 - t0:.1f} s")


if __name__ == "__main__":
    main()

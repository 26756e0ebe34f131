"""Where does the host sit inside HIP calls? Reads an AMD_LOG_LEVEL=4 log (stderr of any run) and prints, for every gap of more
than --ms between two consecutive lines of the SAME thread, the lines around it.
python tools/hip_log_gaps.py log.txt [--ms 3] [--context 6]"""
import argparse
import re
import collections

ap = argparse.ArgumentParser()
ap.add_argument("log")
ap.add_argument("--ms", type=float, default=3.0)
ap.add_argument("--context", type=int, default=6)
a = ap.parse_args()
pat = re.compile(r"^:(\d):(\S+)\s*:(\d+)\s*: (\d+) us:\s*\[pid:(\d+)\s+tid:\s*(0x[0-9a-f]+)\]\s?(.*)$")
last = {}
recent = collections.defaultdict(lambda: collections.deque(maxlen=a.context))
n = 0
after = {}
with open(a.log, errors="replace") as f:
    for line in f:
        m = pat.match(line.rstrip("\n"))
        if not m:
            continue
        n += 1
        ts, tid, text = int(m.group(4)), m.group(6), f"{m.group(2)}:{m.group(3)} {m.group(7)}"
        if tid in after and after[tid] > 0:
            print(f"      {tid} {ts} {text[:200]}")
            after[tid] -= 1
        if tid in last and ts - last[tid] > a.ms * 1000:
            print(f"--- thread {tid}: {(ts - last[tid]) / 1000:.2f} ms between")
            for t0, l0 in recent[tid]:
                print(f"      {tid} {t0} {l0[:200]}")
            print(f"  >>> {tid} {ts} {text[:200]}")
            after[tid] = 3
        last[tid] = ts
        recent[tid].append((ts, text))
print(f"{n} log lines")

#!/bin/bash
# Samples rocm-smi power / clocks (read-only queries) while a command runs:  tools/power_probe.sh <label> <command ...>
# Prints the command's last output line and the (sclk MHz, socket power W) samples; "under load" = samples above 600 W.
label=$1; shift
( for i in $(seq 400); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo; sleep 0.2; done ) > /tmp/power_$label.txt &
sp=$!
"$@" > /tmp/power_cmd_$label.txt 2>&1
kill $sp 2>/dev/null
python3 - "$label" <<'PY'
import re, sys
label = sys.argv[1]
res = open(f"/tmp/power_cmd_{label}.txt").read().strip().split("\n")[-1]
m = re.search(r'"ms_per_step": ([0-9.]+)', res)
if m: res = f"{float(m.group(1)):.2f} ms per step"
s = []
for r in open(f"/tmp/power_{label}.txt"):
    a = re.search(r"sclk clock level: \S+ \((\d+)Mhz\).*?Power \(W\): ([0-9.]+)", r)
    if a: s.append((int(a.group(1)), float(a.group(2))))
busy = [x for x in s if x[1] > 600]
print(f"== {label}: {res[:100]}")
print("   samples (sclk MHz, W): " + " ".join(f"({a},{int(b)})" for a, b in s))
if busy:
    print(f"   under load: sclk {min(a for a, _ in busy)}-{max(a for a, _ in busy)} MHz, power {int(min(b for _, b in busy))}-{int(max(b for _, b in busy))} W ({len(busy)} samples)")
PY

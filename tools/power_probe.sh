#!/bin/bash
# Samples rocm-smi power / clocks while a command runs:  tools/power_probe.sh <label> <command ...>   (read-only queries)
label=$1; shift
( for i in $(seq 200); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.25; done ) > /tmp/power_$label.txt &
sp=$!
"$@" > /tmp/power_cmd_$label.txt 2>&1
kill $sp 2>/dev/null
echo "== $label: $(tail -1 /tmp/power_cmd_$label.txt | cut -c1-200)"
sort /tmp/power_$label.txt | uniq -c | sort -rn | head -8

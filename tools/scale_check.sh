#!/bin/bash
# First thing to run on a box with more than one GPU: the 1 / 2 / 4 / 8-GPU curve of the bench workload (m-mix) and of
# configs[2] (c3: control-only data-parallel training, the configuration whose gradient traffic is largest relative to its step), with
# what RCCL chose for the gradient all-reduce (algorithm / protocol / channels from NCCL_DEBUG=INFO) and the exposed communication time
# per step for every N -- so that one lease answers whether the collective is ring- or link-bound on the 7-link xGMI mesh.
#   tools/scale_check.sh [outdir]            (nothing is faked: with one visible GPU it records that and stops)
# Reads: bench.py's JSON line (value, ms_per_step, exposed_comm_ms_per_step).  Writes: <outdir>/scale_<workload>_<payload>.txt + the JSON lines.
out=${1:-gpurun_out/scale_check}
mkdir -p "$out"
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ngpu=$(python3 -c 'import torch; print(torch.cuda.device_count())')
echo "visible GPUs: $ngpu" | tee "$out/summary.txt"
if [ "$ngpu" -lt 2 ]; then
  echo "one GPU: no curve can be measured here (the one-GPU anchor of the reducer's fixed cost is bench.py --force-dp, profiles/r0*_forcedp_*)" | tee -a "$out/summary.txt"
  exit 0
fi
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
for wl in m-mix c3; do
  for payload in fp32 bf16 fp32+rs_ag; do          # "+rs_ag": every message as reduce-scatter + all-gather (NEKO_DP_COLLECTIVE=rs_ag)
    coll=allreduce; case $payload in *+rs_ag) coll=rs_ag;; esac
    export NEKO_DP_COLLECTIVE=$coll
    pl=${payload%%+*}
    for n in 1 2 4 8; do
      [ "$n" -gt "$ngpu" ] && continue
      tag=${wl}_${payload}_n$n
      port=$((29500 + RANDOM % 2000))
      if [ "$n" = 1 ]; then
        NEKO_DP_PAYLOAD=$pl python3 bench.py --workload $wl --gpus 1 --no-cpu-baseline > "$out/$tag.json" 2> "$out/$tag.err"
      else
        NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL,TUNING NEKO_DP_PAYLOAD=$pl python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
          --master-addr 127.0.0.1 --master-port $port bench.py --workload $wl --gpus $n --no-cpu-baseline > "$out/$tag.json" 2> "$out/$tag.err"
      fi
      python3 - "$out/$tag.json" "$out/$tag.err" "$wl" "$payload" "$n" <<'PY' | tee -a "$out/summary.txt"
import json, re, sys
line = next((l for l in open(sys.argv[1]) if l.startswith("{")), None)
wl, payload, n = sys.argv[3], sys.argv[4], int(sys.argv[5])
if line is None:
    print(f"{wl:6s} {payload} N={n}: no JSON line (see {sys.argv[2]})"); sys.exit(0)
d = json.loads(line)
err = open(sys.argv[2], errors="replace").read()
algo = sorted(set(re.findall(r"(?:Algo|algorithm)\s*[:=]?\s*(\w+)", err)))[:4]
proto = sorted(set(re.findall(r"(?:Proto|protocol)\s*[:=]?\s*(\w+)", err)))[:4]
chans = re.findall(r"(\d+) coll channels", err)
print(f"{wl:6s} {payload} N={n}: {d['value']:12.0f} {d['unit']}  {d['ms_per_step']:7.2f} ms/step  exposed comm {d.get('exposed_comm_ms_per_step', 'n/a')} ms  "
      f"RCCL algo {algo or '?'} proto {proto or '?'} channels {chans[:1] or '?'}")
PY
    done
  done
done
python3 - "$out" <<'PY' | tee -a "$out/summary.txt"
# scaling efficiency per workload / payload from the JSON lines (the driver computes its own from SCALE_rNN.json; this is for the notes)
import glob, json, os, re, sys
rows = {}
for f in glob.glob(os.path.join(sys.argv[1], "*_n*.json")):
    m = re.match(r"(.+)_(fp32|bf16|fp32\+rs_ag)_n(\d+)\.json", os.path.basename(f))
    line = next((l for l in open(f) if l.startswith("{")), None)
    if m and line:
        rows.setdefault((m.group(1), m.group(2)), {})[int(m.group(3))] = json.loads(line)["value"]
for (wl, pl), v in sorted(rows.items()):
    if 1 in v:
        print(f"{wl} {pl}: " + "  ".join(f"N={n}: x{v[n] / v[1]:.2f}" for n in sorted(v)))
PY

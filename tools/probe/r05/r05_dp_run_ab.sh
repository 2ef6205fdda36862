cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_dp_gpu.py -x -q -m gpu 2>&1 | tail -3
for v in "run64=NEKO_DP_MIN_RUN_MB=64" "perrange=NEKO_DP_MIN_RUN_MB=0"; do name=${v%%=*}; envs=${v#*=}
 for w in c3 m-mix; do
  env $envs python bench.py --workload $w --force-dp --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$name $w %.2f ms/step exposed %.3f' % (j['ms_per_step'], j['exposed_comm_ms_per_step']))"
 done
done
python bench.py --workload c3 --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('no reducer c3 %.2f ms/step' % j['ms_per_step'])"

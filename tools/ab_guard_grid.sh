cd /tmp
R=$GRAFT_REPO_ROOT
for i in 1 2; do
for v in 32 64 128 256 512; do
  echo "GRID=$v: $(env TSDR_GUARD_GRID=$v python3 $R/bench.py --quick --repeats 5 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'fused', d['fused']['ms_per_step'], d['roofline']['kernels_ms_per_step'].get('sync_guard'))")"
done
echo "OFF: $(env TSDR_SYNC_GUARD_PPB=0 python3 $R/bench.py --quick --repeats 5 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'fused', d['fused']['ms_per_step'])")"
done

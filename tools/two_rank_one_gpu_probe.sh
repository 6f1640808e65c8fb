cd ${GRAFT_REPO_ROOT:-$PWD}
PORT=29713
for r in 0 1; do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT OMP_NUM_THREADS=1 CSIM_BENCH_PHASE_TIMEOUT=20 \
    timeout -k 5 200 python3 bench.py --gpus 2 --nx 1536 --ny 1024 --steps 37 --warmup 7 --ramp-seconds 0.02 > gpurun_out/two_rank_$r.out 2> gpurun_out/two_rank_$r.err &
done
wait
for r in 0 1; do echo "== rank $r rc"; tail -c 1500 gpurun_out/two_rank_$r.err | tail -8; done
python3 -c "
import json
d=json.loads(open('gpurun_out/two_rank_0.out').read().strip().splitlines()[-1]); c=d['config']
print(round(d['value']), c['halo_transport'][:80], c['exchange_schedules_ms_per_step'], c['value_is'][:60], c['parity_preflight']['ok'])"

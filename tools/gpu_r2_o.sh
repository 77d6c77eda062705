#!/bin/bash
# bench.py N = 8 flow at FULL size, eight ranks sharing the one GPU (gloo rendezvous, exchange staged
# through host memory): k-way partition with k = 8, shards, schedule, parity, variants, JSON
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2o
for r in 0 1 2 3 4 5 6 7; do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=8 MASTER_ADDR=127.0.0.1 MASTER_PORT=29641 DGLA_BENCH_BACKEND=gloo \
    timeout 800 python bench.py --gpus 8 --steps 2 --warmup 1 > gpurun_out/r2o/rank$r.out 2> gpurun_out/r2o/rank$r.err &
done
wait
grep '^{' gpurun_out/r2o/rank0.out | python -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d['config']
print('n_gpus', d['n_gpus'], 'cut', round(c['cut_fraction'], 4), 'partition_s', round(c['partition_seconds'], 1), 'halo_rows_max', c['halo_rows_max'])
print('parity', d['parity_max_rel_err_vs_single_gpu_launch'])
for k, v in d['variants'].items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a != 'note'})
print('partition_stats', c.get('partition_stats'))
"
for r in 0 1 2 3 4 5 6 7; do tail -n 2 gpurun_out/r2o/rank$r.err; done

#!/bin/bash
# bench.py N > 1 flow at FULL size, two ranks sharing the one GPU (gloo rendezvous, exchange staged
# through host memory): exercises partition + shards + schedule + JSON at the driver's sizes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
for r in 0 1; do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 DGLA_BENCH_BACKEND=gloo \
    timeout 800 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r2k/rank$r.out 2> gpurun_out/r2k/rank$r.err &
done
wait
grep '^{' gpurun_out/r2k/rank0.out | cut -c1-3000
tail -3 gpurun_out/r2k/rank0.err gpurun_out/r2k/rank1.err

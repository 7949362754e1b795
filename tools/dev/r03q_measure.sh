#!/bin/bash
# config 5 at 12 beacons per tag: bench line, rocprofv3 kernel stats, PMC traffic; and the two-rank rehearsal of the multi-rank bench line
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
timeout -k 10 200 python3 bench.py --workload config5 --slam-bcns 12 --no-cpu-baseline --steps 60 --warmup 12 2>/dev/null > $o/${tag}_config5_b12.json
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_sprof -- python3 bench.py --workload config5 --slam-bcns 12 --no-cpu-baseline --steps 60 --warmup 12 > $o/${tag}_config5_b12_under_rocprof.json 2> $o/${tag}_b12.err
find $o/${tag}_sprof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $o/${tag}_slam_config5_b12_rocprofv3_kernel_stats.csv
rm -rf $o/${tag}_sprof
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/${tag}_spmc_fetch -- python3 bench.py --workload config5 --slam-bcns 12 --steps 30 --warmup 12 --no-cpu-baseline > /dev/null 2>> $o/${tag}_b12.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/${tag}_spmc_write -- python3 bench.py --workload config5 --slam-bcns 12 --steps 30 --warmup 12 --no-cpu-baseline > /dev/null 2>> $o/${tag}_b12.err
python3 tools/pmc_traffic.py $o/${tag}_spmc_fetch $o/${tag}_spmc_write slam > $o/${tag}_pmc_traffic_slam_b12.json
git -C . checkout profiles/pmc_traffic_slam.json 2>/dev/null || true
rm -rf $o/${tag}_spmc_fetch $o/${tag}_spmc_write
BENCH_PG=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --particles 4000000 > $o/${tag}_w2.json 2> $o/${tag}_w2.err
echo done

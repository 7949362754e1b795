#!/bin/bash
# measurement set of a round step: tools/dev/r02_measure.sh TAG   (run on the GPU box from the repo root)
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out
timeout -k 10 400 python3 bench.py > $o/${tag}_bench_default.json 2> $o/${tag}_bench_default.err
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $o/${tag}_bench_short.json 2>> $o/${tag}_bench_default.err
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof -- python3 bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline > $o/${tag}_bench_under_rocprof.json 2> $o/${tag}_prof.err
find $o/${tag}_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $o/${tag}_rocprofv3_kernel_stats.csv
rm -rf $o/${tag}_prof
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/${tag}_pmc_fetch -- python3 bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-kernel-events > /dev/null 2> $o/${tag}_pmc.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/${tag}_pmc_write -- python3 bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-kernel-events > /dev/null 2>> $o/${tag}_pmc.err
python3 tools/pmc_traffic.py $o/${tag}_pmc_fetch $o/${tag}_pmc_write 16000000 > $o/${tag}_pmc_traffic.json
cp profiles/pmc_traffic.json $o/pmc_traffic_latest.json
rm -rf $o/${tag}_pmc_fetch $o/${tag}_pmc_write

#!/bin/bash
# usage: ab_prof.sh [REF]  -- as ab.sh, plus one rocprofv3 kernel trace per library: prints every kernel's average / min / max.
ref=${1:-HEAD}
set -e
rm -rf /tmp/ab_src && mkdir -p /tmp/ab_src
git -C /root/repo archive "$ref" slam3d_amd/csrc include | tar -x -C /tmp/ab_src
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result -Wno-unused-value"
(cd /tmp/ab_src/slam3d_amd/csrc && /opt/rocm/bin/hipcc $F -shared -o /root/repo/slam3d_amd/lib/libexp_ref.so pf_loc.hip pf_slam.hip)
make -C /root/repo/slam3d_amd/csrc >/dev/null
cd /root/repo
bash tools/dev/gpu_when_free.sh 1000 'cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && for r in 1 2; do for v in exp_ref particlefilter; do PFX_LIB_PATH=$PWD/slam3d_amd/lib/lib$v.so timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline > gpurun_out/ab_${v}_$r.json 2> gpurun_out/ab_${v}_$r.err || exit 1; done; done; for v in exp_ref particlefilter; do rm -rf gpurun_out/abp_$v; export PFX_LIB_PATH=$PWD/slam3d_amd/lib/lib$v.so; timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abp_$v -- python3 bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline > /dev/null 2> gpurun_out/abp_$v.err || exit 1; find gpurun_out/abp_$v -name "*kernel_stats.csv" | xargs -I{} cp {} gpurun_out/abp_$v.csv; rm -rf gpurun_out/abp_$v; done' > /tmp/ab_gpurun.log 2>&1 || { tail -5 /tmp/ab_gpurun.log; exit 1; }
rm -f slam3d_amd/lib/libexp_ref.so
python - <<'PY'
import json, csv
for v in ("exp_ref", "particlefilter"):
    for r in (1, 2):
        d = json.loads([l for l in open(f"gpurun_out/ab_{v}_{r}.json") if l.startswith("{")][-1])
        print(f"{v:15s} run {r}: {d['ms_per_step']:.4f} ms/step  {d['value']/1e9:6.1f} G/s ",
              {k: round(x["avg_ms"], 4) for k, x in d["kernels"].items()})
for v in ("exp_ref", "particlefilter"):
    print(v)
    for row in csv.DictReader(open(f"gpurun_out/abp_{v}.csv")):
        if int(row["Calls"]) > 10:
            print(f"   {row['Name'][:40]:40s} calls {row['Calls']:>5s} avg {float(row['AverageNs'])/1e3:8.2f} us  min {int(row['MinNs'])/1e3:7.2f}  max {int(row['MaxNs'])/1e3:7.2f}")
PY

#!/bin/bash
# usage: ab.sh [REF]   -- same-box A/B of the resampling kernel: builds the library of git REF (default HEAD) next to the
# working tree's, runs bench.py on both alternately in ONE gpurun call (boxes differ by a few per cent) and prints the
# per-kernel averages.
ref=${1:-HEAD}
set -e
rm -rf /tmp/ab_src && mkdir -p /tmp/ab_src
git -C /root/repo archive "$ref" slam3d_amd/csrc include | tar -x -C /tmp/ab_src
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result -Wno-unused-value"
(cd /tmp/ab_src/slam3d_amd/csrc && /opt/rocm/bin/hipcc $F -shared -o /root/repo/slam3d_amd/lib/libexp_ref.so pf_loc.hip pf_slam.hip)
make -C /root/repo/slam3d_amd/csrc >/dev/null
cd /root/repo
bash tools/dev/gpu_when_free.sh 900 'for r in 1 2; do for v in exp_ref particlefilter; do PFX_LIB_PATH=$PWD/slam3d_amd/lib/lib$v.so timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline > gpurun_out/ab_${v}_$r.json 2> gpurun_out/ab_${v}_$r.err || exit 1; done; done' > /tmp/ab_gpurun.log 2>&1 || { tail -5 /tmp/ab_gpurun.log; exit 1; }
rm -f slam3d_amd/lib/libexp_ref.so
python - <<'PY'
import json
for v in ("exp_ref", "particlefilter"):
    for r in (1, 2):
        d = json.loads([l for l in open(f"gpurun_out/ab_{v}_{r}.json") if l.startswith("{")][-1])
        print(f"{v:15s} run {r}: {d['ms_per_step']:.4f} ms/step  {d['value']/1e9:6.1f} G/s ",
              {k: round(x["avg_ms"], 4) for k, x in d["kernels"].items()})
PY

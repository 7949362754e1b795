#!/bin/bash
# usage: ab_slam.sh [REF] -- same-box A/B of the SLAM row kernel: the library of git REF (default HEAD) against the working
# tree's; config-5 bench and the branch split of k_slam_gate_rows for both, in ONE gpurun call.
ref=${1:-HEAD}
set -e
rm -rf /tmp/ab_src && mkdir -p /tmp/ab_src
git -C /root/repo archive "$ref" slam3d_amd/csrc include | tar -x -C /tmp/ab_src
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result -Wno-unused-value -Wno-unused-function"
(cd /tmp/ab_src/slam3d_amd/csrc && /opt/rocm/bin/hipcc $F -shared -o /root/repo/slam3d_amd/lib/libexp_ref.so pf_loc.hip pf_loc_batch.hip pf_slam.hip)
make -C /root/repo/slam3d_amd/csrc >/dev/null
cd /root/repo
/usr/local/graft/bin/gpurun --timeout 900 -- 'for r in 1 2; do for v in exp_ref particlefilter; do export PFX_LIB_PATH=$PWD/slam3d_amd/lib/lib$v.so; bash tools/dev/slam_split.sh ab_${v}_$r || exit 1; done; done' 2>&1 | grep -E "config5|only_norm|status"
rm -f slam3d_amd/lib/libexp_ref.so

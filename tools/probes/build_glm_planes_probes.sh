#!/bin/bash
# ablation builds of tools/probes/glm_planes_probe (developer tool; binaries are git-ignored)
cd "$(dirname "$0")"
H=/opt/rocm/bin/hipcc
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-value"
$H $F glm_planes_probe.hip -o glm_planes_probe &
$H $F -DPA_GLMP_ABL_NOPRIO glm_planes_probe.hip -o glm_planes_probe_noprio &
$H $F -DPA_GLMP_ABL_NOTRANS glm_planes_probe.hip -o glm_planes_probe_notrans &
$H $F -DPA_GLMP_ABL_ONETRANS glm_planes_probe.hip -o glm_planes_probe_onetrans &
$H $F -DPA_GLMP_ABL_NOSPLIT glm_planes_probe.hip -o glm_planes_probe_nosplit &
$H $F -DPA_GLMP_ABL_NOGEMM2 glm_planes_probe.hip -o glm_planes_probe_nogemm2 &
$H $F -DPA_GLMP_ABL_NOGEMM1 -DPA_GLMP_ABL_NOGEMM2 glm_planes_probe.hip -o glm_planes_probe_nomfma &
$H $F -DPA_GLMP_ABL_NOTRANS -DPA_GLMP_ABL_NOSPLIT glm_planes_probe.hip -o glm_planes_probe_novalu &
$H $F -DPA_GLMP_ABL_NOTRANS -DPA_GLMP_ABL_NOSPLIT -DPA_GLMP_ABL_NOGEMM1 -DPA_GLMP_ABL_NOGEMM2 glm_planes_probe.hip -o glm_planes_probe_alloff &
$H $F -DPA_GLMP_ABL_NOTRANS -DPA_GLMP_ABL_NOSPLIT -DPA_GLMP_ABL_NOGEMM1 -DPA_GLMP_ABL_NOGEMM2 -DPA_GLMP_ABL_NODMA glm_planes_probe.hip -o glm_planes_probe_alloff_nodma &
$H $F -DPA_GLMP_ABL_NODMA glm_planes_probe.hip -o glm_planes_probe_nodma &
wait
ls glm_planes_probe*

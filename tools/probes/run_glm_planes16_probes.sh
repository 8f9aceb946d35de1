#!/bin/bash
# runs every ablation build of glm_planes16_probe twice (wg/CU = 2), on the GPU box
cd "$(dirname "$0")"
for v in base notrans nosplit noelem novalu nogemm1 nogemm2 nomfma notr nodma onlymfma onlyvalu alloff alloff_nodma base; do
  ./glm_planes16_probe_$v 2 $v
done
# fixed cost: the NODMA build with an empty tile loop, then with 1/4 and 1/2 of the tiles
./glm_planes16_probe_alloff_nodma 2 "alloff_nodma, no tiles" 0
./glm_planes16_probe_alloff_nodma 2 "alloff_nodma, 1/4 tiles" 3906
./glm_planes16_probe_alloff_nodma 2 "alloff_nodma, 1/2 tiles" 7812
./glm_planes16_probe_nodma 2 "nodma, no tiles" 0
./glm_planes16_probe_nodma 2 "nodma, 1/2 tiles" 7812
./glm_planes16_probe_nodma 1 "nodma wg/CU=1"
./glm_planes16_probe_nodma 3 "nodma wg/CU=3"
./glm_planes16_probe_base 1 "base wg/CU=1"
./glm_planes16_probe_base 3 "base wg/CU=3"
./glm_planes16_probe_alloff 1 "alloff wg/CU=1"
./glm_planes16_probe_alloff 3 "alloff wg/CU=3"
./glm_planes16_probe_alloff 4 "alloff wg/CU=4"

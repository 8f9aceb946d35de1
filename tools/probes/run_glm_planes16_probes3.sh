#!/bin/bash
cd "$(dirname "$0")"
for v in base wide wide4 wide_nodma wide_noelem wide_novalu wide_nosplit wide_notrans wide_nogemm2 wide_alloff wide_alloff_nodma base wide; do
  ./glm_planes16_probe_$v 2 $v
done
./glm_planes16_probe_wide 1 "wide wg/CU=1"
./glm_planes16_probe_wide_alloff_nodma 2 "wide_alloff_nodma, no tiles" 0

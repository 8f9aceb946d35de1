#!/bin/bash
cd "$(dirname "$0")"
for v in base noprio priotile alloff_nodma alloff_nodma_noprio alloff_nodma_priv alloff_nodma_priv_noprio priv priv_priotile occ2 occ2_priotile nodma nodma_priotile base; do
  ./glm_planes16_probe_$v 2 $v
done

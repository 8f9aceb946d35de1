#!/bin/bash
# ablation builds of tools/probes/glm_planes16_probe (developer tool; binaries are git-ignored)
cd "$(dirname "$0")"
H=/opt/rocm/bin/hipcc
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-value -Wno-unused-variable -I../../include"
mk() { name=$1; shift; $H $F "$@" glm_planes16_probe.hip -o glm_planes16_probe_$name 2>&1 | grep -v "warning\|hip-link\|^ \|^$\|generated" & }
mk base
mk notrans -DPA_GLMH_ABL_NOTRANS
mk nosplit -DPA_GLMH_ABL_NOSPLIT
mk noelem -DPA_GLMH_ABL_NOELEM
mk novalu -DPA_GLMH_ABL_NOELEM -DPA_GLMH_ABL_NOSPLIT
mk nogemm1 -DPA_GLMH_ABL_NOGEMM1
mk nogemm2 -DPA_GLMH_ABL_NOGEMM2
mk nomfma -DPA_GLMH_ABL_NOGEMM1 -DPA_GLMH_ABL_NOGEMM2
mk notr -DPA_GLMH_ABL_NOTR
mk nodma -DPA_GLMH_ABL_NODMA
mk onlymfma -DPA_GLMH_ABL_NOELEM -DPA_GLMH_ABL_NOSPLIT -DPA_GLMH_ABL_NOTR
mk onlyvalu -DPA_GLMH_ABL_NOGEMM1 -DPA_GLMH_ABL_NOGEMM2 -DPA_GLMH_ABL_NOTR
mk alloff -DPA_GLMH_ABL_NOELEM -DPA_GLMH_ABL_NOSPLIT -DPA_GLMH_ABL_NOGEMM1 -DPA_GLMH_ABL_NOGEMM2 -DPA_GLMH_ABL_NOTR
mk alloff_nodma -DPA_GLMH_ABL_NOELEM -DPA_GLMH_ABL_NOSPLIT -DPA_GLMH_ABL_NOGEMM1 -DPA_GLMH_ABL_NOGEMM2 -DPA_GLMH_ABL_NOTR -DPA_GLMH_ABL_NODMA
mk noprio -DPA_GLMH_ABL_NOPRIO
mk priotile -DPA_GLMH_PRIO_BY_TILE
mk alloff_nodma_noprio -DPA_GLMH_ABL_NOELEM -DPA_GLMH_ABL_NOSPLIT -DPA_GLMH_ABL_NOGEMM1 -DPA_GLMH_ABL_NOGEMM2 -DPA_GLMH_ABL_NOTR -DPA_GLMH_ABL_NODMA -DPA_GLMH_ABL_NOPRIO
mk alloff_nodma_priv -DPA_GLMH_ABL_NOELEM -DPA_GLMH_ABL_NOSPLIT -DPA_GLMH_ABL_NOGEMM1 -DPA_GLMH_ABL_NOGEMM2 -DPA_GLMH_ABL_NOTR -DPA_GLMH_ABL_NODMA -DPROBE_PRIV=true
mk alloff_nodma_priv_noprio -DPA_GLMH_ABL_NOELEM -DPA_GLMH_ABL_NOSPLIT -DPA_GLMH_ABL_NOGEMM1 -DPA_GLMH_ABL_NOGEMM2 -DPA_GLMH_ABL_NOTR -DPA_GLMH_ABL_NODMA -DPROBE_PRIV=true -DPA_GLMH_ABL_NOPRIO
mk priv -DPROBE_PRIV=true
mk priv_priotile -DPROBE_PRIV=true -DPA_GLMH_PRIO_BY_TILE
mk occ2 -DPROBE_OCC=2
mk occ2_priotile -DPROBE_OCC=2 -DPA_GLMH_PRIO_BY_TILE
mk nodma_priotile -DPA_GLMH_ABL_NODMA -DPA_GLMH_PRIO_BY_TILE
wait
ls glm_planes16_probe_*

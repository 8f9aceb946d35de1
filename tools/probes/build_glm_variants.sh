#!/bin/bash
# builds tools/probes/glm_variants (developer tool; binary is git-ignored)
set -e
cd "$(dirname "$0")"
H=/opt/rocm/bin/hipcc
F="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value"
$H $F -DPROBE_NAME=base -c glm_variants_unit.hip -o /tmp/gv_base.o &
$H $F -DPROBE_NAME=noelem -DPA_GLM_PROBE_NOELEM -c glm_variants_unit.hip -o /tmp/gv_noelem.o &
$H $F -DPROBE_NAME=nosplitg -DPA_GLM_PROBE_NOSPLITG -c glm_variants_unit.hip -o /tmp/gv_nosplitg.o &
$H $F -DPROBE_NAME=nosplitx -DPA_GLM_PROBE_NOSPLITX -c glm_variants_unit.hip -o /tmp/gv_nosplitx.o &
$H $F -DPROBE_NAME=nosplit -DPA_GLM_PROBE_NOSPLITX -DPA_GLM_PROBE_NOSPLITG -c glm_variants_unit.hip -o /tmp/gv_nosplit.o &
$H $F -DPROBE_NAME=nogemm2 -DPA_GLM_PROBE_NOGEMM2 -c glm_variants_unit.hip -o /tmp/gv_nogemm2.o &
$H $F -DPROBE_NAME=noxb -DPA_GLM_PROBE_NOXB -c glm_variants_unit.hip -o /tmp/gv_noxb.o &
$H $F -DPROBE_NAME=nogemm1 -DPA_GLM_PROBE_NOGEMM1 -c glm_variants_unit.hip -o /tmp/gv_nogemm1.o &
$H $F -DPROBE_NAME=minimal -DPA_GLM_PROBE_NOELEM -DPA_GLM_PROBE_NOSPLITG -DPA_GLM_PROBE_NOSPLITX -DPA_GLM_PROBE_NOGEMM1 -DPA_GLM_PROBE_NOGEMM2 -c glm_variants_unit.hip -o /tmp/gv_minimal.o &
$H $F -DPROBE_NAME=nomfma -DPA_GLM_PROBE_NOGEMM1 -DPA_GLM_PROBE_NOGEMM2 -c glm_variants_unit.hip -o /tmp/gv_nomfma.o &
$H $F -DPROBE_NAME=novalu -DPA_GLM_PROBE_NOELEM -DPA_GLM_PROBE_NOSPLITG -DPA_GLM_PROBE_NOSPLITX -c glm_variants_unit.hip -o /tmp/gv_novalu.o &
$H $F -DPROBE_NAME=sched6 -DPA_GLM_SCHED_PIPELINE=6 -c glm_variants_unit.hip -o /tmp/gv_sched6.o &
$H $F -DPROBE_NAME=sched8 -DPA_GLM_SCHED_PIPELINE=8 -c glm_variants_unit.hip -o /tmp/gv_sched8.o &
$H $F -DPROBE_NAME=sched10 -DPA_GLM_SCHED_PIPELINE=10 -c glm_variants_unit.hip -o /tmp/gv_sched10.o &
$H $F -DPROBE_NAME=sched11 -DPA_GLM_SCHED_PIPELINE=11 -c glm_variants_unit.hip -o /tmp/gv_sched11.o &
$H $F -DPROBE_NAME=sched13 -DPA_GLM_SCHED_PIPELINE=13 -c glm_variants_unit.hip -o /tmp/gv_sched13.o &
wait
$H $F -c glm_variants_main.cpp -o /tmp/gvmain.o
$H --offload-arch=gfx950 /tmp/gvmain.o /tmp/gv_*.o -o glm_variants

export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out/r02b
./tools/probes/issue_probe > gpurun_out/r02b/issue_probe.txt 2>&1
cat gpurun_out/r02b/issue_probe.txt
bash tools/pmc_glm.sh r02b 2>&1 | tail -80

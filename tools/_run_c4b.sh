#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | tail -6
import sys, torch
sys.path.insert(0, "tools")
import bench_configs as bc
dev = torch.device("cuda:0")
for bs in (32, 4096, None):
    print(bs, bc.config4(dev, steps=30, batch_size=bs))
PY

#!/bin/bash
# final evidence of round 4 on the final sources: GPU suite, smoke(), kernel stats, PMC passes, the bench line with traffic
export TAG=r04
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -rf 2>&1 | tail -15 > gpurun_out/${TAG}_pytest_gpu.txt
cat gpurun_out/${TAG}_pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/${TAG}_smoke.txt
TAG=$TAG bash tools/profile_round.sh

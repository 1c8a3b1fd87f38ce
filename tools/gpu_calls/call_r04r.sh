#!/bin/bash
# comment-only change of mk_solver.hip after the evidence call: the kernel-source hash moved, so the kernel statistics, the PMC
# passes and the bench line are taken again (same kernels)
export TAG=r04
mkdir -p gpurun_out
TAG=$TAG bash tools/profile_round.sh

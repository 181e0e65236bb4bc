#!/bin/bash
# usage (in the build container, from the repo root): tools/gpu_prof.sh <name> [gpurun timeout]
# One gpurun call: the kernel-trace + PMC passes of tools/prof_pmc.sh at the current HEAD, stamped with its revision (the GPU box
# gets a snapshot without .git).  Afterwards: cp gpurun_out/<name>/pmc_summary.json profiles/ and summary.txt -> profiles/rNN_rocprofv3_summary.txt.
REV=$(git rev-parse --short HEAD)$(git diff --quiet || echo +dirty)
exec /usr/local/graft/bin/gpurun --timeout ${2:-1100} -- "mkdir -p gpurun_out/$1 && GIT_REV=$REV bash tools/prof_pmc.sh $1 > gpurun_out/$1/prof.log 2>&1; tail -5 gpurun_out/$1/prof.log"

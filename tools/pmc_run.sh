#!/bin/bash
# usage: tools/pmc_run.sh <tag> <counters...> -- <python script args>
# collects PMC counters in their own pass (kernel-trace only), output under gpurun_out/<tag>
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" -d gpurun_out/$tag -o $tag -- python3 "$@" > gpurun_out/$tag.log 2>&1
echo "rc=$?"

#!/bin/bash
# One gpurun call's worth of work: tools/run_gpu.sh <tag> <command...> -> logs under gpurun_out/<tag>/
tag=$1; shift
mkdir -p gpurun_out/$tag
( eval "$@" ) > gpurun_out/$tag/out.log 2> gpurun_out/$tag/err.log
echo "rc=$?" >> gpurun_out/$tag/out.log
tail -c 3000 gpurun_out/$tag/out.log

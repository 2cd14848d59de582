#!/bin/bash
# The round's measurement pipeline on the GPU box (one gpurun call): bench line, rocprofv3 kernel trace of the same command, the
# dominant kernel by launch shape, the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate, with --kernel-trace only) and the frame
# schedule from the model's own events.  Summaries land under gpurun_out/<tag>_*; copy what is to be judged into profiles/.
#   bash tools/profile_round.sh r04
set -o pipefail
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_640_default.json 2> $O/${TAG}_bench_640_default.log || exit 1
echo "bench done"
rocprofv3 --kernel-trace --stats -d $O/prof_${TAG} -o ${TAG} -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-variants --no-cpu-baseline > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof_bench.log || exit 1
python3 tools/kernel_stats.py $O/prof_${TAG}/${TAG}_results.db $O/${TAG}_bench_640_kernel_stats.csv
python3 tools/kernel_shape_stats.py $O/prof_${TAG}/${TAG}_results.db 'conv_igemm_kernelILi64ELi64ELi32ELb0ELb0' $O/${TAG}_dominant_kernel_rocprof.json
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${TAG}_fetch -o f -- python3 bench.py --steps 6 --warmup 3 --no-variants --no-cpu-baseline --no-kernel-events > /dev/null 2> $O/${TAG}_pmc_fetch.log || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${TAG}_write -o w -- python3 bench.py --steps 6 --warmup 3 --no-variants --no-cpu-baseline --no-kernel-events > /dev/null 2> $O/${TAG}_pmc_write.log || exit 1
python3 tools/traffic_json.py $O/pmc_${TAG}_fetch $O/pmc_${TAG}_write $O/${TAG}
python3 tools/pmc_summary.py $O/pmc_${TAG}_fetch FETCH_SIZE $O/${TAG}_pmc_fetch_size.csv
python3 tools/pmc_summary.py $O/pmc_${TAG}_write WRITE_SIZE $O/${TAG}_pmc_write_size.csv
echo "pmc done"
python3 tools/frame_schedule.py > $O/${TAG}_frame_schedule.txt 2>&1
cat $O/${TAG}_frame_schedule.txt

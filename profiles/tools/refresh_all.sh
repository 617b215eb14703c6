#!/bin/bash
# Everything profiles/ holds for one build, in the order it has to be made (GPU box, repo root):
#   1. bash profiles/tools/refresh_all.sh r04            kernel trace + the four PMC passes on the default workload, kernel traces of the dense / k = 31
#                                                        workloads, bench lines of every workload
#   2. (in the container) grep '"metric"' gpurun_out/r04_kt.log | tail -n 1 > gpurun_out/r04_bench.json
#      python3 profiles/make_summary.py r04 gpurun_out/r04_kt gpurun_out/r04_fetch gpurun_out/r04_write gpurun_out/r04_sq gpurun_out/r04_req
#      cp the newest gpurun_out/r04_<workload>_kt/*/*kernel_stats.csv and gpurun_out/r04_bench_line_<workload>.json into profiles/
#   3. python bench.py > gpurun_out/r04_bench_line.json once more ON THE GPU BOX: only now does profiles/traffic.json carry this build's fingerprint
#      and the line quote `roofline.traffic`; cp it to profiles/r04_bench_line.json
TAG=${1:?tag}
set -o pipefail
mkdir -p gpurun_out
bash profiles/run_profiles.sh $TAG all > gpurun_out/${TAG}_top.txt 2>&1
BENCH_ARGS="--workload dense-repeats-8th" bash profiles/run_profiles.sh ${TAG}_dense8th kt > gpurun_out/${TAG}_dense8th_top.txt 2>&1
BENCH_ARGS="--workload hifi-k31" bash profiles/run_profiles.sh ${TAG}_k31 kt > gpurun_out/${TAG}_k31_top.txt 2>&1
BENCH_ARGS="--workload dense-repeats-25th" bash profiles/run_profiles.sh ${TAG}_dense25th kt > gpurun_out/${TAG}_dense25th_top.txt 2>&1
bash profiles/tools/final_lines.sh $TAG

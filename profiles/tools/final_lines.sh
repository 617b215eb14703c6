TAG=${1:-r05}
set -o pipefail
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_line.err
for w in ecsample30x-like hifi-half dense-repeats-25th dense-repeats-8th hifi-k31; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-full > gpurun_out/${TAG}_bench_line_$w.json 2> gpurun_out/${TAG}_bench_line_$w.err
done

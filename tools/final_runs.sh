set -x
python bench.py --workload cfg4 --layers-json gpurun_out/r03x_layers_cfg4.json --no-cpu-baseline --no-extras > gpurun_out/r03x_bench_cfg4_1gpu.json 2> gpurun_out/r03x_bench_cfg4.err
python bench.py --workload cfg2 --no-cpu-baseline --no-extras > gpurun_out/r03x_bench_cfg2_1gpu.json 2> gpurun_out/r03x_bench_cfg2.err
python bench.py --train --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r03x_bench_cfg3_train_graph.json 2> gpurun_out/r03x_bench_train.err
tools/profile_train.sh r03x 5 > gpurun_out/r03x_profile_train.log 2>&1
python bench.py --workload cfg5 --steps 2 --warmup 1 --layers-json gpurun_out/r03x_layers_cfg5.json --no-cpu-baseline --no-extras > gpurun_out/r03x_bench_cfg5_1gpu.json 2> gpurun_out/r03x_bench_cfg5.err
tail -c 400 gpurun_out/r03x_bench_cfg4_1gpu.json; echo; tail -c 300 gpurun_out/r03x_bench_cfg2_1gpu.json; echo; cut -c1-400 gpurun_out/r03x_bench_cfg3_train_graph.json; echo; cut -c1-300 gpurun_out/r03x_bench_cfg5_1gpu.json

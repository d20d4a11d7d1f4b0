mkdir -p gpurun_out/r02n
( timeout 900 python -m pytest tests/test_hip_parity.py -q -k "full_size_vs_oracle" > gpurun_out/r02n/pytest_100k.log 2>&1; echo "rc=$?" >> gpurun_out/r02n/pytest_100k.log ) &
BG=$!
NSR_LIB_PATH=$PWD/nice_slam_amd/_ab/libnsr_ts.so timeout 300 python tests/perf/ts_probe.py 1000 color fused > gpurun_out/r02n/ts_color_fused.txt 2>&1
grep -E "^pass|kernel span|tile [0-9]:" gpurun_out/r02n/ts_color_fused.txt
timeout 300 python bench.py --no-cpu-baseline --windows 1 > gpurun_out/r02n/bench_1.json 2> gpurun_out/r02n/bench_1.err; cut -c1-900 gpurun_out/r02n/bench_1.json
timeout 600 python -m pytest tests -q -m gpu -x --deselect tests/test_hip_parity.py::test_synthetic_stress_full_size_vs_oracle > gpurun_out/r02n/pytest_rest.log 2>&1; tail -5 gpurun_out/r02n/pytest_rest.log
wait $BG
tail -15 gpurun_out/r02n/pytest_100k.log

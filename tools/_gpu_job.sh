mkdir -p gpurun_out/r02y
S=$(date +%s)
timeout 1200 python -m pytest tests/test_hip_parity.py -q -k "full_size_vs_oracle" -s > gpurun_out/r02y/pytest_100k.log 2>&1
echo "wall $(( $(date +%s) - S )) s" >> gpurun_out/r02y/pytest_100k.log
tail -8 gpurun_out/r02y/pytest_100k.log

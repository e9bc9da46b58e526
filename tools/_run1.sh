cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r03a/pytest.txt; cat gpurun_out/r03a/pytest.txt
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-extras > gpurun_out/r03a/bench_mixed.json 2> gpurun_out/r03a/bench_mixed.err; cat gpurun_out/r03a/bench_mixed.json | cut -c1-600
timeout 900 tools/record_others.sh r03a 2>&1 | tail -40

cd /root/repo
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/collect_profiles.sh r1_j > gpurun_out/r1_j_collect.log 2>&1
tail -3 gpurun_out/r1_j_collect.log
cat gpurun_out/r1_j_bench.json | cut -c1-900

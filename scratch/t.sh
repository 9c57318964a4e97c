cd /root/repo
timeout 300 python scratch/bench_k1.py 2>&1 | grep drop-in

cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_b_kt; mkdir -p $out
args="bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --no-pipelined --no-fast --no-calibrated --no-one-pass --no-extras --no-rccl"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $args > $out/bench.json 2> $out/kt.log
ks=$(find $out/kt -name '*kernel_stats.csv' | head -1)
python3 az-net_amd/tools/summarize_prof.py "$ks" $out/kernel_stats.csv "level loop"
rm -rf $out/kt
grep -E "^k_|name" $out/kernel_stats.csv

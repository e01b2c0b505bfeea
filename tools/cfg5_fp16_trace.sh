# where the parity-grade (fp16 mode) cfg5 render spends its time: kernel stats of tools/cfg_bench.py cfg5 65536 fp16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/cfg5fp16; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o r -- python3 tools/cfg_bench.py cfg5 65536 fp16 > $O/run.log 2>&1
tail -2 $O/run.log
python tools/kstats.py $(find $O/ks -name "*kernel_stats.csv" | head -1) 25

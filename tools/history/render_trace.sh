# kernel stats of the forward render at a given ray count: bash tools/render_trace.sh <outdir> <rays> [mode]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-rtrace}; N=${2:-8192}; M=${3:-bf16}
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$N -o render -- python3 tools/x3_bench.py $N $M > $O/kt_$N.log 2>&1
python tools/kstats.py $(find $O/kt_$N -name "*kernel_stats.csv" | head -1) 40 | sed 's/^/   /'
tail -2 $O/kt_$N.log

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "16384 16" "65536 64"; do set -- $cfg
export LPHQ_SEARCH_LOOKAHEAD=$1 LPHQ_SEARCH_STEPS=$2
d=$R/gpurun_out/trace_build_$1
timeout 900 rocprofv3 --kernel-trace --stats -d $d -o t -- python3 $R/tools/search_sweep_child.py $R 63 20 128 20000 /tmp/x.lph > /dev/null 2> $R/gpurun_out/trace_build_$1.err
db=$(find $d -name '*_results.db' | head -1)
[ -n "$db" ] && python3 $R/tools/rocpd_summary.py $db $R/gpurun_out/r05_build_repair_$1_kernel_stats.txt > /dev/null
rm -rf $d
echo "== lookahead $1 steps $2"; head -12 $R/gpurun_out/r05_build_repair_$1_kernel_stats.txt | cut -c1-150
done

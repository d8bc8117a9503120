set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp1
for v in ilp2 prio priop; do
  LPHQ_LIB_PATH=$PWD/lphash_amd/liblphq_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "packed or contig or ragged or golden" > gpurun_out/exp1/pytest_$v.txt 2>&1; tail -2 gpurun_out/exp1/pytest_$v.txt
done
bash tools/ab_multi.sh exp1_c2 "" lphash_amd/liblphq.so lphash_amd/liblphq_ilp2.so lphash_amd/liblphq_priop.so
bash tools/ab_multi.sh exp1_c3 "--mode stream" lphash_amd/liblphq.so lphash_amd/liblphq_prio.so
bash tools/ab_multi.sh exp1_c5 "--kind u" lphash_amd/liblphq.so lphash_amd/liblphq_ilp2.so

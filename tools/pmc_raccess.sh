#!/bin/bash
# PMC counters of the Raccess kernels on 32 x 2 kb (the latency mode: helper wavefronts, windows): one rocprofv3 pass per
# counter set.  usage (through gpurun): bash tools/pmc_raccess.sh > gpurun_out/pmc_raccess.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_raccess
mkdir -p $OUT
for set in "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex 'k_inside|k_outside|k_biloop|k_access' --output-format csv -d $OUT/$tag -- python3 $R/tools/gpu_probe_raccess.py 32 2000 > $OUT/$tag.log 2>&1 || { tail -5 $OUT/$tag.log; exit 1; }
  echo "done $tag"
done
for d in $OUT/*/; do python3 $R/tools/pmc_summary.py $(find $d -name "*counter_collection.csv"); done

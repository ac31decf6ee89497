#!/bin/bash
# PMC passes over a short bench run, one rocprofv3 run per counter set (never mixed with trace domains).
#   tools/pmc.sh TAG [bench args...]    ->  gpurun_out/pmc_TAG_<set>/..., summary by tools/pmc_summary.py
tag=$1; shift
args="--steps 1 --warmup 0 --spp 8 --no-cpu-baseline $*"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
declare -A SETS
SETS[sq1]="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT"
SETS[sq2]="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_SCA"
SETS[tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
SETS[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
for s in sq1 sq2 tcp tcc; do
  rm -rf gpurun_out/pmc_${tag}_$s
  timeout -s KILL 240 rocprofv3 --pmc ${SETS[$s]} -d gpurun_out/pmc_${tag}_$s --output-format csv -- python3 bench.py $args > gpurun_out/pmc_${tag}_$s.log 2>&1 || { echo "pass $s failed"; tail -5 gpurun_out/pmc_${tag}_$s.log; exit 1; }
done
python3 tools/pmc_summary.py --json gpurun_out/pmc_${tag}_limiters.json gpurun_out/pmc_${tag}_* | tee gpurun_out/pmc_${tag}_summary.txt
# keep only the summaries (the raw per-dispatch CSVs are large)
for s in sq1 sq2 tcp tcc; do rm -rf gpurun_out/pmc_${tag}_$s; done

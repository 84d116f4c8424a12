#!/bin/bash
# Where the waves of the dominant kernel spend their cycles: SQ / GRBM counter passes over one N = 65 536 solve
# (tools/quick_perf.py), kernel-trace only, summed per kernel by tools/pmc_sq_derive.py -> gpurun_out/pmc_sq.txt
# Run on the GPU box:  bash tools/pmc_sq.sh
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmcsq_*
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export TGP_SYNC_EVENTS=1          # see tools/pmc_bench.sh
rocprofv3 -L > $R/gpurun_out/pmcsq_avail.txt 2>&1
P1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
P2="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES"
P3="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/pmcsq_$i -- python3 $R/tools/quick_perf.py 65536 > $R/gpurun_out/pmcsq_$i.log 2>&1
  echo "pass $i rc=$?"
done
python3 $R/tools/pmc_sq_derive.py > $R/gpurun_out/pmc_sq.txt 2>&1
cat $R/gpurun_out/pmc_sq.txt

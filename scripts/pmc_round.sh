#!/bin/bash
# PMC passes (each its own rocprofv3 run) for every kernel of a frame -> gpurun_out/pmc_<tag>.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
W=${WORKLOAD:-c2}
bash scripts/pmc_once.sh sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" --workload $W > gpurun_out/pmc_sq.txt 2>&1
bash scripts/pmc_once.sh fetch "FETCH_SIZE" --workload $W > gpurun_out/pmc_fetch.txt 2>&1
bash scripts/pmc_once.sh write "WRITE_SIZE" --workload $W > gpurun_out/pmc_write.txt 2>&1
bash scripts/pmc_once.sh lds "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY" --workload $W > gpurun_out/pmc_lds.txt 2>&1
tail -n 40 gpurun_out/pmc_*.txt

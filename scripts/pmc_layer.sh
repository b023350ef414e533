#!/bin/bash
# usage: scripts/pmc_layer.sh <tag> <layer_bench args...>   -- three PMC passes over one layer_bench invocation
# (separate runs per counter group: rocprofv3 does not split counters over passes)
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $R/$out/p1 -- python3 $R/scripts/layer_bench.py "$@" > $R/$out/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d $R/$out/p2 -- python3 $R/scripts/layer_bench.py "$@" > $R/$out/p2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $R/$out/p3 -- python3 $R/scripts/layer_bench.py "$@" > $R/$out/p3.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $R/$out/p4 -- python3 $R/scripts/layer_bench.py "$@" > $R/$out/p4.log 2>&1 || exit 1

# GPU box: SQ counters of the MLP-chain kernels at the FP1 shape (two passes of 8 SQ counters)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_gemm
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1 /tmp/p2
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/p1 -o a -- python3 $R/tools/gemm_only.py > /dev/null 2> $O/err1.txt
python3 $R/tools/pmc_summary.py "$(find /tmp/p1 -name '*counter_collection.csv' | head -1)" gemm_kernel > $O/pass1.txt
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d /tmp/p2 -o b -- python3 $R/tools/gemm_only.py > /dev/null 2> $O/err2.txt
python3 $R/tools/pmc_summary.py "$(find /tmp/p2 -name '*counter_collection.csv' | head -1)" gemm_kernel > $O/pass2.txt
echo done

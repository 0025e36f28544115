set -o pipefail
export TMPDIR=/tmp PYTHONPATH=$PWD
O=gpurun_out/pmc_ab; rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1 || true
grep -o "SQ_INSTS_VMEM[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_ACTIVE_INST_[A-Z_]*\|TA_BUSY[a-z_A-Z]*\|TA_TA_BUSY[a-zA-Z_]*\|SQ_WAIT_INST_[A-Z_]*\|SQ_INSTS_FLAT[A-Z_]*\|TCP_[A-Z_]*STALL[A-Z_]*\|TA_ADDR_STALL[A-Z_a-z]*\|TA_DATA_STALL[A-Za-z_]*" $O/avail.txt | sort -u > $O/names.txt
cat $O/names.txt | tr '\n' ' '

# Development: three SQ counter passes over tools/${PROF_SCRIPT:-prof_fwd.py}; results under gpurun_out/$1/
out=gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAIT_ANY -d $R/$out -o p1 --output-format csv -- python3 $R/tools/${PROF_SCRIPT:-prof_fwd.py} > $R/$out/p1.log 2>&1 &&
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD -d $R/$out -o p2 --output-format csv -- python3 $R/tools/${PROF_SCRIPT:-prof_fwd.py} > $R/$out/p2.log 2>&1 &&
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES -d $R/$out -o p3 --output-format csv -- python3 $R/tools/${PROF_SCRIPT:-prof_fwd.py} > $R/$out/p3.log 2>&1
cd $R && python3 tools/pmc_ratio.py $(find $out -name "*counter_collection.csv")

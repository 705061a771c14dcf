#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ctr; mkdir -p $out
rocprofv3 -L > $out/list.txt 2>&1
grep -i -E "icache|ifetch|SQC_|INST_CACHE|SQ_INSTS_|SQ_WAIT|SQ_THREAD|SQ_INST_CYCLES|LEVEL" $out/list.txt | head -80

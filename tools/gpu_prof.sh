#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_prof.sh <outdir> <op> [ENV=VAL ...] — kernel-trace times and one PMC pass
# (SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES) of tools/run_op.py <op>, summaries to <outdir>/{kt,pmc}_<op><suffix>.txt (suffix = the env settings)
out=$1; op=$2; shift 2
suffix=$(echo "$*" | tr -c 'A-Za-z0-9=\n' '_' | sed 's/_*$//')
[ -n "$suffix" ] && suffix="_$suffix"
mkdir -p $out
export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace -d /tmp/prof_kt_$$ -o r -- python tools/run_op.py $op 20 > $out/kt_$op$suffix.log 2>&1
db=$(find /tmp/prof_kt_$$ -name '*.db' | head -1); [ -n "$db" ] && python tools/prof_summary.py $db > $out/kt_$op$suffix.txt 2>&1
env "$@" rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d /tmp/prof_pmc_$$ -o r -- python tools/run_op.py $op 5 > $out/pmc_$op$suffix.log 2>&1
db=$(find /tmp/prof_pmc_$$ -name '*.db' | head -1); [ -n "$db" ] && python tools/pmc_summary.py $db zg > $out/pmc_$op$suffix.txt 2>&1
rm -rf /tmp/prof_kt_$$ /tmp/prof_pmc_$$
echo "== $op $*"; grep -v "^#" $out/kt_$op$suffix.txt | cut -c1-70,111-160 | head -4; cat $out/pmc_$op$suffix.txt | cut -c1-130 | head -8

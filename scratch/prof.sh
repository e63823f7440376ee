#!/bin/bash
# usage: prof.sh <tag> [bench args...]   -- kernel trace + two PMC passes of bench.py
set -u
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-images 0 --no-targets "$@" > $O/trace.json 2> $O/trace.err || { echo trace failed; tail -5 $O/trace.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-images 0 --no-targets "$@" > $O/pmc_fetch.json 2> $O/pmc_fetch.err || { echo fetch failed; tail -5 $O/pmc_fetch.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-images 0 --no-targets "$@" > $O/pmc_write.json 2> $O/pmc_write.err || { echo write failed; tail -5 $O/pmc_write.err; exit 1; }
find $O -name "*.csv" | head -20
ls -la $O/*/* | head

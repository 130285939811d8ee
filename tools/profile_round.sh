#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the default bench line, the same command under rocprofv3 (kernel stats, then the
# two PMC passes, each in its own run as MI355X_MICROARCH.md prescribes), the rc_ladder line and an AC sweep profile.
# Everything lands in gpurun_out/prof_round/; tools/collect_profiles.py turns it into profiles/rNN_*.
set -o pipefail
OUT=gpurun_out/prof_round
rm -rf $OUT; mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
step() { echo "== $1"; }
step bench;       timeout -k 10 600 python3 bench.py --single-instance > $OUT/bench_line.json 2> $OUT/bench.err || exit 1
step rc_ladder;   timeout -k 10 300 python3 bench.py --workload rc_ladder --no-cpu-baseline > $OUT/bench_line_rc_ladder.json 2>> $OUT/bench.err || exit 1
step stats;       timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_line_under_rocprof.json 2> $OUT/stats.err || exit 1
step pmc_fetch;   timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || exit 1
step pmc_write;   timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.err || exit 1
step ac;          timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ac_stats -- python3 tools/ac_probe.py --n 1000 --inst 64 --freqs 201 > $OUT/ac_probe.json 2> $OUT/ac.err || exit 1
find $OUT -name "*.csv" | head -40
cat $OUT/bench_line.json

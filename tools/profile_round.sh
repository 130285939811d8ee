#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the default bench line, the same command under rocprofv3 (kernel stats, then the
# PMC passes, each in its own run as MI355X_MICROARCH.md prescribes), the rc_ladder line, an AC sweep profile and
# BASELINE config 5 (full length + profiled shorter runs).  Everything lands in gpurun_out/prof_round/;
# tools/collect_profiles.py turns it into profiles/rNN_*.   Usage: profile_round.sh [part]   part in {bench, config5, all}
set -o pipefail
PART=${1:-all}
OUT=gpurun_out/prof_round
mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
step() { echo "== $1"; }
if [ "$PART" = bench ] || [ "$PART" = all ]; then
step bench;       timeout -k 10 600 python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.err || exit 1
step rc_ladder;   timeout -k 10 300 python3 bench.py --workload rc_ladder --no-cpu-baseline > $OUT/bench_line_rc_ladder.json 2>> $OUT/bench.err || exit 1
step stats;       timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-instance > $OUT/bench_line_under_rocprof.json 2> $OUT/stats.err || exit 1
step pmc_fetch;   timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-single-instance > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || exit 1
step pmc_write;   timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-single-instance > $OUT/pmc_write.json 2> $OUT/pmc_write.err || exit 1
step pmc_sq;      timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-single-instance > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err || exit 1
step pmc_lds;     timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_lds -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-single-instance > $OUT/pmc_lds.json 2> $OUT/pmc_lds.err || exit 1
step ac;          timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ac_stats -- python3 tools/ac_probe.py --n 1000 --inst 64 --freqs 201 > $OUT/ac_probe.json 2> $OUT/ac.err || exit 1
step ac_fetch;    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/ac_fetch -- python3 tools/ac_probe.py --n 1000 --inst 64 --freqs 201 > $OUT/ac_fetch.json 2> $OUT/ac_fetch.err || exit 1
step ac_write;    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/ac_write -- python3 tools/ac_probe.py --n 1000 --inst 64 --freqs 201 > $OUT/ac_write.json 2> $OUT/ac_write.err || exit 1
fi
if [ "$PART" = config5 ] || [ "$PART" = all ]; then
step c5_full;     timeout -k 10 900 python3 tools/config5_full.py > $OUT/config5_full.json 2> $OUT/config5.err || exit 1
step c5_tasklist; timeout -k 10 300 python3 tools/config5_full.py --steps 2000 --front-cut -1 --wgs 16 > $OUT/config5_tasklists_2000.json 2>> $OUT/config5.err || exit 1
step c5_x4;       timeout -k 10 300 python3 tools/config5_full.py --steps 5000 --inst 4 --check-steps 0 > $OUT/config5_x4.json 2>> $OUT/config5.err || exit 1
step c5_x16;      timeout -k 10 300 python3 tools/config5_full.py --steps 3000 --inst 16 --no-currents --check-steps 0 > $OUT/config5_x16.json 2>> $OUT/config5.err || exit 1
step c5_stats;    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_stats -- python3 tools/config5_full.py --steps 5000 --check-steps 0 > $OUT/config5_5000.json 2> $OUT/c5_stats.err || exit 1
step c5_fetch;    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/c5_fetch -- python3 tools/config5_full.py --steps 5000 --check-steps 0 > $OUT/c5_fetch.json 2> $OUT/c5_fetch.err || exit 1
step c5_write;    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/c5_write -- python3 tools/config5_full.py --steps 5000 --check-steps 0 > $OUT/c5_write.json 2> $OUT/c5_write.err || exit 1
fi
find $OUT -name "*.csv" | head -60
cat $OUT/bench_line.json 2>/dev/null | cut -c1-600
cat $OUT/config5_full.json 2>/dev/null

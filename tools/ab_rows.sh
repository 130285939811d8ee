# same-session A/B: row records on / off, both geometries (interleaved repetitions)
pick='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print("geom",d["geom"],"T",d["T"],"solves/s %.4g"%d["solves_per_s"],"us/step %.3f"%d["us_per_step"])'
for rep in 1 2 3; do for nr in 0 1; do
  echo -n "no_rows=$nr "; python tools/perf_probe.py --configs 512:1:512 --packed 1 --steps 6000 --reps 2 --no-rows $nr 2>/dev/null | python -c "$pick" || exit 1
done; done
for rep in 1 2; do for nr in 0 1; do
  echo -n "no_rows=$nr "; python tools/perf_probe.py --configs 1:1:1024 --steps 6000 --reps 2 --no-rows $nr 2>/dev/null | python -c "$pick" || exit 1
done; done
